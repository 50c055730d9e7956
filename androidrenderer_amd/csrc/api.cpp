// C ABI implementation (include/sah_hip.h): argument validation, host-side digestion of the uniform blocks,
// kernel launches on the context's HIP stream.  Compiled by hipcc with -ffp-contract=off: the fp32 expressions
// evaluated here are the uniform sub-expressions of the reference shaders and must round exactly as the
// per-pixel code would.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sah_hip.h"
#include "params.hpp"

namespace sah {
hipError_t launch_lighting(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const CacheArgs& cache, const RtgiArgs& rtgi,
                           const SkyArgs& sky, const FastArgs* fast, int sun_mode, int gi, int ppt, bool brute_force_lights, hipStream_t st);
hipError_t launch_colx_table(const LightingArgs& a, const FastArgs& f, float* out, uint32_t stride, uint32_t row_stride, hipStream_t st);
hipError_t launch_probe_irr_unpack(const VolumeArg& src, uint8_t* dst, hipStream_t st);
hipError_t launch_copy_scene(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                             uint32_t row_end, hipStream_t st);
hipError_t launch_bloom_downsample(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                                   uint32_t row_end, hipStream_t st);
struct TonemapArgs;
hipError_t launch_tonemap(const TonemapArgs& t, hipStream_t st);
hipError_t launch_lpv_clear(const VolumeArg* vols, int n, uint32_t num_cascades, hipStream_t st);
}  // namespace sah

#include "ctx.hpp"
#include "post_args.hpp"

namespace {


bool vec_ok(const sah_plane* p, uint32_t bytes) { return ((uintptr_t)p->ptr % bytes) == 0 && (p->row_pitch_bytes % bytes) == 0; }

// fp32 helpers that mirror the shader expressions (individually rounded; this TU is built with -ffp-contract=off)
void normalize3(const float in[3], float out[3]) {
    const float d = in[0] * in[0] + in[1] * in[1] + in[2] * in[2];
    const float inv = 1.0f / std::sqrt(d);
    for (int i = 0; i < 3; i++) out[i] = in[i] * inv;
}
void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - b[1] * a[2];
    o[1] = a[2] * b[0] - b[2] * a[0];
    o[2] = a[0] * b[1] - b[0] * a[1];
}
float round_to_half(float f) { return (float)(_Float16)f; }

bool is_zero(float x) { return x == 0.0f; }
bool all_finite(const float* m, int n) {
    for (int i = 0; i < n; i++)
        if (!std::isfinite(m[i])) return false;
    return true;
}
// |m[i]| <= 2^40 (NaN fails): with |world position| <= 2^41 (the kernel defers anything farther from the camera than 2^40) every
// product and partial sum of an affine transform stays finite, so cascade / shadow coordinates are never NaN for shaded pixels.
bool all_bounded(const float* m, int n) {
    for (int i = 0; i < n; i++)
        if (!(std::fabs(m[i]) <= 0x1p+40f)) return false;
    return true;
}

// Decides whether the uniform blocks have the structure the fast kernel assumes (DESIGN.md "Fast path proofs").
// Column-major m[col*4 + row].
bool detect_fast_path(const sah_lighting_desc* d, uint32_t sun_mode, uint32_t gi_kind, const sah::CsmArgs& csm, sah::FastArgs* f) {
    const float* P = d->view->inverse_projection;
    const float* V = d->view->inverse_view;
    if (!all_finite(P, 16) || !all_finite(V, 16)) return false;
    if (!std::isfinite(d->view->render_resolution[0]) || !std::isfinite(d->view->render_resolution[1])) return false;
    // inverse_projection: row 0 depends on X only, row 1 on Y only, rows 2-3 on depth only
    if (!(is_zero(P[4]) && is_zero(P[8]) && is_zero(P[1]) && is_zero(P[9]) && is_zero(P[2]) && is_zero(P[6]) && is_zero(P[3]) && is_zero(P[7])))
        return false;
    // inverse_view: affine
    if (!(is_zero(V[3]) && is_zero(V[7]) && is_zero(V[11]) && V[15] == 1.0f)) return false;
    if (!all_bounded(d->view->view + 12, 3)) return false;  // camera position (-view[3].xyz)
    f->p0 = P[0]; f->p12 = P[12]; f->p5 = P[5]; f->p13 = P[13];
    f->p10 = P[10]; f->p14 = P[14]; f->p11 = P[11]; f->p15 = P[15];
    {
        // Shared-reciprocal divide for vs = (X, Y, Z) / w (lighting_fast.hpp): numerators must be +0 or 2^-40 <= |n| <= 2^40.
        // X = p0 * ndc.x + p12 with ndc.x = tx * 2 - 1 either 0 or >= 2^-24 in magnitude (tx * 2 is exact, the subtraction is exact
        // by Sterbenz near 1 and >= 0.5 in magnitude elsewhere) and <= 2^9 when width <= 256 * render_resolution; so
        // 2^-16 <= |p0| <= 2^30 and p12 == +0 keep X in the domain (a -0 product plus +0 is +0).  Same for Y.  Z is the constant p14
        // when p10 == 0, which holds for every perspective projection's inverse.
        auto in = [](float v, float lo, float hi) { return std::fabs(v) >= lo && std::fabs(v) <= hi; };
        const float r0 = d->view->render_resolution[0], r1 = d->view->render_resolution[1];
        f->pos_div_nr = P[10] == 0.0f && P[12] == 0.0f && !std::signbit(P[12]) && P[13] == 0.0f && !std::signbit(P[13]) &&
                        in(P[0], 0x1p-16f, 0x1p+30f) && in(P[5], 0x1p-16f, 0x1p+30f) && in(P[14], 0x1p-40f, 0x1p+40f) && r0 > 0.f && r1 > 0.f &&
                        (float)d->lit->width <= 256.0f * r0 && (float)d->lit->height <= 256.0f * r1;
    }
    if (sun_mode == SAH_SHADOW_MODE_CSM) {
        if (!csm.shadowmap.ptr || (uint64_t)csm.shadowmap.slice_pitch * csm.shadowmap.depth >= (1ull << 32)) return false;
        if (!csm.is_d16 || !csm.d16_recip_ok) return false;
        for (int c = 0; c < 4; c++) {
            const float* b = csm.biased[c];
            if (!all_bounded(b, 16)) return false;
            if (!(is_zero(b[3]) && is_zero(b[7]) && is_zero(b[11]) && b[15] == 1.0f)) return false;
        }
    }
    if (gi_kind == SAH_GI_LPV) {
        const sah_gi& gi = *d->gi;
        if (gi.lpv_red.width != gi.lpv_green.width || gi.lpv_red.width != gi.lpv_blue.width || gi.lpv_red.height != gi.lpv_green.height ||
            gi.lpv_red.height != gi.lpv_blue.height || gi.lpv_red.depth != gi.lpv_green.depth || gi.lpv_red.depth != gi.lpv_blue.depth ||
            gi.lpv_red.row_pitch_bytes != gi.lpv_green.row_pitch_bytes || gi.lpv_red.row_pitch_bytes != gi.lpv_blue.row_pitch_bytes ||
            gi.lpv_red.slice_pitch_bytes != gi.lpv_green.slice_pitch_bytes || gi.lpv_red.slice_pitch_bytes != gi.lpv_blue.slice_pitch_bytes ||
            (uint64_t)gi.lpv_red.slice_pitch_bytes * gi.lpv_red.depth >= (1ull << 32))
            return false;
        for (uint32_t c = 0; c < gi.lpv_num_cascades; c++) {
            const float* m = gi.lpv_cascades[c].world_to_cascade;
            if (!all_bounded(m, 16)) return false;
            if (!(is_zero(m[1]) && is_zero(m[2]) && is_zero(m[3]) && is_zero(m[4]) && is_zero(m[6]) && is_zero(m[7]) && is_zero(m[8]) &&
                  is_zero(m[9]) && is_zero(m[11]) && m[15] == 1.0f))
                return false;
            if (is_zero(m[0]) || is_zero(m[5]) || is_zero(m[10])) return false;
            f->lpv_s[c][0] = m[0]; f->lpv_s[c][1] = m[5]; f->lpv_s[c][2] = m[10];
            f->lpv_t[c][0] = m[12]; f->lpv_t[c][1] = m[13]; f->lpv_t[c][2] = m[14];
        }
        const uint32_t n = gi.lpv_num_cascades;
        f->ncasc_pow2 = (n & (n - 1)) == 0;
        f->inv_ncasc = 1.0f / (float)n;
        if (!std::isfinite(gi.lpv_exposure)) return false;
    }
    return true;
}

}  // namespace

void SahRange::resolve(push_fn& push, pop_fn& pop) {
    const char* on = getenv("SAH_ROCTX");
    if (!on || atoi(on) == 0) return;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        push = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr;
        pop = nullptr;
    }
}

bool sah_ipc_timed_out(const sah_ctx* ctx);  // api_ipc.cpp

extern "C" {

int sah_abi_version(void) { return SAH_ABI_VERSION; }

const char* sah_status_string(int s) {
    switch (s) {
        case SAH_OK: return "ok";
        case SAH_ERR_INVALID_ARGUMENT: return "invalid argument";
        case SAH_ERR_UNSUPPORTED_FORMAT: return "unsupported format";
        case SAH_ERR_HIP: return "HIP error";
        case SAH_ERR_NO_DEVICE: return "no HIP device";
        case SAH_ERR_COMM: return "communicator error";
        case SAH_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown";
}

const char* sah_last_error(const sah_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int sah_comm_init(sah_ctx* ctx, const void* comm_id);
void sah_comm_destroy(sah_ctx* ctx);
void sah_ipc_destroy(sah_ctx* ctx);

int sah_create(sah_ctx** out, int device, int rank, int world, const void* comm_id) {
    if (!out || world < 1 || rank < 0 || rank >= world) return SAH_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return SAH_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return SAH_ERR_INVALID_ARGUMENT;
    sah_ctx* ctx = new sah_ctx();
    ctx->device = device;
    ctx->rank = rank;
    ctx->world = world;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return SAH_ERR_HIP; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return SAH_ERR_HIP; }
    ctx->own_stream = true;
    // format tables: sRGB8 -> linear (Vulkan sRGB EOTF, evaluated in double, rounded to fp32) and UNORM8 -> float
    float host[512];
    for (int i = 0; i < 256; i++) {
        const double c = (double)i / 255.0;
        host[i] = (float)((c <= 0.04045) ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
        host[256 + i] = (float)i / 255.0f;
    }
    if (hipMalloc((void**)&ctx->luts, sizeof(host)) != hipSuccess ||
        hipMemcpy(ctx->luts, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) {
        sah_destroy(ctx);
        return SAH_ERR_HIP;
    }
    if (comm_id) {  // world == 1 with an id builds a one-rank communicator: the N-rank exchange path, runnable on one GPU
        int rc = sah_comm_init(ctx, comm_id);
        if (rc != SAH_OK) { sah_destroy(ctx); return rc; }
    }
    if (hipMalloc((void**)&ctx->state, sizeof(sah::FrameState)) != hipSuccess ||
        hipMemset(ctx->state, 0, sizeof(sah::FrameState)) != hipSuccess) {
        sah_destroy(ctx);
        return SAH_ERR_HIP;
    }
    const char* ppt = getenv("SAH_FORCE_PPT");
    if (ppt) ctx->force_ppt = atoi(ppt);
    const char* mc = getenv("SAH_RASTER_MERGE_CAPACITY");
    if (mc) ctx->raster_merge_cap = (uint32_t)atoi(mc);
    const char* gen = getenv("SAH_FORCE_GENERAL");
    if (gen) ctx->force_general = atoi(gen) != 0;
    *out = ctx;
    return SAH_OK;
}

// Test hook (tests/test_abi_fuzz.py): a context on a machine WITHOUT a HIP device.  Every entry point validates its arguments as it always
// does and then fails at its first HIP call with SAH_ERR_HIP — which is what the argument fuzz wants to see for every combination of extents,
// pitches, formats, row ranges and null sub-pointers: a status code, never a fault.  Refused where a device exists (a launch with the fuzz's
// made-up addresses must not reach a GPU).
int sah_debug_create_detached(sah_ctx** out) {
    if (!out) return SAH_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) == hipSuccess && n > 0) return SAH_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    sah_ctx* ctx = new sah_ctx();
    ctx->device = -1;
    *out = ctx;
    return SAH_OK;
}

void sah_destroy(sah_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    sah_comm_destroy(ctx);  // (synchronises both streams first)
    sah_ipc_destroy(ctx);
    if (ctx->comm_ready) (void)hipEventDestroy(ctx->comm_ready);
    if (ctx->comm_done) (void)hipEventDestroy(ctx->comm_done);
    if (ctx->luts) (void)hipFree(ctx->luts);
    if (ctx->probe_slots) (void)hipFree(ctx->probe_slots);
    if (ctx->probe_done) (void)hipEventDestroy(ctx->probe_done);
    if (ctx->state) (void)hipFree(ctx->state);
    if (ctx->list) (void)hipFree(ctx->list);
    if (ctx->lpv_packed) (void)hipFree(ctx->lpv_packed);
    if (ctx->irr32) (void)hipFree(ctx->irr32);
    if (ctx->colx_table) (void)hipFree(ctx->colx_table);
    if (ctx->tm_thresholds) (void)hipFree(ctx->tm_thresholds);
    if (ctx->tm_code_table) (void)hipFree(ctx->tm_code_table);
    if (ctx->tm_axis) (void)hipFree(ctx->tm_axis);
    for (SahCacheGuard* g : {&ctx->guard_lighting, &ctx->guard_tonemap, &ctx->guard_raster, &ctx->guard_rt})
        if (g->done) (void)hipEventDestroy(g->done);
    for (void* p : ctx->raster.ptr)
        if (p) (void)hipFree(p);
    for (void* p : ctx->rt.ptr)
        if (p) (void)hipFree(p);
    if (ctx->raster.half_to_srgb8) (void)hipFree(ctx->raster.half_to_srgb8);
    if (ctx->raster.host_counters) (void)hipHostFree(ctx->raster.host_counters);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int sah_set_stream(sah_ctx* ctx, void* hip_stream) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    // Switching streams does not order them as a whole: callers that pipeline two work streams switch per frame
    // (androidrenderer_amd/chain.py).  What IS ordered is the context-wide device state a pass re-uses across calls — the LPV gather copy,
    // the fp32 irradiance atlas, the column table and the deferred-pixel lists of sah_lighting, the axis tables of sah_tonemap_ex, the
    // rasteriser's scratch, the ray-tracing structure (SahCacheGuard, ctx.hpp) and the probe-slot table (sah_probe_update): a pass that
    // moves to another stream starts behind its own last use on the old one.
    if ((hipStream_t)hip_stream == ctx->stream && !ctx->own_stream) return SAH_OK;
    const bool drained = ctx->own_stream && ctx->stream;
    if (drained) (void)hipStreamSynchronize(ctx->stream);
    SahCacheGuard* guards[] = {&ctx->guard_lighting, &ctx->guard_tonemap, &ctx->guard_raster, &ctx->guard_rt};
    for (SahCacheGuard* g : guards) HIP_TRY(ctx, sah_guard_leave(ctx, *g, drained));
    if (drained) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return SAH_OK;
}

void* sah_get_stream(sah_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// Testing / tuning hooks (not part of the reference-facing ABI): force the general kernel, force pixels-per-thread.
int sah_debug_set(sah_ctx* ctx, int force_general, int force_ppt) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    ctx->force_general = force_general != 0;
    ctx->force_ppt = force_ppt;
    return SAH_OK;
}

// Debug / analysis hook: number of pixels the last fast-path sah_lighting call sent to the fix-up kernel (synchronises the stream).
int sah_debug_deferred_pixels(sah_ctx* ctx, uint64_t* out) {
    if (!ctx || !out) return SAH_ERR_INVALID_ARGUMENT;
    *out = 0;
    if (!ctx->last_seg_count || !ctx->last_num_segments) return SAH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint16_t> counts(ctx->last_num_segments);
    HIP_TRY(ctx, hipMemcpy(counts.data(), ctx->last_seg_count, counts.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    uint64_t n = 0;
    for (uint16_t c : counts) n += c;  // (the general list only: sky pixels are not counted)
    *out = n;
    return SAH_OK;
}

// Debug / test hook: how many times sah_lighting has rebuilt its gather copy of the LPV volumes (k_lpv_pack) and its fp32 copy of the
// irradiance atlas (k_probe_irr_unpack) in full since the context was made — the change counters of sah_gi exist to keep these at rest.
int sah_debug_copy_rebuilds(sah_ctx* ctx, uint32_t out[2]) {
    if (!ctx || !out) return SAH_ERR_INVALID_ARGUMENT;
    out[0] = ctx->dbg_lpv_packs;
    out[1] = ctx->dbg_irr_unpacks;
    return SAH_OK;
}

int sah_sync(sah_ctx* ctx) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream && ctx->comm_stream != ctx->stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    // a gather of the direct exchange whose wait gave up has skipped its copies: what the caller is about to read is stale or torn
    if (sah_ipc_timed_out(ctx)) return fail(ctx, SAH_ERR_COMM, "direct exchange: a peer did not arrive within 2 s; the gathered rows are not valid");
    return SAH_OK;
}

int sah_lighting(sah_ctx* ctx, const sah_lighting_desc* d) {
    SAH_RANGE();
    using namespace sah;
    if (!ctx || !d) return SAH_ERR_INVALID_ARGUMENT;
    if (!d->gbuffer || !d->lit || !d->view) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "gbuffer, lit and view are required");
    const sah_gbuffer& g = *d->gbuffer;
    const uint32_t W = d->lit->width, H = d->lit->height;
    if (W == 0 || H == 0) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "empty render target");
    if (!plane_ok(d->lit, SAH_FORMAT_R16G16B16A16_SFLOAT, SAH_FORMAT_R16G16B16A16_SFLOAT, W, H))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "lit must be R16G16B16A16_SFLOAT");
    if (!plane_ok(&g.color, SAH_FORMAT_R8G8B8A8_SRGB, SAH_FORMAT_R8G8B8A8_SRGB, W, H) ||
        !plane_ok(&g.normals, SAH_FORMAT_R16G16B16A16_SFLOAT, SAH_FORMAT_R16G16B16A16_SFLOAT, W, H) ||
        !plane_ok(&g.data, SAH_FORMAT_R8G8B8A8_UNORM, SAH_FORMAT_R8G8B8A8_UNORM, W, H) ||
        !plane_ok(&g.emission, SAH_FORMAT_R8G8B8A8_SRGB, SAH_FORMAT_R8G8B8A8_SRGB, W, H) ||
        !plane_ok(&g.depth, SAH_FORMAT_D32_SFLOAT, SAH_FORMAT_R32_SFLOAT, W, H))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "g-buffer planes must match the reference formats and the lit extent");
    uint32_t r0 = d->row_begin, r1 = d->row_end;
    if (r0 == 0 && r1 == 0) r1 = H;
    if (r1 > H || r0 > r1) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "row range [%u,%u) outside image height %u", r0, r1, H);

    const uint32_t sun_mode = d->sun ? d->sun->shadow_mode : SAH_SHADOW_MODE_OFF;
    if (sun_mode > SAH_SHADOW_MODE_RT) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad shadow_mode %u", sun_mode);
    const uint32_t gi_kind = d->gi ? d->gi->kind : SAH_GI_NONE;
    if (gi_kind > SAH_GI_RTGI) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad gi kind %u", gi_kind);
    if ((d->sky || d->lights) && !d->sun) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sun constants are required with sky");

    LightingArgs a;
    memset(&a, 0, sizeof(a));
    a.color = parg(&g.color);
    a.normals = parg(&g.normals);
    a.data = parg(&g.data);
    a.emission = parg(&g.emission);
    a.depth = parg(&g.depth);
    a.lit = parg(d->lit);
    a.width = W;
    a.height = H;
    a.row_begin = r0;
    a.row_end = r1;
    a.flags = d->flags;
    a.res[0] = d->view->render_resolution[0];
    a.res[1] = d->view->render_resolution[1];
    memcpy(a.inv_proj, d->view->inverse_projection, 64);
    memcpy(a.inv_view, d->view->inverse_view, 64);
    for (int i = 0; i < 3; i++) a.view_pos[i] = -d->view->view[12 + i];  // `-view[3].xyz` (directional_light.frag:112)
    a.luts = ctx->luts;
    if (d->sun) {
        const float neg[3] = {-d->sun->direction_and_tan_size[0], -d->sun->direction_and_tan_size[1], -d->sun->direction_and_tan_size[2]};
        normalize3(neg, a.sun_L);
        for (int i = 0; i < 3; i++) a.sun_color[i] = d->sun->color[i];
    }
    bool vec4ok = (W % 4 == 0) && vec_ok(&g.color, 16) && vec_ok(&g.normals, 16) && vec_ok(&g.data, 16) && vec_ok(&g.emission, 16) &&
                  vec_ok(&g.depth, 16) && vec_ok(d->lit, 16);

    if (gi_kind == SAH_GI_LPV && d->ao && d->ao->ptr) {
        if (!plane_ok(d->ao, SAH_FORMAT_R32_SFLOAT, SAH_FORMAT_R32_SFLOAT, W, H)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "ao must be R32_SFLOAT");
        a.ao = parg(d->ao);
        a.has_ao = 1;
        vec4ok = vec4ok && vec_ok(d->ao, 16);
    }
    if (sun_mode == SAH_SHADOW_MODE_RT && d->shadow_mask && d->shadow_mask->ptr) {
        if (!plane_ok(d->shadow_mask, SAH_FORMAT_R32_SFLOAT, SAH_FORMAT_R32_SFLOAT, W, H))
            return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "shadow_mask must be R32_SFLOAT");
        a.shadow_mask = parg(d->shadow_mask);
        a.has_mask = 1;
        vec4ok = vec4ok && vec_ok(d->shadow_mask, 16);
    }

    CsmArgs csm;
    memset(&csm, 0, sizeof(csm));
    if (sun_mode == SAH_SHADOW_MODE_CSM) {
        if (d->shadowmap && d->shadowmap->ptr) {
            if (d->shadowmap->format != SAH_FORMAT_D16_UNORM && d->shadowmap->format != SAH_FORMAT_D32_SFLOAT)
                return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "shadowmap must be D16_UNORM or D32_SFLOAT");
            if (d->shadowmap->depth < 4) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "shadowmap needs 4 layers");
            csm.shadowmap = varg(*d->shadowmap);
            csm.is_d16 = d->shadowmap->format == SAH_FORMAT_D16_UNORM;
            csm.d16_recip = 1.0f / 65535.0f;
            static const bool recip_ok = [] {  // exhaustive check of the reciprocal sequence used by shadow_pcf()
                const float y = 1.0f / 65535.0f;
                for (uint32_t i = 0; i < 65536; i++) {
                    const float v = (float)i;
                    const float q = v * y;
                    const float q2 = std::fmaf(std::fmaf(-q, 65535.0f, v), y, q);
                    if (q2 != v / 65535.0f) return false;
                }
                return true;
            }();
            csm.d16_recip_ok = recip_ok;
        }
        for (int c = 0; c < 4; c++) {
            csm.splits[c] = d->sun->data[c][0];
            // biasMat * cascade_matrices[c] (directional_light.frag:55-65), evaluated as the matrix product it is
            const float* M = d->sun->cascade_matrices[c];
            for (int col = 0; col < 4; col++) {
                const float* m = M + col * 4;
                csm.biased[c][col * 4 + 0] = ((0.5f * m[0] + 0.0f * m[1]) + 0.0f * m[2]) + 0.5f * m[3];
                csm.biased[c][col * 4 + 1] = ((0.0f * m[0] + 0.5f * m[1]) + 0.0f * m[2]) + 0.5f * m[3];
                csm.biased[c][col * 4 + 2] = ((0.0f * m[0] + 0.0f * m[1]) + 1.0f * m[2]) + 0.0f * m[3];
                csm.biased[c][col * 4 + 3] = ((0.0f * m[0] + 0.0f * m[1]) + 0.0f * m[2]) + 1.0f * m[3];
            }
        }
    }

    LpvArgs lpv;
    memset(&lpv, 0, sizeof(lpv));
    CacheArgs cache;
    memset(&cache, 0, sizeof(cache));
    RtgiArgs rtgi;
    memset(&rtgi, 0, sizeof(rtgi));
    if (gi_kind == SAH_GI_LPV) {
        const sah_gi& gi = *d->gi;
        if (gi.lpv_num_cascades == 0 || gi.lpv_num_cascades > 4 || !gi.lpv_cascades)
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "LPV needs 1..4 cascades and their matrices");
        const sah_volume* vols[3] = {&gi.lpv_red, &gi.lpv_green, &gi.lpv_blue};
        for (const sah_volume* v : vols) {
            if (!v->ptr || v->format != SAH_FORMAT_R16G16B16A16_SFLOAT) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "LPV volumes must be RGBA16F");
            if ((uint64_t)v->row_pitch_bytes < (uint64_t)v->width * 8 || (uint64_t)v->slice_pitch_bytes < (uint64_t)v->row_pitch_bytes * v->height)
                return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "LPV volume pitches too small");
        }
        lpv.red = varg(gi.lpv_red);
        lpv.green = varg(gi.lpv_green);
        lpv.blue = varg(gi.lpv_blue);
        for (uint32_t c = 0; c < gi.lpv_num_cascades; c++) memcpy(lpv.world_to_cascade[c], gi.lpv_cascades[c].world_to_cascade, 64);
        lpv.num_cascades = gi.lpv_num_cascades;
        lpv.num_cascades_f = (float)gi.lpv_num_cascades;
        lpv.exposure = gi.lpv_exposure;
    } else if (gi_kind == SAH_GI_CACHE) {
        const sah_gi& gi = *d->gi;
        if (!gi.probe_irradiance.ptr || gi.probe_irradiance.format != SAH_FORMAT_B10G11R11_UFLOAT_PACK32 || !gi.probe_depth.ptr ||
            gi.probe_depth.format != SAH_FORMAT_R16G16_SFLOAT || !gi.probe_validity.ptr || gi.probe_validity.format != SAH_FORMAT_R8_UNORM)
            return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "probe atlases must be B10G11R11 / R16G16F / R8_UNORM arrays");
        cache.irradiance = varg(gi.probe_irradiance);
        cache.depth = varg(gi.probe_depth);
        cache.validity = varg(gi.probe_validity);
        for (int c = 0; c < 4; c++) {
            const float ext[3] = {32.f, 8.f, 32.f};
            for (int i = 0; i < 3; i++) {
                cache.cascade_min[c][i] = gi.probe_cascades[c].min[i];
                cache.cascade_max[c][i] = gi.probe_cascades[c].min[i] + ext[i] * gi.probe_cascades[c].probe_spacing;
            }
            cache.spacing[c] = gi.probe_cascades[c].probe_spacing;
        }
        cache.spacing_pow2 = 1;
        for (int c = 0; c < 4; c++) {
            int e = 0;
            const float sp = cache.spacing[c], m = frexpf(sp, &e);
            // 2^-60 .. 2^60: neither the reciprocal nor a quotient of a finite coordinate's difference can leave the normal range unevenly
            cache.spacing_pow2 = cache.spacing_pow2 && m == 0.5f && e > -60 && e < 60;
            cache.inv_spacing[c] = 1.0f / sp;
        }
        cache.probe_size[0] = gi.probe_size[0];
        cache.probe_size[1] = gi.probe_size[1];
        for (int i = 0; i < 2; i++) cache.inv_tex[i] = 1.0f / (((float)gi.probe_size[i] + 2.0f) * 32.0f);
        cache.debug_mode = gi.cache_debug_mode;
        auto bytes = [](const sah_volume& v) { return (uint64_t)v.slice_pitch_bytes * v.depth; };
        cache.hot_ok = bytes(gi.probe_irradiance) < (1ull << 32) && bytes(gi.probe_depth) < (1ull << 32) && bytes(gi.probe_validity) < (1ull << 32) &&
                       // div_const() is proven bit-exact for probe indices below 32 only (tools/microbench/div_const_check.c); the
                       // reference's grid is exactly 32 x 32 x 32 (irradiance_cache.cpp:94-183)
                       gi.probe_validity.width <= 32 && gi.probe_validity.height <= 32 && gi.probe_validity.depth <= 32 &&
                       gi.probe_size[0] >= 1 && gi.probe_size[0] <= 30 && gi.probe_size[1] >= 1 && gi.probe_size[1] <= 30 &&
                       // atlases exactly 32 blocks wide, as get_probe_uv assumes: texcoords then never reach the REPEAT seam
                       gi.probe_irradiance.width == 32u * (gi.probe_size[0] + 2u) && gi.probe_irradiance.height == 32u * (gi.probe_size[1] + 2u) &&
                       gi.probe_depth.width == 32u * 12u && gi.probe_depth.height == 32u * 12u &&
                       // the widened copy is addressed with 4 x the atlas's 32-bit offsets
                       bytes(gi.probe_irradiance) < (1ull << 30) && gi.probe_irradiance.row_pitch_bytes % 4 == 0 && gi.probe_irradiance.slice_pitch_bytes % 4 == 0 &&
                       ((uintptr_t)gi.probe_irradiance.ptr % 4) == 0;
        if (cache.hot_ok) {
            const size_t need = 4 * (size_t)bytes(gi.probe_irradiance);
            if (ctx->irr32_bytes < need) {
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->irr32) (void)hipFree(ctx->irr32);
                ctx->irr32 = nullptr;
                ctx->irr32_bytes = 0;
                if (hipMalloc((void**)&ctx->irr32, need) == hipSuccess) {
                    ctx->irr32_bytes = need;
                    ctx->irr32_generation = 0;
                    ctx->cache_epoch++;
                } else {  // no room for the widened copy: the general gather needs none
                    (void)hipGetLastError();
                    ctx->irr32 = nullptr;
                    cache.hot_ok = 0;
                }
            }
        }
        if (cache.hot_ok) {
            const bool reuse = gi.probe_generation != 0 && gi.probe_generation == ctx->irr32_generation &&
                               same_volume(cache.irradiance, ctx->irr32_source);
            if (!reuse) {
                if (gi.probe_generation != 0) ctx->cache_epoch++;  // (with 0 every call rebuilds: the same launches every time)
                HIP_TRY(ctx, launch_probe_irr_unpack(cache.irradiance, ctx->irr32, ctx->stream));
                ctx->dbg_irr_unpacks++;
                ctx->irr32_generation = gi.probe_generation;
                ctx->irr32_source = cache.irradiance;
            }
            cache.irr32 = ctx->irr32;
        }
    } else if (gi_kind == SAH_GI_RTGI) {
        const sah_gi& gi = *d->gi;
        if (!plane_ok(&gi.ray_buffer, SAH_FORMAT_R16G16B16A16_SFLOAT, SAH_FORMAT_R16G16B16A16_SFLOAT, W, H) ||
            !plane_ok(&gi.ray_irradiance, SAH_FORMAT_R16G16B16A16_SFLOAT, SAH_FORMAT_R16G16B16A16_SFLOAT, W, H))
            return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "ray_buffer / ray_irradiance must be RGBA16F at render resolution");
        rtgi.ray_buffer = parg(&gi.ray_buffer);
        rtgi.ray_irradiance = parg(&gi.ray_irradiance);
        if (gi.num_extra_rays) {
            if (!gi.noise.ptr || gi.noise.format != SAH_FORMAT_R8G8B8A8_UNORM || gi.noise.width < 128 || gi.noise.height < 128)
                return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "noise must be R8G8B8A8_UNORM, at least 128x128");
            rtgi.noise = parg(&gi.noise);
            rtgi.noise_w = gi.noise.width;
            rtgi.noise_h = gi.noise.height;
        }
        rtgi.num_extra_rays = gi.num_extra_rays;
        rtgi.extra_ray_radius = gi.extra_ray_radius;
    }

    SkyArgs sky;
    memset(&sky, 0, sizeof(sky));
    if (d->sky) {
        // the sky fill's sun direction: -normalize(direction) (sky_unified.slang:199)
        const float dirn[3] = {d->sun->direction_and_tan_size[0], d->sun->direction_and_tan_size[1], d->sun->direction_and_tan_size[2]};
        float nd[3];
        normalize3(dirn, nd);
        const float sun_dir[3] = {-nd[0], -nd[1], -nd[2]};
        if (!fill_sky_args(*d->sky, sun_dir, &sky)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "sky LUTs must be RGBA16F");
    }

    // pixels per thread: 4 (16 B/lane plane loads) whenever pitches and width allow (measured on MI355X, DESIGN.md §7: with the
    // packed LPV gather the 4-pixel body fits 106 VGPRs without spills and beats 2 px/thread by ~4 %)
    int ppt = vec4ok ? 4 : 1;
    // ... unless the launch would then be too few workgroups to keep the chip's 1,024 workgroup slots (256 CUs x 4) busy for more than a
    // round and a half: a 1280 x 720 frame, or one rank's 270 rows of a 4K frame, is ~900 workgroups of 1,024 pixels, and the kernel then takes
    // as long as one workgroup lives.  Fewer pixels per thread: more and shorter workgroups (1280 x 720, RT sun only: 0.0356 ms at 4, 0.0233 at
    // 2, 0.0222 at 1; 1920 x 1080 is equal at all three; at 4K 4 wins by 20 %).  The three bodies produce the same bits (tests/test_lighting_gpu.py).
    while (ppt > 1 && (uint64_t)(W / (uint32_t)ppt) * (r1 - r0) < 1536ull * 256ull) ppt /= 2;
    if (ctx->force_ppt == 1 || ctx->force_ppt == 2 || ctx->force_ppt == 4) {
        if (ctx->force_ppt == 1 || (vec4ok && W % ctx->force_ppt == 0)) ppt = ctx->force_ppt;
    }
    if (d->lights && d->lights->count) {
        if (!d->lights->lights) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "light list pointer is null");
        a.lights = d->lights->lights;
        a.num_lights = d->lights->count;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_lighting));
    FastArgs fast;
    memset(&fast, 0, sizeof(fast));
    const bool fast_kind = (gi_kind == SAH_GI_NONE || gi_kind == SAH_GI_LPV) && a.num_lights == 0;
    const bool use_fast = fast_kind && !ctx->force_general && detect_fast_path(d, sun_mode, gi_kind, csm, &fast);
    // the tiled kernel (light list, cache / RTGI overlays) borrows the fast kernel's geometry and CSM sun when the uniform blocks allow
    // (its LPV overlay, if any, stays the general one: the LPV part of the check is skipped)
    static const bool no_tiled_fast_geom = getenv("SAH_TILED_GENERAL_GEOMETRY") != nullptr;  // A/B switch (tools/ab.sh)
    const bool tiled_fast_geom = !fast_kind && !ctx->force_general && !no_tiled_fast_geom && detect_fast_path(d, sun_mode, SAH_GI_NONE, csm, &fast);
    // ... and, round 6, the fast kernel's LPV overlay (gather from the packed copy) where the LPV part of the check holds as well: a light list
    // over an LPV frame (configs[4]) no longer pays the general overlay's nine trilinear fetches per pixel
    static const bool no_tiled_fast_lpv = getenv("SAH_TILED_GENERAL_LPV") != nullptr;  // A/B switch
    const bool tiled_fast_lpv = tiled_fast_geom && gi_kind == SAH_GI_LPV && !no_tiled_fast_lpv && ctx->state && detect_fast_path(d, sun_mode, SAH_GI_LPV, csm, &fast);
    fast.lpv_fast = tiled_fast_lpv ? 1u : 0u;
    // the gather copy of the LPV volumes for a kernel that reads it: rebuilt by k_lpv_pack (in front of the kernel: lighting.hip) unless the caller's
    // change counter says it stands (SAH_GENERATION_TRACKED: the last step of sah_lpv_propagate has written it — api_post.cpp)
    auto prepare_lpv_copy = [&]() -> int {
        const SahLpvPackLayout pk = sah_lpv_pack_layout(lpv.red.width, lpv.red.height, lpv.red.depth);
        if (pk.total >= (1ull << 32)) return fail(ctx, SAH_ERR_UNSUPPORTED, "LPV volumes too large for the packed gather copy");
        HIP_TRY(ctx, sah_lpv_pack_reserve(ctx, pk.total));
        fast.lpv_packed = ctx->lpv_packed;
        fast.pk_row_pitch = pk.row_pitch;
        fast.pk_slice_pitch = pk.slice_pitch;
        // the copy of the previous call is kept when the caller's change counter says the volumes are the ones it was made from
        const sah::VolumeArg src[3] = {lpv.red, lpv.green, lpv.blue};
        const uint32_t gen = d->gi->lpv_generation;
        const bool reuse = gen != 0 && gen == ctx->lpv_pack_generation && same_volume(src[0], ctx->lpv_pack_source[0]) &&
                           same_volume(src[1], ctx->lpv_pack_source[1]) && same_volume(src[2], ctx->lpv_pack_source[2]);
        fast.repack = reuse ? 0u : 1u;
        if (!reuse) {
            if (gen != 0) ctx->cache_epoch++;  // (with 0 every call rebuilds: the same launches every time)
            ctx->dbg_lpv_packs++;
            ctx->lpv_pack_generation = gen;
            for (int i = 0; i < 3; i++) ctx->lpv_pack_source[i] = src[i];
            sah_lpv_pack_written_by_pack(ctx, lpv.red.width, lpv.red.height, lpv.red.depth);
        }
        return SAH_OK;
    };
    if (tiled_fast_lpv && r1 > r0) {
        if (const int rc = prepare_lpv_copy(); rc != SAH_OK) return rc;
        fast.state = ctx->state;
    }
    if (use_fast) {
        // deferred-pixel segments (params.hpp): one per wave of the fast kernel, 64 * ppt byte codes + a 16-bit count each
        const uint64_t groups = (uint64_t)(W / (uint32_t)ppt) * (r1 - r0);
        const uint32_t nseg = (uint32_t)((groups + 63) / 64);
        const uint32_t seg_stride = 64u * (uint32_t)ppt;
        const size_t codes_bytes = ((size_t)nseg * seg_stride + 255) & ~(size_t)255;
        const size_t need = codes_bytes + 2 * (size_t)nseg * sizeof(uint16_t) + 256;  // two counts per segment: general (front), sky (back)
        if (ctx->list_bytes < need) {  // grow-only workspace
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->list) (void)hipFree(ctx->list);
            ctx->list = nullptr;
            ctx->list_bytes = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->list, need));
            ctx->list_bytes = need;
            ctx->cache_epoch++;
        }
        fast.seg_list = (uint8_t*)ctx->list;
        fast.seg_count = (uint16_t*)((uint8_t*)ctx->list + codes_bytes);
        fast.num_segments = nseg;
        fast.seg_stride = seg_stride;
        if (gi_kind == SAH_GI_LPV) {
            if (const int rc = prepare_lpv_copy(); rc != SAH_OK) return rc;
        }
        fast.sky_enabled = sky.enabled;
        {   // The sky workgroups of the fast kernel (lighting.hip): they LEAD the grid, one per `sky_ratio` surface workgroups.  Leading: a thread
            // of a sky workgroup walks sky_ratio * ppt pixels one after the other (~900 instructions per sky pixel), and a walk that starts with the
            // launch's last workgroups is the launch's tail (the atrium's sky is its last rows: 1280 x 720 22.8 -> 17.8 us, 1920 x 1080 55.5 -> 50.6,
            // 4K CSM only 121.7 -> 113.8 leading instead of interleaved 1 : 4).  How many: about 512 of them — half a round of the chip's 1,024
            // workgroup slots, two sky waves on every SIMD of a sky-heavy frame; more only hold slots in front of the surface workgroups to find
            // nothing (4K: 2,025 sky workgroups 0.1649 ms, 506 0.1615), fewer make the walk the critical path of a short launch (1920 x 1080: 127
            // sky workgroups 0.058 ms against 0.0507 with 506).  tools/experiments/r6/README.md §2.
            static const int env_ratio = getenv("SAH_SKY_RATIO") ? atoi(getenv("SAH_SKY_RATIO")) : 0;  // experiments (tools/experiments/r6)
            static const int env_interleaved = getenv("SAH_SKY_INTERLEAVED") ? atoi(getenv("SAH_SKY_INTERLEAVED")) : 0;
            const uint64_t blocks = ((uint64_t)(W / (uint32_t)ppt) * (r1 - r0) + 255) / 256;
            const uint64_t want = (blocks + 511) / 512;
            fast.sky_ratio = (uint32_t)(want < 4 ? 4 : (want > 32 ? 32 : want));
            if (env_ratio > 0) fast.sky_ratio = (uint32_t)env_ratio;
            fast.sky_first = env_interleaved ? 0u : 1u;  // (lighting.hip's launcher turns the flag into the number of sky workgroups)
            if (env_interleaved && env_ratio <= 0) fast.sky_ratio = 4;  // (rounds 2-5: every fifth workgroup)
        }
        {  // thread index -> (row, group in row) by a multiply-high: exact while gid * groups_per_row < 2^32 (magic = floor(2^32 / d) + 1)
            const uint64_t gpr = W / (uint32_t)ppt, threads = (gpr * (r1 - r0) + 255) / 256 * 256;
            fast.row_magic = (gpr >= 2 && threads * gpr < (1ull << 32)) ? (uint32_t)((1ull << 32) / gpr) + 1u : 0u;
        }
        fast.state = ctx->state;
        fast.hint_slot = ctx->hint_slot;  // (FrameState::deferred_hint: consecutive calls take the two words in turn)
        ctx->hint_slot ^= 1u;
    }
    // per-column numerators of the view-space x and per-row ones of y (IEEE divides per thread / per pixel otherwise; the tiled kernel reads
    // them too): a function of the extent, the render resolution
    // and two entries of the inverse projection — rebuilt when one of them changes
    if ((use_fast && ppt == 4 && (sun_mode != SAH_SHADOW_MODE_OFF || gi_kind == SAH_GI_LPV)) || tiled_fast_geom) {
        const float key[7] = {a.res[0], fast.p0, fast.p12, a.res[1], fast.p5, fast.p13, (float)H};
        const uint32_t stride = (W + 63u) & ~63u, row_stride = (H + 63u) & ~63u;
        const uint32_t need = 2 * stride + 2 * row_stride;
        if (ctx->colx_capacity < need) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->colx_table) (void)hipFree(ctx->colx_table);
            ctx->colx_table = nullptr;
            ctx->colx_capacity = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->colx_table, (size_t)need * sizeof(float)));
            ctx->colx_capacity = need;
            ctx->cache_epoch++;
            ctx->colx_width = 0;
        }
        if (ctx->colx_width != W || memcmp(key, ctx->colx_key, sizeof(key)) != 0) {
            HIP_TRY(ctx, launch_colx_table(a, fast, ctx->colx_table, stride, row_stride, ctx->stream));
            ctx->cache_epoch++;
            ctx->colx_width = W;
            memcpy(ctx->colx_key, key, sizeof(key));
        }
        fast.colx_tab = ctx->colx_table;
        fast.colx_stride = stride;
        fast.rowy_stride = row_stride;
    }
    HIP_TRY(ctx, launch_lighting(a, csm, lpv, cache, rtgi, sky, (use_fast || tiled_fast_geom) ? &fast : nullptr, (int)sun_mode, (int)gi_kind, ppt,
                                 (d->flags & SAH_LIGHTING_BRUTE_FORCE_LIGHTS) != 0, ctx->stream));
    if (use_fast) {
        ctx->last_seg_count = fast.seg_count;
        ctx->last_num_segments = fast.num_segments;
    } else {
        ctx->last_seg_count = nullptr;
        ctx->last_num_segments = 0;
    }
    return SAH_OK;
}

}  // extern "C"
