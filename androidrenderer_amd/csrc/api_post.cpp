// C ABI: post chain, LPV maintenance and the RCCL row all-gather.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "../../include/sah_hip.h"
#include "ctx.hpp"
#include "post_args.hpp"

namespace sah {
hipError_t launch_copy_scene(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                             uint32_t row_end, hipStream_t st);
hipError_t launch_bloom_downsample(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                                   uint32_t row_end, hipStream_t st);
bool launch_copy_bloom_mip0(const PlaneArg& lit, uint32_t lw, uint32_t lh, const PlaneArg& aa, uint32_t aw, uint32_t ah, const PlaneArg& mip0, uint32_t mw, uint32_t mh,
                            uint32_t mip_row_begin, uint32_t mip_row_end, uint32_t aa_row_begin, uint32_t aa_row_end, hipStream_t st, hipError_t* err);
bool launch_bloom_pair(const PlaneArg& s, uint32_t sw, uint32_t sh, const PlaneArg& a, uint32_t aw, uint32_t ah, const PlaneArg& b, uint32_t bw, uint32_t bh,
                       hipStream_t st, hipError_t* err);
hipError_t launch_tonemap(const TonemapArgs& t, hipStream_t st);
hipError_t launch_tonemap_tol(const TonemapArgs& t, hipStream_t st);
hipError_t launch_tonemap_axis_tables(const TonemapArgs& t, TmAxis* out, hipStream_t st);
hipError_t launch_lpv_clear(const VolumeArg* vols, int n, uint32_t num_cascades, hipStream_t st);
hipError_t launch_lpv_propagate(const VolumeArg src[3], const VolumeArg dst[3], uint32_t num_cascades, const LpvPackEmit* emit, int mode, hipStream_t st);
hipError_t launch_lpv_build_tables(hipStream_t st, bool* hot_structure);
hipError_t launch_sky_luts(const PlaneArg& transmittance, const PlaneArg& multiscattering, const PlaneArg& sky_view, const float light_vector[3], hipStream_t st);
hipError_t launch_fill_r32f(const PlaneArg& dst, uint32_t w, uint32_t h, float value, hipStream_t st);
hipError_t launch_probe_copy(const ProbeAtlasArgs& src, const ProbeAtlasArgs& dst, const float movement[4][3], hipStream_t st);
hipError_t launch_probe_irr_unpack_probes(const VolumeArg& src, uint8_t* dst, const uint32_t* probes, uint32_t num_probes, hipStream_t st);  // lighting_tiled.hip
hipError_t launch_probe_update(const ProbeAtlasArgs& atl, const VolumeArg& trace, const uint32_t* probes, uint32_t num_probes, uint32_t* slots,
                               hipStream_t st);
}  // namespace sah

namespace {
bool rgba16f_ok(const sah_plane* p) {
    return p && p->ptr && p->format == SAH_FORMAT_R16G16B16A16_SFLOAT && p->width && p->height &&
           (uint64_t)p->row_pitch_bytes >= (uint64_t)p->width * 8 && ((uintptr_t)p->ptr % 8) == 0 && (p->row_pitch_bytes % 8) == 0;
}
// Output code of the tonemap tail for one channel value x = colour * luma/(luma+1):
// pow(x, 1/2.2) (fp32 result of the fp64 libm value), then the sRGB OETF the swap chain applies, then UNORM8
// (scene_upsample.frag:66-72 + hardware sRGB write; DESIGN.md "Numerics").  Monotone non-decreasing in x.
uint32_t tonemap_code(float x) {
    if (!(x > 0.0f)) return 0u;
    const float g = (float)std::pow((double)x, (double)(1.f / 2.2f));
    if (!(g > 0.0f)) return 0u;
    if (g >= 1.0f) return 255u;
    const double d = (double)g;
    const float s = (float)((d <= 0.0031308) ? 12.92 * d : 1.055 * std::pow(d, 1.0 / 2.4) - 0.055);
    if (!(s > 0.0f)) return 0u;
    if (s >= 1.0f) return 255u;
    return (uint32_t)(s * 255.0f + 0.5f);
}
// thr[k] = smallest positive float whose code is >= k, found by bisection on the bit pattern (positive floats order like
// their bits); thr[0] is never read.
void build_tonemap_thresholds(float thr[256]) {
    thr[0] = 0.0f;
    for (uint32_t k = 1; k < 256; k++) {
        uint32_t lo = 0u, hi = 0x7f800000u;  // code(+0) = 0 < k <= 255 = code(+inf)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            float f;
            memcpy(&f, &mid, 4);
            if (tonemap_code(f) >= k) hi = mid; else lo = mid;
        }
        memcpy(&thr[k], &hi, 4);
    }
}
// First-level table of the device's code search: bucket b holds the positive floats whose bit pattern >> kTmBucketShift equals
// base + b (16 buckets per octave, from thr[1] to thr[255]); first[b] = min(code(first float of the bucket), 252).  A bucket spans at
// most three codes (checked), so code(x) = first[b] + [x >= thr[first + 1]] + [x >= thr[first + 2]] + [x >= thr[first + 3]].
bool build_tonemap_buckets(const float thr[256], uint8_t first[sah::kTmMaxBuckets], uint32_t* base, uint32_t* count) {
    uint32_t tb[256];
    memcpy(tb, thr, sizeof(tb));
    const uint32_t b0 = tb[1] >> sah::kTmBucketShift, b1 = tb[255] >> sah::kTmBucketShift;
    const uint32_t n = b1 - b0 + 1;
    if (n > sah::kTmMaxBuckets) return false;
    auto code_of_bits = [&](uint32_t bits) {  // number of thresholds <= the float
        uint32_t c = 0;
        for (uint32_t k = 1; k < 256; k++) c += tb[k] <= bits ? 1u : 0u;
        return c;
    };
    for (uint32_t b = 0; b < n; b++) {
        const uint32_t start = (b0 + b) << sah::kTmBucketShift, end = ((b0 + b + 1) << sah::kTmBucketShift) - 1u;
        // bucket 0 also takes everything below thr[1] (clamped there by the kernel): code 0
        const uint32_t lo = b == 0 ? 0u : code_of_bits(start);
        const uint32_t f = lo < 252u ? lo : 252u;
        if (code_of_bits(end) - f > 3u) return false;
        first[b] = (uint8_t)f;
    }
    for (uint32_t b = n; b < sah::kTmMaxBuckets; b++) first[b] = 252;
    *base = b0;
    *count = n;
    return true;
}
bool lpv_vol_ok(const sah_volume* v) {
    return v && v->ptr && v->format == SAH_FORMAT_R16G16B16A16_SFLOAT && (uint64_t)v->row_pitch_bytes >= (uint64_t)v->width * 8 &&
           (uint64_t)v->slice_pitch_bytes >= (uint64_t)v->row_pitch_bytes * v->height && ((uintptr_t)v->ptr % 8) == 0 &&
           (v->row_pitch_bytes % 8) == 0 && (v->slice_pitch_bytes % 8) == 0;
}
}  // namespace

int sah_ipc_find(const sah_ctx* ctx, const void* ptr, uint64_t bytes);
int sah_ipc_gather(sah_ctx* ctx, uint32_t id, uint8_t* buffer, uint64_t bytes_per_rank, bool reversed, hipStream_t st);

bool sah_ipc_timed_out(const sah_ctx* ctx);  // api_ipc.cpp

extern "C" {

int sah_copy_scene_rows(sah_ctx* ctx, const sah_plane* lit, const sah_plane* out, uint32_t row_begin, uint32_t row_end) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!rgba16f_ok(lit) || !rgba16f_ok(out)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "copy_scene needs RGBA16F planes");
    if (row_begin == 0 && row_end == 0) row_end = out->height;
    if (row_end > out->height || row_begin > row_end) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad row range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah::launch_copy_scene(parg(lit), lit->width, lit->height, parg(out), out->width, out->height, row_begin, row_end, ctx->stream));
    return SAH_OK;
}

int sah_copy_scene(sah_ctx* ctx, const sah_plane* lit, const sah_plane* out) { return sah_copy_scene_rows(ctx, lit, out, 0, 0); }

static int bloom_range(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom, uint32_t first_mip, uint32_t last_mip, uint32_t row_begin,
                       uint32_t row_end) {
    if ((first_mip == 0 && !rgba16f_ok(scene)) || !bloom || bloom->num_mips == 0 || bloom->num_mips > SAH_MAX_BLOOM_MIPS)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bloom needs an RGBA16F scene and 1..%d mips", SAH_MAX_BLOOM_MIPS);
    for (uint32_t m = 0; m < bloom->num_mips; m++)
        if (!rgba16f_ok(&bloom->mips[m])) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "bloom mip %u must be RGBA16F", m);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (uint32_t m = first_mip; m <= last_mip && m < bloom->num_mips; m++) {  // bloomer.cpp:50-151: scene -> mip0, mip i -> mip i+1
        const sah_plane* src = m == 0 ? scene : &bloom->mips[m - 1];
        const sah_plane* dst = &bloom->mips[m];
        const uint32_t r0 = m == first_mip ? row_begin : 0u, r1 = m == first_mip ? row_end : dst->height;
        // the small mips of the chain, two per launch (post.hip: k_bloom_pair): whole mips only, and only where the texels of the first
        // one that neighbouring tiles compute twice cost less than the launch they save (mip 2 and beyond of a 4K chain)
        const bool whole = r0 == 0 && r1 == dst->height;
        if (whole && m + 1 <= last_mip && m + 1 < bloom->num_mips && (uint64_t)dst->width * dst->height <= (1u << 18)) {
            const sah_plane* nxt = &bloom->mips[m + 1];
            hipError_t e = hipSuccess;
            if (sah::launch_bloom_pair(parg(src), src->width, src->height, parg(dst), dst->width, dst->height, parg(nxt), nxt->width, nxt->height, ctx->stream, &e)) {
                HIP_TRY(ctx, e);
                m++;
                continue;
            }
        }
        HIP_TRY(ctx, sah::launch_bloom_downsample(parg(src), src->width, src->height, parg(dst), dst->width, dst->height, r0, r1, ctx->stream));
    }
    return SAH_OK;
}

// "Copy scene" + the first bloom dispatch in one pass over lit_scene (post.hip: k_copy_bloom_mip0).  Rows of `antialiased` that the fused
// launch cannot own (farther than 3 rows from the rows of mip 0's sources) are copied by plain launches; extents that do not suit it take
// the two passes one after the other.
int sah_copy_scene_bloom_mip0_rows(sah_ctx* ctx, const sah_plane* lit, const sah_plane* out, const sah_mipchain* bloom, uint32_t aa_row_begin, uint32_t aa_row_end,
                                   uint32_t mip_row_begin, uint32_t mip_row_end) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!rgba16f_ok(lit) || !rgba16f_ok(out)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "copy_scene needs RGBA16F planes");
    if (!bloom || bloom->num_mips == 0 || bloom->num_mips > SAH_MAX_BLOOM_MIPS || !rgba16f_ok(&bloom->mips[0]))
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bloom needs 1..%d RGBA16F mips", SAH_MAX_BLOOM_MIPS);
    const sah_plane* m0 = &bloom->mips[0];
    if (aa_row_begin == 0 && aa_row_end == 0) aa_row_end = out->height;
    if (mip_row_begin == 0 && mip_row_end == 0) mip_row_end = m0->height;
    if (aa_row_end > out->height || aa_row_begin > aa_row_end || mip_row_end > m0->height || mip_row_begin > mip_row_end)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad row range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (mip_row_end > mip_row_begin && aa_row_end > aa_row_begin) {
        // what the fused launch can own: from 3 rows above the first source row of the mip rows to 3 rows below the last
        const uint32_t lo = (uint32_t)std::max<int64_t>(std::max<int64_t>((int64_t)2 * mip_row_begin - 3, (int64_t)aa_row_begin), 0);
        const uint32_t hi = (uint32_t)std::min<uint64_t>((uint64_t)2 * mip_row_end + 2, aa_row_end);
        if (hi > lo) {
            hipError_t e = hipSuccess;
            if (sah::launch_copy_bloom_mip0(parg(lit), lit->width, lit->height, parg(out), out->width, out->height, parg(m0), m0->width, m0->height, mip_row_begin,
                                            mip_row_end, lo, hi, ctx->stream, &e)) {
                HIP_TRY(ctx, e);
                if (lo > aa_row_begin) HIP_TRY(ctx, sah::launch_copy_scene(parg(lit), lit->width, lit->height, parg(out), out->width, out->height, aa_row_begin, lo, ctx->stream));
                if (aa_row_end > hi) HIP_TRY(ctx, sah::launch_copy_scene(parg(lit), lit->width, lit->height, parg(out), out->width, out->height, hi, aa_row_end, ctx->stream));
                return SAH_OK;
            }
        }
    }
    // the two passes
    if (aa_row_end > aa_row_begin)
        HIP_TRY(ctx, sah::launch_copy_scene(parg(lit), lit->width, lit->height, parg(out), out->width, out->height, aa_row_begin, aa_row_end, ctx->stream));
    if (mip_row_end > mip_row_begin)
        HIP_TRY(ctx, sah::launch_bloom_downsample(parg(out), out->width, out->height, parg(m0), m0->width, m0->height, mip_row_begin, mip_row_end, ctx->stream));
    return SAH_OK;
}

int sah_bloom(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!bloom || bloom->num_mips == 0) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bloom needs 1..%d mips", SAH_MAX_BLOOM_MIPS);
    return bloom_range(ctx, scene, bloom, 0, bloom->num_mips - 1, 0, bloom->mips[0].height);
}

int sah_bloom_mip0_rows(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom, uint32_t row_begin, uint32_t row_end) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!bloom || bloom->num_mips == 0) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bloom needs 1..%d mips", SAH_MAX_BLOOM_MIPS);
    if (row_end > bloom->mips[0].height || row_begin > row_end) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad mip 0 row range");
    return bloom_range(ctx, scene, bloom, 0, 0, row_begin, row_end);
}

int sah_bloom_from_mip0(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!bloom || bloom->num_mips == 0) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bloom needs 1..%d mips", SAH_MAX_BLOOM_MIPS);
    if (bloom->num_mips == 1) return SAH_OK;
    return bloom_range(ctx, scene, bloom, 1, bloom->num_mips - 1, 0, bloom->mips[1].height);
}

int sah_bloom_mip_rows(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom, uint32_t mip, uint32_t row_begin, uint32_t row_end) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!bloom || bloom->num_mips == 0 || bloom->num_mips > SAH_MAX_BLOOM_MIPS || mip >= bloom->num_mips)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bloom needs 1..%d mips and a mip index below their number", SAH_MAX_BLOOM_MIPS);
    if (row_end > bloom->mips[mip].height || row_begin > row_end) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad row range");
    return bloom_range(ctx, scene, bloom, mip, mip, row_begin, row_end);
}

int sah_bloom_from_mip(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom, uint32_t mip) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!bloom || bloom->num_mips == 0 || bloom->num_mips > SAH_MAX_BLOOM_MIPS || mip >= bloom->num_mips)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bloom needs 1..%d mips and a mip index below their number", SAH_MAX_BLOOM_MIPS);
    if (mip + 1 >= bloom->num_mips) return SAH_OK;
    return bloom_range(ctx, scene, bloom, mip + 1, bloom->num_mips - 1, 0, bloom->mips[mip + 1].height);
}

int sah_bloom_source_rows(uint32_t src_height, uint32_t dst_height, uint32_t row_begin, uint32_t row_end, uint32_t out[2]) {
    if (!out || src_height == 0 || dst_height == 0 || row_begin > row_end || row_end > dst_height) return SAH_ERR_INVALID_ARGUMENT;
    out[0] = out[1] = 0;
    if (row_begin == row_end) return SAH_OK;
    // exact integer floors of c -+ 2 with c = ((2 j + 1) * hs - hd) / (2 hd); with hs = 2 hd every tap coordinate is an integer + 0.5 and no
    // rounding can move its floor, otherwise one row of slack either side (androidrenderer_amd/shard.py: _downsample_sources)
    const int64_t hs = src_height, hd = dst_height, slack = (hs == 2 * hd) ? 0 : 1;
    auto floordiv = [](int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    const int64_t lo = floordiv((2 * (int64_t)row_begin + 1) * hs - hd - 4 * hd, 2 * hd) - slack;
    const int64_t hi = floordiv((2 * ((int64_t)row_end - 1) + 1) * hs - hd + 4 * hd, 2 * hd) + 1 + slack + 1;
    out[0] = (uint32_t)std::min<int64_t>(std::max<int64_t>(lo, 0), hs);
    out[1] = (uint32_t)std::min<int64_t>(std::max<int64_t>(hi, 0), hs);
    return SAH_OK;
}

int sah_tonemap(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom, const sah_plane* out, uint32_t row_begin, uint32_t row_end) {
    return sah_tonemap_ex(ctx, scene, bloom, out, row_begin, row_end, 0u);
}

int sah_tonemap_ex(sah_ctx* ctx, const sah_plane* scene, const sah_mipchain* bloom, const sah_plane* out, uint32_t row_begin, uint32_t row_end,
                   uint32_t flags) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (flags & ~SAH_TONEMAP_TOLERANCE_1CODE) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "unknown tonemap flags %#x", flags);
    if (!rgba16f_ok(scene) || !bloom || bloom->num_mips > SAH_MAX_BLOOM_MIPS || !out || !out->ptr)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "tonemap needs scene, bloom chain and output");
    if (out->format != SAH_FORMAT_R8G8B8A8_SRGB && out->format != SAH_FORMAT_R8G8B8A8_UNORM)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "tonemap output must be R8G8B8A8");
    if ((uint64_t)out->row_pitch_bytes < (uint64_t)out->width * 4 || ((uintptr_t)out->ptr % 4) || (out->row_pitch_bytes % 4))
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad output pitch/alignment");
    if (row_begin == 0 && row_end == 0) row_end = out->height;
    if (row_end > out->height || row_begin > row_end) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "bad row range");
    sah::TonemapArgs t;
    memset(&t, 0, sizeof(t));
    t.scene = parg(scene);
    t.scene_w = scene->width;
    t.scene_h = scene->height;
    t.num_mips = bloom->num_mips;
    for (uint32_t m = 0; m < bloom->num_mips; m++) {
        if (!rgba16f_ok(&bloom->mips[m])) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "bloom mip %u must be RGBA16F", m);
        t.mips[m] = parg(&bloom->mips[m]);
        t.mip_w[m] = bloom->mips[m].width;
        t.mip_inv_w[m] = 1.0f / (float)bloom->mips[m].width;
        t.mip_inv_h[m] = 1.0f / (float)bloom->mips[m].height;
        t.mip_h[m] = bloom->mips[m].height;
    }
    for (uint32_t m = bloom->num_mips; m < 6; m++) {  // the kernel's staging loads are unconditional: absent mips alias the scene (nothing of them is used)
        t.mips[m] = t.scene;
        t.mip_w[m] = scene->width;
        t.mip_h[m] = scene->height;
    }
    if (!ctx->tm_thresholds || !ctx->tm_code_table) {  // built once per context (~15k libm pow calls)
        struct {
            float thr[256];
            uint8_t first[sah::kTmMaxBuckets];
        } tab;
        uint32_t bucket_base = 0, bucket_count = 0;
        build_tonemap_thresholds(tab.thr);
        if (!build_tonemap_buckets(tab.thr, tab.first, &bucket_base, &bucket_count))
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "tonemap code table: a bucket spans more than three codes (internal)");
        // the same two-level search with one read per look-up: per bucket the three thresholds behind its first code, and that code
        float code_tab[sah::kTmMaxBuckets][4];
        for (uint32_t b = 0; b < sah::kTmMaxBuckets; b++) {
            const uint32_t f = tab.first[b];  // <= 252
            code_tab[b][0] = tab.thr[f + 1];
            code_tab[b][1] = tab.thr[f + 2];
            code_tab[b][2] = tab.thr[f + 3];
            memcpy(&code_tab[b][3], &f, 4);
        }
        // both tables are uploaded into locals and published together: a context whose second upload failed must not be left with the
        // first table set and the second one null (the next call would skip this block and launch with a null code table)
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        float *d_thr = nullptr, *d_code = nullptr;
        hipError_t e = hipMalloc((void**)&d_thr, sizeof(tab));
        if (e == hipSuccess) e = hipMemcpy(d_thr, &tab, sizeof(tab), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&d_code, sizeof(code_tab));
        if (e == hipSuccess) e = hipMemcpy(d_code, code_tab, sizeof(code_tab), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (d_thr) (void)hipFree(d_thr);
            if (d_code) (void)hipFree(d_code);
            return fail(ctx, SAH_ERR_HIP, "tonemap code tables: %s", hipGetErrorString(e));
        }
        if (ctx->tm_thresholds) (void)hipFree(ctx->tm_thresholds);
        if (ctx->tm_code_table) (void)hipFree(ctx->tm_code_table);
        ctx->tm_thresholds = d_thr;
        ctx->cache_epoch++;
        ctx->tm_code_table = d_code;
        ctx->tm_bucket_base = bucket_base;
        ctx->tm_bucket_count = bucket_count;
        ctx->tm_thr_lo = tab.thr[1];
        ctx->tm_thr_hi = tab.thr[255];
    }
    t.thresholds = ctx->tm_thresholds;
    t.code_table = ctx->tm_code_table;
    t.bucket_base = ctx->tm_bucket_base;
    t.thr_lo = ctx->tm_thr_lo;
    t.thr_hi = ctx->tm_thr_hi;
    t.out = parg(out);
    t.out_w = out->width;
    t.out_h = out->height;
    t.row_begin = row_begin;
    t.row_end = row_end;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_tonemap));
    // (the tolerance kernel stages six mips; a longer chain takes the strict kernel, whose result is inside the tolerance by definition)
    bool tol_ok = (flags & SAH_TONEMAP_TOLERANCE_1CODE) && bloom->num_mips <= 6;
    for (uint32_t m = 0; m < bloom->num_mips; m++) tol_ok = tol_ok && bloom->mips[m].width <= 65536u && bloom->mips[m].height <= 65536u;  // (16-bit extents in its LDS table)
    if (tol_ok) {
        // per-column / per-row axis set-ups: a function of the extents only, kept across calls
        uint32_t key[2 + 2 * 8 + 1] = {out->width, out->height};
        for (uint32_t m = 0; m < bloom->num_mips; m++) {
            key[2 + 2 * m] = bloom->mips[m].width;
            key[3 + 2 * m] = bloom->mips[m].height;
        }
        key[18] = bloom->num_mips;
        t.axis_stride = (std::max(out->width, out->height) + 63u) & ~63u;
        const size_t need = (size_t)6 * 2 * 4 * t.axis_stride * sizeof(sah::TmAxis);
        const bool rebuild = !ctx->tm_axis || ctx->tm_axis_bytes < need || memcmp(key, ctx->tm_axis_key, sizeof(key)) != 0;
        if (!ctx->tm_axis || ctx->tm_axis_bytes < need) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->tm_axis) (void)hipFree(ctx->tm_axis);
            ctx->tm_axis = nullptr;
            ctx->tm_axis_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->tm_axis, need));
            ctx->tm_axis_bytes = need;
        }
        t.axis_tables = (const sah::TmAxis*)ctx->tm_axis;
        if (rebuild) {
            HIP_TRY(ctx, sah::launch_tonemap_axis_tables(t, (sah::TmAxis*)ctx->tm_axis, ctx->stream));
            ctx->cache_epoch++;
            memcpy(ctx->tm_axis_key, key, sizeof(key));
        }
        HIP_TRY(ctx, sah::launch_tonemap_tol(t, ctx->stream));
    } else {
        HIP_TRY(ctx, sah::launch_tonemap(t, ctx->stream));
    }
    return SAH_OK;
}

int sah_lpv_clear(sah_ctx* ctx, const sah_volume* red, const sah_volume* green, const sah_volume* blue, const sah_volume* geometry,
                  uint32_t num_cascades) {
    SAH_RANGE();
    if (ctx) sah_drop_lpv_copy(ctx);
    if (!ctx || num_cascades == 0 || num_cascades > 4) return SAH_ERR_INVALID_ARGUMENT;
    const sah_volume* in[4] = {red, green, blue, geometry};
    sah::VolumeArg v[4];
    int n = 0;
    for (const sah_volume* p : in) {
        if (!p || !p->ptr) continue;
        if (!lpv_vol_ok(p)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "LPV volumes must be RGBA16F, 8-byte aligned");
        v[n++] = varg(*p);
    }
    if (n == 0) return SAH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah::launch_lpv_clear(v, n, num_cascades, ctx->stream));
    return SAH_OK;
}

int sah_lpv_propagate(sah_ctx* ctx, const sah_volume a_rgb[3], const sah_volume b_rgb[3], uint32_t num_cascades, uint32_t steps) {
    SAH_RANGE();
    if (!ctx || !a_rgb || !b_rgb || num_cascades == 0 || num_cascades > 4) return SAH_ERR_INVALID_ARGUMENT;
    sah::VolumeArg a[3], b[3];
    for (int i = 0; i < 3; i++) {
        if (!lpv_vol_ok(&a_rgb[i]) || !lpv_vol_ok(&b_rgb[i])) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "LPV volumes must be RGBA16F");
        if (a_rgb[i].width < 32 * num_cascades || a_rgb[i].height < 32 || a_rgb[i].depth < 32 || b_rgb[i].width < 32 * num_cascades ||
            b_rgb[i].height < 32 || b_rgb[i].depth < 32)
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "LPV volumes must be at least (32*cascades)x32x32");
        a[i] = varg(a_rgb[i]);
        b[i] = varg(b_rgb[i]);
    }
    // (arguments are in order: from here on the volumes change, and the Lighting pass's gather copy of them is stale.  The epoch moves WITH the
    // drop: an early return below — a failed launch, a failed reserve — leaves a dropped or half-written copy, and a captured Lighting half that
    // reads it must not be replayed: ADVICE r5.  Only the success path that ends where it began takes the move back, at the end.)
    const uint32_t prev_gen = ctx->lpv_pack_generation;
    const uint64_t epoch_in = ctx->cache_epoch;
    sah::VolumeArg prev_src[3];
    for (int i = 0; i < 3; i++) prev_src[i] = ctx->lpv_pack_source[i];
    sah_drop_lpv_copy(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->lpv_tables_built) {  // the 30 direction pairs' SH / lobe vectors, into this device's constant memory, once per context
        HIP_TRY(ctx, sah::launch_lpv_build_tables(ctx->stream, &ctx->lpv_hot_structure));
        ctx->lpv_tables_built = true;
    }
    // lpv.hip: the hot form of the 30 direction pairs (finite coefficients; the three colours of a cell in one thread) when the tables the device
    // built have the structure it relies on; the general form for a context under sah_debug_set(force_general) — the tests' cross-check
    static const int env_mode = getenv("SAH_LPV_MODE") ? atoi(getenv("SAH_LPV_MODE")) : -1;  // experiments: 0 general, 1 hot, 3 hot + three colours per thread
    bool offsets32 = true;  // (the hot kernels address a volume by 32-bit byte offsets)
    for (int i = 0; i < 3; i++)
        offsets32 = offsets32 && (uint64_t)a[i].slice_pitch * a[i].depth < (1ull << 32) && (uint64_t)b[i].slice_pitch * b[i].depth < (1ull << 32);
    const int mode = (!ctx->lpv_hot_structure || ctx->force_general || !offsets32) ? 0 : ((env_mode == 0 || env_mode == 1 || env_mode == 3) ? env_mode : 1);
    // light_propagation_volume.cpp:1016-1034: `steps` dispatches ping-ponging A -> B -> A ...  (Two steps per launch — 8^3 bricks with
    // their halo in LDS, bit-identical — were measured: 28 us per pair against 2 x 9.3 us, 1.5x the arithmetic in longer dependency
    // chains; not kept.)
    // The LAST step also writes the Lighting pass's gather copy of the volumes it stores (sah_gi::lpv_generation, SAH_GENERATION_TRACKED) when
    // the propagated cells are the whole volume — (32 * cascades) x 32 x 32, the reference's extent: a larger volume has texels no step
    // writes.  The copy's buffer belongs to the state sah_lighting builds and reads, possibly on another stream: same guard.
    sah::LpvPackEmit emit = {};
    bool emits = steps > 0;
    const sah::VolumeArg* last = (steps & 1) ? b : a;  // where the last step stores
    for (int i = 0; i < 3; i++) emits = emits && last[i].width == 32 * num_cascades && last[i].height == 32 && last[i].depth == 32;
    if (emits) {
        const SahLpvPackLayout pk = sah_lpv_pack_layout(last[0].width, last[0].height, last[0].depth);
        HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_lighting));
        HIP_TRY(ctx, sah_lpv_pack_reserve(ctx, pk.total));
        if (!ctx->state) emits = false;  // (made by sah_create; a context without it has no fast Lighting path either)
        if (emits) HIP_TRY(ctx, sah_lpv_pack_borders_for(ctx, last[0].width, last[0].height, last[0].depth, pk.total));
        emit = {ctx->lpv_packed, pk.row_pitch, pk.slice_pitch, ctx->state};
    }
    for (uint32_t s = 0; s < steps; s++) {
        const sah::LpvPackEmit* e = (emits && s + 1 == steps) ? &emit : nullptr;
        if ((s & 1) == 0) HIP_TRY(ctx, sah::launch_lpv_propagate(a, b, num_cascades, e, mode, ctx->stream));
        else HIP_TRY(ctx, sah::launch_lpv_propagate(b, a, num_cascades, e, mode, ctx->stream));
    }
    if (emits) {
        ctx->lpv_pack_generation = SAH_GENERATION_TRACKED;
        for (int i = 0; i < 3; i++) ctx->lpv_pack_source[i] = last[i];
    }
    // a frame loop that propagates into the same volumes every frame leaves the state as it found it (copy tracked, made from `last`):
    // the Lighting pass that follows enqueues what it enqueued the frame before.  Anything else is a change.
    bool same = emits && prev_gen == SAH_GENERATION_TRACKED;
    for (int i = 0; i < 3 && same; i++) same = same_volume(prev_src[i], last[i]);
    if (same && ctx->cache_epoch == epoch_in + 1) ctx->cache_epoch = epoch_in;  // (nothing but the drop above has moved it: no reallocation, no other layout)
    else if (!same && emits && prev_gen == 0) ctx->cache_epoch++;               // a copy came into being
    return SAH_OK;
}

int sah_sky_update_luts(sah_ctx* ctx, const sah_plane* transmittance, const sah_plane* multiscattering, const sah_plane* sky_view,
                        const float light_vector[3]) {
    SAH_RANGE();
    if (!ctx || !light_vector) return SAH_ERR_INVALID_ARGUMENT;
    auto ok = [](const sah_plane* p, uint32_t w, uint32_t h) { return rgba16f_ok(p) && p->width == w && p->height == h; };
    if (!ok(transmittance, 256, 64) || !ok(multiscattering, 32, 32) || !ok(sky_view, 200, 200))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "sky LUTs must be RGBA16F 256x64 (transmittance), 32x32 (multiple scattering), 200x200 (sky view)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah::launch_sky_luts(parg(transmittance), parg(multiscattering), parg(sky_view), light_vector, ctx->stream));
    return SAH_OK;
}

int sah_ao_clear(sah_ctx* ctx, const sah_plane* ao) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!ao || !ao->ptr || ao->format != SAH_FORMAT_R32_SFLOAT || !ao->width || !ao->height || (uint64_t)ao->row_pitch_bytes < (uint64_t)ao->width * 4 ||
        ((uintptr_t)ao->ptr % 4) || (ao->row_pitch_bytes % 4))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "the AO target must be an R32_SFLOAT plane");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah::launch_fill_r32f(parg(ao), ao->width, ao->height, 1.0f, ctx->stream));
    return SAH_OK;
}

// ---- irradiance-cache probe maintenance (a11) ---------------------------------------------------------------------------
static bool probe_vol_ok(const sah_volume& v, uint32_t format, uint32_t bpp, uint32_t w, uint32_t h) {
    return v.ptr && v.format == format && v.width == w && v.height == h && v.depth == 32 && (uint64_t)v.row_pitch_bytes >= (uint64_t)w * bpp &&
           (uint64_t)v.slice_pitch_bytes >= (uint64_t)v.row_pitch_bytes * h && ((uintptr_t)v.ptr % 4) == 0 && (bpp == 1 || (v.row_pitch_bytes % 4) == 0) &&
           (bpp == 1 || (v.slice_pitch_bytes % 4) == 0);
}
static int probe_atlases_args(sah_ctx* ctx, const sah_probe_atlases* a, sah::ProbeAtlasArgs* out) {
    if (!a) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "null probe atlases");
    // extents of irradiance_cache.cpp:94-183: probe grid 32 x (8 * 4) x 32, blocks 7x8 / 13x13 / 12x12 / 1x1
    if (!probe_vol_ok(a->rtgi, SAH_FORMAT_B10G11R11_UFLOAT_PACK32, 4, 32 * 7, 32 * 8) ||
        !probe_vol_ok(a->light_cache, SAH_FORMAT_B10G11R11_UFLOAT_PACK32, 4, 32 * 13, 32 * 13) ||
        !probe_vol_ok(a->depth, SAH_FORMAT_R16G16_SFLOAT, 4, 32 * 12, 32 * 12) || !probe_vol_ok(a->average, SAH_FORMAT_B10G11R11_UFLOAT_PACK32, 4, 32, 32) ||
        !probe_vol_ok(a->validity, SAH_FORMAT_R8_UNORM, 1, 32, 32))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT,
                    "probe atlases must be rtgi 224x256x32 B10G11R11, light cache 416x416x32 B10G11R11, depth 384x384x32 R16G16F, "
                    "average 32x32x32 B10G11R11, validity 32x32x32 R8_UNORM");
    out->rtgi = varg(a->rtgi);
    out->light_cache = varg(a->light_cache);
    out->depth = varg(a->depth);
    out->average = varg(a->average);
    out->validity = varg(a->validity);
    return SAH_OK;
}

int sah_probe_copy(sah_ctx* ctx, const sah_probe_atlases* src, const sah_probe_atlases* dst, const float cascade_movement[4][3]) {
    SAH_RANGE();
    if (!ctx || !cascade_movement) return SAH_ERR_INVALID_ARGUMENT;
    sah::ProbeAtlasArgs s, d;
    int rc = probe_atlases_args(ctx, src, &s);
    if (rc != SAH_OK) return rc;
    rc = probe_atlases_args(ctx, dst, &d);
    if (rc != SAH_OK) return rc;
    if (s.rtgi.ptr == d.rtgi.ptr || s.light_cache.ptr == d.light_cache.ptr || s.depth.ptr == d.depth.ptr || s.average.ptr == d.average.ptr ||
        s.validity.ptr == d.validity.ptr)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "probe copy: source and destination atlases must not alias");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    sah_drop_irr32_copy(ctx);  // the Lighting pass's fp32 copy of an irradiance atlas is stale from here on
    HIP_TRY(ctx, sah::launch_probe_copy(s, d, cascade_movement, ctx->stream));
    return SAH_OK;
}

int sah_probe_update(sah_ctx* ctx, const sah_probe_atlases* atlases, const sah_volume* trace_results, const uint32_t* probes_to_update,
                     uint32_t num_probes) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    sah::ProbeAtlasArgs a;
    const int rc = probe_atlases_args(ctx, atlases, &a);
    if (rc != SAH_OK) return rc;
    if (num_probes == 0) return SAH_OK;
    if (!probes_to_update) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "probes_to_update is null");
    if (!trace_results || !trace_results->ptr || trace_results->format != SAH_FORMAT_R16G16B16A16_SFLOAT || trace_results->width != 20 ||
        trace_results->height != 20 || trace_results->depth < num_probes || trace_results->row_pitch_bytes < 20 * 8 ||
        (uint64_t)trace_results->slice_pitch_bytes < (uint64_t)trace_results->row_pitch_bytes * 20 || ((uintptr_t)trace_results->ptr % 8) ||
        (trace_results->row_pitch_bytes % 8) || (trace_results->slice_pitch_bytes % 8))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "trace_results must be R16G16B16A16_SFLOAT 20 x 20 x >= num_probes, 8-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the Lighting pass's fp32 copy of this irradiance atlas: kept current probe by probe when the context tracks it
    // (SAH_GENERATION_TRACKED, made from this very atlas), stale otherwise
    const bool patch_irr32 = ctx->irr32 && ctx->irr32_generation == SAH_GENERATION_TRACKED && same_volume(a.rtgi, ctx->irr32_source);
    if (!patch_irr32) sah_drop_irr32_copy(ctx);
    if (!ctx->probe_slots) {  // probe cell -> position in the update list (probes.hip: ordered_stores); all zero between calls
        HIP_TRY(ctx, hipMalloc((void**)&ctx->probe_slots, 32 * 32 * 32 * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->probe_slots, 0, 32 * 32 * 32 * sizeof(uint32_t), ctx->stream));
    }
    // the slot table is context-wide: an update enqueued on another stream than the previous one starts behind that one's clear pass
    if (!ctx->probe_done) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->probe_done, hipEventDisableTiming));
    if (ctx->probe_stream && ctx->probe_stream != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->probe_done, 0));
    HIP_TRY(ctx, sah::launch_probe_update(a, varg(*trace_results), probes_to_update, num_probes, ctx->probe_slots, ctx->stream));
    if (patch_irr32) {
        HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_lighting));
        HIP_TRY(ctx, sah::launch_probe_irr_unpack_probes(a.rtgi, ctx->irr32, probes_to_update, num_probes, ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->probe_done, ctx->stream));
    ctx->probe_stream = ctx->stream;
    return SAH_OK;
}

int sah_probe_notify_updated(sah_ctx* ctx, const sah_volume* probe_irradiance, const uint32_t* probes, uint32_t num_probes) {
    SAH_RANGE();
    if (!ctx || !probe_irradiance) return SAH_ERR_INVALID_ARGUMENT;
    if (num_probes == 0) return SAH_OK;
    if (!probes) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "probes is null");
    const sah::VolumeArg irr = varg(*probe_irradiance);
    if (!ctx->irr32 || ctx->irr32_generation != SAH_GENERATION_TRACKED || !same_volume(irr, ctx->irr32_source)) return SAH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_lighting));
    HIP_TRY(ctx, sah::launch_probe_irr_unpack_probes(irr, ctx->irr32, probes, num_probes, ctx->stream));
    return SAH_OK;
}

// ---- RCCL (resolved at run time so that a process which already carries an RCCL — e.g. PyTorch's — shares it) ----
// The function-pointer types come from <rccl/rccl.h> itself (decltype of the declarations), so a prototype that drifts from the
// installed library is a compile error, not a silent ABI mismatch; only the symbol lookup is deferred to dlopen / dlsym.
#define RCCL_SYM(lib, fn) reinterpret_cast<decltype(&fn)>(dlsym(lib, #fn))

static void* open_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) return h;
    }
    return nullptr;
}

int sah_comm_unique_id(void* out) {
    if (!out) return SAH_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == 128, "sah_comm_unique_id hands out 128 bytes");
    void* h = open_rccl();
    if (!h) return SAH_ERR_COMM;
    auto f = RCCL_SYM(h, ncclGetUniqueId);
    if (!f) return SAH_ERR_COMM;
    ncclUniqueId id;
    if (f(&id) != ncclSuccess) return SAH_ERR_COMM;
    memcpy(out, &id, sizeof(id));
    return SAH_OK;
}

int sah_comm_init(sah_ctx* ctx, const void* comm_id) {
    ctx->rccl = open_rccl();
    if (!ctx->rccl) return fail(ctx, SAH_ERR_COMM, "librccl not found: %s", dlerror());
    auto init = RCCL_SYM(ctx->rccl, ncclCommInitRank);
    if (!init) return fail(ctx, SAH_ERR_COMM, "ncclCommInitRank not found");
    ncclUniqueId id;
    memcpy(&id, comm_id, sizeof(id));
    if (hipSetDevice(ctx->device) != hipSuccess) return SAH_ERR_HIP;
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = init(&comm, ctx->world, id, ctx->rank);
    if (rc != ncclSuccess) return fail(ctx, SAH_ERR_COMM, "ncclCommInitRank failed: %d", (int)rc);
    ctx->comm = comm;
    // The reversed-rank communicator of sah_allgather_rows_reversed is made here, while nothing is in flight on the parent (a split
    // is a collective over the parent and must not overlap its other operations).  If the installed RCCL cannot split, the reversed
    // exchange falls back to grouped point-to-point transfers on the parent communicator.
    ctx->comm_reversed = nullptr;
    const char* no_split = getenv("SAH_COMM_NO_SPLIT");
    auto split = RCCL_SYM(ctx->rccl, ncclCommSplit);
    ncclComm_t rev = nullptr;
    if (split && !(no_split && no_split[0] == '1')) {
        if (split(comm, 0, ctx->world - 1 - ctx->rank, &rev, nullptr) != ncclSuccess) rev = nullptr;
    }
    // Which path the reversed exchange takes must be ONE decision for the whole job: a rank on the split communicator and a rank on
    // the send / recv fallback would wait for each other forever.  So the ranks agree (minimum of "my split succeeded" over the parent
    // communicator) and the reversed communicator is used only if every rank has one.
    int mine = rev ? 1 : 0, all = 0;
    int* d_flag = nullptr;
    auto allreduce = RCCL_SYM(ctx->rccl, ncclAllReduce);
    bool agreed = false;
    if (allreduce && hipMalloc((void**)&d_flag, sizeof(int)) == hipSuccess) {
        if (hipMemcpy(d_flag, &mine, sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
            allreduce(d_flag, d_flag, 1, ncclInt32, ncclMin, comm, ctx->stream) == ncclSuccess &&
            hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(&all, d_flag, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
            agreed = true;
        (void)hipFree(d_flag);
    }
    if (!agreed) {
        if (rev) {
            auto destroy = RCCL_SYM(ctx->rccl, ncclCommDestroy);
            if (destroy) destroy(rev);
        }
        return fail(ctx, SAH_ERR_COMM, "the ranks could not agree on the reversed-exchange path (ncclAllReduce on the parent communicator failed)");
    }
    if (all == 1) {
        ctx->comm_reversed = rev;
        ctx->last_error = "reversed exchange: split communicator";
    } else {
        if (rev) {
            auto destroy = RCCL_SYM(ctx->rccl, ncclCommDestroy);
            if (destroy) destroy(rev);
        }
        ctx->last_error = "reversed exchange: grouped ncclSend / ncclRecv on the parent communicator";
    }
    return SAH_OK;
}

void sah_comm_destroy(sah_ctx* ctx) {
    // nothing of this context may still be in flight on either stream when the communicators and events go away
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->rccl) {
        auto f = RCCL_SYM(ctx->rccl, ncclCommDestroy);
        if (f && ctx->comm_reversed) f((ncclComm_t)ctx->comm_reversed);
        if (f && ctx->comm) f((ncclComm_t)ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->comm_reversed = nullptr;
}

int sah_comm_set_stream(sah_ctx* ctx, void* hip_stream) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_pending) {  // a gather is still in flight on the old side stream: the work stream joins it before the streams change
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_done, 0));
        ctx->comm_pending = false;
    }
    ctx->comm_stream = (hipStream_t)hip_stream;
    if (ctx->comm_stream && !ctx->comm_ready) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->comm_ready, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->comm_done, hipEventDisableTiming));
    }
    ctx->comm_pending = false;
    return SAH_OK;
}

int sah_comm_wait(sah_ctx* ctx) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (ctx->comm_pending) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_done, 0));
        ctx->comm_pending = false;
    }
    // (what is known on the host now: a wait of an EARLIER gather that gave up.  The gather just joined may still be running; sah_sync
    // reports its outcome)
    if (sah_ipc_timed_out(ctx)) return fail(ctx, SAH_ERR_COMM, "direct exchange: a peer did not arrive within 2 s; the gathered rows are not valid");
    return SAH_OK;
}

// `reversed`: the exchange runs on a second communicator in which this process has rank world - 1 - rank (made by sah_comm_init with
// ncclCommSplit: same devices, key = reversed rank), so that the in-place slot of rank r is block world - 1 - r.  Without that
// communicator the same blocks travel as grouped ncclSend / ncclRecv pairs on the parent.
static int allgather_bytes_impl(sah_ctx* ctx, void* buffer, uint64_t bytes_per_rank, bool reversed) {
    if (!ctx || !buffer) return SAH_ERR_INVALID_ARGUMENT;
    if (bytes_per_rank == 0) return SAH_OK;
    if (const int id = sah_ipc_find(ctx, buffer, (uint64_t)ctx->world * bytes_per_rank); id >= 0) {
        // direct exchange (api_ipc.cpp): same stream discipline as the RCCL path below
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const bool side = ctx->comm_stream && ctx->comm_stream != ctx->stream;
        if (side) {
            HIP_TRY(ctx, hipEventRecord(ctx->comm_ready, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0));
            st = ctx->comm_stream;
        }
        if (int rc = sah_ipc_gather(ctx, (uint32_t)id, (uint8_t*)buffer, bytes_per_rank, reversed, st); rc != SAH_OK) return rc;
        if (side) {
            HIP_TRY(ctx, hipEventRecord(ctx->comm_done, ctx->comm_stream));
            ctx->comm_pending = true;
        }
        return SAH_OK;
    }
    if (!ctx->comm) {
        if (ctx->world == 1) return SAH_OK;  // one rank and no communicator: the buffer already is the gathered result
        return fail(ctx, SAH_ERR_COMM, "context was created without a communicator");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    int slot = ctx->rank;
    const bool p2p = reversed && !ctx->comm_reversed;
    if (reversed) {
        if (ctx->comm_reversed) comm = (ncclComm_t)ctx->comm_reversed;
        slot = ctx->world - 1 - ctx->rank;
    }
    auto ag = RCCL_SYM(ctx->rccl, ncclAllGather);
    if (!ag) return fail(ctx, SAH_ERR_COMM, "ncclAllGather not found");
    const uint8_t* send = (const uint8_t*)buffer + (size_t)slot * bytes_per_rank;
    hipStream_t st = ctx->stream;
    const bool side = ctx->comm_stream && ctx->comm_stream != ctx->stream;
    if (side) {  // the gather runs behind everything enqueued so far on the work stream, and beside whatever is enqueued next
        HIP_TRY(ctx, hipEventRecord(ctx->comm_ready, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0));
        st = ctx->comm_stream;
    }
    if (p2p) {
        auto gs = RCCL_SYM(ctx->rccl, ncclGroupStart);
        auto ge = RCCL_SYM(ctx->rccl, ncclGroupEnd);
        auto snd = RCCL_SYM(ctx->rccl, ncclSend);
        auto rcv = RCCL_SYM(ctx->rccl, ncclRecv);
        if (!gs || !ge || !snd || !rcv) return fail(ctx, SAH_ERR_COMM, "ncclSend / ncclRecv / ncclGroup* not found");
        ncclResult_t rc = gs();
        for (int p = 0; p < ctx->world && rc == ncclSuccess; p++) {
            if (p == ctx->rank) continue;  // this rank's block is already in its slot
            rc = snd(send, (size_t)bytes_per_rank, ncclUint8, p, comm, st);
            if (rc == ncclSuccess) rc = rcv((uint8_t*)buffer + (size_t)(ctx->world - 1 - p) * bytes_per_rank, (size_t)bytes_per_rank, ncclUint8, p, comm, st);
        }
        const ncclResult_t rc_end = ge();
        if (rc != ncclSuccess || rc_end != ncclSuccess) return fail(ctx, SAH_ERR_COMM, "grouped ncclSend / ncclRecv failed: %d / %d", (int)rc, (int)rc_end);
    } else {
        // in place: the send buffer is this rank's slot of the receive buffer
        const ncclResult_t rc = ag(send, buffer, (size_t)bytes_per_rank, ncclUint8, comm, st);
        if (rc != ncclSuccess) return fail(ctx, SAH_ERR_COMM, "ncclAllGather failed: %d", (int)rc);
    }
    if (side) {
        HIP_TRY(ctx, hipEventRecord(ctx->comm_done, ctx->comm_stream));
        ctx->comm_pending = true;
    }
    return SAH_OK;
}

int sah_allgather_bytes(sah_ctx* ctx, void* buffer, uint64_t bytes_per_rank) { return allgather_bytes_impl(ctx, buffer, bytes_per_rank, false); }

static int allgather_rows_impl(sah_ctx* ctx, const sah_plane* image, uint32_t rows_per_rank, uint32_t allocated_rows, bool reversed) {
    if (!ctx || !image || !image->ptr) return SAH_ERR_INVALID_ARGUMENT;
    const uint64_t slots = (uint64_t)rows_per_rank * ctx->world;
    if (slots < image->height)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "rows_per_rank * world = %llu leaves rows of a %u-row image ungathered", (unsigned long long)slots,
                    image->height);
    if (slots > allocated_rows)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "the allocation holds %u rows, the gather needs %llu equal slots (pad it to rows_per_rank * world)",
                    allocated_rows, (unsigned long long)slots);
    return allgather_bytes_impl(ctx, image->ptr, (uint64_t)rows_per_rank * image->row_pitch_bytes, reversed);
}

int sah_allgather_rows(sah_ctx* ctx, const sah_plane* image, uint32_t rows_per_rank, uint32_t allocated_rows) {
    SAH_RANGE();
    return allgather_rows_impl(ctx, image, rows_per_rank, allocated_rows, false);
}

int sah_allgather_rows_reversed(sah_ctx* ctx, const sah_plane* image, uint32_t rows_per_rank, uint32_t allocated_rows) {
    SAH_RANGE();
    return allgather_rows_impl(ctx, image, rows_per_rank, allocated_rows, true);
}

}  // extern "C"
