// C ABI: ray tracing — sah_rt_build, sah_rtao, sah_sun_shadow_mask (include/sah_hip.h "ray tracing"; kernels in rt.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "../../include/sah_hip.h"
#include "ctx.hpp"
#include "rt_args.hpp"

namespace sah {
hipError_t launch_rt_scan(const sah_primitive* prims, uint32_t n, uint32_t* tri_base, RtBuildState* st, hipStream_t s);
hipError_t launch_rt_world(const RtScene& sc, const uint32_t* tri_base, uint32_t total, RtTriangle* out, RtBuildState* st, hipStream_t s);
hipError_t launch_rt_sort(const RtTriangle* tris, const RtBuildState* st, unsigned long long* keys, uint32_t padded, hipStream_t s);
hipError_t launch_rt_nodes(const RtTriangle* unsorted, const unsigned long long* keys, RtTriangle* sorted, RtNodeGroup* nodes, const RtBvh& bvh, hipStream_t s);
hipError_t launch_rtao(const RtaoArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s);
hipError_t launch_sun_shadow_mask(const ShadowMaskArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s);
hipError_t launch_noise_dirs(const PlaneArg& noise, const float* luts, float* out, hipStream_t s);
hipError_t launch_probe_trace(const ProbeTraceArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s);
hipError_t launch_rtgi_trace(const RtgiTraceArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s);
}  // namespace sah

namespace {
enum Slot { R_TRI_BASE, R_STATE, R_UNSORTED, R_SORTED, R_KEYS, R_NODES, R_NOISE_DIRS };

int ensure(sah_ctx* ctx, int slot, size_t bytes) {
    auto& r = ctx->rt;
    if (r.bytes[slot] >= bytes && r.ptr[slot]) return SAH_OK;
    if (r.ptr[slot]) (void)hipFree(r.ptr[slot]);
    r.ptr[slot] = nullptr;
    r.bytes[slot] = 0;
    const size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(ctx, hipMalloc(&r.ptr[slot], want));
    r.bytes[slot] = want;
    return SAH_OK;
}

bool plane_fmt(const sah_plane* p, uint32_t fmt, uint32_t bpp) {
    return p && p->ptr && p->format == fmt && p->width && p->height && (uint64_t)p->row_pitch_bytes >= (uint64_t)p->width * bpp &&
           ((uintptr_t)p->ptr % bpp) == 0 && (p->row_pitch_bytes % bpp) == 0;
}
}  // namespace

// the row window of sah_rt_set_rows clipped to a plane of `height` rows ((0, 0): every row)
static void rt_rows(const sah_ctx* ctx, uint32_t height, uint32_t* begin, uint32_t* end) {
    const bool all = ctx->rt.row_begin == 0 && ctx->rt.row_end == 0;
    *begin = all ? 0u : (ctx->rt.row_begin < height ? ctx->rt.row_begin : height);
    *end = all ? height : (ctx->rt.row_end < height ? ctx->rt.row_end : height);
}

extern "C" {

int sah_rt_set_rows(sah_ctx* ctx, uint32_t row_begin, uint32_t row_end) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (row_end < row_begin) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "row range [%u, %u)", row_begin, row_end);
    ctx->rt.row_begin = row_begin;
    ctx->rt.row_end = row_end;
    return SAH_OK;
}


int sah_rt_set_bounces(sah_ctx* ctx, uint32_t num_bounces) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (num_bounces > (uint32_t)sah::kRtMaxBounces) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "at most %d bounces (asked for %u)", sah::kRtMaxBounces, num_bounces);
    ctx->rt.num_bounces = num_bounces;
    return SAH_OK;
}

int sah_rt_build(sah_ctx* ctx, const sah_scene_geometry* scene, uint32_t* stats) {
    SAH_RANGE();
    using namespace sah;
    if (!ctx || !scene) return SAH_ERR_INVALID_ARGUMENT;
    auto& rt = ctx->rt;
    rt.built = false;
    if (scene->num_primitives && (!scene->primitives || !scene->indices || !scene->vertex_positions))
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "scene arrays are null");
    if (scene->num_primitives >= (1u << 24)) return fail(ctx, SAH_ERR_UNSUPPORTED, "too many primitives");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_rt));
    RtScene sc;
    memset(&sc, 0, sizeof(sc));
    sc.positions = scene->vertex_positions;
    sc.vertex_data = scene->vertex_data;
    sc.indices = scene->indices;
    sc.primitives = scene->primitives;
    sc.materials = scene->materials;
    sc.num_primitives = scene->num_primitives;
    sc.num_indices = scene->num_indices;
    sc.num_vertices = scene->num_vertices;
    sc.num_materials = scene->num_materials;
    sc.textures = scene->num_textures ? scene->textures : nullptr;
    sc.material_textures = (scene->num_textures && scene->textures) ? scene->material_textures : nullptr;
    sc.num_textures = scene->num_textures;
    sc.luts = ctx->luts;
    RtBvh bvh;
    memset(&bvh, 0, sizeof(bvh));
    RtBuildState host;
    memset(&host, 0, sizeof(host));
    if (scene->num_primitives) {
        if (int rc = ensure(ctx, R_TRI_BASE, (size_t)(scene->num_primitives + 1) * sizeof(uint32_t)); rc != SAH_OK) return rc;
        if (int rc = ensure(ctx, R_STATE, sizeof(RtBuildState)); rc != SAH_OK) return rc;
        auto* st = (RtBuildState*)rt.ptr[R_STATE];
        HIP_TRY(ctx, launch_rt_scan(scene->primitives, scene->num_primitives, (uint32_t*)rt.ptr[R_TRI_BASE], st, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t total = host.total;
        if (total >= kRtMaxTriangles) return fail(ctx, SAH_ERR_UNSUPPORTED, "%u triangles: the structure holds fewer than %u", total, kRtMaxTriangles);
        if (total) {
            uint32_t padded = kRtSortChunk;
            while (padded < total) padded <<= 1;
            if (int rc = ensure(ctx, R_UNSORTED, (size_t)total * sizeof(RtTriangle)); rc != SAH_OK) return rc;
            // (+ kRtFanout spare entries behind the triangles and the nodes: the walk loads groups of four without a predicate)
            if (int rc = ensure(ctx, R_SORTED, (size_t)(total + kRtFanout) * sizeof(RtTriangle)); rc != SAH_OK) return rc;
            if (int rc = ensure(ctx, R_KEYS, (size_t)padded * sizeof(unsigned long long)); rc != SAH_OK) return rc;
            HIP_TRY(ctx, launch_rt_world(sc, (const uint32_t*)rt.ptr[R_TRI_BASE], total, (RtTriangle*)rt.ptr[R_UNSORTED], st, ctx->stream));
            HIP_TRY(ctx, launch_rt_sort((const RtTriangle*)rt.ptr[R_UNSORTED], st, (unsigned long long*)rt.ptr[R_KEYS], padded, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            bvh.num_tris = host.kept;
            float S;
            memcpy(&S, &host.max_abs_bits, 4);
            bvh.pad = S * 0x1p-16f;
            uint32_t count = bvh.num_tris, offset = 0, levels = 0;  // level 0: one box per triangle
            while (bvh.num_tris) {
                bvh.level_offset[levels] = offset;  // in groups of four nodes
                bvh.level_count[levels] = count;
                offset += (count + kRtFanout - 1) / kRtFanout;
                levels++;
                if (count == 1) break;
                count = (count + kRtFanout - 1) / kRtFanout;
            }
            bvh.num_levels = levels;
            if (bvh.num_tris) {
                if (int rc = ensure(ctx, R_NODES, (size_t)offset * sizeof(RtNodeGroup)); rc != SAH_OK) return rc;
                bvh.tris = (const RtTriangle*)rt.ptr[R_SORTED];
                bvh.nodes = (const RtNodeGroup*)rt.ptr[R_NODES];
                HIP_TRY(ctx, launch_rt_nodes((const RtTriangle*)rt.ptr[R_UNSORTED], (const unsigned long long*)rt.ptr[R_KEYS], (RtTriangle*)rt.ptr[R_SORTED],
                                             (RtNodeGroup*)rt.ptr[R_NODES], bvh, ctx->stream));
            }
        }
    }
    rt.bvh = bvh;
    rt.scene = sc;
    rt.built = true;
    if (stats) {
        stats[0] = bvh.num_tris;
        stats[1] = host.dropped;
        stats[2] = bvh.num_levels;
        stats[3] = 0;
    }
    return SAH_OK;
}

static int check_cutout_inputs(sah_ctx* ctx) {
    // the any-hit stage reads vertex colours, texcoords and materials: a structure built without them cannot shade CUTOUT candidates
    const sah::RtScene& sc = ctx->rt.scene;
    if (ctx->rt.bvh.num_tris && (!sc.vertex_data || !sc.materials || sc.num_materials == 0))
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "the scene given to sah_rt_build has no vertex data / materials (needed by the any-hit stage)");
    return SAH_OK;
}

int sah_rtao(sah_ctx* ctx, const sah_view_data* view, const sah_plane* depth, const sah_plane* normals, const sah_plane* noise,
             uint32_t samples_per_pixel, float max_ray_distance, const sah_plane* ao_out) {
    SAH_RANGE();
    using namespace sah;
    if (!ctx || !view) return SAH_ERR_INVALID_ARGUMENT;
    if (!ctx->rt.built) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sah_rt_build has not been called on this context");
    if (!plane_fmt(ao_out, SAH_FORMAT_R32_SFLOAT, 4)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "ao_out must be an R32_SFLOAT plane");
    const uint32_t W = ao_out->width, H = ao_out->height;
    if (!plane_fmt(depth, depth ? depth->format : 0, 4) || (depth->format != SAH_FORMAT_D32_SFLOAT && depth->format != SAH_FORMAT_R32_SFLOAT) ||
        depth->width != W || depth->height != H)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "depth must be D32_SFLOAT of ao_out's extent");
    if (!plane_fmt(normals, SAH_FORMAT_R16G16B16A16_SFLOAT, 8) || normals->width != W || normals->height != H)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "normals must be R16G16B16A16_SFLOAT of ao_out's extent");
    if (!plane_fmt(noise, SAH_FORMAT_R8G8B8A8_UNORM, 4) || noise->width > 65535 || noise->height > 65535)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "noise must be R8G8B8A8_UNORM with an extent that fits uint16");
    if (samples_per_pixel > 4096) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "samples_per_pixel > 4096");
    RtaoArgs a;
    memset(&a, 0, sizeof(a));
    a.depth = parg(depth);
    a.normals = parg(normals);
    a.noise = parg(noise);
    a.out = parg(ao_out);
    a.width = W;
    a.height = H;
    a.noise_w = noise->width;
    a.noise_h = noise->height;
    memcpy(a.inv_proj, view->inverse_projection, 64);
    memcpy(a.inv_view, view->inverse_view, 64);
    a.res[0] = view->render_resolution[0];
    a.res[1] = view->render_resolution[1];
    a.samples = samples_per_pixel;
    a.max_distance = max_ray_distance;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_rt));
    rt_rows(ctx, H, &a.row_begin, &a.row_end);
    HIP_TRY(ctx, launch_rtao(a, ctx->rt.bvh, ctx->rt.scene, ctx->stream));
    return SAH_OK;
}

int sah_sun_shadow_mask(sah_ctx* ctx, const sah_view_data* view, const sah_sun_light_constants* sun, const sah_plane* depth,
                        const sah_plane* normals, const sah_plane* noise, const sah_plane* mask_out) {
    SAH_RANGE();
    using namespace sah;
    if (!ctx || !view || !sun) return SAH_ERR_INVALID_ARGUMENT;
    if (!ctx->rt.built) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sah_rt_build has not been called on this context");
    if (int rc = check_cutout_inputs(ctx); rc != SAH_OK) return rc;
    if (!plane_fmt(mask_out, SAH_FORMAT_R32_SFLOAT, 4)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "mask_out must be an R32_SFLOAT plane");
    const uint32_t W = mask_out->width, H = mask_out->height;
    if (!plane_fmt(depth, depth ? depth->format : 0, 4) || (depth->format != SAH_FORMAT_D32_SFLOAT && depth->format != SAH_FORMAT_R32_SFLOAT) ||
        depth->width != W || depth->height != H)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "depth must be D32_SFLOAT of mask_out's extent");
    if (!plane_fmt(normals, SAH_FORMAT_R16G16B16A16_SFLOAT, 8) || normals->width != W || normals->height != H)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "normals must be R16G16B16A16_SFLOAT of mask_out's extent");
    if (!plane_fmt(noise, SAH_FORMAT_R8G8B8A8_UNORM, 4) || noise->width < 128 || noise->height < 128)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "noise must be R8G8B8A8_UNORM, at least 128 x 128");
    if (!(sun->num_shadow_samples >= 0.0f && sun->num_shadow_samples <= 4096.0f))
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "num_shadow_samples must be in [0, 4096]");
    ShadowMaskArgs a;
    memset(&a, 0, sizeof(a));
    a.depth = parg(depth);
    a.normals = parg(normals);
    a.noise = parg(noise);
    a.out = parg(mask_out);
    a.width = W;
    a.height = H;
    memcpy(a.inv_proj, view->inverse_projection, 64);
    memcpy(a.inv_view, view->inverse_view, 64);
    a.res[0] = view->render_resolution[0];
    a.res[1] = view->render_resolution[1];
    {  // normalize(-direction): x * (1 / sqrt(dot)), every operator rounded (this file is built with -ffp-contract=off)
        const float n[3] = {-sun->direction_and_tan_size[0], -sun->direction_and_tan_size[1], -sun->direction_and_tan_size[2]};
        const float d = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
        const float inv = 1.0f / std::sqrt(d);
        for (int i = 0; i < 3; i++) a.L[i] = n[i] * inv;
    }
    a.tan_size = sun->direction_and_tan_size[3];
    a.num_samples = sun->num_shadow_samples;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_rt));
    rt_rows(ctx, H, &a.row_begin, &a.row_end);
    // every sample of every pixel reads one of 16 384 noise directions: normalised once per call instead of once per ray
    if (int rc = ensure(ctx, R_NOISE_DIRS, 128u * 128u * 16u); rc != SAH_OK) return rc;
    a.noise_dirs = static_cast<const float*>(ctx->rt.ptr[R_NOISE_DIRS]);
    HIP_TRY(ctx, launch_noise_dirs(a.noise, ctx->rt.scene.luts, static_cast<float*>(ctx->rt.ptr[R_NOISE_DIRS]), ctx->stream));
    HIP_TRY(ctx, launch_sun_shadow_mask(a, ctx->rt.bvh, ctx->rt.scene, ctx->stream));
    return SAH_OK;
}

// what both GI generators hand to the hit / miss stages
static int fill_gi_args(sah_ctx* ctx, const sah_sun_light_constants* sun, const sah_sky_luts* sky, const sah_plane* noise, sah::GiArgs* g) {
    if (!sun || !sky) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sun constants and sky LUTs are required (hit and miss stages)");
    if (!plane_fmt(noise, SAH_FORMAT_R8G8B8A8_UNORM, 4) || noise->width < 128 || noise->height < 128)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "noise must be R8G8B8A8_UNORM, at least 128 x 128");
    memset(g, 0, sizeof(*g));
    for (int i = 0; i < 3; i++) g->sun_dir[i] = sun->direction_and_tan_size[i];
    for (int i = 0; i < 3; i++) g->sun_color[i] = sun->color[i];
    g->tan_size = sun->direction_and_tan_size[3];
    g->noise = parg(noise);
    g->num_bounces = ctx->rt.num_bounces;
    // GI miss stage: get_sky_color(WorldRayDirection(), sun_light.direction_and_tan_size.xyz, ...) — the direction as stored (sky_unified.slang:229)
    const float sun_dir[3] = {sun->direction_and_tan_size[0], sun->direction_and_tan_size[1], sun->direction_and_tan_size[2]};
    if (!fill_sky_args(*sky, sun_dir, &g->sky)) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "sky LUTs must be RGBA16F");
    return SAH_OK;
}

int sah_probe_trace(sah_ctx* ctx, const sah_probe_trace_desc* d) {
    SAH_RANGE();
    using namespace sah;
    if (!ctx || !d) return SAH_ERR_INVALID_ARGUMENT;
    if (!ctx->rt.built) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sah_rt_build has not been called on this context");
    if (int rc = check_cutout_inputs(ctx); rc != SAH_OK) return rc;
    if (d->num_probes == 0) return SAH_OK;
    if (!d->probes_to_update) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "probes_to_update is null");
    const sah_volume& tr = d->trace_results;
    if (!tr.ptr || tr.format != SAH_FORMAT_R16G16B16A16_SFLOAT || tr.width != 20 || tr.height != 20 || tr.depth < d->num_probes || tr.row_pitch_bytes < 160 ||
        (uint64_t)tr.slice_pitch_bytes < (uint64_t)tr.row_pitch_bytes * 20 || ((uintptr_t)tr.ptr % 8) || (tr.row_pitch_bytes % 8) || (tr.slice_pitch_bytes % 8))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "trace_results must be R16G16B16A16_SFLOAT 20 x 20 x >= num_probes, 8-byte aligned");
    if (!d->probe_irradiance.ptr || d->probe_irradiance.format != SAH_FORMAT_B10G11R11_UFLOAT_PACK32 || !d->probe_depth.ptr ||
        d->probe_depth.format != SAH_FORMAT_R16G16_SFLOAT || !d->probe_validity.ptr || d->probe_validity.format != SAH_FORMAT_R8_UNORM)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "probe atlases must be B10G11R11 / R16G16F / R8_UNORM arrays");
    ProbeTraceArgs a;
    memset(&a, 0, sizeof(a));
    if (int rc = fill_gi_args(ctx, d->sun, d->sky, d->noise, &a.gi); rc != SAH_OK) return rc;
    a.cache.irradiance = varg(d->probe_irradiance);
    a.cache.depth = varg(d->probe_depth);
    a.cache.validity = varg(d->probe_validity);
    for (int c = 0; c < 4; c++) {
        for (int i = 0; i < 3; i++) a.cache.cascade_min[c][i] = d->cascades[c].min[i];
        a.cache.spacing[c] = d->cascades[c].probe_spacing;
    }
    a.cache.probe_size[0] = d->probe_size[0];
    a.cache.probe_size[1] = d->probe_size[1];
    a.probes = d->probes_to_update;
    a.num_probes = d->num_probes;
    a.out = varg(tr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_rt));
    HIP_TRY(ctx, launch_probe_trace(a, ctx->rt.bvh, ctx->rt.scene, ctx->stream));
    return SAH_OK;
}

int sah_rtgi_trace(sah_ctx* ctx, const sah_view_data* view, const sah_sun_light_constants* sun, const sah_sky_luts* sky, const sah_plane* depth,
                   const sah_plane* normals, const sah_plane* noise, const sah_plane* ray_buffer, const sah_plane* ray_irradiance) {
    SAH_RANGE();
    using namespace sah;
    if (!ctx || !view) return SAH_ERR_INVALID_ARGUMENT;
    if (!ctx->rt.built) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sah_rt_build has not been called on this context");
    if (int rc = check_cutout_inputs(ctx); rc != SAH_OK) return rc;
    if (!plane_fmt(ray_buffer, SAH_FORMAT_R16G16B16A16_SFLOAT, 8) || !plane_fmt(ray_irradiance, SAH_FORMAT_R16G16B16A16_SFLOAT, 8))
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "ray_buffer / ray_irradiance must be R16G16B16A16_SFLOAT planes");
    const uint32_t W = ray_buffer->width, H = ray_buffer->height;
    if (ray_irradiance->width != W || ray_irradiance->height != H) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "ray_buffer and ray_irradiance differ in extent");
    if (!plane_fmt(depth, depth ? depth->format : 0, 4) || (depth->format != SAH_FORMAT_D32_SFLOAT && depth->format != SAH_FORMAT_R32_SFLOAT) ||
        depth->width != W || depth->height != H)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "depth must be D32_SFLOAT of the ray buffers' extent");
    if (!plane_fmt(normals, SAH_FORMAT_R16G16B16A16_SFLOAT, 8) || normals->width != W || normals->height != H)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "normals must be R16G16B16A16_SFLOAT of the ray buffers' extent");
    RtgiTraceArgs a;
    memset(&a, 0, sizeof(a));
    if (int rc = fill_gi_args(ctx, sun, sky, noise, &a.gi); rc != SAH_OK) return rc;
    a.depth = parg(depth);
    a.normals = parg(normals);
    a.ray_buffer = parg(ray_buffer);
    a.ray_irradiance = parg(ray_irradiance);
    a.width = W;
    a.height = H;
    memcpy(a.inv_proj, view->inverse_projection, 64);
    memcpy(a.inv_view, view->inverse_view, 64);
    a.res[0] = view->render_resolution[0];
    a.res[1] = view->render_resolution[1];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_rt));
    rt_rows(ctx, H, &a.row_begin, &a.row_end);
    HIP_TRY(ctx, launch_rtgi_trace(a, ctx->rt.bvh, ctx->rt.scene, ctx->stream));
    return SAH_OK;
}

}  // extern "C"
