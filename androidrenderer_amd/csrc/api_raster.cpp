// C ABI: the scene rasteriser (sun shadow cascades, depth + G-buffer) — argument checks, scratch buffers, the two launch stages.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/sah_hip.h"
#include "ctx.hpp"
#include "raster_args.hpp"

namespace sah {
hipError_t launch_raster_setup(const RasterArgs& a, bool gbuffer, hipStream_t st);
hipError_t launch_raster_tiles(const RasterArgs& a, bool gbuffer, hipStream_t st);
}  // namespace sah

namespace {
constexpr uint32_t kTile = 64;
constexpr uint32_t kMaxExtent = 8192;  // keeps every snapped coordinate inside the guard band below 2^24.1
enum Scratch { S_COUNTERS, S_TRI_BASE, S_RECORDS, S_ATTRS, S_TILES, S_PAIRS, S_SEQ, S_CLIPQ };

int ensure(sah_ctx* ctx, int slot, size_t bytes) {
    auto& r = ctx->raster;
    if (r.bytes[slot] >= bytes && r.ptr[slot]) return SAH_OK;
    if (r.ptr[slot]) (void)hipFree(r.ptr[slot]);
    r.ptr[slot] = nullptr;
    r.bytes[slot] = 0;
    const size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(ctx, hipMalloc(&r.ptr[slot], want));
    r.bytes[slot] = want;
    return SAH_OK;
}

// fp16 bit pattern -> sRGB8 code of an R8G8B8A8_SRGB store: the OETF evaluated in fp64 and rounded to fp32, then UNORM8 as
// floor(s * 255 + 0.5) in fp32 (DESIGN.md §3 "stores")
int ensure_srgb_table(sah_ctx* ctx) {
    if (ctx->raster.half_to_srgb8) return SAH_OK;
    std::vector<uint8_t> table(65536);
    for (uint32_t bits = 0; bits < 65536; bits++) {
        const uint32_t sign = bits >> 15, ex = (bits >> 10) & 31u, man = bits & 1023u;
        double v;
        if (ex == 31) v = man ? NAN : INFINITY;
        else if (ex == 0) v = std::ldexp((double)man, -24);
        else v = std::ldexp((double)(man | 1024u), (int)ex - 25);
        if (sign) v = -v;
        uint8_t code = 0;
        if (v > 0.0) {  // NaN and non-positive values encode to 0
            if (v >= 1.0) code = 255;
            else {
                const float s = (float)((v <= 0.0031308) ? 12.92 * v : 1.055 * std::pow(v, 1.0 / 2.4) - 0.055);
                code = !(s > 0.0f) ? 0 : (s >= 1.0f ? 255 : (uint8_t)(s * 255.0f + 0.5f));
            }
        }
        table[bits] = code;
    }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->raster.half_to_srgb8, 65536));
    HIP_TRY(ctx, hipMemcpy(ctx->raster.half_to_srgb8, table.data(), 65536, hipMemcpyHostToDevice));
    return SAH_OK;
}

bool geometry_ok(const sah_scene_geometry* g, bool need_attributes) {
    if (!g) return false;
    if (g->num_primitives == 0) return true;
    if (!g->primitives || !g->indices || !g->vertex_positions) return false;
    if (need_attributes && (!g->vertex_data || !g->materials || g->num_materials == 0)) return false;
    return g->num_primitives < (1u << 24);
}

// Runs both stages; grows the record buffer and repeats stage 1 when the first guess was too small.
int run(sah_ctx* ctx, sah::RasterArgs& a, const sah_scene_geometry* scene, bool gbuffer, uint32_t* stats) {
    auto& r = ctx->raster;
    const uint32_t ntiles = a.tiles_x * a.tiles_y * a.num_views;
    if (!r.host_counters) HIP_TRY(ctx, hipHostMalloc((void**)&r.host_counters, 16 * sizeof(uint32_t)));
    if (int rc = ensure(ctx, S_COUNTERS, 16 * sizeof(uint32_t)); rc != SAH_OK) return rc;
    if (int rc = ensure(ctx, S_TRI_BASE, (size_t)(scene->num_primitives + 1) * sizeof(uint32_t)); rc != SAH_OK) return rc;
    if (int rc = ensure(ctx, S_TILES, (size_t)ntiles * 3 * sizeof(uint32_t)); rc != SAH_OK) return rc;
    // first guess: every index triple is drawn once per view and survives; instanced index ranges or clipping can exceed it
    size_t want_records = (size_t)(scene->num_indices / 3) * a.num_views + 1024, want_clipped = want_records / 8 + 1024;
    for (int attempt = 0; attempt < 3; attempt++) {
        if (int rc = ensure(ctx, S_CLIPQ, want_clipped * sizeof(uint2)); rc != SAH_OK) return rc;
        a.clip_queue = (uint2*)r.ptr[S_CLIPQ];
        a.clip_capacity = (uint32_t)std::min<size_t>(r.bytes[S_CLIPQ] / sizeof(uint2), 0xffffffffu);
        if (int rc = ensure(ctx, S_RECORDS, want_records * sizeof(sah::RasterRecord)); rc != SAH_OK) return rc;
        if (gbuffer)
            if (int rc = ensure(ctx, S_ATTRS, want_records * sizeof(sah::RasterAttr)); rc != SAH_OK) return rc;
        a.counters = (uint32_t*)r.ptr[S_COUNTERS];
        a.tri_base = (uint32_t*)r.ptr[S_TRI_BASE];
        a.records = (sah::RasterRecord*)r.ptr[S_RECORDS];
        a.attrs = (sah::RasterAttr*)r.ptr[S_ATTRS];
        a.record_capacity = (uint32_t)std::min<size_t>(r.bytes[S_RECORDS] / sizeof(sah::RasterRecord), 0xffffffffu);
        if (gbuffer) a.record_capacity = (uint32_t)std::min<size_t>(a.record_capacity, r.bytes[S_ATTRS] / sizeof(sah::RasterAttr));
        a.tile_count = (uint32_t*)r.ptr[S_TILES];
        a.tile_cursor = a.tile_count + ntiles;
        a.tile_offset = a.tile_count + 2 * (size_t)ntiles;
        HIP_TRY(ctx, sah::launch_raster_setup(a, gbuffer, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(r.host_counters, a.counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // records: one slot per (view, triangle) plus the appended fans of the clipped ones
        const size_t need_records = (size_t)r.host_counters[0] * a.num_views + r.host_counters[1];
        if (need_records <= a.record_capacity && r.host_counters[3] <= a.clip_capacity) break;
        if (attempt == 2) return fail(ctx, SAH_ERR_HIP, "rasteriser: scratch buffers still too small after regrowing");
        // a short clip queue also hides records: size both for the worst case of what was seen
        want_clipped = std::max<size_t>(want_clipped, r.host_counters[3]);
        want_records = std::max<size_t>(want_records, need_records + 7 * (size_t)r.host_counters[3]);
    }
    const uint32_t total_tris = r.host_counters[0], pairs = r.host_counters[2];
    if (total_tris >= (1u << 28)) return fail(ctx, SAH_ERR_UNSUPPORTED, "rasteriser: more than 2^28 triangles in one pass");
    if (int rc = ensure(ctx, S_PAIRS, (size_t)(pairs + 1) * sizeof(uint32_t)); rc != SAH_OK) return rc;
    a.pairs = (uint32_t*)r.ptr[S_PAIRS];
    if (gbuffer) {
        if (int rc = ensure(ctx, S_SEQ, ((size_t)total_tris * 8 + 1) * sizeof(uint32_t)); rc != SAH_OK) return rc;
        a.seq_to_record = (uint32_t*)r.ptr[S_SEQ];
    }
    HIP_TRY(ctx, sah::launch_raster_tiles(a, gbuffer, ctx->stream));
    if (stats) HIP_TRY(ctx, hipMemcpyAsync(stats, a.counters + 4, SAH_RASTER_STATS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return SAH_OK;
}

void fill_scene(sah::RasterArgs& a, const sah_scene_geometry* scene) {
    a.positions = scene->vertex_positions;
    a.vertex_data = scene->vertex_data;
    a.indices = scene->indices;
    a.primitives = scene->primitives;
    a.materials = scene->materials;
    a.num_primitives = scene->num_primitives;
    a.num_indices = scene->num_indices;
    a.num_vertices = scene->num_vertices;
    a.num_materials = scene->num_materials;
}
}  // namespace

extern "C" {

int sah_shadow_render(sah_ctx* ctx, const sah_scene_geometry* scene, const sah_sun_light_constants* sun, uint32_t num_cascades,
                      const sah_volume* shadowmap, uint32_t* stats) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!geometry_ok(scene, false) || !sun || num_cascades == 0 || num_cascades > 4) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "shadow_render: bad scene or cascade count");
    if (!shadowmap || !shadowmap->ptr || shadowmap->format != SAH_FORMAT_D16_UNORM) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "shadow_render: the shadow map must be D16_UNORM");
    if (shadowmap->width == 0 || shadowmap->height == 0 || shadowmap->width > kMaxExtent || shadowmap->height > kMaxExtent || shadowmap->depth < num_cascades ||
        (uint64_t)shadowmap->row_pitch_bytes < (uint64_t)shadowmap->width * 2 || (uint64_t)shadowmap->slice_pitch_bytes < (uint64_t)shadowmap->row_pitch_bytes * shadowmap->height ||
        ((uintptr_t)shadowmap->ptr % 2) || (shadowmap->row_pitch_bytes % 2) || (shadowmap->slice_pitch_bytes % 2))
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "shadow_render: shadow map extent (1..%u), layers or pitches", kMaxExtent);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    sah::RasterArgs a{};
    fill_scene(a, scene);
    a.num_views = num_cascades;
    for (uint32_t c = 0; c < num_cascades; c++) std::memcpy(a.clip_matrix[c], sun->cascade_matrices[c], 64);
    a.width = shadowmap->width;
    a.height = shadowmap->height;
    a.half_w = (float)a.width * 0.5f;
    a.half_h = (float)a.height * 0.5f;
    a.tiles_x = (a.width + kTile - 1) / kTile;
    a.tiles_y = (a.height + kTile - 1) / kTile;
    a.shadowmap = varg(*shadowmap);
    return run(ctx, a, scene, false, stats);
}

int sah_gbuffer_render(sah_ctx* ctx, const sah_scene_geometry* scene, const sah_view_data* view, const sah_gbuffer* out, uint32_t* stats) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!geometry_ok(scene, true) || !view || !out) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "gbuffer_render: bad scene, view or targets");
    const uint32_t W = out->depth.width, H = out->depth.height;
    if (W == 0 || H == 0 || W > kMaxExtent || H > kMaxExtent) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "gbuffer_render: extent must be 1..%u", kMaxExtent);
    const struct { const sah_plane* p; uint32_t fmt; uint32_t align; const char* name; } targets[5] = {
        {&out->color, SAH_FORMAT_R8G8B8A8_SRGB, 4, "color"},   {&out->normals, SAH_FORMAT_R16G16B16A16_SFLOAT, 8, "normals"},
        {&out->data, SAH_FORMAT_R8G8B8A8_UNORM, 4, "data"},    {&out->emission, SAH_FORMAT_R8G8B8A8_SRGB, 4, "emission"},
        {&out->depth, SAH_FORMAT_D32_SFLOAT, 4, "depth"}};
    for (const auto& t : targets)
        if (!plane_ok(t.p, t.fmt, t.fmt, W, H) || ((uintptr_t)t.p->ptr % t.align) || (t.p->row_pitch_bytes % t.align))
            return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "gbuffer_render: target '%s' has the wrong format, extent or alignment", t.name);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure_srgb_table(ctx); rc != SAH_OK) return rc;
    sah::RasterArgs a{};
    fill_scene(a, scene);
    a.num_views = 1;
    std::memcpy(a.view_matrix, view->view, 64);
    std::memcpy(a.clip_matrix[0], view->projection, 64);
    a.width = W;
    a.height = H;
    a.half_w = (float)W * 0.5f;
    a.half_h = (float)H * 0.5f;
    a.tiles_x = (W + kTile - 1) / kTile;
    a.tiles_y = (H + kTile - 1) / kTile;
    a.half_to_srgb8 = ctx->raster.half_to_srgb8;
    a.out_color = parg(&out->color);
    a.out_normals = parg(&out->normals);
    a.out_data = parg(&out->data);
    a.out_emission = parg(&out->emission);
    a.out_depth = parg(&out->depth);
    return run(ctx, a, scene, true, stats);
}

}  // extern "C"
