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
hipError_t launch_extract_vpls(const VolumeArg& flux, const VolumeArg& normals, const VolumeArg& depth, const sah_lpv_cascade_matrices& c, uint32_t cascade,
                               float grid_cell_size, const float* luts, sah_packed_vpl* list, uint32_t* count, void* scratch, hipStream_t st);
hipError_t launch_inject_vpls(const sah_packed_vpl* list, const uint32_t* count, uint32_t capacity, const sah_lpv_cascade_matrices& c, uint32_t cascade,
                              uint32_t num_cascades, const VolumeArg rgb[3], uint32_t* cells_scratch, hipStream_t st);
}  // namespace sah

namespace {
constexpr uint32_t kTile = sah::kRasterTile;
constexpr uint32_t kMaxExtent = 8192;  // keeps every snapped coordinate inside the guard band below 2^24.1
enum Scratch { S_COUNTERS, S_TRI_BASE, S_RECORDS, S_ATTRS, S_TILES, S_PAIRS, S_SEQ, S_CLIPQ, S_VPL_CELLS, S_VPL_CANDIDATES, S_HEAVY, S_EXTRA, S_TICKETS, S_MERGE };

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

// Runs both stages; grows the scratch buffers and repeats the pass when a guess was too small.
int run(sah_ctx* ctx, sah::RasterArgs& a, const sah_scene_geometry* scene, bool gbuffer, uint32_t* stats) {
    auto& r = ctx->raster;
    const uint32_t ntiles = a.tiles_x * a.tiles_y * a.num_views;
    if (!r.host_counters) HIP_TRY(ctx, hipHostMalloc((void**)&r.host_counters, 16 * sizeof(uint32_t)));
    if (int rc = ensure(ctx, S_COUNTERS, 16 * sizeof(uint32_t)); rc != SAH_OK) return rc;
    if (int rc = ensure(ctx, S_TRI_BASE, (size_t)(scene->num_primitives + 1) * sizeof(uint32_t)); rc != SAH_OK) return rc;
    if (int rc = ensure(ctx, S_TILES, (size_t)ntiles * 3 * sizeof(uint32_t)); rc != SAH_OK) return rc;
    // First guesses: every index triple is drawn once per view and survives (instanced index ranges or clipping can exceed it); the
    // bin list and the draw-order table as large as the last call needed.  Both stages are launched back to back and the counters are
    // read once, at the end: if any buffer turned out too small (the kernels never write past a buffer, they only count), it is grown
    // and the pass repeated.  From the second frame of a scene on this is one iteration with no idle gap on the GPU.
    size_t want_records = (size_t)(scene->num_indices / 3) * a.num_views + 1024, want_clipped = want_records / 8 + 1024;
    size_t want_pairs = std::max<size_t>(r.bytes[S_PAIRS] / sizeof(uint32_t), 2 * want_records + 4 * (size_t)ntiles);
    size_t want_seq = gbuffer ? std::max<size_t>(r.bytes[S_SEQ] / sizeof(uint32_t), (size_t)(scene->num_indices / 3) * 8 * a.num_views + 64) : 0;
    for (int attempt = 0; attempt < 4; attempt++) {
        if (int rc = ensure(ctx, S_CLIPQ, want_clipped * sizeof(uint2)); rc != SAH_OK) return rc;
        if (int rc = ensure(ctx, S_RECORDS, want_records * sizeof(sah::RasterRecord)); rc != SAH_OK) return rc;
        // shadow pass: the alpha test of CUTOUT primitives needs their vertex colours and materials (lean records, raster_args.hpp)
        const bool shadow_attrs = !gbuffer && scene->vertex_data && scene->materials && scene->num_materials;
        if (gbuffer)
            if (int rc = ensure(ctx, S_ATTRS, want_records * sizeof(sah::RasterAttr)); rc != SAH_OK) return rc;
        if (shadow_attrs)
            if (int rc = ensure(ctx, S_ATTRS, want_records * sizeof(sah::ShadowAttr)); rc != SAH_OK) return rc;
        if (int rc = ensure(ctx, S_PAIRS, want_pairs * sizeof(uint32_t)); rc != SAH_OK) return rc;
        if (gbuffer)
            if (int rc = ensure(ctx, S_SEQ, want_seq * sizeof(uint32_t)); rc != SAH_OK) return rc;
        a.clip_queue = (uint2*)r.ptr[S_CLIPQ];
        a.clip_capacity = (uint32_t)std::min<size_t>(r.bytes[S_CLIPQ] / sizeof(uint2), 0xffffffffu);
        a.counters = (uint32_t*)r.ptr[S_COUNTERS];
        a.tri_base = (uint32_t*)r.ptr[S_TRI_BASE];
        a.records = (sah::RasterRecord*)r.ptr[S_RECORDS];
        a.attrs = (sah::RasterAttr*)r.ptr[S_ATTRS];
        a.shadow_attrs = shadow_attrs ? (sah::ShadowAttr*)r.ptr[S_ATTRS] : nullptr;
        a.record_capacity = (uint32_t)std::min<size_t>(r.bytes[S_RECORDS] / sizeof(sah::RasterRecord), 0xffffffffu);
        if (gbuffer) a.record_capacity = (uint32_t)std::min<size_t>(a.record_capacity, r.bytes[S_ATTRS] / sizeof(sah::RasterAttr));
        if (shadow_attrs) a.record_capacity = (uint32_t)std::min<size_t>(a.record_capacity, r.bytes[S_ATTRS] / sizeof(sah::ShadowAttr));
        a.tile_count = (uint32_t*)r.ptr[S_TILES];
        a.tile_cursor = a.tile_count + ntiles;
        a.tile_offset = a.tile_count + 2 * (size_t)ntiles;
        a.pairs = (uint32_t*)r.ptr[S_PAIRS];
        a.pairs_capacity = (uint32_t)std::min<size_t>(r.bytes[S_PAIRS] / sizeof(uint32_t), 0xffffffffu);
        a.seq_to_record = (uint32_t*)r.ptr[S_SEQ];
        a.seq_capacity = gbuffer ? r.bytes[S_SEQ] / sizeof(uint32_t) : 0;
        // long bin lists are cut into parts of kRasterSplit entries: at most pairs / kRasterSplit further parts, and a merge buffer per split tile (the
        // number of those is capped: tiles beyond it are processed whole)
        a.extra_capacity = a.pairs_capacity / sah::kRasterSplit + 1u;
        a.merge_capacity = std::min<uint32_t>(a.extra_capacity, ctx->raster_merge_cap);
        const size_t tile_bytes = (size_t)kTile * kTile * (gbuffer ? 8 : 4);
        if (int rc = ensure(ctx, S_HEAVY, (size_t)ntiles * sizeof(uint32_t)); rc != SAH_OK) return rc;
        if (int rc = ensure(ctx, S_EXTRA, (size_t)a.extra_capacity * sizeof(uint2)); rc != SAH_OK) return rc;
        if (int rc = ensure(ctx, S_TICKETS, (size_t)a.merge_capacity * sizeof(uint32_t)); rc != SAH_OK) return rc;
        if (int rc = ensure(ctx, S_MERGE, (size_t)a.merge_capacity * tile_bytes); rc != SAH_OK) return rc;
        a.heavy_slot = (uint32_t*)r.ptr[S_HEAVY];
        a.extra_parts = (uint2*)r.ptr[S_EXTRA];
        a.tickets = (uint32_t*)r.ptr[S_TICKETS];
        a.merge_depth = (uint32_t*)r.ptr[S_MERGE];
        a.merge_keys = (unsigned long long*)r.ptr[S_MERGE];
        HIP_TRY(ctx, sah::launch_raster_setup(a, gbuffer, ctx->stream));
        HIP_TRY(ctx, sah::launch_raster_tiles(a, gbuffer, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(r.host_counters, a.counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const size_t total_tris = r.host_counters[0], clipped = r.host_counters[3], pairs = r.host_counters[2];
        if (total_tris >= (1u << 28)) return fail(ctx, SAH_ERR_UNSUPPORTED, "rasteriser: more than 2^28 triangles in one pass");
        if (r.host_counters[13])
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "rasteriser: %u invalid texture slots or bindings (index beyond the table, more than %d levels, a level "
                        "that is not R8G8B8A8_UNORM / _SRGB, a sampler enum out of range)", r.host_counters[13], SAH_MAX_TEXTURE_MIPS);
        if (r.host_counters[12])
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "shadow_render: the scene has CUTOUT primitives (%u triangles): vertex_data and materials are needed "
                        "for their alpha test (shadow_masked pipeline)", r.host_counters[12]);
        // records: one slot per (view, triangle) plus the appended fans of the clipped ones
        const size_t need_records = total_tris * a.num_views + r.host_counters[1];
        const size_t need_seq = gbuffer ? total_tris * 8 * a.num_views + 1 : 0;
        if (need_records <= a.record_capacity && clipped <= a.clip_capacity && pairs <= a.pairs_capacity && need_seq <= a.seq_capacity) break;
        if (attempt == 3) return fail(ctx, SAH_ERR_HIP, "rasteriser: scratch buffers still too small after regrowing");
        // a short clip queue hides records and a short record buffer hides bin entries: size for the worst case of what was seen
        want_clipped = std::max<size_t>(want_clipped, clipped);
        want_records = std::max<size_t>(want_records, need_records + 7 * clipped);
        want_pairs = std::max<size_t>(want_pairs, pairs + pairs / 4 + 16);
        want_seq = std::max<size_t>(want_seq, need_seq);
    }
    if (stats) HIP_TRY(ctx, hipMemcpyAsync(stats, a.counters + 4, SAH_RASTER_STATS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return SAH_OK;
}

void fill_scene(sah_ctx* ctx, sah::RasterArgs& a, const sah_scene_geometry* scene) {
    a.luts = ctx->luts;
    a.positions = scene->vertex_positions;
    a.vertex_data = scene->vertex_data;
    a.indices = scene->indices;
    a.primitives = scene->primitives;
    a.materials = scene->materials;
    a.num_primitives = scene->num_primitives;
    a.num_indices = scene->num_indices;
    a.num_vertices = scene->num_vertices;
    a.num_materials = scene->num_materials;
    const bool textured = scene->textures && scene->material_textures && scene->num_textures;
    a.textures = textured ? scene->textures : nullptr;
    a.material_textures = textured ? scene->material_textures : nullptr;
    a.num_textures = textured ? scene->num_textures : 0;
    a.shader_mip_bias = 0.0f;
}
}  // namespace

extern "C" {

int sah_shadow_render(sah_ctx* ctx, const sah_scene_geometry* scene, const sah_sun_light_constants* sun, uint32_t num_cascades,
                      const sah_volume* shadowmap, uint32_t* stats) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!geometry_ok(scene, false) || !sun || num_cascades == 0 || num_cascades > 4) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "shadow_render: bad scene or cascade count");
    if (!shadowmap || !shadowmap->ptr || shadowmap->format != SAH_FORMAT_D16_UNORM) return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "shadow_render: the shadow map must be D16_UNORM");
    if (shadowmap->width == 0 || shadowmap->height == 0 || shadowmap->width > kMaxExtent || shadowmap->height > kMaxExtent || shadowmap->depth < num_cascades ||
        (uint64_t)shadowmap->row_pitch_bytes < (uint64_t)shadowmap->width * 2 || (uint64_t)shadowmap->slice_pitch_bytes < (uint64_t)shadowmap->row_pitch_bytes * shadowmap->height ||
        ((uintptr_t)shadowmap->ptr % 2) || (shadowmap->row_pitch_bytes % 2) || (shadowmap->slice_pitch_bytes % 2))
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "shadow_render: shadow map extent (1..%u), layers or pitches", kMaxExtent);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_raster));
    sah::RasterArgs a{};
    fill_scene(ctx, a, scene);
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
    SAH_RANGE();
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
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_raster));
    if (int rc = ensure_srgb_table(ctx); rc != SAH_OK) return rc;
    sah::RasterArgs a{};
    fill_scene(ctx, a, scene);
    a.num_views = 1;
    a.shader_mip_bias = view->material_texture_mip_bias;
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

int sah_rsm_render(sah_ctx* ctx, const sah_scene_geometry* scene, const sah_sun_light_constants* sun, const sah_lpv_cascade_matrices* cascades,
                   uint32_t num_cascades, const sah_rsm_targets* rsm, uint32_t* stats) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!geometry_ok(scene, true) || !sun || !cascades || !rsm || num_cascades == 0 || num_cascades > 4)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "rsm_render: bad scene, sun, cascades or targets");
    const uint32_t W = rsm->depth.width, H = rsm->depth.height;
    if (W == 0 || H == 0 || W > kMaxExtent || H > kMaxExtent) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "rsm_render: extent must be 1..%u", kMaxExtent);
    const struct { const sah_volume* v; uint32_t fmt; uint32_t bpp; const char* name; } targets[3] = {
        {&rsm->flux, SAH_FORMAT_R8G8B8A8_SRGB, 4, "flux"}, {&rsm->normals, SAH_FORMAT_R8G8B8A8_UNORM, 4, "normals"}, {&rsm->depth, SAH_FORMAT_D16_UNORM, 2, "depth"}};
    for (const auto& t : targets)
        if (!t.v->ptr || t.v->format != t.fmt || t.v->width != W || t.v->height != H || t.v->depth < num_cascades ||
            (uint64_t)t.v->row_pitch_bytes < (uint64_t)W * t.bpp || (uint64_t)t.v->slice_pitch_bytes < (uint64_t)t.v->row_pitch_bytes * H ||
            ((uintptr_t)t.v->ptr % t.bpp) || (t.v->row_pitch_bytes % t.bpp) || (t.v->slice_pitch_bytes % t.bpp))
            return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "rsm_render: target '%s' has the wrong format, extent, layers or alignment", t.name);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_raster));
    if (int rc = ensure_srgb_table(ctx); rc != SAH_OK) return rc;
    sah::RasterArgs a{};
    fill_scene(ctx, a, scene);
    a.num_views = num_cascades;
    for (uint32_t c = 0; c < num_cascades; c++) std::memcpy(a.clip_matrix[c], cascades[c].rsm_vp, 64);
    a.width = W;
    a.height = H;
    a.half_w = (float)W * 0.5f;
    a.half_h = (float)H * 0.5f;
    a.tiles_x = (W + kTile - 1) / kTile;
    a.tiles_y = (H + kTile - 1) / kTile;
    a.half_to_srgb8 = ctx->raster.half_to_srgb8;
    a.rsm = 1;
    for (int k = 0; k < 3; k++) a.sun_direction[k] = sun->direction_and_tan_size[k];
    a.rsm_flux = varg(rsm->flux);
    a.rsm_normals = varg(rsm->normals);
    a.rsm_depth = varg(rsm->depth);
    return run(ctx, a, scene, true, stats);
}

int sah_lpv_extract_vpls(sah_ctx* ctx, const sah_rsm_targets* rsm, const sah_lpv_cascade_matrices* cascades, uint32_t cascade_index,
                         float grid_cell_size, sah_packed_vpl* vpl_list, uint32_t* vpl_count) {
    SAH_RANGE();
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!rsm || !cascades || !vpl_list || !vpl_count || cascade_index >= 4) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "lpv_extract_vpls: null argument or cascade index");
    const uint32_t res = rsm->depth.width;
    if (res == 0 || (res % 2) || rsm->depth.height != res || rsm->flux.width != res || rsm->flux.height != res || rsm->normals.width != res ||
        rsm->normals.height != res || cascade_index >= rsm->depth.depth || cascade_index >= rsm->flux.depth || cascade_index >= rsm->normals.depth)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "lpv_extract_vpls: the RSM layers must be square, of even size and cover the cascade");
    if (!rsm->flux.ptr || !rsm->normals.ptr || !rsm->depth.ptr || rsm->flux.format != SAH_FORMAT_R8G8B8A8_SRGB || rsm->normals.format != SAH_FORMAT_R8G8B8A8_UNORM ||
        rsm->depth.format != SAH_FORMAT_D16_UNORM)
        return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "lpv_extract_vpls: RSM formats are RGBA8_SRGB / RGBA8_UNORM / D16_UNORM");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_raster));
    const size_t invocations = (size_t)(res / 2) * (res / 2);
    if (int rc = ensure(ctx, S_VPL_CANDIDATES, invocations * (sizeof(sah_packed_vpl) + sizeof(uint32_t))); rc != SAH_OK) return rc;
    HIP_TRY(ctx, sah::launch_extract_vpls(varg(rsm->flux), varg(rsm->normals), varg(rsm->depth), cascades[cascade_index], cascade_index, grid_cell_size, ctx->luts,
                                          vpl_list, vpl_count, ctx->raster.ptr[S_VPL_CANDIDATES], ctx->stream));
    return SAH_OK;
}

int sah_lpv_inject_vpls(sah_ctx* ctx, const sah_packed_vpl* vpl_list, const uint32_t* vpl_count, uint32_t capacity, const sah_lpv_cascade_matrices* cascades,
                        uint32_t cascade_index, uint32_t num_cascades, const sah_volume rgb[3]) {
    SAH_RANGE();
    if (ctx) sah_drop_lpv_copy(ctx);
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    if (!vpl_list || !vpl_count || !cascades || !rgb || num_cascades == 0 || num_cascades > 4 || cascade_index >= num_cascades)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "lpv_inject_vpls: null argument or cascade index");
    sah::VolumeArg v[3];
    for (int c = 0; c < 3; c++) {
        if (!rgb[c].ptr || rgb[c].format != SAH_FORMAT_R16G16B16A16_SFLOAT || rgb[c].width != rgb[0].width || rgb[c].height != rgb[0].height ||
            rgb[c].depth != rgb[0].depth || rgb[c].width == 0 || (uint64_t)rgb[c].row_pitch_bytes < (uint64_t)rgb[c].width * 8 ||
            (uint64_t)rgb[c].slice_pitch_bytes < (uint64_t)rgb[c].row_pitch_bytes * rgb[c].height || ((uintptr_t)rgb[c].ptr % 8) || (rgb[c].row_pitch_bytes % 8) ||
            (rgb[c].slice_pitch_bytes % 8) || (uint64_t)rgb[c].width * rgb[c].height * rgb[c].depth >= 0xffffffffull)
            return fail(ctx, SAH_ERR_UNSUPPORTED_FORMAT, "lpv_inject_vpls: the three volumes must be RGBA16F of one extent, 8-byte aligned");
        v[c] = varg(rgb[c]);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sah_guard_touch(ctx, ctx->guard_raster));
    // cell index per light, then (16-byte aligned) the 12 blend sources of up to 4096 sorted lights (vpl.hip: k_inject_sorted)
    if (int rc = ensure(ctx, S_VPL_CELLS, ((size_t)capacity + 8) * sizeof(uint32_t) + (size_t)4096 * 12 * sizeof(float)); rc != SAH_OK) return rc;
    HIP_TRY(ctx, sah::launch_inject_vpls(vpl_list, vpl_count, capacity, cascades[cascade_index], cascade_index, num_cascades, v,
                                         (uint32_t*)ctx->raster.ptr[S_VPL_CELLS], ctx->stream));
    return SAH_OK;
}

}  // extern "C"
