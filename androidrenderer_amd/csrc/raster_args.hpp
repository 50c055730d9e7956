// Kernel-argument block and scratch records of the scene rasteriser (raster.hip, api_raster.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sah_hip.h"
#include "params.hpp"

#ifndef SAH_RASTER_TILE
#define SAH_RASTER_TILE 64  // pixels per tile edge (a multiple of 8, at most 64); 32 measured slower on every scene
#endif

namespace sah {

constexpr uint32_t kRasterTile = SAH_RASTER_TILE;
#ifndef SAH_RASTER_SPLIT
#define SAH_RASTER_SPLIT 256  // bin lists longer than this are cut into parts of this many entries, one workgroup each (64 / 128 measured: dense shadow cascades 50 % / 15 % slower)
#endif
constexpr uint32_t kRasterSplit = SAH_RASTER_SPLIT;

// One window-space triangle of one view: clipped, fanned, snapped to 1/256 pixel, oriented so that its area is positive.
struct RasterRecord {
    int32_t X[3], Y[3];
    float z[3];
    uint32_t view;
    uint16_t x0, x1, y0, y1;  // candidate pixels (centres inside the bounding box), clipped to the viewport; x0 > x1: empty slot
    uint32_t seq;             // G-buffer: draw order, (running triangle number) * 8 + fan index
    uint32_t cutout;          // alpha-tested primitive (every pass: the reference draws masked geometry with the *_masked pipelines)
};
static_assert(sizeof(RasterRecord) == 56, "RasterRecord layout");

// G-buffer pass only: what the fragment stage needs to interpolate the INPUT triangle's varyings.
struct RasterAttr {
    float inv_w[3];
    float bary[3][3];      // barycentric coordinates of the record's vertices in the input triangle (identity unless clipped)
    uint32_t primitive;
    uint32_t material;
    uint32_t seq;          // draw order: (running triangle number) * 8 + fan index
    uint32_t cutout;
    uint16_t vout[3][12];  // fp16 varyings of the input triangle's vertices: colour rgba, normal xyz, tangent xyzw, pad
    float uv[3][2];        // the float2 texcoord varying
};
static_assert(sizeof(RasterAttr) == 160, "RasterAttr layout");

// Shadow pass, CUTOUT records only: what the alpha test of the shadow_masked fragment stage needs (gltf_basic_pbr.slang:181-196 with
// SAH_DEPTH_ONLY, SAH_MASKED: tinted_base_color.a = texel.a * vertex colour.a * tint.a against the opacity threshold).
struct ShadowAttr {
    float inv_w[3];
    float bary[3][3];
    uint16_t alpha[3];  // fp16 vertex colour alpha of the input triangle's vertices
    uint16_t pad;
    uint32_t material;
    uint32_t pad2;
    float uv[3][2];     // texcoords (the alpha may come from the base-colour texture)
    uint32_t pad3[2];
};
static_assert(sizeof(ShadowAttr) == 96, "ShadowAttr layout");

struct RasterArgs {
    // scene
    const float* positions;
    const sah_vertex_data* vertex_data;
    const uint32_t* indices;
    const sah_primitive* primitives;
    const sah_material* materials;
    uint32_t num_primitives, num_indices, num_vertices, num_materials;
    const sah_texture* textures;                     // null: every slot is the material's constant texel
    const sah_material_textures* material_textures;
    uint32_t num_textures;
    float shader_mip_bias;                           // view->material_texture_mip_bias in the G-buffer pass, 0 in the shadow and RSM passes
    const float* luts;                               // 256 sRGB8 -> linear, then 256 UNORM8 -> float
    // views
    uint32_t num_views;
    float view_matrix[16];     // G-buffer: world -> view
    float clip_matrix[4][16];  // shadow: world -> clip per cascade; G-buffer: [0] = projection
    uint32_t width, height;
    float half_w, half_h;
    uint32_t tiles_x, tiles_y;
    // scratch (device)
    uint32_t* counters;  // 16 words: [0] triangles, [1] records, [2] pairs, [3] clip queue, [4..11] stats (of which [9] extra list parts, [10] split tiles), [12] CUTOUT triangles seen by a shadow pass that has no attributes, [13] invalid texture slots / bindings
    uint32_t* tri_base;  // num_primitives
    RasterRecord* records;
    RasterAttr* attrs;
    ShadowAttr* shadow_attrs;  // shadow pass: one per record, written for CUTOUT records; null when the scene came without vertex data / materials
    uint32_t record_capacity;
    uint2* clip_queue;  // (view, running triangle number) of the triangles that cross a clipping plane
    uint32_t clip_capacity;
    uint32_t* tile_count;   // num_views * tiles_y * tiles_x, followed by tile_cursor
    uint32_t* tile_cursor;
    uint32_t* tile_offset;
    uint32_t* pairs;
    uint32_t pairs_capacity;
    uint32_t* seq_to_record;
    uint64_t seq_capacity;
    // long bin lists are cut into parts (k_split): one slot per split tile
    uint32_t* heavy_slot;   // per tile: slot, or ~0 when its list is processed whole
    uint2* extra_parts;     // (tile, part) of the parts beyond the first
    uint32_t extra_capacity;
    uint32_t* tickets;      // per slot: parts that have merged so far
    uint32_t* merge_depth;  // per slot: kRasterTile^2 depth codes (shadow cascades)
    unsigned long long* merge_keys;  // per slot: kRasterTile^2 visibility keys (G-buffer, RSM)
    uint32_t merge_capacity;
    const uint8_t* half_to_srgb8;  // 65536 entries: fp16 bit pattern -> sRGB8 code
    // RSM variant of the G-buffer path (sah_rsm_render): per-view clip matrices, D16 LESS, flux / normal targets
    uint32_t rsm;
    float sun_direction[3];
    VolumeArg rsm_flux, rsm_normals, rsm_depth;
    // outputs
    VolumeArg shadowmap;
    PlaneArg out_color, out_normals, out_data, out_emission, out_depth;
};

}  // namespace sah
