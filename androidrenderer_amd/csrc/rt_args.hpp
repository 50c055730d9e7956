// Argument blocks of the ray tracer (rt.hip, api_rt.cpp): the scene arrays captured by sah_rt_build, the acceleration structure, and
// the two ray generators.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sah_hip.h"
#include "params.hpp"

namespace sah {

constexpr uint32_t kRtFanout = 4;       // children per node, triangles per leaf node
constexpr uint32_t kRtMaxLevels = 15;   // level 0 = one box per triangle, then 4^13 >= kRtMaxTriangles; the walk keeps 4 bits per level in 64
constexpr uint32_t kRtMaxTriangles = 1u << 26;
constexpr uint32_t kRtSortChunk = 2048; // keys one workgroup sorts in LDS

// One world-space triangle, as the traversal reads it: three 16-byte words
//   {v0.xyz, primitive} {v1.xyz, triangle-in-primitive} {v2.xyz, flags}      flags bit 0: CUTOUT primitive
struct RtTriangle {
    float v0[3];
    uint32_t primitive;
    float v1[3];
    uint32_t triangle;
    float v2[3];
    uint32_t flags;
};
static_assert(sizeof(RtTriangle) == 48, "RtTriangle layout");

// The boxes of four sibling nodes (the children of one node of the level above), component by component: what one step of the walk
// loads, as six 16-byte words with nothing unused in them (a 32-byte box per node had 8 bytes of padding: a quarter of the traffic
// through the texture path, which is what the walk saturates)
struct RtNodeGroup {
    float lo[3][kRtFanout], hi[3][kRtFanout];
};
static_assert(sizeof(RtNodeGroup) == 96, "RtNodeGroup layout");

struct RtScene {  // device pointers of the sah_scene_geometry given to sah_rt_build
    const float* positions;
    const sah_vertex_data* vertex_data;
    const uint32_t* indices;
    const sah_primitive* primitives;
    const sah_material* materials;
    uint32_t num_primitives, num_indices, num_vertices, num_materials;
    const sah_texture* textures;
    const sah_material_textures* material_textures;
    uint32_t num_textures;
    const float* luts;  // 256 sRGB8 -> linear, 256 UNORM8 -> float
};

struct RtBvh {
    const RtTriangle* tris;  // Morton order
    const RtNodeGroup* nodes;  // level 0 first (one padded box per triangle, same index): level L has level_count[L] nodes in
                               // ceil(level_count[L] / 4) groups from group level_offset[L] on; node n is lane n % 4 of group n / 4 and
                               // covers the four nodes of group n of level L - 1; the top level has one node
    uint32_t num_tris, num_levels;
    uint32_t level_offset[kRtMaxLevels], level_count[kRtMaxLevels];
    float pad;  // S * 2^-16 (sah_hip.h "pad")
};

// device scalars of a build: [0] running triangles kept, [1] left out, [2] max |coordinate| bits, [3..5] min / [6..8] max of the triangle
// box centres as order-preserving integers
struct RtBuildState {
    uint32_t kept, dropped, max_abs_bits;
    uint32_t cmin[3], cmax[3];
    uint32_t total;  // triangles of all primitives (scan total)
};

struct RtaoArgs {
    PlaneArg depth, normals, noise, out;
    uint32_t width, height, noise_w, noise_h;
    uint32_t row_begin, row_end;  // output rows traced (sah_rt_set_rows)
    float inv_proj[16], inv_view[16];
    float res[2];
    uint32_t samples;
    float max_distance;
};

struct ShadowMaskArgs {
    PlaneArg depth, normals, noise, out;
    uint32_t width, height;
    uint32_t row_begin, row_end;
    float inv_proj[16], inv_view[16];
    float res[2];
    float L[3];  // normalize(-direction), fp32
    float tan_size, num_samples;
    const float* noise_dirs;  // float4 per texel of the noise plane's 128 x 128 corner: normalize(texel.rgb * 2 - 1), 0 (k_noise_dirs, every call)
};

struct GiArgs {  // what the GI hit / miss stages read besides the scene (rt.hip: trace_gi)
    float sun_dir[3];   // direction_and_tan_size.xyz as stored
    float sun_color[3];
    float tan_size;
    PlaneArg noise;     // R8G8B8A8_UNORM, >= 128 x 128
    SkyArgs sky;        // get_sky_color with the sun direction as stored (sky_unified.slang:229)
    uint32_t num_bounces;  // payload.remaining_bounces of the generators' rays (sah_rt_set_bounces; the reference: 0)
};
constexpr int kRtMaxBounces = 2;

struct ProbeTraceArgs {
    GiArgs gi;
    CacheArgs cache;        // cascades + atlases the misses sample from
    const uint32_t* probes; // device: uint3 per probe
    uint32_t num_probes;
    VolumeArg out;          // RGBA16F 20 x 20 x num_probes
};

struct RtgiTraceArgs {
    GiArgs gi;
    PlaneArg depth, normals, ray_buffer, ray_irradiance;
    uint32_t width, height;
    uint32_t row_begin, row_end;
    float inv_proj[16], inv_view[16];
    float res[2];
};

}  // namespace sah
