#pragma once
#include <stdint.h>

#include "params.hpp"

namespace sah {

struct TonemapArgs {
    PlaneArg scene;
    uint32_t scene_w, scene_h;
    PlaneArg mips[8];
    uint32_t mip_w[8], mip_h[8];
    float mip_inv_w[8], mip_inv_h[8];  // 1.0f / (float)mip_w, 1.0f / (float)mip_h (IEEE on the host: the shader's `1.0 / textureSize`)
    uint32_t num_mips;
    PlaneArg out;
    uint32_t out_w, out_h;
    uint32_t row_begin, row_end;
    const float* thresholds;  // device, 256 floats then kTmMaxBuckets bytes: see api_post.cpp tonemap_code(), build_tonemap_buckets()
    const float* code_table;  // device, kTmMaxBuckets x float4: {thresholds[first + 1], [first + 2], [first + 3], first (integer bits)} per bucket — the
                              // same search with one 16-byte read per channel (tonemap_tol.hip reads it from global memory)
    uint32_t bucket_base;     // bit pattern >> kTmBucketShift of thresholds[1]
    float thr_lo, thr_hi;     // thresholds[1], thresholds[255]
    // tolerance mode only (tonemap_tol.hip): the axis set-ups of every output column / row, built once per (output extent, chain extents) by
    // k_tonemap_axis_tables: entry [(mip * 2 + axis) * 4 + variant][column or row], axis 0 = x, stride `axis_stride` entries
    const struct TmAxis* axis_tables;
    uint32_t axis_stride;
};

struct TmAxis {  // texel index of the first of the two taps (unclamped) and the fraction f of the second (y: divided by 16): 8 bytes, the
    int i;       // weights are 1 - f, f (y: 1/16 - f/16, f/16 — the bits of (1 - f) / 16)
    float f;
};

constexpr uint32_t kTmBucketShift = 19, kTmMaxBuckets = 512;

}  // namespace sah
