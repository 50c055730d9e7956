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
    uint32_t bucket_base;     // bit pattern >> kTmBucketShift of thresholds[1]
    float thr_lo, thr_hi;     // thresholds[1], thresholds[255]
};

constexpr uint32_t kTmBucketShift = 19, kTmMaxBuckets = 512;

}  // namespace sah
