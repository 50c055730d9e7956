// Post chain for gfx950: "Copy scene" (a13), bloom downsample pyramid (a7), tonemap composite (a8).
//   RenderCore/shaders/util/copy_with_sampler.frag.slang:9-12
//   RenderCore/shaders/postprocessing/bloom_downsample.comp:16-52   (host: RenderCore/render/bloomer.cpp:38-262)
//   RenderCore/shaders/ui/scene_upsample.frag:20-72                 (host: RenderCore/render/phase/ui_phase.cpp:98-113)
// Bilinear filtering is emulated in fp32 exactly as DESIGN.md "Sampling" defines it (CDNA has no filtering
// hardware we could use anyway): p = uv*size - 0.5, i0 = floor(p), f = p - i0, taps clamped per axis.
#include <hip/hip_runtime.h>

#include "../../include/sah_hip.h"
#include "numerics.hpp"
#include "post_args.hpp"

namespace sah {

enum { ADDR_REPEAT = 0, ADDR_CLAMP = 1 };

struct Rgba {
    float c[4];
};

SAH_DEV Rgba load_rgba16f(const PlaneArg& p, int x, int y) {
    const uint2 q = *reinterpret_cast<const uint2*>(p.ptr + (size_t)y * p.pitch + (size_t)x * 8);
    Rgba r;
    r.c[0] = h2f((uint16_t)(q.x & 0xffffu));
    r.c[1] = h2f((uint16_t)(q.x >> 16));
    r.c[2] = h2f((uint16_t)(q.y & 0xffffu));
    r.c[3] = h2f((uint16_t)(q.y >> 16));
    return r;
}

template <int MODE> SAH_DEV int wrap(int i, int n) {
    if (MODE == ADDR_REPEAT) {
        i %= n;
        return i < 0 ? i + n : i;
    }
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}

template <int MODE> SAH_DEV Rgba bilinear(const PlaneArg& p, uint32_t W, uint32_t H, float u, float v) {
    const float px = u * (float)W - 0.5f, py = v * (float)H - 0.5f;
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    const float fx = px - fx0, fy = py - fy0;
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    const int x0 = (int)__builtin_fminf(__builtin_fmaxf(fx0, -1.0e9f), 1.0e9f), y0 = (int)__builtin_fminf(__builtin_fmaxf(fy0, -1.0e9f), 1.0e9f);
    const int xa = wrap<MODE>(x0, (int)W), xb = wrap<MODE>(x0 + 1, (int)W);
    const int ya = wrap<MODE>(y0, (int)H), yb = wrap<MODE>(y0 + 1, (int)H);
    const Rgba t00 = load_rgba16f(p, xa, ya), t10 = load_rgba16f(p, xb, ya), t01 = load_rgba16f(p, xa, yb), t11 = load_rgba16f(p, xb, yb);
    // Vulkan weighted-sum formula, fma chain in tap order (DESIGN.md "Sampling")
    const float w00 = wx0 * wy0, w10 = fx * wy0, w01 = wx0 * fy, w11 = fx * fy;
    Rgba r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float a = __builtin_fmaf(w00, t00.c[i], 0.0f);
        a = __builtin_fmaf(w10, t10.c[i], a);
        a = __builtin_fmaf(w01, t01.c[i], a);
        a = __builtin_fmaf(w11, t11.c[i], a);
        r.c[i] = a;
    }
    return r;
}

SAH_DEV void store_rgba16f(const PlaneArg& p, int x, int y, float r, float g, float b, float a) {
    uint2 q;
    q.x = (uint32_t)f2h(r) | ((uint32_t)f2h(g) << 16);
    q.y = (uint32_t)f2h(b) | ((uint32_t)f2h(a) << 16);
    *reinterpret_cast<uint2*>(const_cast<uint8_t*>(p.ptr) + (size_t)y * p.pitch + (size_t)x * 8) = q;
}

// ---- a13 -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_copy_scene(PlaneArg src, uint32_t sw, uint32_t sh, PlaneArg dst, uint32_t dw, uint32_t dh) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const float inv_w = 1.0f / (float)dw, inv_h = 1.0f / (float)dh;
    const float u = ((float)x + 0.5f) * inv_w, v = ((float)y + 0.5f) * inv_h;
    const Rgba t = bilinear<ADDR_REPEAT>(src, sw, sh, u, v);
    store_rgba16f(dst, (int)x, (int)y, t.c[0], t.c[1], t.c[2], t.c[3]);
}

// ---- a7 --------------------------------------------------------------------------------------------------
struct C3 {
    float r, g, b;
};
SAH_DEV C3 operator+(C3 a, C3 b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
SAH_DEV C3 operator*(C3 a, float s) { return {a.r * s, a.g * s, a.b * s}; }

SAH_DEV C3 tap_clamp(const PlaneArg& p, uint32_t W, uint32_t H, float u, float v) {
    const Rgba t = bilinear<ADDR_CLAMP>(p, W, H, u, v);
    return {t.c[0], t.c[1], t.c[2]};
}

SAH_DEV C3 box_blur(const PlaneArg& p, uint32_t W, uint32_t H, float u, float v, float ix, float iy) {
    const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
    const C3 s = tap_clamp(p, W, H, u + ox, v + oy) + tap_clamp(p, W, H, u + oz, v + oy) + tap_clamp(p, W, H, u + ox, v + ow) +
                 tap_clamp(p, W, H, u + oz, v + ow);
    return s * 0.25f;
}

__global__ void __launch_bounds__(256) k_bloom_downsample(PlaneArg src, uint32_t sw, uint32_t sh, PlaneArg dst, uint32_t dw, uint32_t dh) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const float ix = 1.0f / (float)sw, iy = 1.0f / (float)sh;
    const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
    const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
    const C3 s = box_blur(src, sw, sh, u, v, ix, iy) * 0.5f + box_blur(src, sw, sh, u + ox, v + oy, ix, iy) * 0.125f +
                 box_blur(src, sw, sh, u + oz, v + oy, ix, iy) * 0.125f + box_blur(src, sw, sh, u + ox, v + ow, ix, iy) * 0.125f +
                 box_blur(src, sw, sh, u + oz, v + ow, ix, iy) * 0.125f;
    store_rgba16f(dst, (int)x, (int)y, s.r, s.g, s.b, 0.0f);
}

// ---- a8 --------------------------------------------------------------------------------------------------
SAH_DEV C3 tent_blur(const PlaneArg& p, uint32_t W, uint32_t H, float u, float v) {
    const float ix = 1.0f / (float)W, iy = 1.0f / (float)H;
    const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
    const C3 s = tap_clamp(p, W, H, u, v) * 4.0f + tap_clamp(p, W, H, u + ox, v + 0.f) * 2.0f + tap_clamp(p, W, H, u + oy, v + 0.f) * 2.0f +
                 tap_clamp(p, W, H, u + 0.f, v + oz) * 2.0f + tap_clamp(p, W, H, u + 0.f, v + ow) * 2.0f +
                 tap_clamp(p, W, H, u + ox, v + oy) * 1.0f + tap_clamp(p, W, H, u + oz, v + oy) * 1.0f +
                 tap_clamp(p, W, H, u + ox, v + ow) * 1.0f + tap_clamp(p, W, H, u + oz, v + ow) * 1.0f;
    return {s.r / 16.f, s.g / 16.f, s.b / 16.f};
}

// linear -> sRGB OETF then UNORM8 (hardware write to an sRGB swapchain)
SAH_DEV uint32_t encode_srgb8(float c) {
    if (!(c > 0.0f)) return 0u;  // NaN, negatives, zero
    if (c >= 1.0f) return 255u;
    const double d = (double)c;
    const float s = (float)((d <= 0.0031308) ? 12.92 * d : 1.055 * pow(d, 1.0 / 2.4) - 0.055);
    if (!(s > 0.0f)) return 0u;
    if (s >= 1.0f) return 255u;
    return (uint32_t)(s * 255.0f + 0.5f);
}

__global__ void __launch_bounds__(256) k_tonemap(TonemapArgs t) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = t.row_begin + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= t.out_w || y >= t.row_end) return;
    const float u = ((float)x + 0.5f) / (float)t.out_w;
    const float v = 1.0f - ((float)y + 0.5f) / (float)t.out_h;
    C3 bloom = {0.f, 0.f, 0.f};
    for (uint32_t m = 0; m < 6 && m < t.num_mips; m++) bloom = bloom + tent_blur(t.mips[m], t.mip_w[m], t.mip_h[m], u, v);
    const Rgba sc = bilinear<ADDR_CLAMP>(t.scene, t.scene_w, t.scene_h, u, v);
    const C3 c = {sc.c[0] + bloom.r * 0.014159f, sc.c[1] + bloom.g * 0.014159f, sc.c[2] + bloom.b * 0.014159f};
    const float luma = c.r * 0.2126f + c.g * 0.7152f + c.b * 0.0722f;
    const float factor = luma / (luma + 1.f);
    const C3 mapped = c * factor;
    const double e = (double)(1.f / 2.2f);
    const float rgb[3] = {(float)pow((double)mapped.r, e), (float)pow((double)mapped.g, e), (float)pow((double)mapped.b, e)};
    const uint32_t px = encode_srgb8(rgb[0]) | (encode_srgb8(rgb[1]) << 8) | (encode_srgb8(rgb[2]) << 16) | (255u << 24);
    *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(t.out.ptr) + (size_t)y * t.out.pitch + (size_t)x * 4) = px;
}

// ---- launchers ---------------------------------------------------------------------------------------------
hipError_t launch_copy_scene(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, hipStream_t st) {
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(k_copy_scene, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh);
    return hipGetLastError();
}
hipError_t launch_bloom_downsample(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, hipStream_t st) {
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(k_bloom_downsample, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh);
    return hipGetLastError();
}
hipError_t launch_tonemap(const TonemapArgs& t, hipStream_t st) {
    const uint32_t rows = t.row_end - t.row_begin;
    if (rows == 0) return hipSuccess;
    const dim3 grid((t.out_w + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(k_tonemap, grid, dim3(256), 0, st, t);
    return hipGetLastError();
}

}  // namespace sah
