// Device helpers shared by the post-chain kernels (post.hip, tonemap.hip): RGBA16F texel access, the emulated bilinear sampler
// (DESIGN.md "Sampling": p = uv * size - 0.5, i0 = floor(p), f = p - i0, weights formed left to right, fma chain in tap order),
// and the tent / box filters of RenderCore/shaders/ui/scene_upsample.frag:20-39 and postprocessing/bloom_downsample.comp:16-36.
#pragma once
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
        // texcoords in (0, 1) put the index in [-1, n]: one conditional add / subtract instead of an integer division (~35 instructions
        // each, four per sample: the copy pass was bound by them); anything further out takes the modulo
        if ((uint32_t)(i + n) < 3u * (uint32_t)n) return i < 0 ? i + n : (i >= n ? i - n : i);
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

// One axis of a CLAMP_TO_EDGE bilinear tap: clamped texel indices and the two weights (1-f, f).
struct Axis {
    int i0, i1;
    float w0, w1;
};
SAH_DEV Axis axis_setup(float coord, uint32_t size) {
    const float p = coord * (float)size - 0.5f;
    const float f0 = __builtin_floorf(p);
    const float f = p - f0;
    const int i = (int)__builtin_fminf(__builtin_fmaxf(f0, -1.0e9f), 1.0e9f);
    Axis a;
    a.i0 = min(max(i, 0), (int)size - 1);
    a.i1 = min(max(i + 1, 0), (int)size - 1);
    a.w0 = 1.0f - f;
    a.w1 = f;
    return a;
}

// unclamped axis set-up: floor index (cells staged in LDS carry the clamp-to-edge replication) and the two weights
struct AxisU {
    int i;
    float w0, w1;
};
SAH_DEV AxisU axis_unclamped(float coord, uint32_t size) {
    const float p = coord * (float)size - 0.5f;
    const float f0 = __builtin_floorf(p);
    const float f = p - f0;
    return {(int)__builtin_fminf(__builtin_fmaxf(f0, -1.0e9f), 1.0e9f), 1.0f - f, f};
}

// Axis set-up as the LDS tap loop consumes it: byte offsets of the two (clamped) texel columns / rows inside the staged
// rectangle and the two weights.  Built once per workgroup for the 32 columns x 4 x-variants and 8 rows x 4 y-variants of
// every mip (tile-shared), instead of 8 set-ups per mip per pixel.
struct AxisE {
    int o0, o1;
    float w0, w1;
};

// One tent tap = one bilinear sample from texels held in LDS (8 bytes each):
// acc = fma(w_k, t_k, acc) from +0 in tap order (t00, t10, t01, t11), conversions folded into v_fma_mix_f32.
SAH_DEV C3 tap_lds(const char* tex, const AxisE& ax, const AxisE& ay) {
    const uint2 t00 = *reinterpret_cast<const uint2*>(tex + (ay.o0 + ax.o0)), t10 = *reinterpret_cast<const uint2*>(tex + (ay.o0 + ax.o1));
    const uint2 t01 = *reinterpret_cast<const uint2*>(tex + (ay.o1 + ax.o0)), t11 = *reinterpret_cast<const uint2*>(tex + (ay.o1 + ax.o1));
    const float w00 = ax.w0 * ay.w0, w10 = ax.w1 * ay.w0, w01 = ax.w0 * ay.w1, w11 = ax.w1 * ay.w1;
    C3 c;
    c.r = fma_mix_lo(w11, t11.x, fma_mix_lo(w01, t01.x, fma_mix_lo(w10, t10.x, fma_mix_lo(w00, t00.x, 0.0f))));
    c.g = fma_mix_hi(w11, t11.x, fma_mix_hi(w01, t01.x, fma_mix_hi(w10, t10.x, fma_mix_hi(w00, t00.x, 0.0f))));
    c.b = fma_mix_lo(w11, t11.y, fma_mix_lo(w01, t01.y, fma_mix_lo(w10, t10.y, fma_mix_lo(w00, t00.y, 0.0f))));
    return c;
}


}  // namespace sah
