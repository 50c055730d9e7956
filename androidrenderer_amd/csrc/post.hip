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

// ---- a7, hot form ------------------------------------------------------------------------------------------------------------
// The 5 boxes x 4 bilinear taps of bloom_downsample.comp:16-52 use only 6 distinct x coordinates (u -+ ix, and those -+ ix again)
// and 6 distinct y coordinates: 12 axis set-ups per output pixel instead of 40, taps accumulated with v_fma_mix_f32 straight from
// the packed fp16 texels.  Same taps, same order, same operators as k_bloom_downsample (kept for planes >= 2 GiB).
__global__ void __launch_bounds__(256) k_bloom_downsample_shared(PlaneArg src, uint32_t sw, uint32_t sh, PlaneArg dst, uint32_t dw, uint32_t dh) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const float ix = 1.0f / (float)sw, iy = 1.0f / (float)sh;
    const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
    const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
    const int pitch = (int)src.pitch;
    auto ax_of = [&](float c) {
        const Axis a = axis_setup(c, sw);
        return AxisE{a.i0 * 8, a.i1 * 8, a.w0, a.w1};
    };
    auto ay_of = [&](float c) {
        const Axis a = axis_setup(c, sh);
        return AxisE{a.i0 * pitch, a.i1 * pitch, a.w0, a.w1};
    };
    const float ua = u + ox, ub = u + oz, vc = v + oy, vd = v + ow;  // box centres of the four corner boxes / taps of the centre box
    const AxisE xa = ax_of(ua), xb = ax_of(ub), xaa = ax_of(ua + ox), xab = ax_of(ua + oz), xba = ax_of(ub + ox), xbb = ax_of(ub + oz);
    const AxisE yc = ay_of(vc), yd = ay_of(vd), ycc = ay_of(vc + oy), ycd = ay_of(vc + ow), ydc = ay_of(vd + oy), ydd = ay_of(vd + ow);
    const char* tex = reinterpret_cast<const char*>(src.ptr);
    auto box = [&](const AxisE& xl, const AxisE& xr, const AxisE& yt, const AxisE& yb) {
        const C3 s = tap_lds(tex, xl, yt) + tap_lds(tex, xr, yt) + tap_lds(tex, xl, yb) + tap_lds(tex, xr, yb);
        return s * 0.25f;
    };
    const C3 s = box(xa, xb, yc, yd) * 0.5f + box(xaa, xab, ycc, ycd) * 0.125f + box(xba, xbb, ycc, ycd) * 0.125f + box(xaa, xab, ydc, ydd) * 0.125f +
                 box(xba, xbb, ydc, ydd) * 0.125f;
    store_rgba16f(dst, (int)x, (int)y, s.r, s.g, s.b, 0.0f);
}

// Tonemap composite, LDS-staged.  A 256-thread workgroup produces a 32x8 output tile.  For every bloom mip the texel
// rectangle the tile can touch (tile bounds mapped into the mip, plus the reach of the tent offsets — which are -ix, -iy
// and +ix in x and +ix, +-iy in y because scene_upsample.frag:29-32 mixes the components of `o`) is copied into LDS once,
// and the 4 + 4 distinct axis set-ups of every column / row of the tile are tabulated in LDS once (they depend on x or on y
// only); every thread then evaluates its 9 taps x 6 mips from those tables.  If any set-up of a mip indexes outside the
// staged rectangle (never for in-range tiles; kept as a guarantee) the whole workgroup takes the global-memory path for
// that mip.  Same operator sequence per tap as tent_blur(): results are bit-identical.
constexpr int kTmTileW = 32, kTmTileH = 8;
constexpr int kTmMip0Texels = 640, kTmMipTexels = 224, kTmLdsTexels = kTmMip0Texels + 5 * kTmMipTexels;
constexpr int kTmAxisPerMip = 4 * kTmTileW + 4 * kTmTileH;

__global__ void __launch_bounds__(256) k_tonemap(TonemapArgs t) {
    __shared__ uint2 s_tex[kTmLdsTexels];
    __shared__ AxisE s_ax[6][kTmAxisPerMip];  // [m][k*32 + column] (x variants k = 0..3), [m][128 + k*8 + row] (y variants)
    __shared__ int s_rect[6][5];              // x0, y0, w, h, lds offset (w == 0: not staged)
    __shared__ int s_bad[6];                  // 1: some set-up of mip m leaves the staged rectangle -> global path
    __shared__ float s_thr[256];              // s_thr[k] = smallest x whose output code is >= k (k = 1..255); s_thr[0] unused
    s_thr[threadIdx.x] = t.thresholds[threadIdx.x];
    const uint32_t bx = blockIdx.x * kTmTileW, by = t.row_begin + blockIdx.y * kTmTileH;
    const uint32_t x_last = min(bx + kTmTileW - 1, t.out_w - 1), y_last = min(by + kTmTileH - 1, t.row_end - 1);
    if (threadIdx.x < 6) {
        const uint32_t m = threadIdx.x;
        int* r = s_rect[m];
        r[0] = r[1] = r[2] = r[3] = 0;
        r[4] = m == 0 ? 0 : kTmMip0Texels + (int)(m - 1) * kTmMipTexels;
        if (m < t.num_mips) {
            const float W = (float)t.mip_w[m], H = (float)t.mip_h[m];
            // conservative texel bounds: tile extent in mip texels, widened by the largest tap offset (in texels) + 2
            const float reach_x = __builtin_fmaxf(1.0f, W / H) + 2.0f, reach_y = __builtin_fmaxf(1.0f, H / W) + 2.0f;
            const float pu0 = ((float)bx + 0.5f) / (float)t.out_w * W - 0.5f, pu1 = ((float)x_last + 0.5f) / (float)t.out_w * W - 0.5f;
            const float pv0 = (1.0f - ((float)y_last + 0.5f) / (float)t.out_h) * H - 0.5f, pv1 = (1.0f - ((float)by + 0.5f) / (float)t.out_h) * H - 0.5f;
            const int x0 = max((int)__builtin_floorf(pu0 - reach_x), 0), x1 = min((int)__builtin_floorf(pu1 + reach_x) + 1, (int)t.mip_w[m] - 1);
            const int y0 = max((int)__builtin_floorf(pv0 - reach_y), 0), y1 = min((int)__builtin_floorf(pv1 + reach_y) + 1, (int)t.mip_h[m] - 1);
            const int w = x1 - x0 + 1, h = y1 - y0 + 1;
            if (w > 0 && h > 0 && w * h <= (m == 0 ? kTmMip0Texels : kTmMipTexels)) {
                r[0] = x0; r[1] = y0; r[2] = w; r[3] = h;
            }
        }
        s_bad[m] = r[2] == 0;
    }
    __syncthreads();
    for (uint32_t m = 0; m < 6 && m < t.num_mips; m++) {
        const int x0 = s_rect[m][0], y0 = s_rect[m][1], w = s_rect[m][2], h = s_rect[m][3], off = s_rect[m][4];
        for (int i = threadIdx.x; i < w * h; i += 256) {
            const int ty = i / w, tx = i - ty * w;
            s_tex[off + i] = *reinterpret_cast<const uint2*>(t.mips[m].ptr + (size_t)(y0 + ty) * t.mips[m].pitch + (size_t)(x0 + tx) * 8);
        }
    }
    // axis tables: entry e of mip m; columns / rows past the image edge re-use the last valid one (those threads exit below)
    for (uint32_t e = threadIdx.x; e < 6u * kTmAxisPerMip; e += 256) {
        const uint32_t m = e / kTmAxisPerMip, i = e - m * kTmAxisPerMip;
        if (m >= t.num_mips) break;
        const uint32_t W = t.mip_w[m], H = t.mip_h[m];
        const float ix = 1.0f / (float)W, iy = 1.0f / (float)H;
        const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
        const int rx0 = s_rect[m][0], ry0 = s_rect[m][1], rw = s_rect[m][2], rh_ = s_rect[m][3], off = s_rect[m][4];
        AxisE en;
        bool inside;
        if (i < 4u * kTmTileW) {
            // x variants (scene_upsample.frag:28-36): u, u + o.x, u + o.y, u + o.z
            const uint32_t k = i / kTmTileW, x = min(bx + (i & (kTmTileW - 1)), x_last);
            const float u = ((float)x + 0.5f) / (float)t.out_w;
            const float c = k == 0 ? u : u + (k == 1 ? ox : k == 2 ? oy : oz);
            const Axis a = axis_setup(c, W);
            en = {(a.i0 - rx0) * 8, (a.i1 - rx0) * 8, a.w0, a.w1};
            inside = a.i0 >= rx0 && a.i1 < rx0 + rw;
        } else {
            // y variants: v (+ 0.f), v + o.z, v + o.w, v + o.y
            const uint32_t j = i - 4u * kTmTileW, k = j / kTmTileH, y = min(by + (j & (kTmTileH - 1)), y_last);
            const float v = 1.0f - ((float)y + 0.5f) / (float)t.out_h;
            const float c = v + (k == 0 ? 0.f : k == 1 ? oz : k == 2 ? ow : oy);
            const Axis a = axis_setup(c, H);
            en = {((a.i0 - ry0) * rw + off) * 8, ((a.i1 - ry0) * rw + off) * 8, a.w0, a.w1};
            inside = a.i0 >= ry0 && a.i1 < ry0 + rh_;
        }
        s_ax[m][i] = en;
        if (!inside) s_bad[m] = 1;
    }
    __syncthreads();

    const uint32_t col = threadIdx.x & (kTmTileW - 1), row = threadIdx.x / kTmTileW;
    const uint32_t x = bx + col, y = by + row;
    if (x >= t.out_w || y >= t.row_end) return;
    const float u = ((float)x + 0.5f) / (float)t.out_w;
    const float v = 1.0f - ((float)y + 0.5f) / (float)t.out_h;
    const char* tex = reinterpret_cast<const char*>(s_tex);
    C3 bloom = {0.f, 0.f, 0.f};
    for (uint32_t m = 0; m < 6 && m < t.num_mips; m++) {
        C3 s;
        if (!s_bad[m]) {
            const AxisE* ax = s_ax[m];
            const AxisE xa = ax[col], xb = ax[kTmTileW + col], xc = ax[2 * kTmTileW + col], xd = ax[3 * kTmTileW + col];
            const AxisE* ayp = ax + 4 * kTmTileW + row;
            const AxisE ya = ayp[0], yb = ayp[kTmTileH], yc = ayp[2 * kTmTileH], yd = ayp[3 * kTmTileH];
            s = tap_lds(tex, xa, ya) * 4.0f + tap_lds(tex, xb, ya) * 2.0f + tap_lds(tex, xc, ya) * 2.0f + tap_lds(tex, xa, yb) * 2.0f +
                tap_lds(tex, xa, yc) * 2.0f + tap_lds(tex, xb, yd) * 1.0f + tap_lds(tex, xd, yd) * 1.0f + tap_lds(tex, xb, yc) * 1.0f +
                tap_lds(tex, xd, yc) * 1.0f;
            s = {s.r / 16.f, s.g / 16.f, s.b / 16.f};
        } else {
            s = tent_blur(t.mips[m], t.mip_w[m], t.mip_h[m], u, v);
        }
        bloom = bloom + s;
    }
    const Rgba sc = bilinear<ADDR_CLAMP>(t.scene, t.scene_w, t.scene_h, u, v);
    const C3 c = {sc.c[0] + bloom.r * 0.014159f, sc.c[1] + bloom.g * 0.014159f, sc.c[2] + bloom.b * 0.014159f};
    const float luma = c.r * 0.2126f + c.g * 0.7152f + c.b * 0.0722f;
    const float factor = luma / (luma + 1.f);
    const C3 mapped = c * factor;
    // pow(x, 1/2.2) -> sRGB OETF -> UNORM8 is a monotone map from fp32 to 256 codes: the host tabulates, by bisection on the
    // exact composite (api_post.cpp: tonemap_code), the smallest input that reaches each code; the device counts thresholds.
    // Two fp64 pow() per channel (~600 issue slots) become an 8-step binary search in LDS.
    const float rgb[3] = {mapped.r, mapped.g, mapped.b};
    uint32_t code[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        uint32_t lo = 0;  // invariant: threshold[lo] <= x, with threshold[0] = -inf; NaN compares false everywhere -> code 0
#pragma unroll
        for (uint32_t step = 128; step >= 1; step >>= 1) lo = (rgb[ch] >= s_thr[lo + step]) ? lo + step : lo;
        code[ch] = lo;
    }
    const uint32_t px = code[0] | (code[1] << 8) | (code[2] << 16) | (255u << 24);
    *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(t.out.ptr) + (size_t)y * t.out.pitch + (size_t)x * 4) = px;
}

// ---- a12 (AO mode Off): clear the R32F target to 1.0 — ambient_occlusion_phase.cpp:167-179 ------------------------------
__global__ void __launch_bounds__(256) k_fill_r32f(PlaneArg dst, uint32_t w, uint32_t h, float value) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    *reinterpret_cast<float*>(const_cast<uint8_t*>(dst.ptr) + (size_t)y * dst.pitch + (size_t)x * 4) = value;
}
hipError_t launch_fill_r32f(const PlaneArg& dst, uint32_t w, uint32_t h, float value, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_r32f, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, st, dst, w, h, value);
    return hipGetLastError();
}

// ---- launchers ---------------------------------------------------------------------------------------------
hipError_t launch_copy_scene(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, hipStream_t st) {
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(k_copy_scene, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh);
    return hipGetLastError();
}
hipError_t launch_bloom_downsample(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, hipStream_t st) {
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    if ((uint64_t)src.pitch * sh < (1ull << 31)) hipLaunchKernelGGL(k_bloom_downsample_shared, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh);
    else hipLaunchKernelGGL(k_bloom_downsample, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh);
    return hipGetLastError();
}
hipError_t launch_tonemap(const TonemapArgs& t, hipStream_t st) {
    const uint32_t rows = t.row_end - t.row_begin;
    if (rows == 0) return hipSuccess;
    const dim3 grid((t.out_w + kTmTileW - 1) / kTmTileW, (rows + kTmTileH - 1) / kTmTileH);
    hipLaunchKernelGGL(k_tonemap, grid, dim3(256), 0, st, t);
    return hipGetLastError();
}

}  // namespace sah
