// Tonemap composite (a8) for gfx950: RenderCore/shaders/ui/scene_upsample.frag:20-72, host RenderCore/render/phase/ui_phase.cpp:98-113.
#include <hip/hip_runtime.h>

#include "post_common.hpp"

namespace sah {

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

hipError_t launch_tonemap(const TonemapArgs& t, hipStream_t st) {
    const uint32_t rows = t.row_end - t.row_begin;
    if (rows == 0) return hipSuccess;
    const dim3 grid((t.out_w + kTmTileW - 1) / kTmTileW, (rows + kTmTileH - 1) / kTmTileH);
    hipLaunchKernelGGL(k_tonemap, grid, dim3(256), 0, st, t);
    return hipGetLastError();
}

}  // namespace sah
