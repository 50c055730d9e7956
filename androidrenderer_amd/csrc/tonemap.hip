// Tonemap composite (a8) for gfx950: RenderCore/shaders/ui/scene_upsample.frag:20-72, host RenderCore/render/phase/ui_phase.cpp:98-113.
#include <hip/hip_runtime.h>

#include "post_common.hpp"

namespace sah {

// Tonemap composite, LDS-staged.  A 256-thread workgroup produces a 32x32 output tile, four pixels of one column per thread.
// For every bloom mip the texel rectangle the tile can touch (tile bounds mapped into the mip, plus the reach of the tent offsets —
// which are -ix, -iy and +ix in x and +ix, +-iy in y because scene_upsample.frag:29-32 mixes the components of `o`) is copied into
// LDS once, WITH the clamp-to-edge replication applied (cell j holds texel clamp(j), for j from below 0 to beyond the last texel), so
// that the two columns of a bilinear tap are always adjacent cells (one 16-byte LDS read) and no index is clamped per tap; the 4 + 4
// distinct axis set-ups of every column / row of the tile are tabulated in LDS once (they depend on x or on y only); every thread
// then evaluates its 9 taps x 6 mips x 4 pixels from those tables.  The kernel is VALU-issue bound (PMC: ~83 % issue-busy): of the
// ~230 instructions per pixel and mip 108 are the taps' v_fma_mix, 75 the weight products and the tent sum the contract pins; the
// staging and table set-up (~450 instructions per wave) is what the four pixels per thread amortise.
// If any set-up of a mip indexes outside the staged rectangle (never for in-range tiles; kept as a guarantee) the whole workgroup
// takes the global-memory path for that mip.  Same operator sequence per tap as tent_blur(): results are bit-identical.
constexpr int kTmTileW = 32, kTmTileH = 32, kTmPpt = 4, kTmRowStep = kTmTileH / kTmPpt;
constexpr int kTmMipTexels[6] = {704, 320, 192, 192, 192, 192};
constexpr int kTmMaxRows[6] = {32, 24, 16, 16, 16, 16};  // staged rectangles are at most 32 cells wide and this many rows high
constexpr int kTmLdsTexels = 704 + 320 + 4 * 192;
constexpr int kTmAxisPerMip = 4 * kTmTileW + 4 * kTmTileH;

// one bilinear tap from the staged (edge-replicated) rectangle: columns o0 and o0 + 8 bytes, rows ay.o0 and ay.o1
SAH_DEV C3 tap_rep(const char* tex, int xo, float xw0, float xw1, const AxisE& ay) {
    const uint2* r0 = reinterpret_cast<const uint2*>(tex + (ay.o0 + xo));
    const uint2* r1 = reinterpret_cast<const uint2*>(tex + (ay.o1 + xo));
    const uint2 t00 = r0[0], t10 = r0[1], t01 = r1[0], t11 = r1[1];
    const float w00 = xw0 * ay.w0, w10 = xw1 * ay.w0, w01 = xw0 * ay.w1, w11 = xw1 * ay.w1;
    C3 c;
    c.r = fma_mix_lo(w11, t11.x, fma_mix_lo(w01, t01.x, fma_mix_lo(w10, t10.x, fma_mix_lo(w00, t00.x, 0.0f))));
    c.g = fma_mix_hi(w11, t11.x, fma_mix_hi(w01, t01.x, fma_mix_hi(w10, t10.x, fma_mix_hi(w00, t00.x, 0.0f))));
    c.b = fma_mix_lo(w11, t11.y, fma_mix_lo(w01, t01.y, fma_mix_lo(w10, t10.y, fma_mix_lo(w00, t00.y, 0.0f))));
    return c;
}


__global__ void __launch_bounds__(256) k_tonemap(TonemapArgs t) {
    __shared__ uint2 s_tex[kTmLdsTexels];
    __shared__ AxisE s_ax[6][kTmAxisPerMip];  // [m][k*32 + column] (x variants k = 0..3: o0 = column offset, o1 unused), [m][128 + k*32 + row] (y variants)
    __shared__ int s_rect[6][5];              // x0, y0 (may be negative: replicated cells), w, h, lds offset (w == 0: not staged)
    __shared__ int s_bad[6];                  // 1: some set-up of mip m leaves the staged rectangle -> global path
    __shared__ float s_thr[256];              // s_thr[k] = smallest x whose output code is >= k (k = 1..255); s_thr[0] unused
    s_thr[threadIdx.x] = t.thresholds[threadIdx.x];
    const uint32_t bx = blockIdx.x * kTmTileW, by = t.row_begin + blockIdx.y * kTmTileH;
    const uint32_t x_last = min(bx + kTmTileW - 1, t.out_w - 1), y_last = min(by + kTmTileH - 1, t.row_end - 1);
    if (threadIdx.x < 6) {
        const uint32_t m = threadIdx.x;
        int* r = s_rect[m];
        r[0] = r[1] = r[2] = r[3] = 0;
        int off = 0;
        for (uint32_t k = 0; k < m; k++) off += kTmMipTexels[k];
        r[4] = off;
        if (m < t.num_mips) {
            const float W = (float)t.mip_w[m], H = (float)t.mip_h[m];
            // conservative cell bounds: tile extent in mip texels, widened by the largest tap offset (in texels) + 2; cells beyond the
            // image replicate its edge, at most `reach` + 2 of them on a side
            const float reach_x = __builtin_fmaxf(1.0f, W / H) + 2.0f, reach_y = __builtin_fmaxf(1.0f, H / W) + 2.0f;
            const float pu0 = ((float)bx + 0.5f) / (float)t.out_w * W - 0.5f, pu1 = ((float)x_last + 0.5f) / (float)t.out_w * W - 0.5f;
            const float pv0 = (1.0f - ((float)y_last + 0.5f) / (float)t.out_h) * H - 0.5f, pv1 = (1.0f - ((float)by + 0.5f) / (float)t.out_h) * H - 0.5f;
            const int x0 = (int)__builtin_floorf(pu0 - reach_x), x1 = (int)__builtin_floorf(pu1 + reach_x) + 1;
            const int y0 = (int)__builtin_floorf(pv0 - reach_y), y1 = (int)__builtin_floorf(pv1 + reach_y) + 1;
            const int w = x1 - x0 + 1, h = y1 - y0 + 1;
            if (w > 0 && h > 0 && w <= 32 && h <= kTmMaxRows[m] && w * h <= kTmMipTexels[m]) {
                r[0] = x0; r[1] = y0; r[2] = w; r[3] = h;
            }
        }
        s_bad[m] = r[2] == 0;
    }
    __syncthreads();
    // staging: a thread owns column tid % 32 of the rectangle (they are at most 32 cells wide) and every 8th row; all of a thread's
    // (up to 15) texels are requested before the first is stored, instead of one round trip to memory per cell
    {
        constexpr int kIters[6] = {(kTmMaxRows[0] + 7) / 8, (kTmMaxRows[1] + 7) / 8, (kTmMaxRows[2] + 7) / 8, (kTmMaxRows[3] + 7) / 8, (kTmMaxRows[4] + 7) / 8,
                                   (kTmMaxRows[5] + 7) / 8};
        constexpr int kTotal = kIters[0] + kIters[1] + kIters[2] + kIters[3] + kIters[4] + kIters[5];
        uint2 staged[kTotal];
        const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
        int n = 0;
#pragma unroll
        for (int m = 0; m < 6; m++) {
            const bool live = (uint32_t)m < t.num_mips;
            const int x0 = s_rect[m][0], y0 = s_rect[m][1], w = live ? s_rect[m][2] : 0, h = s_rect[m][3];
            const int wmax = (int)t.mip_w[m] - 1, hmax = (int)t.mip_h[m] - 1;
            const int sx = min(max(x0 + tx, 0), wmax);  // CLAMP_TO_EDGE, once per cell
#pragma unroll
            for (int j = 0; j < kIters[m]; j++, n++) {
                const int ty = ty0 + 8 * j;
                const int sy = min(max(y0 + ty, 0), hmax);
                staged[n] = make_uint2(0u, 0u);
                if (tx < w && ty < h) staged[n] = *reinterpret_cast<const uint2*>(t.mips[m].ptr + (size_t)sy * t.mips[m].pitch + (size_t)sx * 8);
            }
        }
        n = 0;
#pragma unroll
        for (int m = 0; m < 6; m++) {
            const int w = (uint32_t)m < t.num_mips ? s_rect[m][2] : 0, h = s_rect[m][3], off = s_rect[m][4];
#pragma unroll
            for (int j = 0; j < kIters[m]; j++, n++) {
                const int ty = ty0 + 8 * j;
                if (tx < w && ty < h) s_tex[off + ty * w + tx] = staged[n];
            }
        }
    }
    // axis tables: thread e builds entry e of every mip — its column (x variants) or row (y variants) coordinate is mip independent;
    // columns / rows past the image edge re-use the last valid one (those pixels are not written)
    {
        static_assert(kTmAxisPerMip == 256, "one table entry per thread and mip");
        const uint32_t i = threadIdx.x;
        const bool is_x = i < 4u * kTmTileW;
        const uint32_t j = is_x ? i : i - 4u * kTmTileW;
        const uint32_t k = is_x ? j / kTmTileW : j / kTmTileH;
        float base;  // u of the column, or v of the row
        if (is_x) base = ((float)min(bx + (j & (kTmTileW - 1)), x_last) + 0.5f) / (float)t.out_w;
        else base = 1.0f - ((float)min(by + (j & (kTmTileH - 1)), y_last) + 0.5f) / (float)t.out_h;
        for (uint32_t m = 0; m < 6 && m < t.num_mips; m++) {
            const uint32_t W = t.mip_w[m], H = t.mip_h[m];
            const float ix = t.mip_inv_w[m], iy = t.mip_inv_h[m];
            const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
            const int rx0 = s_rect[m][0], ry0 = s_rect[m][1], rw = s_rect[m][2], rh_ = s_rect[m][3], off = s_rect[m][4];
            AxisE en;
            bool inside;
            if (is_x) {
                // x variants (scene_upsample.frag:28-36): u, u + o.x, u + o.y, u + o.z
                const float c = k == 0 ? base : base + (k == 1 ? ox : k == 2 ? oy : oz);
                const AxisU a = axis_unclamped(c, W);
                en = {(a.i - rx0) * 8, 0, a.w0, a.w1};
                inside = a.i >= rx0 && a.i + 1 < rx0 + rw;
            } else {
                // y variants: v (+ 0.f), v + o.z, v + o.w, v + o.y
                const float c = base + (k == 0 ? 0.f : k == 1 ? oz : k == 2 ? ow : oy);
                const AxisU a = axis_unclamped(c, H);
                en = {((a.i - ry0) * rw + off) * 8, ((a.i + 1 - ry0) * rw + off) * 8, a.w0, a.w1};
                inside = a.i >= ry0 && a.i + 1 < ry0 + rh_;
            }
            s_ax[m][i] = en;
            if (!inside) s_bad[m] = 1;
        }
    }
    __syncthreads();

    const uint32_t col = threadIdx.x & (kTmTileW - 1), row0 = threadIdx.x / kTmTileW;  // tile rows row0 + 8 k
    const uint32_t x = bx + col;
    if (x >= t.out_w || by + row0 >= t.row_end) return;
    const float u = ((float)x + 0.5f) / (float)t.out_w;
    const char* tex = reinterpret_cast<const char*>(s_tex);
    C3 bloom[kTmPpt];
#pragma unroll
    for (int k = 0; k < kTmPpt; k++) bloom[k] = {0.f, 0.f, 0.f};
    for (uint32_t m = 0; m < 6 && m < t.num_mips; m++) {
        if (!s_bad[m]) {
            const AxisE* ax = s_ax[m];
            const AxisE xa = ax[col], xb = ax[kTmTileW + col], xc = ax[2 * kTmTileW + col], xd = ax[3 * kTmTileW + col];
#pragma unroll
            for (int k = 0; k < kTmPpt; k++) {  // (rows past the band re-use the last valid row's set-ups; their result is dropped)
                const AxisE* ayp = ax + 4 * kTmTileW + row0 + k * kTmRowStep;
                const AxisE ya = ayp[0], yb = ayp[kTmTileH], yc = ayp[2 * kTmTileH], yd = ayp[3 * kTmTileH];
                C3 s = tap_rep(tex, xa.o0, xa.w0, xa.w1, ya) * 4.0f + tap_rep(tex, xb.o0, xb.w0, xb.w1, ya) * 2.0f + tap_rep(tex, xc.o0, xc.w0, xc.w1, ya) * 2.0f +
                       tap_rep(tex, xa.o0, xa.w0, xa.w1, yb) * 2.0f + tap_rep(tex, xa.o0, xa.w0, xa.w1, yc) * 2.0f + tap_rep(tex, xb.o0, xb.w0, xb.w1, yd) * 1.0f +
                       tap_rep(tex, xd.o0, xd.w0, xd.w1, yd) * 1.0f + tap_rep(tex, xb.o0, xb.w0, xb.w1, yc) * 1.0f + tap_rep(tex, xd.o0, xd.w0, xd.w1, yc) * 1.0f;
                s = {s.r / 16.f, s.g / 16.f, s.b / 16.f};
                bloom[k] = bloom[k] + s;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kTmPpt; k++) {
                const uint32_t y = min(by + row0 + k * kTmRowStep, t.row_end - 1);
                bloom[k] = bloom[k] + tent_blur(t.mips[m], t.mip_w[m], t.mip_h[m], u, 1.0f - ((float)y + 0.5f) / (float)t.out_h);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kTmPpt; k++) {
        const uint32_t y = by + row0 + k * kTmRowStep;
        if (y >= t.row_end) break;
        const float v = 1.0f - ((float)y + 0.5f) / (float)t.out_h;
        const Rgba sc = bilinear<ADDR_CLAMP>(t.scene, t.scene_w, t.scene_h, u, v);
        const C3 c = {sc.c[0] + bloom[k].r * 0.014159f, sc.c[1] + bloom[k].g * 0.014159f, sc.c[2] + bloom[k].b * 0.014159f};
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
            uint32_t lo4 = 0;  // 4 * lo; invariant: threshold[lo] <= x, with threshold[0] = -inf; NaN compares false everywhere -> code 0
#pragma unroll
            for (uint32_t step4 = 512; step4 >= 4; step4 >>= 1) {
                const float thr = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_thr) + lo4 + step4);
                lo4 = (rgb[ch] >= thr) ? lo4 + step4 : lo4;
            }
            code[ch] = lo4 >> 2;
        }
        const uint32_t px = code[0] | (code[1] << 8) | (code[2] << 16) | (255u << 24);
        *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(t.out.ptr) + (size_t)y * t.out.pitch + (size_t)x * 4) = px;
    }
}

hipError_t launch_tonemap(const TonemapArgs& t, hipStream_t st) {
    const uint32_t rows = t.row_end - t.row_begin;
    if (rows == 0) return hipSuccess;
    const dim3 grid((t.out_w + kTmTileW - 1) / kTmTileW, (rows + kTmTileH - 1) / kTmTileH);
    hipLaunchKernelGGL(k_tonemap, grid, dim3(256), 0, st, t);
    return hipGetLastError();
}

}  // namespace sah
