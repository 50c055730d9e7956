// Post chain for gfx950: "Copy scene" (a13), bloom downsample pyramid (a7); the tonemap composite (a8) is tonemap.hip.
//   RenderCore/shaders/util/copy_with_sampler.frag.slang:9-12
//   RenderCore/shaders/postprocessing/bloom_downsample.comp:16-52   (host: RenderCore/render/bloomer.cpp:38-262)
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>

#include "post_common.hpp"

namespace sah {

// ---- a13 -------------------------------------------------------------------------------------------------
// Four texels of one column per thread, four rows apart, all sixteen taps in flight before the first store: with one texel per thread the
// pass was 32 400 workgroups of a single load -> filter -> store chain each, bound by workgroup turnover (4.7 TB/s where torch's copy
// kernel streams the same bytes at 6.9, tools/microbench/stream_ceiling.py).
constexpr uint32_t kCopyPpt = 4;  // (8: 0.0274 ms, 4: 0.0252, 1: 0.0284)
__global__ void __launch_bounds__(256) k_copy_scene(PlaneArg src, uint32_t sw, uint32_t sh, PlaneArg dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                                                     uint32_t row_end) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y0 = row_begin + blockIdx.y * (4 * kCopyPpt) + (threadIdx.x >> 6);
    if (x >= dw) return;
    const float inv_w = 1.0f / (float)dw, inv_h = 1.0f / (float)dh;
    const float u = ((float)x + 0.5f) * inv_w;
    Rgba t[kCopyPpt];
#pragma unroll
    for (uint32_t q = 0; q < kCopyPpt; q++) {
        const uint32_t y = y0 + 4 * q;
        if (y < row_end) t[q] = bilinear<ADDR_REPEAT>(src, sw, sh, u, ((float)y + 0.5f) * inv_h);
    }
#pragma unroll
    for (uint32_t q = 0; q < kCopyPpt; q++) {
        const uint32_t y = y0 + 4 * q;
        if (y < row_end) store_rgba16f(dst, (int)x, (int)y, t[q].c[0], t[q].c[1], t[q].c[2], t[q].c[3]);
    }
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

// ---- a7, LDS-staged --------------------------------------------------------------------------------------------------------------
// A 256-thread workgroup produces a 64 x (4 PPT) tile of the destination mip (PPT texels of one column per thread).  The source rectangle
// the tile's 20 taps per texel can touch (2x the tile plus 3 texels either side) is copied to LDS once, with the clamp-to-edge
// replication applied (cell j holds texel clamp(j)), so the two columns of a bilinear tap are adjacent cells (one 16-byte LDS read)
// and no tap clamps an index; the 6 + 6 axis set-ups of every tile column / row are tabulated once per workgroup (one thread per
// column / row: its coordinate costs one divide for all six).  The 5 boxes x 4 bilinear taps of bloom_downsample.comp:16-52 use only
// 6 distinct x and 6 distinct y coordinates (u -+ ix, and those -+ ix again); same taps, same order, same operators as
// k_bloom_downsample: bit-identical.  A tile whose rectangle does not fit (extreme aspect ratios) takes the global-memory form.
// `row_begin/row_end`: destination rows to produce (row-sharded pyramids).
// (PPT = 2 for the large mips; small mips take PPT = 1 — 64x4 tiles — so that their few texels spread over more
// workgroups with shorter dependency chains: 10.5 -> 5 us per launch for the 240x135 mip and below)
constexpr int kBlW = 64, kBlPitch = 136;
// LDS row layout: texel tx of a staged row sits in cell tx + tx / 32 (one empty cell after every 32).  Adjacent destination columns
// read source texels two apart, 16 bytes: with the plain layout lanes i and i + 16 of a ds_read_b64 hit the same two banks (44 % of
// the kernel's LDS cycles were bank conflicts); the gap moves the second sixteen lanes on by two banks.
constexpr int kBlPitchCells = kBlPitch + (kBlPitch - 1) / 32 + 1;
SAH_DEV int bl_cell(int tx) { return tx + (tx >> 5); }
// The five boxes of one destination texel from a staged rectangle: set-ups xs[0..5] = columns of u - ix, u + ix, (u - ix) -+ ix, (u + ix) -+ ix,
// ys likewise (weights pre-scaled by the box weights, see the axis tables).  One body for every LDS-staged form of the pass, so that they
// cannot differ in a rounding.
template <int kPitchCells> SAH_DEV C3 tap_rep_p(const char* tex, const AxisE& ax, const AxisE& ay) {
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const volatile v2u* LdsTexel;
    const LdsTexel c0 = (LdsTexel)(tex + (ay.o0 + ax.o0)), c1 = (LdsTexel)(tex + (ay.o0 + ax.o1));
    const v2u t00 = c0[0], t10 = c1[0], t01 = c0[kPitchCells], t11 = c1[kPitchCells];
    const float w00 = ax.w0 * ay.w0, w10 = ax.w1 * ay.w0, w01 = ax.w0 * ay.w1, w11 = ax.w1 * ay.w1;
    C3 c;
    c.r = fma_mix_lo(w11, t11.x, fma_mix_lo(w01, t01.x, fma_mix_lo(w10, t10.x, fma_mix_lo(w00, t00.x, 0.0f))));
    c.g = fma_mix_hi(w11, t11.x, fma_mix_hi(w01, t01.x, fma_mix_hi(w10, t10.x, fma_mix_hi(w00, t00.x, 0.0f))));
    c.b = fma_mix_lo(w11, t11.y, fma_mix_lo(w01, t01.y, fma_mix_lo(w10, t10.y, fma_mix_lo(w00, t00.y, 0.0f))));
    return c;
}
template <int kPitchCells> SAH_DEV C3 bloom_texel(const char* lds, const AxisE (&xs)[6], const AxisE (&ys)[6]) {
    auto box = [&](const AxisE& xl, const AxisE& xr, const AxisE& yt, const AxisE& yb) {  // (weights pre-scaled: see the table)
        return tap_rep_p<kPitchCells>(lds, xl, yt) + tap_rep_p<kPitchCells>(lds, xr, yt) + tap_rep_p<kPitchCells>(lds, xl, yb) + tap_rep_p<kPitchCells>(lds, xr, yb);
    };
    return box(xs[0], xs[1], ys[0], ys[1]) + box(xs[2], xs[3], ys[2], ys[3]) + box(xs[4], xs[5], ys[2], ys[3]) + box(xs[2], xs[3], ys[4], ys[5]) +
           box(xs[4], xs[5], ys[4], ys[5]);
}

template <int kBlPpt>
__global__ void __launch_bounds__(256) k_bloom_downsample_lds(PlaneArg src, uint32_t sw, uint32_t sh, PlaneArg dst, uint32_t dw, uint32_t dh,
                                                               uint32_t row_begin, uint32_t row_end) {
    constexpr int kBlH = 4 * kBlPpt, kBlRows = 2 * kBlH + 8, kBlTexels = kBlPitch * kBlRows;
    __shared__ uint2 s_tex[kBlPitchCells * kBlRows];
    __shared__ AxisE s_ax[6 * kBlW + 6 * kBlH];  // [k * 64 + column] (o0 = column offset, o1 unused), [384 + k * 16 + row]
    __shared__ int s_rect[4];
    __shared__ int s_bad;
    const uint32_t tid = threadIdx.x;
    const uint32_t bx = blockIdx.x * kBlW, by = row_begin + blockIdx.y * kBlH;
    const uint32_t x_last = min(bx + kBlW - 1, dw - 1), y_last = min(by + kBlH - 1, row_end - 1);
    const float ix = 1.0f / (float)sw, iy = 1.0f / (float)sh;
    const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
    if (tid == 0) {
        const float pu0 = ((float)bx + 0.5f) / (float)dw * (float)sw - 0.5f, pu1 = ((float)x_last + 0.5f) / (float)dw * (float)sw - 0.5f;
        const float pv0 = ((float)by + 0.5f) / (float)dh * (float)sh - 0.5f, pv1 = ((float)y_last + 0.5f) / (float)dh * (float)sh - 0.5f;
        const int x0 = (int)__builtin_floorf(pu0 - 3.0f), x1 = (int)__builtin_floorf(pu1 + 3.0f) + 1;  // cells beyond the image replicate its edge
        const int y0 = (int)__builtin_floorf(pv0 - 3.0f), y1 = (int)__builtin_floorf(pv1 + 3.0f) + 1;
        const int w = x1 - x0 + 1, h = y1 - y0 + 1;
        const bool fits = w > 0 && h > 0 && w <= kBlPitch && h <= kBlRows;
        s_rect[0] = x0; s_rect[1] = y0; s_rect[2] = fits ? w : 0; s_rect[3] = fits ? h : 0;
        s_bad = fits ? 0 : 1;
    }
    __syncthreads();
    const int rx0 = s_rect[0], ry0 = s_rect[1], rw = s_rect[2], rh = s_rect[3];
    // fixed LDS pitch: the row / column split is a division by a constant.  Fully unrolled with the loads first: a thread has its
    // (up to 22) texels in flight together instead of one round trip to memory per cell
    constexpr int kIters = (kBlTexels + 255) / 256;
    uint2 staged[kIters];
#pragma unroll
    for (int it = 0; it < kIters; it++) {
        const int i = (int)tid + it * 256;
        const int ty = i / kBlPitch, tx = i - ty * kBlPitch;
        const int sx = min(max(rx0 + tx, 0), (int)sw - 1), sy = min(max(ry0 + ty, 0), (int)sh - 1);  // CLAMP_TO_EDGE, once per cell
        staged[it] = make_uint2(0u, 0u);
        if (ty < rh && tx < rw) staged[it] = *reinterpret_cast<const uint2*>(src.ptr + (size_t)sy * src.pitch + (size_t)sx * 8);
    }
#pragma unroll
    for (int it = 0; it < kIters; it++) {
        const int i = (int)tid + it * 256;
        const int ty = i / kBlPitch, tx = i - ty * kBlPitch;
        if (i < kBlTexels) s_tex[ty * kBlPitchCells + bl_cell(tx)] = staged[it];
    }
    if (tid < (uint32_t)(kBlW + kBlH)) {  // one thread per tile column / row: six set-ups from one coordinate
        const bool is_x = tid < (uint32_t)kBlW;
        const uint32_t j = is_x ? tid : tid - kBlW;
        const float c = is_x ? ((float)min(bx + j, x_last) + 0.5f) / (float)dw : ((float)min(by + j, y_last) + 0.5f) / (float)dh;
        const float lo = is_x ? ox : oy, hi = is_x ? oz : ow;
        const float ca = c + lo, cb = c + hi;
        const float coords[6] = {ca, cb, ca + lo, ca + hi, cb + lo, cb + hi};
        bool inside = true;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const AxisU a = axis_unclamped(coords[k], is_x ? sw : sh);
            if (is_x) {
                s_ax[k * kBlW + j] = AxisE{bl_cell(a.i - rx0) * 8, bl_cell(a.i + 1 - rx0) * 8, a.w0, a.w1};
                inside = inside && a.i >= rx0 && a.i + 1 < rx0 + rw;
            } else {
                // box weights folded in (powers of two commute with every rounding; nothing comes near the denormals): the centre box
                // (set-ups 0, 1) carries 0.25 * 0.5, the four outer boxes 0.25 * 0.125
                const float scale = k < 2 ? 0.125f : 0.03125f;
                s_ax[6 * kBlW + k * kBlH + j] = AxisE{(a.i - ry0) * kBlPitchCells * 8, 0, a.w0 * scale, a.w1 * scale};
                inside = inside && a.i >= ry0 && a.i + 1 < ry0 + rh;
            }
        }
        if (!inside) s_bad = 1;
    }
    __syncthreads();
    const uint32_t col = tid & 63u, x = bx + col;
    if (x >= dw) return;
    const bool bad = s_bad != 0;
    const char* lds = reinterpret_cast<const char*>(s_tex);
    AxisE xs[6];
    if (!bad) {
#pragma unroll
        for (int k = 0; k < 6; k++) xs[k] = s_ax[k * kBlW + col];
    }
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)kBlPpt; q++) {
        const uint32_t row = (tid >> 6) + 4u * q, y = by + row;
        if (y >= row_end) break;
        C3 s;
        if (!bad) {
            const AxisE* rowp = s_ax + 6 * kBlW + row;
            const AxisE ys[6] = {rowp[0], rowp[kBlH], rowp[2 * kBlH], rowp[3 * kBlH], rowp[4 * kBlH], rowp[5 * kBlH]};
            s = bloom_texel<kBlPitchCells>(lds, xs, ys);
        } else {  // global-memory form: the same 6 + 6 set-ups, computed per texel
            const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
            const int pitch = (int)src.pitch;
            auto ax_of = [&](float c) {
                const Axis a = axis_setup(c, sw);
                return AxisE{a.i0 * 8, a.i1 * 8, a.w0, a.w1};
            };
            auto ay_of = [&](float c) {
                const Axis a = axis_setup(c, sh);
                return AxisE{a.i0 * pitch, a.i1 * pitch, a.w0, a.w1};
            };
            const float ua = u + ox, ub = u + oz, vc = v + oy, vd = v + ow;
            const AxisE xa = ax_of(ua), xb = ax_of(ub), xaa = ax_of(ua + ox), xab = ax_of(ua + oz), xba = ax_of(ub + ox), xbb = ax_of(ub + oz);
            const AxisE yc = ay_of(vc), yd = ay_of(vd), ycc = ay_of(vc + oy), ycd = ay_of(vc + ow), ydc = ay_of(vd + oy), ydd = ay_of(vd + ow);
            const char* tex = reinterpret_cast<const char*>(src.ptr);
            auto box = [&](const AxisE& xl, const AxisE& xr, const AxisE& yt, const AxisE& yb) {
                const C3 b = tap_lds(tex, xl, yt) + tap_lds(tex, xr, yt) + tap_lds(tex, xl, yb) + tap_lds(tex, xr, yb);
                return b * 0.25f;
            };
            s = box(xa, xb, yc, yd) * 0.5f + box(xaa, xab, ycc, ycd) * 0.125f + box(xba, xbb, ycc, ycd) * 0.125f + box(xaa, xab, ydc, ydd) * 0.125f +
                box(xba, xbb, ydc, ydd) * 0.125f;
        }
        store_rgba16f(dst, (int)x, (int)y, s.r, s.g, s.b, 0.0f);
    }
}

// ---- a13 + a7 mip 0 in one pass --------------------------------------------------------------------------------------------------
// "Copy scene" writes `antialiased` and the first bloom dispatch reads it straight back (scene_renderer.cpp:502-527, bloomer.cpp:50-72): two
// launches and 8 B/px of traffic that exist only because they are two draws.  Here the workgroup that produces a 64 x 8 tile of mip 0 computes
// the antialiased texels its taps can touch (136 x 24, edge replication applied — the rectangle k_bloom_downsample_lds would load) from
// `lit` with the copy pass's own sampler arithmetic (bilinear<ADDR_REPEAT>: same operators, tabulated per column / row), keeps them in LDS as
// the fp16 bits it would have loaded, stores the ones it OWNS to `antialiased`, and filters its tile from LDS with bloom_texel().  Texels in
// the overlap of neighbouring tiles are computed by each of them, from the same inputs with the same operators: the same bits, and only the
// owner stores.  Ownership: antialiased rows [2 by, 2 by + 16) x columns [2 bx, 2 bx + 128) of the tile at (bx, by); the first / last tile
// row of the launch also owns the rows from aa_row_begin / up to aa_row_end (the rows its neighbours' taps and the composite need from
// beyond the mip rows of a row-sharded frame — they lie inside its rectangle: the host checks), the last tile column the odd column.
struct CopyAxis {  // one axis of the copy's bilinear tap: byte offsets of the two (wrapped) texels and the two weights
    uint32_t o0, o1;
    float w0, w1;
};
struct CopyBloomArgs {
    PlaneArg lit, aa, mip0;
    uint32_t lw, lh, aw, ah, mw, mh;
    uint32_t mip_row_begin, mip_row_end, aa_row_begin, aa_row_end;
};
template <int kBlPpt>
__global__ void __launch_bounds__(256) k_copy_bloom_mip0(const CopyBloomArgs g) {
    constexpr int kBlH = 4 * kBlPpt, kBlRows = 2 * kBlH + 8, kBlTexels = kBlPitch * kBlRows;
    __shared__ uint2 s_tex[kBlPitchCells * kBlRows];
    __shared__ AxisE s_ax[6 * kBlW + 6 * kBlH];
    __shared__ CopyAxis s_cx[kBlPitch], s_cy[kBlRows];
    const uint32_t tid = threadIdx.x;
    const uint32_t bx = blockIdx.x * kBlW, by = g.mip_row_begin + blockIdx.y * kBlH;
    const uint32_t x_last = min(bx + kBlW - 1, g.mw - 1), y_last = min(by + kBlH - 1, g.mip_row_end - 1);
    const uint32_t sw = g.aw, sh = g.ah, dw = g.mw, dh = g.mh;  // the bloom pass's source (antialiased) and destination (mip 0)
    const float ix = 1.0f / (float)sw, iy = 1.0f / (float)sh;
    const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
    // the rectangle of k_bloom_downsample_lds (uniform, every thread)
    const float pu0 = ((float)bx + 0.5f) / (float)dw * (float)sw - 0.5f, pu1 = ((float)x_last + 0.5f) / (float)dw * (float)sw - 0.5f;
    const float pv0 = ((float)by + 0.5f) / (float)dh * (float)sh - 0.5f, pv1 = ((float)y_last + 0.5f) / (float)dh * (float)sh - 0.5f;
    const int rx0 = (int)__builtin_floorf(pu0 - 3.0f), rx1 = (int)__builtin_floorf(pu1 + 3.0f) + 1;
    const int ry0 = (int)__builtin_floorf(pv0 - 3.0f), ry1 = (int)__builtin_floorf(pv1 + 3.0f) + 1;
    const int rw = rx1 - rx0 + 1, rh = ry1 - ry0 + 1;  // (<= kBlPitch x kBlRows: host check)
    // 1. the copy's axis set-ups for the rectangle's columns / rows (cell j stands for antialiased texel clamp(r0 + j)) and the bloom
    //    pass's six set-ups per tile column / row
    if (tid < (uint32_t)(kBlPitch + kBlRows)) {
        const bool is_x = tid < (uint32_t)kBlPitch;
        const int j = is_x ? (int)tid : (int)tid - kBlPitch;
        const int t = is_x ? min(max(rx0 + j, 0), (int)sw - 1) : min(max(ry0 + j, 0), (int)sh - 1);
        // copy_with_sampler.frag.slang: uv = SV_Position.xy * inverse_resolution, linear sampler with REPEAT addressing (k_copy_scene)
        const float inv = 1.0f / (float)(is_x ? sw : sh);
        const float c = ((float)t + 0.5f) * inv;
        const uint32_t n = is_x ? g.lw : g.lh;
        const float p = c * (float)n - 0.5f;
        const float f0 = __builtin_floorf(p);
        const float f = p - f0;
        const int i0 = (int)__builtin_fminf(__builtin_fmaxf(f0, -1.0e9f), 1.0e9f);
        const uint32_t a = (uint32_t)wrap<ADDR_REPEAT>(i0, (int)n), b = (uint32_t)wrap<ADDR_REPEAT>(i0 + 1, (int)n);
        const uint32_t stride = is_x ? 8u : g.lit.pitch;
        (is_x ? s_cx : s_cy)[j] = CopyAxis{a * stride, b * stride, 1.0f - f, f};
    }
    {
        const uint32_t u = tid;  // bloom tables by the first kBlW + kBlH threads, exactly as k_bloom_downsample_lds
        if (u < (uint32_t)(kBlW + kBlH)) {
            const bool is_x = u < (uint32_t)kBlW;
            const uint32_t j = is_x ? u : u - kBlW;
            const float c = is_x ? ((float)min(bx + j, x_last) + 0.5f) / (float)dw : ((float)min(by + j, y_last) + 0.5f) / (float)dh;
            const float lo = is_x ? ox : oy, hi = is_x ? oz : ow;
            const float ca = c + lo, cb = c + hi;
            const float coords[6] = {ca, cb, ca + lo, ca + hi, cb + lo, cb + hi};
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const AxisU a = axis_unclamped(coords[k], is_x ? sw : sh);
                if (is_x) {
                    s_ax[k * kBlW + j] = AxisE{bl_cell(a.i - rx0) * 8, bl_cell(a.i + 1 - rx0) * 8, a.w0, a.w1};
                } else {
                    const float scale = k < 2 ? 0.125f : 0.03125f;
                    s_ax[6 * kBlW + k * kBlH + j] = AxisE{(a.i - ry0) * kBlPitchCells * 8, 0, a.w0 * scale, a.w1 * scale};
                }
            }
        }
    }
    __syncthreads();
    // 2. the antialiased texels of the rectangle: seven cells per round and thread, their taps in flight together
    {
        const int own_x0 = 2 * (int)bx, own_x1 = blockIdx.x + 1 == gridDim.x ? (int)sw : 2 * (int)(bx + kBlW);
        const int own_y0 = blockIdx.y == 0 ? (int)g.aa_row_begin : max(2 * (int)by, (int)g.aa_row_begin);
        const int own_y1 = blockIdx.y + 1 == gridDim.y ? (int)g.aa_row_end : min(2 * (int)(by + kBlH), (int)g.aa_row_end);
        constexpr int kCells = 7, kRounds = (kBlTexels + 256 * kCells - 1) / (256 * kCells);  // 13 cells per thread in two rounds of 7 x 4 taps in flight (4: 57.9 us at 4K)
        const uint8_t* lit = g.lit.ptr;
        for (int round = 0; round < kRounds; round++) {
            uint2 t[kCells][4];
            CopyAxis cx[kCells], cy[kCells];
            int cell_x[kCells], cell_y[kCells];
#pragma unroll
            for (int q = 0; q < kCells; q++) {
                const int i = (int)tid + (round * kCells + q) * 256;
                const int ty = min(i / kBlPitch, kBlRows - 1), tx = i - (i / kBlPitch) * kBlPitch;  // (cells past the rectangle: a valid address, dropped below)
                cell_x[q] = i < kBlTexels && tx < rw && ty < rh ? tx : -1;
                cell_y[q] = ty;
                cx[q] = s_cx[tx];
                cy[q] = s_cy[ty];
                t[q][0] = *reinterpret_cast<const uint2*>(lit + (cy[q].o0 + cx[q].o0));
                t[q][1] = *reinterpret_cast<const uint2*>(lit + (cy[q].o0 + cx[q].o1));
                t[q][2] = *reinterpret_cast<const uint2*>(lit + (cy[q].o1 + cx[q].o0));
                t[q][3] = *reinterpret_cast<const uint2*>(lit + (cy[q].o1 + cx[q].o1));
            }
#pragma unroll
            for (int q = 0; q < kCells; q++) {
                if (cell_x[q] < 0) continue;
                // bilinear<ADDR_REPEAT>: weights formed as there, fma chain in tap order from +0, four channels
                const float w[4] = {cx[q].w0 * cy[q].w0, cx[q].w1 * cy[q].w0, cx[q].w0 * cy[q].w1, cx[q].w1 * cy[q].w1};
                float ch[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float a = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t word = c < 2 ? t[q][k].x : t[q][k].y;
                        a = __builtin_fmaf(w[k], h2f((uint16_t)((c & 1) ? word >> 16 : word & 0xffffu)), a);
                    }
                    ch[c] = a;
                }
                uint2 v;
                v.x = (uint32_t)f2h(ch[0]) | ((uint32_t)f2h(ch[1]) << 16);
                v.y = (uint32_t)f2h(ch[2]) | ((uint32_t)f2h(ch[3]) << 16);
                s_tex[cell_y[q] * kBlPitchCells + bl_cell(cell_x[q])] = v;
                const int ax = rx0 + cell_x[q], ay = ry0 + cell_y[q];  // the texel the cell stands for, when inside the image
                if (ax >= own_x0 && ax < own_x1 && ay >= own_y0 && ay < own_y1)
                    *reinterpret_cast<uint2*>(const_cast<uint8_t*>(g.aa.ptr) + (size_t)ay * g.aa.pitch + (size_t)ax * 8) = v;
            }
        }
    }
    __syncthreads();
    // 3. the mip 0 tile, as k_bloom_downsample_lds filters it
    const uint32_t col = tid & 63u, x = bx + col;
    if (x >= dw) return;
    const char* lds = reinterpret_cast<const char*>(s_tex);
    AxisE xs[6];
#pragma unroll
    for (int k = 0; k < 6; k++) xs[k] = s_ax[k * kBlW + col];
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)kBlPpt; q++) {
        const uint32_t row = (tid >> 6) + 4u * q, y = by + row;
        if (y >= g.mip_row_end) break;
        const AxisE* rowp = s_ax + 6 * kBlW + row;
        const AxisE ys[6] = {rowp[0], rowp[kBlH], rowp[2 * kBlH], rowp[3 * kBlH], rowp[4 * kBlH], rowp[5 * kBlH]};
        const C3 s = bloom_texel<kBlPitchCells>(lds, xs, ys);
        store_rgba16f(g.mip0, (int)x, (int)y, s.r, s.g, s.b, 0.0f);
    }
}

// host: every tile's rectangle fits and holds its set-ups (the kernel's arithmetic), the rows asked for beyond the mip rows lie inside the
// first / last tile row's rectangle, and the extents are those of a half-resolution mip (ownership)
template <int kBlPpt> static bool copy_bloom_fits(const CopyBloomArgs& g) {
    constexpr int kBlH = 4 * kBlPpt, kBlRows = 2 * kBlH + 8;
    if (g.mw == 0 || g.mh == 0 || (g.aw != 2 * g.mw && g.aw != 2 * g.mw + 1) || (g.ah != 2 * g.mh && g.ah != 2 * g.mh + 1)) return false;
    if (g.lw > 32768 || g.lh > 32768 || (uint64_t)g.lit.pitch * g.lh >= (1ull << 31)) return false;  // 32-bit byte offsets in the copy's tables
    auto axis_ok = [](uint32_t d0, uint32_t d1, uint32_t dst, uint32_t src, int cap, int* r0, int* r1) {
        const float p0 = ((float)d0 + 0.5f) / (float)dst * (float)src - 0.5f, p1 = ((float)d1 + 0.5f) / (float)dst * (float)src - 0.5f;
        const int a = (int)__builtin_floorf(p0 - 3.0f), b = (int)__builtin_floorf(p1 + 3.0f) + 1;
        *r0 = a;
        *r1 = b;
        if (b - a + 1 > cap) return false;
        const float i = 1.0f / (float)src, lo = i * -1.0f, hi = i * 1.0f;
        for (uint32_t t = d0; t <= d1; t++) {
            const float c = ((float)t + 0.5f) / (float)dst, ca = c + lo, cb = c + hi;
            const float coords[6] = {ca, cb, ca + lo, ca + hi, cb + lo, cb + hi};
            for (float co : coords) {
                const float p = co * (float)src - 0.5f;
                const int idx = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(p), -1.0e9f), 1.0e9f);
                if (idx < a || idx + 1 > b) return false;
            }
        }
        return true;
    };
    int r0, r1;
    for (uint32_t b0 = 0; b0 < g.mw; b0 += kBlW) {
        const uint32_t b1 = std::min(b0 + kBlW - 1, g.mw - 1);
        if (!axis_ok(b0, b1, g.mw, g.aw, kBlPitch, &r0, &r1)) return false;
        const int own1 = (b0 + kBlW >= g.mw ? (int)g.aw : 2 * (int)(b0 + kBlW)) - 1;
        if (2 * (int)b0 < r0 || own1 > r1) return false;
    }
    for (uint32_t b0 = g.mip_row_begin; b0 < g.mip_row_end; b0 += kBlH) {
        const uint32_t b1 = std::min(b0 + kBlH - 1, g.mip_row_end - 1);
        if (!axis_ok(b0, b1, g.mh, g.ah, kBlRows, &r0, &r1)) return false;
        const int own0 = b0 == g.mip_row_begin ? (int)g.aa_row_begin : 2 * (int)b0;
        const int own1 = (b0 + kBlH >= g.mip_row_end ? (int)g.aa_row_end : 2 * (int)(b0 + kBlH)) - 1;
        if (own0 < r0 || own1 > r1) return false;
    }
    return true;
}

// antialiased rows [aa_row_begin, aa_row_end) and mip 0 rows [mip_row_begin, mip_row_end) in one launch; false: not in this form (the caller
// runs the two passes)
bool launch_copy_bloom_mip0(const PlaneArg& lit, uint32_t lw, uint32_t lh, const PlaneArg& aa, uint32_t aw, uint32_t ah, const PlaneArg& mip0, uint32_t mw, uint32_t mh,
                            uint32_t mip_row_begin, uint32_t mip_row_end, uint32_t aa_row_begin, uint32_t aa_row_end, hipStream_t st, hipError_t* err) {
    *err = hipSuccess;
    if (mip_row_end <= mip_row_begin || (uint64_t)aa.pitch * ah >= (1ull << 31)) return false;
    const CopyBloomArgs g{lit, aa, mip0, lw, lh, aw, ah, mw, mh, mip_row_begin, mip_row_end, aa_row_begin, aa_row_end};
    constexpr int kPpt = 2;
    // the fit check walks every tile column and row of the launch (about 18,000 set-ups at 4K): its answer depends on the extents, the lit
    // pitch and the row ranges only, and a frame loop asks the same question every frame — the last answer is kept per calling thread
    struct FitKey {
        uint32_t v[11];
        bool ok;
    };
    static thread_local FitKey last = {{0}, false};
    const uint32_t key[11] = {lw, lh, aw, ah, mw, mh, mip_row_begin, mip_row_end, aa_row_begin, aa_row_end, lit.pitch};
    if (memcmp(last.v, key, sizeof(key)) != 0) {
        memcpy(last.v, key, sizeof(key));
        last.ok = copy_bloom_fits<kPpt>(g);
    }
    if (!last.ok) return false;
    const uint32_t rows = mip_row_end - mip_row_begin;
    hipLaunchKernelGGL(k_copy_bloom_mip0<kPpt>, dim3((mw + kBlW - 1) / kBlW, (rows + 4 * kPpt - 1) / (4 * kPpt)), dim3(256), 0, st, g);
    *err = hipGetLastError();
    return true;
}

// ---- a7, two mips per launch -------------------------------------------------------------------------------------------------------
// bloomer.cpp:50-151 makes mip i + 1 from mip i, one dispatch each; the small mips of the chain are launches of a few thousand texels
// whose time is their own dependency chain (dispatch -> staging round trip -> filter -> store, ~5 us) plus the boundary to the next.
// k_bloom_pair produces mips A and B = A + 1 from mip S = A - 1 in ONE launch without any exchange between workgroups: a workgroup owns
// a TB x TB tile of B, computes every texel of A that tile's taps can touch (2 TB + 8 square, kept in LDS as fp16 exactly as it would
// be stored) from the texels of S it staged (4 TB + 24 square), stores the part of A it owns, and filters its B tile from the LDS copy.
// Texels of A in the overlap of neighbouring tiles are computed by several workgroups — from the same inputs by the same operators
// (bloom_texel), so all copies are the same bits, and only the owner stores.  The host checks, with the kernel's own rectangle
// arithmetic (pair_rect: the same fp32 operators on both sides), that every rectangle fits before it takes this path.
struct PairRect {
    int x0, w;  // first texel (unclamped) and number of cells along one axis
};
// texels of a source axis (size `src`) that the taps of destination texels [d0, d1] (axis size `dst`) can touch, as
// k_bloom_downsample_lds computes its rectangle: cells beyond the image replicate its edge
__host__ __device__ inline PairRect pair_rect(uint32_t d0, uint32_t d1, uint32_t dst, uint32_t src) {
    const float p0 = ((float)d0 + 0.5f) / (float)dst * (float)src - 0.5f, p1 = ((float)d1 + 0.5f) / (float)dst * (float)src - 0.5f;
    const int a = (int)__builtin_floorf(p0 - 3.0f), b = (int)__builtin_floorf(p1 + 3.0f) + 1;
    return {a, b - a + 1};
}
__host__ __device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int TB> struct PairDims {
    static constexpr int kAW = 2 * TB + 8;               // cells of mip A per axis
    static constexpr int kSW = 2 * kAW + 8;              // cells of mip S per axis
    static constexpr int kAPitch = kAW + (kAW - 1) / 32 + 1, kSPitch = kSW + (kSW - 1) / 32 + 1;  // bl_cell() gaps, as the one-mip kernel
};

struct BloomPairArgs {
    PlaneArg s, a, b;
    uint32_t sw, sh, aw, ah, bw, bh;
    uint32_t tiles_x, tiles_y;
};

// NT threads: 512 for the 8 x 8 tiles, whose 24 x 24 rectangle of A is then one round of work per thread instead of three (the rounds
// are the kernel's critical path: 12 -> ~8 us for mips 2 + 3 of a 4K chain)
template <int TB, int NT>
__global__ void __launch_bounds__(NT) k_bloom_pair(const BloomPairArgs g) {
    using D = PairDims<TB>;
    __shared__ uint2 s_src[D::kSPitch * D::kSW];
    __shared__ uint2 s_mid[D::kAPitch * D::kAW];
    __shared__ AxisE s_ax[6 * D::kAW], s_ay[6 * D::kAW];  // level A set-ups per cell column / row of the A rectangle
    __shared__ AxisE s_bx[6 * TB], s_by[6 * TB];          // level B set-ups per tile column / row
    const uint32_t tid = threadIdx.x;
    const uint32_t bx = blockIdx.x * TB, by = blockIdx.y * TB;
    const uint32_t bx_last = min(bx + TB - 1, g.bw - 1), by_last = min(by + TB - 1, g.bh - 1);
    // rectangles (uniform; every thread computes them: a handful of scalar-like operations against a barrier and an LDS round trip)
    const PairRect ax = pair_rect(bx, bx_last, g.bw, g.aw), ay = pair_rect(by, by_last, g.bh, g.ah);
    const int ax_lo = clampi(ax.x0, 0, (int)g.aw - 1), ax_hi = clampi(ax.x0 + ax.w - 1, 0, (int)g.aw - 1);
    const int ay_lo = clampi(ay.x0, 0, (int)g.ah - 1), ay_hi = clampi(ay.x0 + ay.w - 1, 0, (int)g.ah - 1);
    const PairRect sx = pair_rect((uint32_t)ax_lo, (uint32_t)ax_hi, g.aw, g.sw), sy = pair_rect((uint32_t)ay_lo, (uint32_t)ay_hi, g.ah, g.sh);

    // 1. stage S (edge replication applied), all loads of a thread in flight together
    constexpr int kSCells = D::kSW * D::kSW, kSIters = (kSCells + NT - 1) / NT;
    {
        uint2 staged[kSIters];
#pragma unroll
        for (int it = 0; it < kSIters; it++) {
            const int i = (int)tid + it * NT;
            const int ty = i / D::kSW, tx = i - ty * D::kSW;
            const int gx = clampi(sx.x0 + tx, 0, (int)g.sw - 1), gy = clampi(sy.x0 + ty, 0, (int)g.sh - 1);
            staged[it] = make_uint2(0u, 0u);
            if (ty < sy.w && tx < sx.w) staged[it] = *reinterpret_cast<const uint2*>(g.s.ptr + (size_t)gy * g.s.pitch + (size_t)gx * 8);
        }
#pragma unroll
        for (int it = 0; it < kSIters; it++) {
            const int i = (int)tid + it * NT;
            const int ty = i / D::kSW, tx = i - ty * D::kSW;
            if (i < kSCells) s_src[ty * D::kSPitch + bl_cell(tx)] = staged[it];
        }
    }
    // 2. axis set-ups.  Level A: one thread per cell column / row of the A rectangle (cell j stands for texel clamp(ax.x0 + j)); level B:
    //    one thread per tile column / row.  Six set-ups from one coordinate, exactly as k_bloom_downsample_lds tabulates them
    auto six = [&](float c, float lo, float hi, uint32_t size, int origin, int pitch_cells, bool is_x, AxisE* out, int stride, int j) {
        const float ca = c + lo, cb = c + hi;
        const float coords[6] = {ca, cb, ca + lo, ca + hi, cb + lo, cb + hi};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const AxisU a = axis_unclamped(coords[k], size);
            if (is_x) {
                out[k * stride + j] = AxisE{bl_cell(a.i - origin) * 8, bl_cell(a.i + 1 - origin) * 8, a.w0, a.w1};
            } else {
                const float scale = k < 2 ? 0.125f : 0.03125f;  // box weights folded in (see k_bloom_downsample_lds)
                out[k * stride + j] = AxisE{(a.i - origin) * pitch_cells * 8, 0, a.w0 * scale, a.w1 * scale};
            }
        }
    };
    {
        const float six_ = 1.0f / (float)g.sw, siy = 1.0f / (float)g.sh;
        if (tid < (uint32_t)D::kAW) {
            const int t = clampi(ax.x0 + (int)tid, 0, (int)g.aw - 1);
            six(((float)t + 0.5f) / (float)g.aw, six_ * -1.0f, six_ * 1.0f, g.sw, sx.x0, D::kSPitch, true, s_ax, D::kAW, (int)tid);
        } else if (tid >= 64u && tid < 64u + (uint32_t)D::kAW) {
            const int j = (int)tid - 64;
            const int t = clampi(ay.x0 + j, 0, (int)g.ah - 1);
            six(((float)t + 0.5f) / (float)g.ah, siy * -1.0f, siy * 1.0f, g.sh, sy.x0, D::kSPitch, false, s_ay, D::kAW, j);
        } else if (tid >= 128u && tid < 128u + (uint32_t)TB) {
            const int j = (int)tid - 128;
            const float aix = 1.0f / (float)g.aw;
            six(((float)min(bx + (uint32_t)j, bx_last) + 0.5f) / (float)g.bw, aix * -1.0f, aix * 1.0f, g.aw, ax.x0, D::kAPitch, true, s_bx, TB, j);
        } else if (tid >= 192u && tid < 192u + (uint32_t)TB) {
            const int j = (int)tid - 192;
            const float aiy = 1.0f / (float)g.ah;
            six(((float)min(by + (uint32_t)j, by_last) + 0.5f) / (float)g.bh, aiy * -1.0f, aiy * 1.0f, g.ah, ay.x0, D::kAPitch, false, s_by, TB, j);
        }
    }
    __syncthreads();
    // 3. mip A over its rectangle: LDS copy for step 4, global store by the owner
    {
        const char* lds = reinterpret_cast<const char*>(s_src);
        // the tile owns A texels [2 bx, 2 bx + 2 TB) — the last tile of a row / column also the odd one beyond (aw = 2 bw + 1)
        const int own_x0 = 2 * (int)bx, own_x1 = blockIdx.x + 1 == g.tiles_x ? (int)g.aw : 2 * (int)(bx + TB);
        const int own_y0 = 2 * (int)by, own_y1 = blockIdx.y + 1 == g.tiles_y ? (int)g.ah : 2 * (int)(by + TB);
        constexpr int kACells = D::kAW * D::kAW;
        for (int i = (int)tid; i < kACells; i += NT) {
            const int cy = i / D::kAW, cx = i - cy * D::kAW;
            if (cx >= ax.w || cy >= ay.w) continue;
            AxisE xs[6], ys[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                xs[k] = s_ax[k * D::kAW + cx];
                ys[k] = s_ay[k * D::kAW + cy];
            }
            const C3 v = bloom_texel<D::kSPitch>(lds, xs, ys);
            uint2 q;
            q.x = (uint32_t)f2h(v.r) | ((uint32_t)f2h(v.g) << 16);
            q.y = (uint32_t)f2h(v.b);  // alpha 0.0 (store_rgba16f(..., 0.0f))
            s_mid[cy * D::kAPitch + bl_cell(cx)] = q;
            const int tx = ax.x0 + cx, ty = ay.x0 + cy;  // the texel this cell stands for, when it is inside the image
            if (tx >= own_x0 && tx < own_x1 && ty >= own_y0 && ty < own_y1)
                *reinterpret_cast<uint2*>(const_cast<uint8_t*>(g.a.ptr) + (size_t)ty * g.a.pitch + (size_t)tx * 8) = q;
        }
    }
    __syncthreads();
    // 4. the B tile from the LDS copy of A
    {
        const char* lds = reinterpret_cast<const char*>(s_mid);
        for (int i = (int)tid; i < TB * TB; i += NT) {
            const int cy = i / TB, cx = i - cy * TB;
            const uint32_t x = bx + (uint32_t)cx, y = by + (uint32_t)cy;
            if (x >= g.bw || y >= g.bh) continue;
            AxisE xs[6], ys[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                xs[k] = s_bx[k * TB + cx];
                ys[k] = s_by[k * TB + cy];
            }
            const C3 v = bloom_texel<D::kAPitch>(lds, xs, ys);
            store_rgba16f(g.b, (int)x, (int)y, v.r, v.g, v.b, 0.0f);
        }
    }
}

// host: every rectangle of every tile fits, and every set-up of both levels stays inside its rectangle (the kernel's own arithmetic)
template <int TB> static bool bloom_pair_fits(uint32_t sw, uint32_t aw, uint32_t bw) {
    using D = PairDims<TB>;
    if (bw == 0 || aw < 2 || sw < 2 || (aw != 2 * bw && aw != 2 * bw + 1)) return false;  // ownership assumes the Vulkan mip rule
    auto setups_inside = [](uint32_t t, uint32_t dst, uint32_t src, const PairRect& r) {
        const float i = 1.0f / (float)src, lo = i * -1.0f, hi = i * 1.0f, c = ((float)t + 0.5f) / (float)dst;
        const float ca = c + lo, cb = c + hi;
        const float coords[6] = {ca, cb, ca + lo, ca + hi, cb + lo, cb + hi};
        for (float co : coords) {
            const float p = co * (float)src - 0.5f;
            const float f0 = __builtin_floorf(p);
            const int idx = (int)__builtin_fminf(__builtin_fmaxf(f0, -1.0e9f), 1.0e9f);
            if (idx < r.x0 || idx + 1 >= r.x0 + r.w) return false;
        }
        return true;
    };
    for (uint32_t b0 = 0; b0 < bw; b0 += TB) {
        const uint32_t b1 = std::min(b0 + TB - 1, bw - 1);
        const PairRect a = pair_rect(b0, b1, bw, aw);
        if (a.w <= 0 || a.w > D::kAW) return false;
        const int lo = clampi(a.x0, 0, (int)aw - 1), hi = clampi(a.x0 + a.w - 1, 0, (int)aw - 1);
        const PairRect sr = pair_rect((uint32_t)lo, (uint32_t)hi, aw, sw);
        if (sr.w <= 0 || sr.w > D::kSW) return false;
        for (uint32_t t = b0; t <= b1; t++)
            if (!setups_inside(t, bw, aw, a)) return false;
        for (int t = lo; t <= hi; t++)
            if (!setups_inside((uint32_t)t, aw, sw, sr)) return false;
        // the owner's A texels lie inside its rectangle
        const int own0 = 2 * (int)b0, own1 = (b0 + TB >= bw ? (int)aw : 2 * (int)(b0 + TB)) - 1;
        if (own0 < a.x0 || own1 > a.x0 + a.w - 1) return false;
    }
    return true;
}

// mips A and A + 1 of the chain from mip A - 1 in one launch; false: the extents do not suit it (the caller launches them one by one)
bool launch_bloom_pair(const PlaneArg& s, uint32_t sw, uint32_t sh, const PlaneArg& a, uint32_t aw, uint32_t ah, const PlaneArg& b, uint32_t bw, uint32_t bh,
                       hipStream_t st, hipError_t* err) {
    *err = hipSuccess;
    if ((uint64_t)s.pitch * sh >= (1ull << 31) || (uint64_t)a.pitch * ah >= (1ull << 31)) return false;
    BloomPairArgs g{s, a, b, sw, sh, aw, ah, bw, bh, 0, 0};
    // 8 x 8 tiles of B while that gives the chip at least a workgroup per CU; 4 x 4 for the last mips
    const bool small = (uint64_t)((bw + 7) / 8) * ((bh + 7) / 8) < 256;
    if (!small) {
        if (!bloom_pair_fits<8>(sw, aw, bw) || !bloom_pair_fits<8>(sh, ah, bh)) return false;
        g.tiles_x = (bw + 7) / 8;
        g.tiles_y = (bh + 7) / 8;
        hipLaunchKernelGGL((k_bloom_pair<8, 512>), dim3(g.tiles_x, g.tiles_y), dim3(512), 0, st, g);
    } else {
        if (!bloom_pair_fits<4>(sw, aw, bw) || !bloom_pair_fits<4>(sh, ah, bh)) return false;
        g.tiles_x = (bw + 3) / 4;
        g.tiles_y = (bh + 3) / 4;
        hipLaunchKernelGGL((k_bloom_pair<4, 256>), dim3(g.tiles_x, g.tiles_y), dim3(256), 0, st, g);
    }
    *err = hipGetLastError();
    return true;
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
hipError_t launch_copy_scene(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                             uint32_t row_end, hipStream_t st) {
    if (row_end <= row_begin) return hipSuccess;
    const dim3 grid((dw + 63) / 64, (row_end - row_begin + 4 * kCopyPpt - 1) / (4 * kCopyPpt));
    hipLaunchKernelGGL(k_copy_scene, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh, row_begin, row_end);
    return hipGetLastError();
}
hipError_t launch_bloom_downsample(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                                   uint32_t row_end, hipStream_t st) {
    if (row_end <= row_begin) return hipSuccess;
    if ((uint64_t)src.pitch * sh < (1ull << 31)) {
        const uint32_t cols = (dw + kBlW - 1) / kBlW, rows = row_end - row_begin;
        constexpr int kBig = 2;  // texels per thread in the large mips; 2 and 4 measured alike (0.094 / 0.096 ms for the 4K chain)
        if ((uint64_t)cols * ((rows + 15) / 16) >= 500) {  // enough 64x16 tiles for two per CU (mip 1 of a 4K chain, 510 of them: 15.5 -> 12.7 us)
            hipLaunchKernelGGL(k_bloom_downsample_lds<kBig>, dim3(cols, (rows + 4 * kBig - 1) / (4 * kBig)), dim3(256), 0, st, src, sw, sh, dst, dw, dh, row_begin, row_end);
        } else {
            hipLaunchKernelGGL(k_bloom_downsample_lds<1>, dim3(cols, (rows + 3) / 4), dim3(256), 0, st, src, sw, sh, dst, dw, dh, row_begin, row_end);
        }
    } else {  // planes of 2 GiB and more: 64-bit addressing, one texel per thread (whole mip: the row range only saves work)
        const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
        hipLaunchKernelGGL(k_bloom_downsample, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh);
    }
    return hipGetLastError();
}

}  // namespace sah
