// Post chain for gfx950: "Copy scene" (a13), bloom downsample pyramid (a7); the tonemap composite (a8) is tonemap.hip.
//   RenderCore/shaders/util/copy_with_sampler.frag.slang:9-12
//   RenderCore/shaders/postprocessing/bloom_downsample.comp:16-52   (host: RenderCore/render/bloomer.cpp:38-262)
#include <hip/hip_runtime.h>

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
// (PPT = SAH_BLOOM_PPT for the large mips; small mips take PPT = 1 — 64x4 tiles — so that their few texels spread over more
// workgroups with shorter dependency chains: 10.5 -> 5 us per launch for the 240x135 mip and below)
constexpr int kBlW = 64, kBlPitch = 136;
// LDS row layout: texel tx of a staged row sits in cell tx + tx / 32 (one empty cell after every 32).  Adjacent destination columns
// read source texels two apart, 16 bytes: with the plain layout lanes i and i + 16 of a ds_read_b64 hit the same two banks (44 % of
// the kernel's LDS cycles were bank conflicts); the gap moves the second sixteen lanes on by two banks.
constexpr int kBlPitchCells = kBlPitch + (kBlPitch - 1) / 32 + 1;
SAH_DEV int bl_cell(int tx) { return tx + (tx >> 5); }
// one bilinear tap from the staged rectangle: columns ax.o0 and ax.o1 (byte offsets inside a row), rows ay.o0 and the one below it
SAH_DEV C3 tap_rep(const char* tex, const AxisE& ax, const AxisE& ay) {
    // four ds_read_b64 (kept apart by `volatile`): merged into two ds_read2_b64 they are serviced at half the bytes per clock
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const volatile v2u* LdsTexel;
    constexpr int kRow = kBlPitchCells;  // in cells: the second row is an immediate offset
    const LdsTexel c0 = (LdsTexel)(tex + (ay.o0 + ax.o0)), c1 = (LdsTexel)(tex + (ay.o0 + ax.o1));
    const v2u t00 = c0[0], t10 = c1[0], t01 = c0[kRow], t11 = c1[kRow];
    const float w00 = ax.w0 * ay.w0, w10 = ax.w1 * ay.w0, w01 = ax.w0 * ay.w1, w11 = ax.w1 * ay.w1;
    C3 c;
    c.r = fma_mix_lo(w11, t11.x, fma_mix_lo(w01, t01.x, fma_mix_lo(w10, t10.x, fma_mix_lo(w00, t00.x, 0.0f))));
    c.g = fma_mix_hi(w11, t11.x, fma_mix_hi(w01, t01.x, fma_mix_hi(w10, t10.x, fma_mix_hi(w00, t00.x, 0.0f))));
    c.b = fma_mix_lo(w11, t11.y, fma_mix_lo(w01, t01.y, fma_mix_lo(w10, t10.y, fma_mix_lo(w00, t00.y, 0.0f))));
    return c;
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
            const AxisE yc = rowp[0], yd = rowp[kBlH], ycc = rowp[2 * kBlH], ycd = rowp[3 * kBlH], ydc = rowp[4 * kBlH], ydd = rowp[5 * kBlH];
            auto box = [&](const AxisE& xl, const AxisE& xr, const AxisE& yt, const AxisE& yb) {  // (weights pre-scaled: see the table)
                return tap_rep(lds, xl, yt) + tap_rep(lds, xr, yt) + tap_rep(lds, xl, yb) + tap_rep(lds, xr, yb);
            };
            s = box(xs[0], xs[1], yc, yd) + box(xs[2], xs[3], ycc, ycd) + box(xs[4], xs[5], ycc, ycd) + box(xs[2], xs[3], ydc, ydd) +
                box(xs[4], xs[5], ydc, ydd);
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
#ifndef SAH_BLOOM_PPT
#define SAH_BLOOM_PPT 2  // texels per thread in the large mips; 2 and 4 measured alike (0.094 / 0.096 ms for the 4K chain)
#endif
        constexpr int kBig = SAH_BLOOM_PPT;
        if ((uint64_t)cols * ((rows + 15) / 16) >= 1024) {  // enough 64x16 tiles for four per CU
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
