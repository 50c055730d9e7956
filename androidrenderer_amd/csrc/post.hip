// Post chain for gfx950: "Copy scene" (a13), bloom downsample pyramid (a7); the tonemap composite (a8) is tonemap.hip.
//   RenderCore/shaders/util/copy_with_sampler.frag.slang:9-12
//   RenderCore/shaders/postprocessing/bloom_downsample.comp:16-52   (host: RenderCore/render/bloomer.cpp:38-262)
#include <hip/hip_runtime.h>

#include "post_common.hpp"

namespace sah {

// ---- a13 -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_copy_scene(PlaneArg src, uint32_t sw, uint32_t sh, PlaneArg dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                                                     uint32_t row_end) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = row_begin + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= row_end) return;
    const float inv_w = 1.0f / (float)dw, inv_h = 1.0f / (float)dh;
    const float u = ((float)x + 0.5f) * inv_w, v = ((float)y + 0.5f) * inv_h;
    const Rgba t = bilinear<ADDR_REPEAT>(src, sw, sh, u, v);
    store_rgba16f(dst, (int)x, (int)y, t.c[0], t.c[1], t.c[2], t.c[3]);
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
// A 256-thread workgroup produces a 64x8 tile of the destination mip (two texels per thread).  The source rectangle the tile's 20
// taps per texel can touch (2x the tile plus 3 texels either side) is copied to LDS once — every source texel was fetched about 20
// times from the vector cache before — and the 6 + 6 axis set-ups of every tile column / row are tabulated once per workgroup.
// The 5 boxes x 4 bilinear taps of bloom_downsample.comp:16-52 use only 6 distinct x and 6 distinct y coordinates (u -+ ix, and those
// -+ ix again); same taps, same order, same operators as k_bloom_downsample: bit-identical.  A tile whose rectangle does not fit (extreme
// aspect ratios) takes the global-memory form.  `row_begin/row_end`: destination rows to produce (row-sharded pyramids).
constexpr int kBlW = 64, kBlH = 8, kBlTexels = 136 * 24;
SAH_DEV C3 bloom_texel(const char* tex, const AxisE* col, const AxisE* row) {  // col[k * kBlW], row[k * kBlH], k = 0..5
    auto box = [&](const AxisE& xl, const AxisE& xr, const AxisE& yt, const AxisE& yb) {
        const C3 s = tap_lds(tex, xl, yt) + tap_lds(tex, xr, yt) + tap_lds(tex, xl, yb) + tap_lds(tex, xr, yb);
        return s * 0.25f;
    };
    const AxisE xa = col[0], xb = col[kBlW], xaa = col[2 * kBlW], xab = col[3 * kBlW], xba = col[4 * kBlW], xbb = col[5 * kBlW];
    const AxisE yc = row[0], yd = row[kBlH], ycc = row[2 * kBlH], ycd = row[3 * kBlH], ydc = row[4 * kBlH], ydd = row[5 * kBlH];
    return box(xa, xb, yc, yd) * 0.5f + box(xaa, xab, ycc, ycd) * 0.125f + box(xba, xbb, ycc, ycd) * 0.125f + box(xaa, xab, ydc, ydd) * 0.125f +
           box(xba, xbb, ydc, ydd) * 0.125f;
}
__global__ void __launch_bounds__(256) k_bloom_downsample_lds(PlaneArg src, uint32_t sw, uint32_t sh, PlaneArg dst, uint32_t dw, uint32_t dh,
                                                               uint32_t row_begin, uint32_t row_end) {
    __shared__ uint2 s_tex[kBlTexels];
    __shared__ AxisE s_ax[6 * kBlW + 6 * kBlH];
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
        const int x0 = max((int)__builtin_floorf(pu0 - 3.0f), 0), x1 = min((int)__builtin_floorf(pu1 + 3.0f) + 1, (int)sw - 1);
        const int y0 = max((int)__builtin_floorf(pv0 - 3.0f), 0), y1 = min((int)__builtin_floorf(pv1 + 3.0f) + 1, (int)sh - 1);
        const int w = x1 - x0 + 1, h = y1 - y0 + 1;
        const bool fits = w > 0 && h > 0 && w * h <= kBlTexels;
        s_rect[0] = x0; s_rect[1] = y0; s_rect[2] = fits ? w : 0; s_rect[3] = fits ? h : 0;
        s_bad = fits ? 0 : 1;
    }
    __syncthreads();
    const int rx0 = s_rect[0], ry0 = s_rect[1], rw = s_rect[2], rh = s_rect[3];
    for (int i = (int)tid; i < rw * rh; i += 256) {
        const int ty = i / rw, tx = i - ty * rw;
        s_tex[i] = *reinterpret_cast<const uint2*>(src.ptr + (size_t)(ry0 + ty) * src.pitch + (size_t)(rx0 + tx) * 8);
    }
    for (uint32_t e = tid; e < (uint32_t)(6 * kBlW + 6 * kBlH); e += 256) {
        Axis a;
        AxisE en;
        bool inside;
        if (e < 6u * kBlW) {
            const uint32_t k = e / kBlW, x = min(bx + e % kBlW, x_last);
            const float u = ((float)x + 0.5f) / (float)dw, ua = u + ox, ub = u + oz;
            a = axis_setup(k == 0 ? ua : k == 1 ? ub : k == 2 ? ua + ox : k == 3 ? ua + oz : k == 4 ? ub + ox : ub + oz, sw);
            en = AxisE{(a.i0 - rx0) * 8, (a.i1 - rx0) * 8, a.w0, a.w1};
            inside = a.i0 >= rx0 && a.i1 < rx0 + rw;
        } else {
            const uint32_t j = e - 6u * kBlW, k = j / kBlH, y = min(by + j % kBlH, y_last);
            const float v = ((float)y + 0.5f) / (float)dh, vc = v + oy, vd = v + ow;
            a = axis_setup(k == 0 ? vc : k == 1 ? vd : k == 2 ? vc + oy : k == 3 ? vc + ow : k == 4 ? vd + oy : vd + ow, sh);
            en = AxisE{(a.i0 - ry0) * rw * 8, (a.i1 - ry0) * rw * 8, a.w0, a.w1};
            inside = a.i0 >= ry0 && a.i1 < ry0 + rh;
        }
        s_ax[e] = en;
        if (!inside) s_bad = 1;
    }
    __syncthreads();
    const uint32_t col = tid & 63u, x = bx + col;
    if (x >= dw) return;
    const bool bad = s_bad != 0;
#pragma unroll
    for (uint32_t half = 0; half < 2; half++) {
        const uint32_t row = (tid >> 6) + 4u * half, y = by + row;
        if (y >= row_end) break;
        C3 s;
        if (!bad) {
            s = bloom_texel(reinterpret_cast<const char*>(s_tex), s_ax + col, s_ax + 6 * kBlW + row);
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
    const dim3 grid((dw + 63) / 64, (row_end - row_begin + 3) / 4);
    hipLaunchKernelGGL(k_copy_scene, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh, row_begin, row_end);
    return hipGetLastError();
}
hipError_t launch_bloom_downsample(const PlaneArg& src, uint32_t sw, uint32_t sh, const PlaneArg& dst, uint32_t dw, uint32_t dh, uint32_t row_begin,
                                   uint32_t row_end, hipStream_t st) {
    if (row_end <= row_begin) return hipSuccess;
    if ((uint64_t)src.pitch * sh < (1ull << 31)) {
        const dim3 grid((dw + kBlW - 1) / kBlW, (row_end - row_begin + kBlH - 1) / kBlH);
        hipLaunchKernelGGL(k_bloom_downsample_lds, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh, row_begin, row_end);
    } else {  // planes of 2 GiB and more: 64-bit addressing, one texel per thread (whole mip: the row range only saves work)
        const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
        hipLaunchKernelGGL(k_bloom_downsample, grid, dim3(256), 0, st, src, sw, sh, dst, dw, dh);
    }
    return hipGetLastError();
}

}  // namespace sah
