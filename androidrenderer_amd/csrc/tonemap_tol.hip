// Tonemap composite, tolerance mode (SAH_TONEMAP_TOLERANCE_1CODE): RenderCore/shaders/ui/scene_upsample.frag:20-72 evaluated in a
// different ORDER — same real-number result, fp32 throughout — so the final R8G8B8A8 code may differ from the strict kernel's
// (tonemap.hip) where the strict value sits within re-association error (about 2^-20 relative) of a code threshold, and then by one
// code: inside BASELINE.json's tolerance for the final image (1 ULP of the stored format = 1 UNORM8 code).  Strict stays the default and
// the test oracle; tests/test_post_gpu.py and tools/stress_post.py histogram |code difference| between the two modes (max must be <= 1).
//
// What the re-ordering buys.  Per bloom mip the shader sums nine bilinear taps, tent-weighted:
//     blur = 1/16 * sum_t c_t * tap(u + dx_t, v + dy_t),   taps (x variant, y variant, c):
//     (0,0,4) (1,0,2) (2,0,2) (0,1,2) (0,2,2) (1,3,1) (3,3,1) (1,2,1) (3,2,1)
//     x variants: u, u - ix, u - iy (sic: o.y added to x), u + ix;   y variants: v, v + ix (sic), v + iy, v - iy
// A bilinear tap is (row interpolation) o (column interpolation), and the column interpolation of x variant k depends on the pixel's
// COLUMN and the texel row only.  So per 32 x 32 tile and mip:
//   pass 1  for every (pixel column, staged texel row): H_k = the column interpolation of the four x variants (8 multiply-adds per
//           channel), combined by y variant:  G_0 = 4 H_0 + 2 H_1 + 2 H_2,  G_1 = 2 H_0,  G_2 = 2 H_0 + H_1 + H_3,  G_3 = H_1 + H_3
//           (the columns of the c matrix above) -> LDS, 32 columns x (rows of the staged rectangle) x 4 planes x rgb;
//   pass 2  per pixel: four row interpolations, one per y variant, of its column of G_yv with weights pre-divided by 16:
//           24 multiply-adds per mip.
// 36 texel-weights x 3 channels + 36 weight products per pixel and mip (strict: 144 multiply-class instructions) become 24 + the
// pixel's share of pass 1 (about 2 staged rows per pixel row over the six mips).
// Layout: a thread owns two ADJACENT pixel columns (rows r and r + 16 of the tile), so that its G reads and writes are 24 contiguous,
// 8-byte aligned bytes — three ds_read_b64 / ds_write_b64, the full-rate LDS forms — and staged texels are 16-byte cells (one
// ds_read_b128 per texel).  Staging and the axis tables of mip m + 1 are double-buffered and filled while mip m is filtered: two
// barriers per mip.  LDS per workgroup: G 33.8 KB + texels 8.8 KB + tables 6 KB + code tables 1.5 KB = 3 workgroups per CU.
#include <hip/hip_runtime.h>

#include "post_common.hpp"

namespace sah {
namespace {

constexpr int kTile = 32;
constexpr int kMaxRows = 22, kMaxCols = 25;  // staged rectangle: rows, texel columns (mip 0 of a half-resolution chain needs 21 x 22)
constexpr int kPlane = kMaxRows * kTile * 3;  // floats per G plane
constexpr int kStageIters = (kMaxRows + 7) / 8;

struct AxisV {  // one axis set-up: offset of the first of the two texels / rows (x: float4 cells, y: floats into a G plane) and the two weights
    int o;
    float w0, w1;
    int pad;
};
// LDS reads in the forms that run at the full 256 B/clk: ds_read_b128 for a texel cell (left to itself the compiler reads the three used
// floats as ds_read_b96: 96 B/clk) and ds_read_b64 for the halves of a G entry (merged into ds_read2_b64 they run at 128 B/clk)
SAH_DEV float4 lds_read16(const float4* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = *(__attribute__((address_space(3))) const volatile v4f*)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
SAH_DEV float2 lds_read8(const float* p) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = *(__attribute__((address_space(3))) const volatile v2f*)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
    return make_float2(v.x, v.y);
}

SAH_DEV AxisV lds_axis(const AxisV* p) {  // one ds_read_b128 per table entry (its 12 used bytes would be read as ds_read_b96)
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = *(__attribute__((address_space(3))) const volatile v4i*)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
    const int o = v.x, w0 = v.y, w1 = v.z;
    return {o, __builtin_bit_cast(float, w0), __builtin_bit_cast(float, w1), 0};
}

struct Rect {
    int x0, y0, w, h;  // w cells (w + 1 texel columns), h rows; w == 0: does not fit
};

// texel rectangle of mip (W x H) the tile can touch (as tonemap.hip): tile bounds in mip texels, widened by the tap offsets (x: -max(1,
// W/H) .. +1 texels, y: -+max(1, H/W)) and half a texel for the roundings of the set-ups; a tap also reads the texel to the right / below
SAH_DEV Rect tile_rect(const TonemapArgs& t, uint32_t W, uint32_t H, uint32_t bx, uint32_t by, uint32_t x_last, uint32_t y_last) {
    const float Wf = (float)W, Hf = (float)H;
    const float left = __builtin_fmaxf(1.0f, Wf / Hf) + 0.5f, right = 1.5f, updown = __builtin_fmaxf(1.0f, Hf / Wf) + 0.5f;
    const float pu0 = ((float)bx + 0.5f) / (float)t.out_w * Wf - 0.5f, pu1 = ((float)x_last + 0.5f) / (float)t.out_w * Wf - 0.5f;
    const float pv0 = (1.0f - ((float)y_last + 0.5f) / (float)t.out_h) * Hf - 0.5f, pv1 = (1.0f - ((float)by + 0.5f) / (float)t.out_h) * Hf - 0.5f;
    Rect r;
    r.x0 = (int)__builtin_floorf(pu0 - left);
    r.y0 = (int)__builtin_floorf(pv0 - updown);
    const int x1 = (int)__builtin_floorf(pu1 + right), y1 = (int)__builtin_floorf(pv1 + updown) + 1;
    r.w = x1 - r.x0 + 1;
    r.h = y1 - r.y0 + 1;
    if (!(r.w > 0 && r.h > 0 && r.w + 1 <= kMaxCols && r.h <= kMaxRows)) r.w = r.h = 0;
    return r;
}

}  // namespace

// The axis set-ups of every output column (x variants: u, u + o.x, u + o.y, u + o.z) and row (y variants: v, v + o.z, v + o.w, v + o.y) of every
// mip — the same expressions, operator for operator, as the strict kernel's (the sample positions and bilinear weights are NOT re-associated:
// they are ill-conditioned in the texel values) — once per (output extent, chain extents) instead of once per tile.
__global__ void __launch_bounds__(256) k_tonemap_axis_tables(TonemapArgs t, TmAxis* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // column / row
    const uint32_t m = blockIdx.y >> 3, axis = (blockIdx.y >> 2) & 1u, k = blockIdx.y & 3u;
    if (m >= t.num_mips) return;
    const uint32_t n = axis == 0 ? t.out_w : t.out_h;
    if (i >= n) return;
    const float ix = t.mip_inv_w[m], iy = t.mip_inv_h[m];
    const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
    TmAxis e;
    if (axis == 0) {
        const float base = ((float)i + 0.5f) / (float)t.out_w;
        const float c = k == 0 ? base : base + (k == 1 ? ox : k == 2 ? oy : oz);
        const AxisU a = axis_unclamped(c, t.mip_w[m]);
        e = {a.i, a.w0, a.w1, 0};
    } else {
        const float base = 1.0f - ((float)i + 0.5f) / (float)t.out_h;
        const float c = base + (k == 0 ? 0.f : k == 1 ? oz : k == 2 ? ow : oy);
        const AxisU a = axis_unclamped(c, t.mip_h[m]);
        e = {a.i, a.w0 * 0.0625f, a.w1 * 0.0625f, 0};
    }
    out[(size_t)blockIdx.y * t.axis_stride + i] = e;
}

__global__ void __launch_bounds__(256, 3) k_tonemap_tol(TonemapArgs t) {
    // (s_src and s_xt are read by pass 1 only, which every thread has left before anybody commits the next mip: single buffers; s_yt is read
    // by pass 2, which overlaps the next commit: double buffer)
    __shared__ __attribute__((aligned(16))) float4 s_src[kMaxRows * kMaxCols];  // staged texels, fp32 rgb (w unused), edge replication applied
    __shared__ __attribute__((aligned(16))) float s_g[4 * kPlane];               // [y variant][staged row][pixel column][rgb]
    __shared__ AxisV s_xt[4][kTile], s_yt[2][4][kTile];                          // x set-ups per column, y set-ups per row (weights / 16)
    __shared__ int s_bad[2];                                                     // the mip cannot be staged: strict evaluation from global memory
    __shared__ Rect s_rect[6];                                                   // the six rectangles, by threads 0..5 (six divides each: not per thread and mip)
    __shared__ float s_thr[256];
    __shared__ uint32_t s_first[kTmMaxBuckets / 4];
    s_thr[threadIdx.x] = t.thresholds[threadIdx.x];
    if (threadIdx.x < kTmMaxBuckets / 4) s_first[threadIdx.x] = reinterpret_cast<const uint32_t*>(t.thresholds + 256)[threadIdx.x];
    const uint32_t bx = blockIdx.x * kTile, by = t.row_begin + blockIdx.y * kTile;
    const uint32_t x_last = min(bx + kTile - 1, t.out_w - 1), y_last = min(by + kTile - 1, t.row_end - 1);
    const uint32_t cp = threadIdx.x & 15u, tr = threadIdx.x >> 4;  // column pair (columns 2 cp, 2 cp + 1), tile rows tr and tr + 16
    const uint32_t nmips = min(t.num_mips, 6u);

    if (threadIdx.x < nmips) s_rect[threadIdx.x] = tile_rect(t, t.mip_w[threadIdx.x], t.mip_h[threadIdx.x], bx, by, x_last, y_last);
    __syncthreads();
    uint2 staged[kStageIters];
    TmAxis entry = {0, 0.f, 0.f, 0};
    Rect rc = {0, 0, 0, 0};
    // this thread's table entry: x set-up (variant tid / 32, column tid % 32) for tid < 128, else the y one
    const uint32_t te = threadIdx.x & (kTile - 1), tk = (threadIdx.x >> 5) & 3u, taxis = threadIdx.x >> 7;
    const uint32_t tpos = taxis == 0 ? min(bx + te, x_last) : min(by + te, y_last);  // columns / rows past the edge re-use the last valid one
    // requests the texels of mip m (its rectangle becomes `rc`): a thread owns texel column tid % 32 and rows tid / 32 + 8 j
    auto stage_request = [&](uint32_t m) {
        const uint32_t W = t.mip_w[m], H = t.mip_h[m];
        rc = s_rect[m];
        if (threadIdx.x == 0) s_bad[m & 1u] = rc.w == 0;
        const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
        const PlaneArg mp = t.mips[m];
        const uint32_t sx8 = (uint32_t)min(max(rc.x0 + tx, 0), (int)W - 1) * 8u;  // clamped into the image: never predicated
#pragma unroll
        for (int j = 0; j < kStageIters; j++) {
            const uint32_t sy = (uint32_t)min(max(rc.y0 + ty0 + 8 * j, 0), (int)H - 1);
            staged[j] = *reinterpret_cast<const uint2*>(mp.ptr + (sy * mp.pitch + sx8));
        }
        entry = t.axis_tables[(size_t)((m * 2u + taxis) * 4u + tk) * t.axis_stride + tpos];
    };
    // converts and stores the requested texels, and builds the axis tables of mip m (thread e: x set-up (variant e / 32, column e % 32)
    // for e < 128, else the y one)
    auto stage_commit = [&](uint32_t m) {
        const uint32_t b = m & 1u;
        const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
        if (rc.w > 0) {
#pragma unroll
            for (int j = 0; j < kStageIters; j++) {
                const int ty = ty0 + 8 * j;
                if (tx <= rc.w && ty < rc.h)
                    s_src[ty * kMaxCols + tx] = make_float4(h2f((uint16_t)(staged[j].x & 0xffffu)), h2f((uint16_t)(staged[j].x >> 16)),
                                                               h2f((uint16_t)(staged[j].y & 0xffffu)), 0.f);
            }
            if (taxis == 0) {
                s_xt[tk][te] = {entry.i - rc.x0, entry.w0, entry.w1, 0};
                if (!(entry.i >= rc.x0 && entry.i + 1 <= rc.x0 + rc.w)) s_bad[b] = 1;
            } else {
                s_yt[b][tk][te] = {(entry.i - rc.y0) * (kTile * 3), entry.w0, entry.w1, 0};
                if (!(entry.i >= rc.y0 && entry.i + 1 < rc.y0 + rc.h)) s_bad[b] = 1;
            }
        }
    };

    C3 bloom[2][2];  // [row r / r + 16][column 2 cp / 2 cp + 1]
#pragma unroll
    for (int a = 0; a < 2; a++)
        for (int c = 0; c < 2; c++) bloom[a][c] = {0.f, 0.f, 0.f};
    const uint32_t x0p = bx + 2u * cp;
    const bool live_col[2] = {x0p < t.out_w, x0p + 1u < t.out_w};
    const bool live_row[2] = {by + tr < t.row_end, by + tr + 16u < t.row_end};

    if (nmips) {
        stage_request(0);
        __syncthreads();  // (s_bad[0] reset before anybody sets it)
        stage_commit(0);
    }
    for (uint32_t m = 0; m < nmips; m++) {
        const uint32_t b = m & 1u;
        __syncthreads();  // texels and tables of mip m are in place; the previous mip's pass 2 is done with s_g
        const bool bad = s_bad[b] != 0;
        const int rows = rc.h;  // rectangle of mip m (stage_request(m) was the last one to run)
        if (!bad) {
            // pass 1: items (column pair, staged row): column pair = tid % 16, rows tid / 16 + 16 j
            AxisV xa[2][4];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) xa[c][k] = lds_axis(&s_xt[k][2u * cp + (uint32_t)c]);
            for (int r = (int)tr; r < rows; r += 16) {
                const float4* srow = s_src + r * kMaxCols;
                float g[4][6];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    float h[4][3];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float4 p = lds_read16(srow + xa[c][k].o), q = lds_read16(srow + xa[c][k].o + 1);
                        h[k][0] = __builtin_fmaf(xa[c][k].w1, q.x, xa[c][k].w0 * p.x);
                        h[k][1] = __builtin_fmaf(xa[c][k].w1, q.y, xa[c][k].w0 * p.y);
                        h[k][2] = __builtin_fmaf(xa[c][k].w1, q.z, xa[c][k].w0 * p.z);
                    }
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float g1 = 2.0f * h[0][ch], g3 = h[1][ch] + h[3][ch];
                        g[0][3 * c + ch] = __builtin_fmaf(2.0f, h[1][ch] + h[2][ch], 2.0f * g1);  // 4 H0 + 2 H1 + 2 H2
                        g[1][3 * c + ch] = g1;
                        g[2][3 * c + ch] = g1 + g3;
                        g[3][3 * c + ch] = g3;
                    }
                }
                float* dst = s_g + (r * kTile + 2 * (int)cp) * 3;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    float2* d2 = reinterpret_cast<float2*>(dst + p * kPlane);  // 24-byte stride: 8-byte aligned
                    d2[0] = make_float2(g[p][0], g[p][1]);
                    d2[1] = make_float2(g[p][2], g[p][3]);
                    d2[2] = make_float2(g[p][4], g[p][5]);
                }
            }
        }
        if (m + 1 < nmips) stage_request(m + 1);  // global loads travel during pass 2
        __syncthreads();
        if (!bad) {
            // pass 2: four row interpolations per pixel, two pixels at a time
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const uint32_t pr = min(tr + 16u * (uint32_t)a, y_last - by);  // rows past the band: the last valid one (dropped later)
                float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int yv = 0; yv < 4; yv++) {
                    const AxisV e = lds_axis(&s_yt[b][yv][pr]);
                    const float* g0 = s_g + yv * kPlane + e.o + 6 * (int)cp;
                    const float* g1 = g0 + kTile * 3;
                    const float2 P0 = lds_read8(g0), P1 = lds_read8(g0 + 2), P2 = lds_read8(g0 + 4), Q0 = lds_read8(g1), Q1 = lds_read8(g1 + 2), Q2 = lds_read8(g1 + 4);
                    s[0] = __builtin_fmaf(e.w1, Q0.x, __builtin_fmaf(e.w0, P0.x, s[0]));
                    s[1] = __builtin_fmaf(e.w1, Q0.y, __builtin_fmaf(e.w0, P0.y, s[1]));
                    s[2] = __builtin_fmaf(e.w1, Q1.x, __builtin_fmaf(e.w0, P1.x, s[2]));
                    s[3] = __builtin_fmaf(e.w1, Q1.y, __builtin_fmaf(e.w0, P1.y, s[3]));
                    s[4] = __builtin_fmaf(e.w1, Q2.x, __builtin_fmaf(e.w0, P2.x, s[4]));
                    s[5] = __builtin_fmaf(e.w1, Q2.y, __builtin_fmaf(e.w0, P2.y, s[5]));
                }
                bloom[a][0] = bloom[a][0] + C3{s[0], s[1], s[2]};
                bloom[a][1] = bloom[a][1] + C3{s[3], s[4], s[5]};
            }
        } else {  // (uniform over the workgroup) a rectangle that does not fit: the strict evaluation from global memory
            const PlaneArg mp = t.mips[m];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint32_t y = min(by + tr + 16u * (uint32_t)a, t.row_end - 1), x = min(x0p + (uint32_t)c, t.out_w - 1);
                    bloom[a][c] = bloom[a][c] + tent_blur(mp, t.mip_w[m], t.mip_h[m], ((float)x + 0.5f) / (float)t.out_w, 1.0f - ((float)y + 0.5f) / (float)t.out_h);
                }
        }
        if (m + 1 < nmips) stage_commit(m + 1);
    }
#pragma unroll
    for (int a = 0; a < 2; a++) {
        if (!live_row[a]) continue;
        const uint32_t y = by + tr + 16u * (uint32_t)a;
        const float v = 1.0f - ((float)y + 0.5f) / (float)t.out_h;
        uint32_t px[2] = {0u, 0u};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t x = min(x0p + (uint32_t)c, t.out_w - 1);
            const float u = ((float)x + 0.5f) / (float)t.out_w;
            const Rgba sc = bilinear<ADDR_CLAMP>(t.scene, t.scene_w, t.scene_h, u, v);
            const C3 col = {__builtin_fmaf(bloom[a][c].r, 0.014159f, sc.c[0]), __builtin_fmaf(bloom[a][c].g, 0.014159f, sc.c[1]),
                            __builtin_fmaf(bloom[a][c].b, 0.014159f, sc.c[2])};
            const float luma = __builtin_fmaf(col.b, 0.0722f, __builtin_fmaf(col.g, 0.7152f, col.r * 0.2126f));
            const float factor = luma / (luma + 1.f);
            const float rgb[3] = {col.r * factor, col.g * factor, col.b * factor};
            uint32_t code[3];  // the code search of tonemap.hip: exact for whatever value reaches it
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float xc = __builtin_fminf(__builtin_fmaxf(rgb[ch], t.thr_lo), t.thr_hi);
                const uint32_t bk = (__builtin_bit_cast(uint32_t, xc) >> kTmBucketShift) - t.bucket_base;
                const uint32_t first = reinterpret_cast<const uint8_t*>(s_first)[bk];
                const float* th = s_thr + first;
                code[ch] = first + (rgb[ch] >= th[1] ? 1u : 0u) + (rgb[ch] >= th[2] ? 1u : 0u) + (rgb[ch] >= th[3] ? 1u : 0u);
            }
            px[c] = code[0] | (code[1] << 8) | (code[2] << 16) | (255u << 24);
        }
        uint8_t* dst = const_cast<uint8_t*>(t.out.ptr) + (size_t)y * t.out.pitch + (size_t)x0p * 4;
        if (live_col[1] && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0u) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(px[0], px[1]);
        } else {
            if (live_col[0]) *reinterpret_cast<uint32_t*>(dst) = px[0];
            if (live_col[1]) *reinterpret_cast<uint32_t*>(dst + 4) = px[1];
        }
    }
}

// grid.y = (mip, axis, variant); `out` holds 6 * 2 * 4 * axis_stride entries
hipError_t launch_tonemap_axis_tables(const TonemapArgs& t, TmAxis* out, hipStream_t st) {
    const uint32_t n = max(t.out_w, t.out_h);
    hipLaunchKernelGGL(k_tonemap_axis_tables, dim3((n + 255u) / 256u, 6u * 8u), dim3(256), 0, st, t, out);
    return hipGetLastError();
}

hipError_t launch_tonemap_tol(const TonemapArgs& t, hipStream_t st) {
    const uint32_t rows = t.row_end - t.row_begin;
    if (rows == 0) return hipSuccess;
    const dim3 grid((t.out_w + kTile - 1) / kTile, (rows + kTile - 1) / kTile);
    hipLaunchKernelGGL(k_tonemap_tol, grid, dim3(256), 0, st, t);
    return hipGetLastError();
}

}  // namespace sah
