// Tonemap composite (a8) for gfx950: RenderCore/shaders/ui/scene_upsample.frag:20-72, host RenderCore/render/phase/ui_phase.cpp:98-113.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "post_common.hpp"

// (both overridable from the command line for A/B builds: SAH_EXTRA_HIPCC_FLAGS, androidrenderer_amd/build.py)
#ifndef SAH_TONEMAP_AHEAD
#define SAH_TONEMAP_AHEAD 1
#endif
#ifndef SAH_TONEMAP_MIN_BLOCKS
#define SAH_TONEMAP_MIN_BLOCKS 4
#endif

namespace sah {

// Tonemap composite, LDS-staged.  A 256-thread workgroup produces a 32x32 output tile, four pixels of one column per thread.
// For every bloom mip the texel rectangle the tile can touch (tile bounds mapped into the mip, plus the reach of the tent offsets —
// which are -ix, -iy and +ix in x and +ix, +-iy in y because scene_upsample.frag:29-32 mixes the components of `o`) is copied into
// LDS once, converted to fp32 and WITH the clamp-to-edge replication applied: cell j of a staged row holds the fp32 r, g, b of texel
// clamp(j) followed by those of texel clamp(j + 1) — 24 bytes, 8-byte aligned — so that the two columns of a bilinear tap are three
// ds_read_b64 (the full-rate LDS read), the second row sits at a compile-time distance (immediate offsets: one address add per
// tap), no index is clamped per tap, and the tap's twelve multiply-adds are plain v_fma_f32 (2.3 issue cycles) instead of
// v_fma_mix_f32 on fp16 texels (4.3).  The 4 + 4 distinct axis set-ups of every column / row of the tile are tabulated in LDS once
// (they depend on x or on y only), WITH the tent weight and the final / 16 folded into the bilinear weights: scaling by a power of
// two commutes with every rounding on the way (no value comes near the fp32 denormals: a non-zero weight is >= 2^-25, a non-zero
// fp16 texel >= 2^-24), so sum_k (c_k / 16) tap_k evaluated as a plain sum of taps with pre-scaled weights has the bits of
// tent_blur()'s `(4 t0 + 2 t1 + ... + t8) / 16`.  Every thread then evaluates its 9 taps x 6 mips x 4 pixels from those tables:
// per tap one address add, four weight products, twelve multiply-adds, and one add per channel into the tent sum.
// If any set-up of a mip indexes outside the staged rectangle (never for in-range tiles of 16:9 frames; wider than ~21:9 or a bloom
// chain that is not half the output resolution can) the whole workgroup takes the global-memory path for that mip.
constexpr int kTmTileW = 32, kTmTileH = 32, kTmPpt = 4, kTmRowStep = kTmTileH / kTmPpt;
constexpr int kTmPitch[6] = {24, 16, 12, 10, 10, 10};  // cells per staged row ...
constexpr int kTmRows[6] = {24, 16, 12, 10, 10, 10};   // ... and rows, per mip (fixed, so that row distances are immediates)
constexpr int kTmCellBytes = 24;
// LDS holds one group of mips at a time (mips 0-1, then mips 2-5: 20 KB of cells + 16 KB of tables instead of 31 + 24 KB, which is
// four workgroups per CU instead of two)
constexpr int kTmSplit = 2;
constexpr int tm_group_first(int m) { return m < kTmSplit ? 0 : kTmSplit; }
constexpr int tm_cell_base(int m) {
    int o = 0;
    for (int k = tm_group_first(m); k < m; k++) o += kTmPitch[k] * kTmRows[k];
    return o;
}
constexpr int kTmCellsA = tm_cell_base(kTmSplit - 1) + kTmPitch[kTmSplit - 1] * kTmRows[kTmSplit - 1];
constexpr int kTmCellsB = tm_cell_base(5) + kTmPitch[5] * kTmRows[5];
constexpr int kTmCells = kTmCellsA > kTmCellsB ? kTmCellsA : kTmCellsB;
constexpr int kTmGroupMips = kTmSplit > 6 - kTmSplit ? kTmSplit : 6 - kTmSplit;
constexpr int kTmAxisPerMip = 4 * kTmTileW + 4 * kTmTileH;

// axis set-up as the tap loop consumes it: byte offset of the cell (x: inside a row; y: of the row, mip base included) and the two
// weights, pre-scaled (see above)
struct AxisT {
    int o;
    float w0, w1;
    int padding;
};

typedef __attribute__((address_space(3))) const char* LdsPtr;
SAH_DEV float2 lds_read8(LdsPtr p) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = *(__attribute__((address_space(3))) const volatile v2f*)p;
    return make_float2(v.x, v.y);
}

// one table entry = one ds_read_b128 (a 12-byte read is serviced at 3/8 of its rate)
typedef __attribute__((address_space(3))) const struct AxisT* LdsAxis;
SAH_DEV AxisT lds_axis(LdsAxis p) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = *(__attribute__((address_space(3))) const volatile v4i*)p;
    const int o = v.x, w0 = v.y, w1 = v.z;  // (copies: __builtin_bit_cast of a vector element reads element 0)
    return {o, __builtin_bit_cast(float, w0), __builtin_bit_cast(float, w1), 0};
}

// the six 8-byte reads of one bilinear tap (two cells: rows y and y + 1), and its evaluation: acc = fma(w_k, t_k, acc) from +0 in
// tap order (t00, t10, t01, t11), as sample_bilinear()
struct TapCells {
    float2 a0, a1, a2, b0, b1, b2;
};
template <int ROW_BYTES> SAH_DEV TapCells tap_read(LdsPtr tex, const AxisT& ax, const AxisT& ay) {
    const LdsPtr p = tex + (ax.o + ay.o);
    // volatile: keeps the six reads as ds_read_b64 (256 B/clk/CU); merged into ds_read2_b64 they are serviced at half that rate
    TapCells c;
    c.a0 = lds_read8(p), c.a1 = lds_read8(p + 8), c.a2 = lds_read8(p + 16);
    c.b0 = lds_read8(p + ROW_BYTES), c.b1 = lds_read8(p + ROW_BYTES + 8), c.b2 = lds_read8(p + ROW_BYTES + 16);
    return c;
}
SAH_DEV C3 tap_eval(const TapCells& t, const AxisT& ax, const AxisT& ay) {
    const float w00 = ax.w0 * ay.w0, w10 = ax.w1 * ay.w0, w01 = ax.w0 * ay.w1, w11 = ax.w1 * ay.w1;
    C3 c;
    c.r = __builtin_fmaf(w11, t.b1.y, __builtin_fmaf(w01, t.b0.x, __builtin_fmaf(w10, t.a1.y, __builtin_fmaf(w00, t.a0.x, 0.0f))));
    c.g = __builtin_fmaf(w11, t.b2.x, __builtin_fmaf(w01, t.b0.y, __builtin_fmaf(w10, t.a2.x, __builtin_fmaf(w00, t.a0.y, 0.0f))));
    c.b = __builtin_fmaf(w11, t.b2.y, __builtin_fmaf(w01, t.b1.x, __builtin_fmaf(w10, t.a2.y, __builtin_fmaf(w00, t.a1.x, 0.0f))));
    return c;
}

// The tent filter of mip M for the thread's four pixels, added to their bloom sums.  Tap order of tent_blur(): (x variant, y variant)
// = (0,0) (1,0) (2,0) (0,1) (0,2) (1,3) (3,3) (1,2) (3,2).  The 36 taps run as one software pipeline: the reads of tap i + kTmAhead
// are issued before tap i is evaluated.  Measured at 4K: one tap ahead under a 128-register bound (four workgroups per CU, 9 registers
// spilled off the hot loop) 0.303 ms; two ahead 0.320; three workgroups per CU without the bound 0.317 (one ahead) / 0.308 (two).
constexpr int kTmAhead = SAH_TONEMAP_AHEAD;
constexpr int kTmTapX[9] = {0, 1, 2, 0, 0, 1, 3, 1, 3}, kTmTapY[9] = {0, 0, 0, 1, 2, 3, 3, 2, 2};
template <int M> SAH_DEV void tent_cells(LdsPtr tex, LdsAxis ax, uint32_t col, uint32_t row0, C3 (&bloom)[kTmPpt]) {
    constexpr int RB = kTmPitch[M] * kTmCellBytes;
    AxisT xs[4], ys[kTmPpt][4];  // (a pixel's y set-ups are read just before its first tap is: registers)
#pragma unroll
    for (int v = 0; v < 4; v++) xs[v] = lds_axis(ax + v * kTmTileW + col);
    TapCells buf[kTmAhead + 1];
    C3 s = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9 * kTmPpt + kTmAhead; i++) {
        if (i < 9 * kTmPpt && i % 9 == 0) {  // (rows past the band re-use the last valid row's set-ups; their result is dropped)
#pragma unroll
            for (int v = 0; v < 4; v++) ys[i / 9][v] = lds_axis(ax + 4 * kTmTileW + v * kTmTileH + row0 + (i / 9) * kTmRowStep);
        }
        if (i < 9 * kTmPpt) buf[i % (kTmAhead + 1)] = tap_read<RB>(tex, xs[kTmTapX[i % 9]], ys[i / 9][kTmTapY[i % 9]]);
        if (i >= kTmAhead) {
            const int j = i - kTmAhead, k = j / 9, q = j % 9;
            const C3 c = tap_eval(buf[j % (kTmAhead + 1)], xs[kTmTapX[q]], ys[k][kTmTapY[q]]);
            s = q == 0 ? c : s + c;
            if (q == 8) bloom[k] = bloom[k] + s;
        }
    }
}

__global__ void __launch_bounds__(256, SAH_TONEMAP_MIN_BLOCKS) k_tonemap(TonemapArgs t) {
    __shared__ __attribute__((aligned(16))) char s_tex[kTmCells * kTmCellBytes];
    __shared__ AxisT s_ax[kTmGroupMips][kTmAxisPerMip];  // [m - first of group][k*32 + column] (x variants k = 0..3), [..][128 + k*32 + row] (y variants)
    __shared__ int s_rect[6][4];              // x0, y0 (may be negative: replicated cells), w, h in cells (w == 0: not staged)
    __shared__ int s_bad[6];                  // 1: some set-up of mip m leaves the staged rectangle -> global path
    __shared__ float s_thr[256];              // s_thr[k] = smallest x whose output code is >= k (k = 1..255); s_thr[0] unused
    __shared__ uint32_t s_first[kTmMaxBuckets / 4];  // first-level table of the code search, one byte per bucket
    s_thr[threadIdx.x] = t.thresholds[threadIdx.x];
    if (threadIdx.x < kTmMaxBuckets / 4) s_first[threadIdx.x] = reinterpret_cast<const uint32_t*>(t.thresholds + 256)[threadIdx.x];
    const uint32_t bx = blockIdx.x * kTmTileW, by = t.row_begin + blockIdx.y * kTmTileH;
    const uint32_t x_last = min(bx + kTmTileW - 1, t.out_w - 1), y_last = min(by + kTmTileH - 1, t.row_end - 1);
    if (threadIdx.x < 6) {
        const uint32_t m = threadIdx.x;
        int* r = s_rect[m];
        r[0] = r[1] = r[2] = r[3] = 0;
        if (m < t.num_mips) {
            const float W = (float)t.mip_w[m], H = (float)t.mip_h[m];
            // cell bounds: tile extent in mip texels, widened by the tap offsets (x: -max(1, W/H) .. +1 texels, y: -+max(1, H/W)) and
            // half a texel for the roundings of the set-ups; a cell also holds the texel to its right, a tap also reads the row below
            const float left = __builtin_fmaxf(1.0f, W / H) + 0.5f, right = 1.5f, updown = __builtin_fmaxf(1.0f, H / W) + 0.5f;
            const float pu0 = ((float)bx + 0.5f) / (float)t.out_w * W - 0.5f, pu1 = ((float)x_last + 0.5f) / (float)t.out_w * W - 0.5f;
            const float pv0 = (1.0f - ((float)y_last + 0.5f) / (float)t.out_h) * H - 0.5f, pv1 = (1.0f - ((float)by + 0.5f) / (float)t.out_h) * H - 0.5f;
            const int x0 = (int)__builtin_floorf(pu0 - left), x1 = (int)__builtin_floorf(pu1 + right);
            const int y0 = (int)__builtin_floorf(pv0 - updown), y1 = (int)__builtin_floorf(pv1 + updown) + 1;
            const int w = x1 - x0 + 1, h = y1 - y0 + 1;
            int pitch = 0, rows = 0;
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (m == (uint32_t)k) pitch = kTmPitch[k], rows = kTmRows[k];
            if (w > 0 && h > 0 && w <= pitch && h <= rows) {
                r[0] = x0; r[1] = y0; r[2] = w; r[3] = h;
            }
        }
        s_bad[m] = r[2] == 0;
    }
    __syncthreads();
    // one group of mips [FIRST, LAST) into LDS: staged cells, then the axis tables
    auto stage = [&](auto first_c, auto last_c) {
        constexpr int FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        // staging: a thread owns texel column tid % 32 of the rectangle (w cells need w + 1 <= 25 texel columns) and every 8th row; all of
        // a thread's (up to 13) texels are requested before the first is converted and stored
        {
            constexpr int kIters[6] = {(kTmRows[0] + 7) / 8, (kTmRows[1] + 7) / 8, (kTmRows[2] + 7) / 8, (kTmRows[3] + 7) / 8, (kTmRows[4] + 7) / 8,
                                       (kTmRows[5] + 7) / 8};
            constexpr int kTotal = 2 * kIters[0] + kIters[1] + kIters[2] + kIters[3] + kIters[4] + kIters[5];  // (upper bound for either group)
            uint2 staged[kTotal];
            const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
            int n = 0;
    #pragma unroll
            for (int m = FIRST; m < LAST; m++) {
                {  // every load below is in range — its texel is clamped into the image, absent mips alias the scene (api_post.cpp) — so none is predicated
                    const int x0 = s_rect[m][0], y0 = s_rect[m][1];
                    const int wmax = (int)t.mip_w[m] - 1, hmax = (int)t.mip_h[m] - 1;
                    const uint32_t sx8 = (uint32_t)min(max(x0 + tx, 0), wmax) * 8u;  // CLAMP_TO_EDGE, once per texel
    #pragma unroll
                    for (int j = 0; j < kIters[m]; j++) {
                        const uint32_t sy = (uint32_t)min(max(y0 + ty0 + 8 * j, 0), hmax);
                        staged[n + j] = *reinterpret_cast<const uint2*>(t.mips[m].ptr + (sy * t.mips[m].pitch + sx8));
                    }
                }
                n += kIters[m];
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
            // tent weights 4 2 2 2 2 1 1 1 1 over 16 = (x scale) * (y scale): x variant 0 carries 2, y variant 0 carries 2 / 16, the other
            // y variants 1 / 16
            const float scale = is_x ? (k == 0 ? 2.0f : 1.0f) : (k == 0 ? 0.125f : 0.0625f);
    #pragma unroll
            for (uint32_t m = FIRST; m < LAST; m++) {
                if (m >= t.num_mips) break;
                const uint32_t W = t.mip_w[m], H = t.mip_h[m];
                const float ix = t.mip_inv_w[m], iy = t.mip_inv_h[m];
                const float ox = ix * -1.0f, oy = iy * -1.0f, oz = ix * 1.0f, ow = iy * 1.0f;
                const int rx0 = s_rect[m][0], ry0 = s_rect[m][1], rw = s_rect[m][2], rh_ = s_rect[m][3];
                AxisT en;
                bool inside;
                if (is_x) {
                    // x variants (scene_upsample.frag:28-36): u, u + o.x, u + o.y, u + o.z
                    const float c = k == 0 ? base : base + (k == 1 ? ox : k == 2 ? oy : oz);
                    const AxisU a = axis_unclamped(c, W);
                    en = {(a.i - rx0) * kTmCellBytes, a.w0 * scale, a.w1 * scale, 0};
                    inside = a.i >= rx0 && a.i < rx0 + rw;
                } else {
                    // y variants: v (+ 0.f), v + o.z, v + o.w, v + o.y
                    const float c = base + (k == 0 ? 0.f : k == 1 ? oz : k == 2 ? ow : oy);
                    const AxisU a = axis_unclamped(c, H);
                    en = {((a.i - ry0) * kTmPitch[m] + tm_cell_base(m)) * kTmCellBytes, a.w0 * scale, a.w1 * scale, 0};
                    inside = a.i >= ry0 && a.i + 1 < ry0 + rh_;
                }
                s_ax[m - FIRST][i] = en;
                if (!inside) s_bad[m] = 1;
            }
        }
            // (the tables above were built while the texels travelled) conversion to fp32 and the two half cells of every texel
            n = 0;
    #pragma unroll
            for (int m = FIRST; m < LAST; m++) {
                const int w = (uint32_t)m < t.num_mips ? s_rect[m][2] : 0, h = s_rect[m][3];
    #pragma unroll
                for (int j = 0; j < kIters[m]; j++, n++) {
                    const int ty = ty0 + 8 * j;
                    if (w > 0 && tx <= w && ty < h) {
                        const float r = h2f((uint16_t)(staged[n].x & 0xffffu)), g = h2f((uint16_t)(staged[n].x >> 16)), b = h2f((uint16_t)(staged[n].y & 0xffffu));
                        char* cell = s_tex + (tm_cell_base(m) + ty * kTmPitch[m] + tx) * kTmCellBytes;
                        if (tx < w) {  // first half of its own cell
                            *reinterpret_cast<float2*>(cell) = make_float2(r, g);
                            *reinterpret_cast<float*>(cell + 8) = b;
                        }
                        if (tx > 0) {  // second half of the cell to its left
                            *reinterpret_cast<float*>(cell - 12) = r;
                            *reinterpret_cast<float2*>(cell - 8) = make_float2(g, b);
                        }
                    }
                }
            }
        }
    };
    const LdsPtr tex = (LdsPtr)s_tex;
    const uint32_t col = threadIdx.x & (kTmTileW - 1), row0 = threadIdx.x / kTmTileW;  // tile rows row0 + 8 k
    const uint32_t x = bx + col;
    const bool live = x < t.out_w && by + row0 < t.row_end;  // (the others still stage and build tables)
    const float u = ((float)x + 0.5f) / (float)t.out_w;
    C3 bloom[kTmPpt];
#pragma unroll
    for (int k = 0; k < kTmPpt; k++) bloom[k] = {0.f, 0.f, 0.f};
    auto filter = [&](uint32_t first, uint32_t last) {
        for (uint32_t m = first; m < last && m < t.num_mips; m++) {
            if (!s_bad[m]) {
                const LdsAxis ax = (LdsAxis)s_ax[m - first];
                switch (m) {  // (the row distance inside the staged rectangle is a compile-time constant of the mip)
                case 0: tent_cells<0>(tex, ax, col, row0, bloom); break;
                case 1: tent_cells<1>(tex, ax, col, row0, bloom); break;
                case 2: tent_cells<2>(tex, ax, col, row0, bloom); break;
                case 3: tent_cells<3>(tex, ax, col, row0, bloom); break;
                case 4: tent_cells<4>(tex, ax, col, row0, bloom); break;
                default: tent_cells<5>(tex, ax, col, row0, bloom); break;
                }
            } else {
#pragma unroll
                for (int k = 0; k < kTmPpt; k++) {
                    const uint32_t y = min(by + row0 + k * kTmRowStep, t.row_end - 1);
                    bloom[k] = bloom[k] + tent_blur(t.mips[m], t.mip_w[m], t.mip_h[m], u, 1.0f - ((float)y + 0.5f) / (float)t.out_h);
                }
            }
        }
    };
    stage(std::integral_constant<int, 0>{}, std::integral_constant<int, kTmSplit>{});
    __syncthreads();
    if (live) filter(0, kTmSplit);
    __syncthreads();  // every wave is done with the first group's cells and tables
    stage(std::integral_constant<int, kTmSplit>{}, std::integral_constant<int, 6>{});
    __syncthreads();
    if (!live) return;
    filter(kTmSplit, 6);
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
        // exact composite (api_post.cpp: tonemap_code), the smallest input that reaches each code; the code of x is the number of
        // thresholds <= x.  Two fp64 pow() per channel (~600 issue slots) become a two-level look-up: the float's exponent and top
        // four mantissa bits select a bucket whose first code is tabulated and which spans at most three codes, then three
        // thresholds are counted.  x below thresholds[1], negative or NaN lands in bucket 0 (first code 0) through the clamp — fmaxf
        // returns its non-NaN operand — and every comparison of the original x is then false: code 0, as the shader's result.
        const float rgb[3] = {mapped.r, mapped.g, mapped.b};
        uint32_t code[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float xc = __builtin_fminf(__builtin_fmaxf(rgb[ch], t.thr_lo), t.thr_hi);
            const uint32_t b = (__builtin_bit_cast(uint32_t, xc) >> kTmBucketShift) - t.bucket_base;
            const uint32_t first = reinterpret_cast<const uint8_t*>(s_first)[b];
            const float* th = s_thr + first;
            code[ch] = first + (rgb[ch] >= th[1] ? 1u : 0u) + (rgb[ch] >= th[2] ? 1u : 0u) + (rgb[ch] >= th[3] ? 1u : 0u);
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
