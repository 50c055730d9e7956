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
// ds_read_b128 per texel) in rows of 32 cells: the two rows a ds_read_b128 lane group spans then start on the same bank, and the sixteen
// cells the group reads fall on sixteen different bank quads (with rows of 25 cells a fifth of the kernel's LDS cycles were conflicts).
// Round 4: the mips are filtered in STAGES — {0}, {1, 2}, {3, 4, 5} — whose rectangles, G rows and axis tables lie side by side in the same
// LDS (22 staged rows suffice for each stage of a half-resolution chain), two barriers per stage instead of per mip: the small mips, whose
// pass 1 is a handful of rows, no longer cost a barrier pair each.  The code look-up reads ONE 16-byte entry per channel (the three
// thresholds of the channel's bucket and its first code) from a 7 KB table in global memory — L1-resident, on the otherwise idle texture
// path — instead of four data-dependent LDS reads, which were where the remaining bank conflicts came from.
// LDS per workgroup: G 32.3 KB + texels 10.8 KB + tables 9 KB = 52.3 KB, 3 workgroups per CU.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "post_common.hpp"

namespace sah {
namespace {

constexpr int kTile = 32;     // tile width in pixels
constexpr int kPitch = 32;    // cells per staged row (a rectangle is at most 25 texel columns wide)
constexpr int kMaxCols = 25;
constexpr int kStageMips = 3;
// Tile height and how the mips are grouped into stages.  32 rows: {0}, {1, 2}, {3, 4, 5} — 21 staged rows hold each stage of a
// half-resolution chain (mip 0 needs 20, mips 1 + 2 12 + 8, mips 3..5 6 + 6 + 6); 52.3 KB of LDS, three workgroups per CU (22 rows: 54.4 KB,
// which did NOT fit three times into a CU's 160 KB — two per CU, 1.47 x the time).  16 rows, for row bands whose 32-row tiles would not
// fill the chip twice (a rank's rows of a sharded frame): {0}, {1, 2}, {3, 4}, {5} in 14 rows, 34 KB, four workgroups per CU.
template <int TH> struct TmShape;
template <> struct TmShape<32> {
    static constexpr int kMaxRows = 21, kWaves = 3;
    static SAH_DEV uint32_t first(uint32_t s) { return s == 0 ? 0u : (s == 1 ? 1u : (s == 2 ? 3u : 6u)); }
};
template <> struct TmShape<16> {
    static constexpr int kMaxRows = 14, kWaves = 4;
    static SAH_DEV uint32_t first(uint32_t s) { return s == 0 ? 0u : (s == 1 ? 1u : (s == 2 ? 3u : (s == 3 ? 5u : 6u))); }
};

struct AxisS {  // one axis set-up in LDS: offset of the first of the two texels / rows (x: float4 cells, y: floats into a G plane) and the
    int o;      // fraction f (x) or f / 16 (y); the two weights are 1 - f, f (resp. 1/16 - f/16, f/16: the same bits as (1 - f) / 16)
    float f;
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
SAH_DEV AxisS lds_axis(const AxisS* p) {
    const float2 v = lds_read8(reinterpret_cast<const float*>(p));
    return {__builtin_bit_cast(int, v.x), v.y};
}

struct Rect {
    int x0, y0, w, h;  // w cells (w + 1 texel columns), h rows; w == 0: does not fit
};

// texel rectangle of mip (W x H) the tile can touch (as tonemap.hip): tile bounds in mip texels, widened by the tap offsets (x: -max(1,
// W/H) .. +1 texels, y: -+max(1, H/W)) and half a texel for the roundings of the set-ups; a tap also reads the texel to the right / below
SAH_DEV Rect tile_rect(const TonemapArgs& t, uint32_t W, uint32_t H, uint32_t bx, uint32_t by, uint32_t x_last, uint32_t y_last, int max_rows) {
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
    if (!(r.w > 0 && r.h > 0 && r.w + 1 <= kMaxCols && r.h <= max_rows)) r.w = r.h = 0;
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
        e = {a.i, a.w1};
    } else {
        const float base = 1.0f - ((float)i + 0.5f) / (float)t.out_h;
        const float c = base + (k == 0 ? 0.f : k == 1 ? oz : k == 2 ? ow : oy);
        const AxisU a = axis_unclamped(c, t.mip_h[m]);
        e = {a.i, a.w1 * 0.0625f};
    }
    out[(size_t)blockIdx.y * t.axis_stride + i] = e;
}

// kThreads: 256 (the product: a thread owns two adjacent columns of kTileH / 16 rows) or 512 — round 6's experiment (SAH_TM_THREADS=512; VERDICT r5 item 3:
// "512- or 1,024-thread workgroups, so a SIMD has another wave to run while one waits at a barrier"): the same tile, rectangles, stages and LDS with twice
// the waves, each thread one row of the 32-row tile; 114 VGPRs, so TWO such workgroups per CU = four waves per SIMD where three 256-thread workgroups are
// three.  Same codes (tests, tools/stress_post.py) and SLOWER: 0.2073 ms against 0.1794 for the 4K frame (two tiles in flight per CU instead of three
// outweigh the second wave per SIMD); held to six waves per SIMD (80 VGPRs, 144 B of scratch) 0.352.  tools/experiments/r6/README.md §8.
template <int kTileH, int kThreads>
__global__ void __launch_bounds__(kThreads, (kThreads == 512 ? 4 : TmShape<kTileH>::kWaves)) k_tonemap_tol(TonemapArgs t) {
    using Shape = TmShape<kTileH>;
    constexpr int kRowStep = kThreads / 16;  // pixel rows (and pass-1 items) a sweep of the workgroup covers
    constexpr int kStageRowStep = kThreads / 32;  // staged rows a sweep of the staging loads covers
    constexpr int kMaxRows = Shape::kMaxRows, kPlane = kMaxRows * kTile * 3, kStageIters = (kMaxRows + kStageRowStep - 1) / kStageRowStep;
    constexpr int kA = kTileH / kRowStep;  // pixel rows per thread: tile rows tr + kRowStep a
    static_assert(kA >= 1 && kTileH % kRowStep == 0, "tile height / threads");
    const uint32_t tid8 = threadIdx.x & 255u;  // the 256 table entries of a mip are fetched and stored by every 256-thread half alike (same values)
    // (s_src and s_xt are read by pass 1 only, which every thread has left before anybody commits the next stage: single buffers; s_yt is
    // read by pass 2, which overlaps the next commit: double buffer)
    __shared__ __attribute__((aligned(16))) float4 s_src[kMaxRows * kPitch];  // staged texels, fp32 rgb (w unused), edge replication applied
    __shared__ __attribute__((aligned(16))) float s_g[4 * kPlane];            // [y variant][staged row of the stage][pixel column][rgb]
    __shared__ __attribute__((aligned(8))) AxisS s_xt[kStageMips][4][kTile], s_yt[2][kStageMips][4][kTileH];
    __shared__ int s_bad[2][kStageMips];  // the mip cannot be staged: strict evaluation from global memory
    __shared__ Rect s_rect[6];            // the six rectangles, by threads 0..5 (six divides each: not per thread and mip)
    // base, pitch and last texel of every mip for the staging loads, whose mip differs from lane to lane (a stage's rows lie side by side):
    // indexing the kernel argument arrays per lane makes the compiler select among all their elements, ~70 instructions per load
    __shared__ uint4 s_mip[6];            // {pointer lo, pointer hi, pitch, (W - 1) | (H - 1) << 16}
    const uint32_t bx = blockIdx.x * kTile, by = t.row_begin + blockIdx.y * kTileH;
    const uint32_t x_last = min(bx + kTile - 1, t.out_w - 1), y_last = min(by + kTileH - 1, t.row_end - 1);
    const uint32_t cp = threadIdx.x & 15u, tr = threadIdx.x >> 4;  // column pair (columns 2 cp, 2 cp + 1), tile rows tr and tr + 16
    const uint32_t nmips = min(t.num_mips, 6u);

    if (threadIdx.x < nmips) {
        s_rect[threadIdx.x] = tile_rect(t, t.mip_w[threadIdx.x], t.mip_h[threadIdx.x], bx, by, x_last, y_last, kMaxRows);
        const uint64_t ptr = reinterpret_cast<uint64_t>(t.mips[threadIdx.x].ptr);
        s_mip[threadIdx.x] = make_uint4((uint32_t)ptr, (uint32_t)(ptr >> 32), t.mips[threadIdx.x].pitch,
                                        (min(t.mip_w[threadIdx.x], 65536u) - 1u) | ((min(t.mip_h[threadIdx.x], 65536u) - 1u) << 16));
    }
    __syncthreads();
    // stages: mips [first, first + count) filtered between one pair of barriers.  A mip whose rows do not fit beside the others of its stage
    // (chains that are not half-resolution pyramids) is marked bad for this tile and evaluated strictly
    auto stage_first_of = [](uint32_t s) __attribute__((always_inline)) { return Shape::first(s); };
    uint32_t nstages = 0;
    while (Shape::first(nstages) < nmips) nstages++;
    auto stage_count = [&](uint32_t s) __attribute__((always_inline)) { return min(Shape::first(s + 1u), nmips) - Shape::first(s); };

    // the thread's table entry of a mip: (axis, variant, column / row) = (tid / 128, tid / 32 % 4, tid % 32)
    const uint32_t entry_off = (((tid8 >> 7) * 4u + ((tid8 >> 5) & 3u)) * t.axis_stride) +
                               ((tid8 >> 7) == 0 ? min(bx + (tid8 & 31u), x_last) : min(by + (tid8 & 31u), y_last));  // past the edge: the last valid one
    uint2 staged[kStageIters];
    TmAxis entry0 = {0, 0.f}, entry1 = entry0, entry2 = entry0;
    // rows of the stage's mips as they lie in s_src / s_g: rb_j = first row of mip j of the stage (rb3: one past the last); a mip that is bad
    // has no rows.  (Named scalars, not arrays: a select among array elements becomes a dynamically indexed array in scratch memory.)
    int rb0 = 0, rb1 = 0, rb2 = 0, rb3 = 0;
    Rect rc0 = {0, 0, 0, 0}, rc1 = rc0, rc2 = rc0;
    auto pick = [](int mj, int a, int b, int c) __attribute__((always_inline)) { return mj == 0 ? a : (mj == 1 ? b : c); };
    auto layout = [&](uint32_t s) __attribute__((always_inline)) {
        const uint32_t first = stage_first_of(s), cnt = stage_count(s);
        const Rect none = {0, 0, 0, 0};
        // (wave-uniform values, made scalar: read from LDS they would sit in vector registers and every select, add and compare on them —
        // most of what a stage's set-up does — would be a vector instruction)
        auto uniform_rect = [](const Rect& r) __attribute__((always_inline)) {
            return Rect{__builtin_amdgcn_readfirstlane(r.x0), __builtin_amdgcn_readfirstlane(r.y0), __builtin_amdgcn_readfirstlane(r.w), __builtin_amdgcn_readfirstlane(r.h)};
        };
        rc0 = cnt > 0u ? uniform_rect(s_rect[first]) : none;
        rc1 = cnt > 1u ? uniform_rect(s_rect[first + 1u]) : none;
        rc2 = cnt > 2u ? uniform_rect(s_rect[first + 2u]) : none;
        rb0 = 0;
        rb1 = rc0.h;
        if (rb1 + rc1.h > kMaxRows) rc1 = none;  // does not fit beside the others: bad
        rb2 = rb1 + rc1.h;
        if (rb2 + rc2.h > kMaxRows) rc2 = none;
        rb3 = rb2 + rc2.h;
    };
    // requests the texels and table entries of stage s: a thread owns texel column tid % 32 and rows tid / 32 + 8 j of the stage's rows
    auto stage_request = [&](uint32_t s) __attribute__((always_inline)) {
        const uint32_t first = stage_first_of(s), cnt = stage_count(s);
        layout(s);
        if (threadIdx.x < (uint32_t)kStageMips) s_bad[s & 1u][threadIdx.x] = threadIdx.x < cnt && pick((int)threadIdx.x, rc0.w, rc1.w, rc2.w) == 0;
        const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
#pragma unroll
        for (int j = 0; j < kStageIters; j++) {
            const int rr = ty0 + kStageRowStep * j;  // row of the stage
            const int mj = (rr >= rb1 ? 1 : 0) + (rr >= rb2 ? 1 : 0);
            const uint32_t m = min(first + (uint32_t)mj, nmips - 1u);
            const uint4 mi = s_mip[m];
            const int wm1 = (int)(mi.w & 0xffffu), hm1 = (int)(mi.w >> 16);
            const int rx0 = pick(mj, rc0.x0, rc1.x0, rc2.x0), ry0 = pick(mj, rc0.y0, rc1.y0, rc2.y0), rbase = pick(mj, rb0, rb1, rb2);
            const uint32_t sx8 = (uint32_t)min(max(rx0 + tx, 0), wm1) * 8u;  // clamped into the image: never predicated
            const uint32_t sy = (uint32_t)min(max(ry0 + (rr - rbase), 0), hm1);
            const uint8_t* base = reinterpret_cast<const uint8_t*>((uint64_t)mi.x | ((uint64_t)mi.y << 32));
            staged[j] = *reinterpret_cast<const uint2*>(base + (sy * mi.z + sx8));
        }
        // table entries: the thread's (axis, variant, column / row) of the three mips of the stage
        auto load_entry = [&](uint32_t q) __attribute__((always_inline)) {
            const uint32_t m = min(first + q, nmips - 1u);
            return t.axis_tables[(size_t)(m * 8u) * t.axis_stride + entry_off];
        };
        entry0 = load_entry(0u);
        entry1 = load_entry(1u);
        entry2 = load_entry(2u);
    };
    // converts and stores the requested texels and builds the axis tables of stage s
    auto stage_commit = [&](uint32_t s) __attribute__((always_inline)) {
        const uint32_t b = s & 1u, cnt = stage_count(s);
        const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
#pragma unroll
        for (int j = 0; j < kStageIters; j++) {
            const int rr = ty0 + kStageRowStep * j;
            if (rr < rb3)
                s_src[rr * kPitch + tx] = make_float4(h2f((uint16_t)(staged[j].x & 0xffffu)), h2f((uint16_t)(staged[j].x >> 16)),
                                                       h2f((uint16_t)(staged[j].y & 0xffffu)), 0.f);
        }
        const uint32_t te = tid8 & (kTile - 1), tk = (tid8 >> 5) & 3u, taxis = tid8 >> 7;
        auto put_entry = [&](uint32_t q, const TmAxis& en, const Rect& r, int rbase) __attribute__((always_inline)) {
            if (q < cnt && r.w > 0) {
                if (taxis == 0) {
                    s_xt[q][tk][te] = {en.i - r.x0, en.f};
                    if (!(en.i >= r.x0 && en.i + 1 <= r.x0 + r.w)) s_bad[b][q] = 1;
                } else if (te < (uint32_t)kTileH) {
                    s_yt[b][q][tk][te] = {(en.i - r.y0 + rbase) * (kTile * 3), en.f};
                    if (!(en.i >= r.y0 && en.i + 1 < r.y0 + r.h)) s_bad[b][q] = 1;
                }
            }
        };
        put_entry(0u, entry0, rc0, rb0);
        put_entry(1u, entry1, rc1, rb1);
        put_entry(2u, entry2, rc2, rb2);
    };

    C3 bloom[kA][2];  // [row r / r + 16][column 2 cp / 2 cp + 1]
#pragma unroll
    for (int a = 0; a < kA; a++)
        for (int c = 0; c < 2; c++) bloom[a][c] = {0.f, 0.f, 0.f};
    const uint32_t x0p = bx + 2u * cp;
    const bool live_col[2] = {x0p < t.out_w, x0p + 1u < t.out_w};
    bool live_row[kA];
#pragma unroll
    for (int a = 0; a < kA; a++) live_row[a] = by + tr + (uint32_t)kRowStep * (uint32_t)a < t.row_end;

    // the scene texels of the thread's pixels are sampled at the END (hoisted to the start, their twelve registers spilled across the whole
    // stage loop: 0.202 against 0.193 ms); one texel per pixel row is touched here so that the lines are in L2 by then
    C3 scene_px[kA][2];
    auto sample_scene = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < kA; a++) {
        const uint32_t y = min(by + tr + (uint32_t)kRowStep * (uint32_t)a, t.row_end - 1u);
        const float v = 1.0f - ((float)y + 0.5f) / (float)t.out_h;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t x = min(x0p + (uint32_t)c, t.out_w - 1);
            const Rgba sc = bilinear<ADDR_CLAMP>(t.scene, t.scene_w, t.scene_h, ((float)x + 0.5f) / (float)t.out_w, v);
            scene_px[a][c] = {sc.c[0], sc.c[1], sc.c[2]};
        }
    }
    };
    {
#pragma unroll
        for (int a = 0; a < kA; a++) {
            const uint32_t y = min(by + tr + (uint32_t)kRowStep * (uint32_t)a, t.row_end - 1u), x = min(x0p, t.out_w - 1);
            const float v = 1.0f - ((float)y + 0.5f) / (float)t.out_h;
            const int sy = min(max((int)(v * (float)t.scene_h), 0), (int)t.scene_h - 1), sx = min(max((int)(((float)x + 0.5f) / (float)t.out_w * (float)t.scene_w), 0), (int)t.scene_w - 1);
            uint32_t w = *reinterpret_cast<const uint32_t*>(t.scene.ptr + (size_t)sy * t.scene.pitch + (size_t)sx * 8);
            asm volatile("" ::"v"(w));
        }
    }
    if (nstages) {
        stage_request(0);
        __syncthreads();  // (s_bad[0] reset before anybody sets it)
        stage_commit(0);
    }
    for (uint32_t s = 0; s < nstages; s++) {
        const uint32_t b = s & 1u, first = stage_first_of(s), cnt = stage_count(s);
        __syncthreads();  // texels and tables of stage s are in place; the previous stage's pass 2 is done with s_g
        // (rc / rb describe stage s here: stage_request(s) was the last one to run; they change when the next stage is requested below)
        const int total_rows = rb3, mr1 = rb1, mr2 = rb2;
        const bool bad0 = cnt > 0u && __builtin_amdgcn_readfirstlane(s_bad[b][0]) != 0, bad1 = cnt > 1u && __builtin_amdgcn_readfirstlane(s_bad[b][1]) != 0,
                   bad2 = cnt > 2u && __builtin_amdgcn_readfirstlane(s_bad[b][2]) != 0;
        {
            // pass 1: items (column pair, staged row of the stage): column pair = tid % 16, rows tid / 16 + 16 j
            for (int r = (int)tr; r < total_rows; r += kRowStep) {
                const int mj = (r >= mr1 ? 1 : 0) + (r >= mr2 ? 1 : 0);
                if (mj == 0 ? bad0 : (mj == 1 ? bad1 : bad2)) continue;  // (its rows hold nothing: rc.h == 0, or the tables are not inside)
                const float4* srow = s_src + r * kPitch;
                float g[4][6];
                // (the LDS reads are volatile — see lds_read16 — and therefore issue in program order: all eight set-ups first, then the
                // texels of a column, so that an item waits for two LDS round trips, not for sixteen one behind the other)
                AxisS xa[2][4];
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int k = 0; k < 4; k++) xa[c][k] = lds_axis(&s_xt[mj][k][2u * cp + (uint32_t)c]);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    float4 tp[4], tq[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        tp[k] = lds_read16(srow + xa[c][k].o);
                        tq[k] = lds_read16(srow + xa[c][k].o + 1);
                    }
                    float h[4][3];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float w1 = xa[c][k].f, w0 = 1.0f - xa[c][k].f;
                        h[k][0] = __builtin_fmaf(w1, tq[k].x, w0 * tp[k].x);
                        h[k][1] = __builtin_fmaf(w1, tq[k].y, w0 * tp[k].y);
                        h[k][2] = __builtin_fmaf(w1, tq[k].z, w0 * tp[k].z);
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
        if (s + 1 < nstages) stage_request(s + 1);  // global loads travel during pass 2 (rc / rb now describe stage s + 1)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kStageMips; j++) {
            if ((uint32_t)j >= cnt) break;
            const uint32_t m = first + (uint32_t)j;
            if (!(j == 0 ? bad0 : (j == 1 ? bad1 : bad2))) {
                // pass 2: four row interpolations per pixel, two pixels at a time.  Program order of the (volatile) LDS reads: the eight set-ups,
                // then the 24 reads of a pixel pair, then its arithmetic — three LDS round trips per mip on the thread's critical path
                AxisS ey[kA][4];
#pragma unroll
                for (int a = 0; a < kA; a++) {
                    const uint32_t pr = min(tr + (uint32_t)kRowStep * (uint32_t)a, y_last - by);  // rows past the band: the last valid one (dropped later)
#pragma unroll
                    for (int yv = 0; yv < 4; yv++) ey[a][yv] = lds_axis(&s_yt[b][j][yv][pr]);
                }
#pragma unroll
                for (int a = 0; a < kA; a++) {
                    float2 P[4][3], Q[4][3];
#pragma unroll
                    for (int yv = 0; yv < 4; yv++) {
                        const float* g0 = s_g + yv * kPlane + ey[a][yv].o + 6 * (int)cp;
                        const float* g1 = g0 + kTile * 3;
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            P[yv][q] = lds_read8(g0 + 2 * q);
                            Q[yv][q] = lds_read8(g1 + 2 * q);
                        }
                    }
                    float sacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int yv = 0; yv < 4; yv++) {
                        const float w1 = ey[a][yv].f, w0 = 0.0625f - ey[a][yv].f;
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            sacc[2 * q] = __builtin_fmaf(w1, Q[yv][q].x, __builtin_fmaf(w0, P[yv][q].x, sacc[2 * q]));
                            sacc[2 * q + 1] = __builtin_fmaf(w1, Q[yv][q].y, __builtin_fmaf(w0, P[yv][q].y, sacc[2 * q + 1]));
                        }
                    }
                    bloom[a][0] = bloom[a][0] + C3{sacc[0], sacc[1], sacc[2]};
                    bloom[a][1] = bloom[a][1] + C3{sacc[3], sacc[4], sacc[5]};
                }
            } else {  // (uniform over the workgroup) a rectangle that does not fit: the strict evaluation from global memory
                const PlaneArg mp = t.mips[m];
#pragma unroll
                for (int a = 0; a < kA; a++)
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const uint32_t y = min(by + tr + (uint32_t)kRowStep * (uint32_t)a, t.row_end - 1), x = min(x0p + (uint32_t)c, t.out_w - 1);
                        bloom[a][c] = bloom[a][c] + tent_blur(mp, t.mip_w[m], t.mip_h[m], ((float)x + 0.5f) / (float)t.out_w, 1.0f - ((float)y + 0.5f) / (float)t.out_h);
                    }
            }
        }
        if (s + 1 < nstages) stage_commit(s + 1);
    }
    const float4* code_tab = reinterpret_cast<const float4*>(t.code_table);
#pragma unroll
    for (int a = 0; a < kA; a++) {
        if (!live_row[a]) continue;
        const uint32_t y = by + tr + (uint32_t)kRowStep * (uint32_t)a;
        uint32_t px[2] = {0u, 0u};
        if (a == 0) sample_scene();
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const C3 sc = scene_px[a][c];
            const C3 col = {__builtin_fmaf(bloom[a][c].r, 0.014159f, sc.r), __builtin_fmaf(bloom[a][c].g, 0.014159f, sc.g),
                            __builtin_fmaf(bloom[a][c].b, 0.014159f, sc.b)};
            const float luma = __builtin_fmaf(col.b, 0.0722f, __builtin_fmaf(col.g, 0.7152f, col.r * 0.2126f));
            // (tolerance mode: the quotient through v_rcp_f32 and one Newton step on it, within an ulp of the IEEE divide's — the same order
            // of error as the re-associated sums above, 11 instructions fewer)
            const float den = luma + 1.f, y0 = __builtin_amdgcn_rcpf(den), q0 = luma * y0;
            const float factor = __builtin_fmaf(__builtin_fmaf(-den, q0, luma), y0, q0);
            const float rgb[3] = {col.r * factor, col.g * factor, col.b * factor};
            uint32_t code[3];  // the code search of tonemap.hip, one table entry per channel: exact for whatever value reaches it
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float xc = __builtin_fminf(__builtin_fmaxf(rgb[ch], t.thr_lo), t.thr_hi);
                const uint32_t bk = (__builtin_bit_cast(uint32_t, xc) >> kTmBucketShift) - t.bucket_base;
                const float4 e = code_tab[bk];  // thresholds first + 1 .. first + 3, first
                code[ch] = __builtin_bit_cast(uint32_t, e.w) + (rgb[ch] >= e.x ? 1u : 0u) + (rgb[ch] >= e.y ? 1u : 0u) + (rgb[ch] >= e.z ? 1u : 0u);
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
    const uint32_t cols = (t.out_w + kTile - 1) / kTile;
    // 32-row tiles for every launch (round 6).  Rounds 4-5 gave row bands that would not fill the chip's 768 workgroup slots twice (a rank's rows of a
    // sharded frame) 16-row tiles, four per CU, for the band's latency BY ITSELF — where the two shapes are level at 270 rows (43.6 against 45.6 us) and the
    // 32-row one ahead below (72 / 144 rows: 16.0 / 18.5 against 19.0 / 27.3).  But a band never runs by itself: in the rank's three-stream frame loop the
    // chip is shared with the lighting and reduction of the next frame, what counts is the composite's WORK, and the 16-row shape stages a third more rows
    // per pixel row (whole frame 0.238 against 0.179 ms): one rank of eight 0.1058 -> 0.0930 ms per frame with 32-row tiles on its band
    // (tools/experiments/r6/README.md §9).  SAH_TM_BAND16=1 brings the old rule back (A/B).
    static const int env_threads = getenv("SAH_TM_THREADS") ? atoi(getenv("SAH_TM_THREADS")) : 0;  // experiments (tools/experiments/r6): 256 / 512
    static const int env_band16 = getenv("SAH_TM_BAND16") ? atoi(getenv("SAH_TM_BAND16")) : 0;
    const bool big = (uint64_t)cols * ((rows + 31) / 32) >= 2 * 768;
    if (big || !env_band16) {
        if (env_threads == 512) hipLaunchKernelGGL((k_tonemap_tol<32, 512>), dim3(cols, (rows + 31) / 32), dim3(512), 0, st, t);
        else hipLaunchKernelGGL((k_tonemap_tol<32, 256>), dim3(cols, (rows + 31) / 32), dim3(256), 0, st, t);
    } else hipLaunchKernelGGL((k_tonemap_tol<16, 256>), dim3(cols, (rows + 15) / 16), dim3(256), 0, st, t);
    return hipGetLastError();
}

}  // namespace sah
