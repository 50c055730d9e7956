"""Round-5 experiment builds (the mechanism is round 4's: tools/experiments/r4/variants.py — a variant is the product source with patches
applied to a COPY, built into build_ab/<name>.so, loaded through SAH_HIP_LIBRARY; nothing here touches csrc/).

    python tools/experiments/r5/variants.py            # every round-5 variant
    python tools/experiments/r5/variants.py NAME ...

fast_stats   VERDICT r4 item 3 (a): k_lighting_fast<CSM, LPV, 4> with counters — per pixel `ndotl > 0`, `ndotl > 0 and shadow != 0`; per
             thread "its four pixels share the LPV cascade / the base cell of the trilinear footprint"; per wave how many PCF / BRDF
             evaluations the present wave votes run against how many a dense (compacted) evaluation would.  Same images as the product
             build (the counters only read); read back with sah_debug_fast_stats (tools/experiments/r5/lit_fractions.py).
"""
import concurrent.futures
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("r4_variants", os.path.join(HERE, "..", "r4", "variants.py"))
base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(base)

VARIANTS = {}

_STATS_KERNEL = r"""
    {   // ---- round-5 counters (experiment build only) ----
        const uint32_t lane_ = threadIdx.x & 63u;
        uint32_t n_surface = 0, n_lit = 0, n_unshadowed = 0, slot_pcf = 0, slot_brdf = 0;
        bool all_surface = active;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const bool surf = active && dbg_surface[i];
            all_surface = all_surface && surf;
            const uint64_t ms = __ballot(surf), ml = __ballot(surf && (dbg_flags[i] & 1u)), mu = __ballot(surf && (dbg_flags[i] & 2u));
            n_surface += (uint32_t)__builtin_popcountll(ms);
            n_lit += (uint32_t)__builtin_popcountll(ml);
            n_unshadowed += (uint32_t)__builtin_popcountll(mu);
            slot_pcf += ml ? 1u : 0u;
            slot_brdf += mu ? 1u : 0u;
        }
        bool same_cascade = all_surface, same_cell = all_surface;
#pragma unroll
        for (int i = 1; i < PPT; i++) {
            same_cascade = same_cascade && dbg_sel[i] == dbg_sel[0];
            same_cell = same_cell && dbg_sel[i] == dbg_sel[0] && dbg_cell[i] == dbg_cell[0];
        }
        const uint64_t m_threads = __ballot(all_surface), m_casc = __ballot(same_cascade), m_cell = __ballot(same_cell), m_act = __ballot(active);
        if (lane_ == 0 && m_act) {
            atomicAdd(&g_fast_stats[0], (unsigned long long)__builtin_popcountll(m_act) * PPT);  // pixels
            atomicAdd(&g_fast_stats[1], (unsigned long long)n_surface);                          // surface pixels
            atomicAdd(&g_fast_stats[2], (unsigned long long)n_lit);                              // ... with ndotl > 0
            atomicAdd(&g_fast_stats[3], (unsigned long long)n_unshadowed);                       // ... and shadow != 0
            atomicAdd(&g_fast_stats[4], 1ull);                                                   // waves
            atomicAdd(&g_fast_stats[5], (unsigned long long)slot_pcf);                           // PCF evaluations (wave x pixel slot) as voted today
            atomicAdd(&g_fast_stats[6], (unsigned long long)slot_brdf);                          // BRDF evaluations as voted today
            atomicAdd(&g_fast_stats[7], (unsigned long long)((n_lit + 63u) / 64u));              // PCF evaluations if the wave's lit pixels were dense
            atomicAdd(&g_fast_stats[8], (unsigned long long)((n_unshadowed + 63u) / 64u));       // BRDF evaluations if dense
            atomicAdd(&g_fast_stats[9], (unsigned long long)__builtin_popcountll(m_threads));    // threads whose four pixels are all surface
            atomicAdd(&g_fast_stats[10], (unsigned long long)__builtin_popcountll(m_casc));      // ... and share the LPV cascade
            atomicAdd(&g_fast_stats[11], (unsigned long long)__builtin_popcountll(m_cell));      // ... and the base cell of the footprint
            atomicAdd(&g_fast_stats[12], (unsigned long long)__builtin_popcountll(m_act));       // threads
        }
    }
"""

VARIANTS["fast_stats"] = (["lighting.hip"], [
    ("lighting_fast.hpp", "struct FastPixelOut {\n    uint2 lit;\n    bool deferred;\n};",
     "struct FastPixelOut {\n    uint2 lit;\n    bool deferred;\n    uint32_t dbg_flags, dbg_cell, dbg_sel;\n};"),
    ("lighting_fast.hpp", "const Surface<Fn>& s, const SurfIn& si, bool sky_px, bool& ok, Fn (&sc)[3]) {",
     "const Surface<Fn>& s, const SurfIn& si, bool sky_px, bool& ok, Fn (&sc)[3], uint32_t* dbg = nullptr) {\n    if (dbg) *dbg = 0u;"),
    ("lighting_fast.hpp", "        shadow = ndotl_sun.v > 0.f ? shadow : 1.0f;\n",
     "        shadow = ndotl_sun.v > 0.f ? shadow : 1.0f;\n"
     "        if (dbg) *dbg = ((ok && !sky_px && ndotl_sun.v > 0.f) ? 1u : 0u) | ((ok && !sky_px && ndotl_sun.v > 0.f && shadow != 0.0f) ? 2u : 0u);\n"),
    ("lighting_common.hpp", "                              const Fn (&n)[4], Fn (&out)[3]) {\n    const int W = (int)L.red.width",
     "                              const Fn (&n)[4], Fn (&out)[3], uint32_t* dbg_base = nullptr) {\n    const int W = (int)L.red.width"),
    ("lighting_common.hpp", "    const uint32_t base = z0 * slice_pitch + y0 * row_pitch + x0 * kLpvPackTexel;\n",
     "    const uint32_t base = z0 * slice_pitch + y0 * row_pitch + x0 * kLpvPackTexel;\n    if (dbg_base) *dbg_base = base;\n"),
    ("lighting_fast.hpp", "    Fn nc[4];\n    float lpv_u = 0.f, lpv_v = 0.f, lpv_w = 0.f;\n", "    Fn nc[4];\n    float lpv_u = 0.f, lpv_v = 0.f, lpv_w = 0.f;\n    uint32_t dbg_flags = 0, dbg_cell = 0, dbg_sel = 0;\n"),
    ("lighting_fast.hpp", "        cpx = cpx + Fn((float)selected);\n", "        cpx = cpx + Fn((float)selected);\n        dbg_sel = selected;\n"),
    ("lighting_fast.hpp", "        fast_csm_sun(a, csm, tab, N, ws, vsz, V, L, s, si, sky_px, ok, sc);", "        fast_csm_sun(a, csm, tab, N, ws, vsz, V, L, s, si, sky_px, ok, sc, &dbg_flags);"),
    ("lighting_fast.hpp", "        lpv_fetch_packed(lpv, f.lpv_packed, f.pk_row_pitch, f.pk_slice_pitch, lpv_u, lpv_v, lpv_w, nc, indirect);",
     "        lpv_fetch_packed(lpv, f.lpv_packed, f.pk_row_pitch, f.pk_slice_pitch, lpv_u, lpv_v, lpv_w, nc, indirect, &dbg_cell);"),
    ("lighting_fast.hpp", "    o.deferred = sky_px ? (f.sky_enabled != 0u) : !ok;\n",
     "    o.deferred = sky_px ? (f.sky_enabled != 0u) : !ok;\n    o.dbg_flags = dbg_flags;\n    o.dbg_cell = dbg_cell;\n    o.dbg_sel = dbg_sel;\n"),
    ("lighting.hip", "constexpr uint32_t kSkyRatio = 4;", "__device__ unsigned long long g_fast_stats[16];\nconstexpr uint32_t kSkyRatio = 4;"),
    ("lighting.hip", "    uint32_t out[2 * PPT];\n    uint32_t deferred_mask = 0, sky_mask = 0;\n",
     "    uint32_t out[2 * PPT];\n    uint32_t deferred_mask = 0, sky_mask = 0;\n    uint32_t dbg_flags[PPT] = {}, dbg_cell[PPT] = {}, dbg_sel[PPT] = {};\n    bool dbg_surface[PPT] = {};\n"),
    ("lighting.hip", "        if (p.depth == 0.f) sky_mask |= 1u << i;\n    }\n",
     "        if (p.depth == 0.f) sky_mask |= 1u << i;\n        dbg_flags[i] = r.dbg_flags;\n        dbg_cell[i] = r.dbg_cell;\n        dbg_sel[i] = r.dbg_sel;\n"
     "        dbg_surface[i] = p.depth != 0.f && !r.deferred;\n    }\n" + _STATS_KERNEL),
    ("lighting.hip", "}  // namespace sah\n",
     "}  // namespace sah\n"
     "extern \"C\" __attribute__((visibility(\"default\"))) int sah_debug_fast_stats(unsigned long long* out, int reset) {\n"
     "    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(sah::g_fast_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;\n"
     "    if (reset) {\n        unsigned long long z[16] = {};\n        if (hipMemcpyToSymbol(HIP_SYMBOL(sah::g_fast_stats), z, sizeof(z)) != hipSuccess) return -1;\n    }\n"
     "    return 0;\n}\n"),
])

# ---- XCD-aware workgroup order (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement": blocks are dealt round-robin over the 8 XCDs, each
# with its own 4 MiB L2).  In launch order, neighbouring tiles of a row land on eight different L2s and every XCD gathers from the whole
# of the side tables (48 MB of probe atlases, the 4.1 MB LPV copy, the shadow cascades); remapped, XCD k shades the k-th contiguous eighth
# of the launch's tiles.  Same images (a permutation of which workgroup shades which tile).
_XCD_REMAP = (
    "    // XCD-aware order: physical block L (XCD L % 8 under round-robin dealing) shades logical tile start(L % 8) + L / 8\n"
    "    {\n        const uint32_t T_ = TOTAL_, L_ = LINEAR_, q_ = T_ / 8u, r_ = T_ % 8u, k_ = L_ % 8u;\n"
    "        LOGICAL_ = k_ * q_ + min(k_, r_) + L_ / 8u;\n    }\n")
VARIANTS["tiled_xcd"] = (["lighting_tiled.hip"], [
    ("lighting_tiled.hip",
     "    const uint32_t x = LIGHTS ? blockIdx.x * 16u + (threadIdx.x & 7u) + ((threadIdx.x >> 3) & 8u) : blockIdx.x * 32u + (threadIdx.x & 31u);\n"
     "    const uint32_t y = a.row_begin + (LIGHTS ? blockIdx.y * 16u + ((threadIdx.x >> 3) & 7u) + ((threadIdx.x >> 4) & 8u) : blockIdx.y * 8u + (threadIdx.x >> 5));\n".replace("\\n", "\n"),
     "    uint32_t logical_ = 0;\n" + _XCD_REMAP.replace("TOTAL_", "gridDim.x * gridDim.y").replace("LINEAR_", "blockIdx.y * gridDim.x + blockIdx.x").replace("LOGICAL_", "logical_") +
     "    const uint32_t bx_ = LIGHTS ? blockIdx.x : logical_ % gridDim.x, by_ = LIGHTS ? blockIdx.y : logical_ / gridDim.x;\n"
     "    const uint32_t x = LIGHTS ? bx_ * 16u + (threadIdx.x & 7u) + ((threadIdx.x >> 3) & 8u) : bx_ * 32u + (threadIdx.x & 31u);\n"
     "    const uint32_t y = a.row_begin + (LIGHTS ? by_ * 16u + ((threadIdx.x >> 3) & 7u) + ((threadIdx.x >> 4) & 8u) : by_ * 8u + (threadIdx.x >> 5));\n"),
])
VARIANTS["fast_xcd"] = (["lighting.hip"], [
    ("lighting.hip", "    uint32_t block_id = blockIdx.x;\n    if (SKY) {\n        if (blockIdx.x % (kSkyRatio + 1u) == kSkyRatio) {",
     "    uint32_t vb_ = 0;\n" + _XCD_REMAP.replace("TOTAL_", "gridDim.x").replace("LINEAR_", "blockIdx.x").replace("LOGICAL_", "vb_") +
     "    uint32_t block_id = vb_;\n    if (SKY) {\n        if (vb_ % (kSkyRatio + 1u) == kSkyRatio) {"),
    ("lighting.hip", "((blockIdx.x / (kSkyRatio + 1u)) * kSkyRatio + k)", "((vb_ / (kSkyRatio + 1u)) * kSkyRatio + k)"),
    ("lighting.hip", "        block_id = blockIdx.x - blockIdx.x / (kSkyRatio + 1u);", "        block_id = vb_ - vb_ / (kSkyRatio + 1u);"),
])
VARIANTS["both_xcd"] = (["lighting.hip", "lighting_tiled.hip"], VARIANTS["fast_xcd"][1] + VARIANTS["tiled_xcd"][1])
# the tolerance composite on a row band: 32-row tiles whatever the band's height (the product picks 16-row tiles when 32-row ones would not
# fill the chip's 768 slots twice)
VARIANTS["tm_band32"] = (["tonemap_tol.hip"], [("tonemap_tol.hip", "    if ((uint64_t)cols * ((rows + 31) / 32) >= 2 * 768) hipLaunchKernelGGL", "    if (true) hipLaunchKernelGGL")])
# Timing-only probe of the composite's occupancy (its images are wrong on purpose): what would a fourth workgroup per CU buy the 32-row shape?
# G in three planes (the third read from the second), staged texels as 8-byte fp16 cells converted at the read, __launch_bounds__(256, 4):
# 38.9 KB of LDS and at most 128 VGPRs.  If this does not move the time, the real restructuring (derive G2 = G1 + G3 at the read, fp16 cells)
# is not worth writing.
VARIANTS["tm_occ4"] = (["tonemap_tol.hip"], [
    ("tonemap_tol.hip", "    static constexpr int kMaxRows = 21, kWaves = 3;", "    static constexpr int kMaxRows = 21, kWaves = 4;"),
    ("tonemap_tol.hip", "    __shared__ __attribute__((aligned(16))) float4 s_src[kMaxRows * kPitch];", "    __shared__ __attribute__((aligned(16))) uint2 s_src[kMaxRows * kPitch];"),
    ("tonemap_tol.hip", "    __shared__ __attribute__((aligned(16))) float s_g[4 * kPlane];", "    __shared__ __attribute__((aligned(16))) float s_g[3 * kPlane];"),
    ("tonemap_tol.hip", "                s_src[rr * kPitch + tx] = make_float4(h2f((uint16_t)(staged[j].x & 0xffffu)), h2f((uint16_t)(staged[j].x >> 16)),\n"
                        "                                                       h2f((uint16_t)(staged[j].y & 0xffffu)), 0.f);",
     "                s_src[rr * kPitch + tx] = staged[j];"),
    ("tonemap_tol.hip", "                const float4* srow = s_src + r * kPitch;", "                const uint2* srow = s_src + r * kPitch;"),
    ("tonemap_tol.hip", "                        tp[k] = lds_read16(srow + xa[c][k].o);\n                        tq[k] = lds_read16(srow + xa[c][k].o + 1);",
     "                        { const float2 a_ = lds_read8(reinterpret_cast<const float*>(srow + xa[c][k].o)); const uint32_t ax_ = __builtin_bit_cast(uint32_t, a_.x), ay_ = __builtin_bit_cast(uint32_t, a_.y);\n"
     "                          tp[k] = make_float4(h2f((uint16_t)(ax_ & 0xffffu)), h2f((uint16_t)(ax_ >> 16)), h2f((uint16_t)(ay_ & 0xffffu)), 0.f);\n"
     "                          const float2 b_ = lds_read8(reinterpret_cast<const float*>(srow + xa[c][k].o + 1)); const uint32_t bx_ = __builtin_bit_cast(uint32_t, b_.x), by_ = __builtin_bit_cast(uint32_t, b_.y);\n"
     "                          tq[k] = make_float4(h2f((uint16_t)(bx_ & 0xffffu)), h2f((uint16_t)(bx_ >> 16)), h2f((uint16_t)(by_ & 0xffffu)), 0.f); }"),
    ("tonemap_tol.hip", "                for (int p = 0; p < 4; p++) {\n                    float2* d2 = reinterpret_cast<float2*>(dst + p * kPlane);  // 24-byte stride: 8-byte aligned",
     "                for (int p = 0; p < 4; p++) {\n                    if (p == 2) continue;\n                    float2* d2 = reinterpret_cast<float2*>(dst + (p == 3 ? 2 : p) * kPlane);  // 24-byte stride: 8-byte aligned"),
    ("tonemap_tol.hip", "                        const float* g0 = s_g + yv * kPlane + ey[a][yv].o + 6 * (int)cp;", "                        const float* g0 = s_g + (yv == 3 ? 2 : (yv == 2 ? 1 : yv)) * kPlane + ey[a][yv].o + 6 * (int)cp;"),
])
# Timing-only upper bound of the "wave-uniform probe cell" idea: every lane takes lane 0's probe cell, so the cell's
# indices, validity bytes, atlas origins and layers become scalar work (images wrong wherever a wave straddles a cell).  What this build
# gains over the product is the most a two-path kernel could gain on waves that really share their cell.
VARIANTS["tiled_uniform_cell"] = (["lighting_tiled.hip"], [
    # TIMING ONLY (wrong image): every wave takes its first lane's probe cell, and nothing is re-evaluated — the upper bound of what a
    # wave-uniform cell path can save (the compiler moves what then depends on uniform values alone to the scalar unit)
    ("lighting_gi_ext.hpp", "        const Fn mp = Fn(__builtin_floorf(psa[k].v));\n        const Fn alpha = nclamp(psa[k] - mp, Fn(0.f), Fn(1.f));",
     "        const Fn mp = Fn(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, __builtin_floorf(psa[k].v)))));\n"
     "        const Fn alpha = nclamp(psa[k] - mp, Fn(0.f), Fn(1.f));"),
    ("lighting_gi_ext.hpp", "    if (weight.v == 0.f) return F3(Fn(0.f));\n    // the three quotients through one refined reciprocal",
     "    bad = false;\n    if (weight.v == 0.f) return F3(Fn(0.f));\n    // the three quotients through one refined reciprocal"),
    ("lighting_gi_ext.hpp", "        bad = bad | !((mn >= __builtin_bit_cast(uint32_t, kDivLo) - 1u) & (mx <= __builtin_bit_cast(uint32_t, kDivHi)));",
     "        bad = false;"),
])
# the same with nothing re-evaluated but every lane's own cell: the reference point of the timing probe above
VARIANTS["tiled_no_redo"] = (["lighting_tiled.hip"], VARIANTS["tiled_uniform_cell"][1][1:])
# Beside the lighting of the next frame (three work streams) the band composite is not on the critical path, but its latency-bound waves
# hold registers the VALU-bound lighting waves could use: what if it may keep only two (one) workgroups per CU resident?  LDS padding, same images.
for _n, _pad in (("tm_band_2wg", 10000), ("tm_band_1wg", 24000)):
    VARIANTS[_n] = (["tonemap_tol.hip"], [
        ("tonemap_tol.hip", "    __shared__ int s_bad[2][kStageMips];  // the mip cannot be staged: strict evaluation from global memory",
         "    __shared__ int s_bad[2][kStageMips];  // the mip cannot be staged: strict evaluation from global memory\n"
         "    __shared__ float s_pad_[kTileH == 16 ? %d : 1];\n    if (t.out_w == 0xffffffffu) s_pad_[threadIdx.x & 0] = 1.f;" % _pad),
        ("tonemap_tol.hip", "    const float4* code_tab = reinterpret_cast<const float4*>(t.code_table);", "    const float4* code_tab = reinterpret_cast<const float4*>(t.code_table);\n    if (t.out_w == 0xfffffffeu) bloom[0][0].r += s_pad_[0];"),
    ])
# Counter build of the cache-GI gather ("wave-uniform probe-cell path"): how many waves hold one probe cell only, and how many
# of a pixel's eight probes are evaluated.  tools/experiments/r5/cell_fractions.py reads the counters.
VARIANTS["tiled_cell_stats"] = (["lighting_tiled.hip"], [
    ("lighting_tiled.hip", '#include "lighting_gi_ext.hpp"', 'namespace sah { __device__ unsigned long long g_cell_stats[16]; }\n#include "lighting_gi_ext.hpp"'),
    ("lighting_gi_ext.hpp", "    F3 irradiance = F3(Fn(0.f));\n    Fn weight = Fn(0.f);\n#pragma unroll\n    for (uint32_t i = 0; i < 8; i++) {\n        const int jx = i & 1u",
     "    {\n"
     "        const uint32_t cell = pidx[0][0] | (pidx[1][0] << 8) | (pidx[2][0] << 16) | (cascade_index << 24);\n"
     "        const bool same = cell == (uint32_t)__builtin_amdgcn_readfirstlane((int)cell);\n"
     "        const uint32_t first = (uint32_t)__builtin_ctzll(lanes(true));\n"
     "        uint32_t nvalid = 0;\n"
     "        for (uint32_t i = 0; i < 8; i++) nvalid += (vbyte[i] & vmask[0][i & 1u] & vmask_yz[(i >> 2) & 1u][(i >> 1) & 1u]) != 0u ? 1u : 0u;\n"
     "        const uint32_t active = (uint32_t)__builtin_popcountll(lanes(true)), same_n = (uint32_t)__builtin_popcountll(lanes(same));\n"
     "        uint32_t nv_wave = nvalid;\n"
     "        for (int o = 32; o >= 1; o >>= 1) nv_wave += (uint32_t)__shfl_xor((int)nv_wave, o);\n"
     "        const uint32_t row_same = __builtin_popcountll(lanes(same) & 0xffffffffull), any_valid_m = (uint32_t)__builtin_popcountll(lanes(nvalid != 0u));\n"
     "        if ((threadIdx.x & 63u) == first) {\n"
     "            atomicAdd(&g_cell_stats[0], 1ull);                              // waves that gather\n"
     "            atomicAdd(&g_cell_stats[1], (unsigned long long)active);       // pixels that gather\n"
     "            atomicAdd(&g_cell_stats[2], same_n == active ? 1ull : 0ull);    // waves whose pixels share one cell (and cascade)\n"
     "            atomicAdd(&g_cell_stats[3], (unsigned long long)same_n);       // pixels in the cell of their wave's first pixel\n"
     "            atomicAdd(&g_cell_stats[4], (unsigned long long)nv_wave);      // probes evaluated (sum over pixels)\n"
     "            atomicAdd(&g_cell_stats[5], active == 64u ? 1ull : 0ull);       // full waves\n"
     "            atomicAdd(&g_cell_stats[6], (unsigned long long)any_valid_m);  // pixels with at least one probe\n"
     "        }\n"
     "    }\n"
     "    F3 irradiance = F3(Fn(0.f));\n    Fn weight = Fn(0.f);\n#pragma unroll\n    for (uint32_t i = 0; i < 8; i++) {\n        const int jx = i & 1u"),
    ("lighting_tiled.hip", "}  // namespace sah\n",
     "}  // namespace sah\n"
     "extern \"C\" __attribute__((visibility(\"default\"))) int sah_debug_cell_stats(unsigned long long* out, int reset) {\n"
     "    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(sah::g_cell_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;\n"
     "    if (reset) {\n        unsigned long long z[16] = {};\n        if (hipMemcpyToSymbol(HIP_SYMBOL(sah::g_cell_stats), z, sizeof(z)) != hipSuccess) return -1;\n    }\n"
     "    return 0;\n}\n"),
])
# The general sample_cascade() — the inline re-evaluation of pixels outside the hot form's domains — as a real function call instead of 1,400
# inlined instructions in the middle of the kernel (tiled_no_redo above says what its mere presence costs: 2 %)
VARIANTS["tiled_redo_call"] = (["lighting_tiled.hip"], [
    ("lighting_gi_ext.hpp", "SAH_DEV F3 sample_cascade(const CacheArgs& c, F3 location, F3 direction, uint32_t cascade_index) {",
     "__device__ __attribute__((noinline)) F3 sample_cascade(const CacheArgs& c, F3 location, F3 direction, uint32_t cascade_index) {"),
])
# The sky workgroups of the fast kernel at the END of the grid instead of interleaved 1 : 4 — would their work fill the tail of the surface
# workgroups (the launch's last ~6 us run at falling occupancy)?  Same images.
VARIANTS["fast_sky_last"] = (["lighting.hip"], [
    ("lighting.hip", "        if (blockIdx.x % (kSkyRatio + 1u) == kSkyRatio) {", "        const uint32_t nsurf_ = gridDim.x / (kSkyRatio + 1u) * kSkyRatio;\n        if (blockIdx.x >= nsurf_) {"),
    ("lighting.hip", "                const uint32_t gid = ((blockIdx.x / (kSkyRatio + 1u)) * kSkyRatio + k) * 256u + threadIdx.x;",
     "                const uint32_t gid = ((blockIdx.x - nsurf_) * kSkyRatio + k) * 256u + threadIdx.x;"),
    ("lighting.hip", "        block_id = blockIdx.x - blockIdx.x / (kSkyRatio + 1u);", "        block_id = blockIdx.x;"),
])
# ... and at the START (their depth reads warm nothing, but their arithmetic runs beside the surface workgroups' first loads)
VARIANTS["fast_sky_first"] = (["lighting.hip"], [
    ("lighting.hip", "        if (blockIdx.x % (kSkyRatio + 1u) == kSkyRatio) {", "        const uint32_t nsky_ = gridDim.x / (kSkyRatio + 1u);\n        if (blockIdx.x < nsky_) {"),
    ("lighting.hip", "                const uint32_t gid = ((blockIdx.x / (kSkyRatio + 1u)) * kSkyRatio + k) * 256u + threadIdx.x;",
     "                const uint32_t gid = (blockIdx.x * kSkyRatio + k) * 256u + threadIdx.x;"),
    ("lighting.hip", "        block_id = blockIdx.x - blockIdx.x / (kSkyRatio + 1u);", "        block_id = blockIdx.x - nsky_;"),
])
for _k, _v in list(VARIANTS.items()):  # (the patch texts above are written with escaped newlines for readability)
    VARIANTS[_k] = (_v[0], [(f, o.replace("\\n", "\n"), n.replace("\\n", "\n")) for f, o, n in _v[1]])


def build_variant(name):
    base.VARIANTS[name] = VARIANTS[name]
    return base.build_variant(name)


if __name__ == "__main__":
    names = sys.argv[1:] or sorted(VARIANTS)
    base.base_build.build()
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        for n in ex.map(build_variant, names):
            print("built", n, flush=True)
