// Fused "Lighting" pass for gfx950 (MI355X): one pass over the G-buffer replaces the reference's clear + sun + GI overlay
// + emissive + sky draws and the RT-mode sun dispatch (RenderCore/render/phase/lighting_phase.cpp:99-134), reading each
// plane once and writing lit_scene once, while reproducing the intermediate RGBA16F blend roundings (DESIGN.md "a0").
//
// Three kernels:
//   k_lighting_fast     the hot kernel.  One thread shades PPT horizontally adjacent pixels (PPT=4: 16 B/lane plane loads,
//                       16 B/lane lit stores).  It exploits what the host verified about the uniform blocks (projection
//                       inverse separable in x/y/depth, affine view inverse, scale+translate LPV cascades, affine shadow
//                       matrices) and skips sub-expressions that are provably inert for the pixel at hand (DESIGN.md "Fast
//                       path proofs").  Pixels it cannot prove anything about (sky, non-finite inputs, roughness 0 under
//                       LPV GI, ...) are appended to a per-frame list with one wave-aggregated atomic.
//   k_lighting_fixup    shades the listed pixels with the general restatement (lighting_common.hpp).
//   k_lighting_general  the general restatement over the whole image: used when the uniform blocks do not have the
//                       structure the fast kernel needs, and as the in-library cross-check of the fast path.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/sah_hip.h"
#include "lighting_common.hpp"
#include "lighting_fast.hpp"
#include "numerics.hpp"
#include "params.hpp"

namespace sah {

// ---- general kernel -------------------------------------------------------------------------------------------
template <int SUN, int GI, int PPT>
__global__ void __launch_bounds__(256) k_lighting_general(const LightingArgs a, const CsmArgs csm, const LpvArgs lpv, const SkyArgs sky) {
    __shared__ float s_lut[512];
    s_lut[threadIdx.x] = a.luts[threadIdx.x];
    s_lut[threadIdx.x + 256] = a.luts[threadIdx.x + 256];
    __syncthreads();

    const uint32_t groups_per_row = a.width / PPT;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t rows = a.row_end - a.row_begin;
    if (gid >= groups_per_row * rows) return;
    const uint32_t ry = gid / groups_per_row;
    const uint32_t y = a.row_begin + ry;
    const uint32_t x0 = (gid - ry * groups_per_row) * PPT;

    uint32_t wc[PPT], wd[PPT], we[PPT], wz[PPT], wn[2 * PPT], wao[PPT], wm[PPT];
    load_words<PPT>(a.color.ptr + (size_t)y * a.color.pitch + (size_t)x0 * 4, wc);
    load_words<PPT>(a.data.ptr + (size_t)y * a.data.pitch + (size_t)x0 * 4, wd);
    load_words<PPT>(a.emission.ptr + (size_t)y * a.emission.pitch + (size_t)x0 * 4, we);
    load_words<PPT>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x0 * 4, wz);
    load_words<2 * PPT>(a.normals.ptr + (size_t)y * a.normals.pitch + (size_t)x0 * 8, wn);
    if (GI == SAH_GI_LPV && a.has_ao) load_words<PPT>(a.ao.ptr + (size_t)y * a.ao.pitch + (size_t)x0 * 4, wao);
    if (SUN == SAH_SHADOW_MODE_RT && a.has_mask) load_words<PPT>(a.shadow_mask.ptr + (size_t)y * a.shadow_mask.pitch + (size_t)x0 * 4, wm);

    uint32_t out[2 * PPT];
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        Px p;
        p.color = wc[i];
        p.data = wd[i];
        p.emission = we[i];
        p.n01 = wn[2 * i];
        p.n23 = wn[2 * i + 1];
        p.depth = __uint_as_float(wz[i]);
        p.ao = (GI == SAH_GI_LPV && a.has_ao) ? __uint_as_float(wao[i]) : 1.0f;
        p.mask = (SUN == SAH_SHADOW_MODE_RT && a.has_mask) ? __uint_as_float(wm[i]) : 1.0f;
        const uint2 r = shade_pixel_general<SUN, GI>(a, csm, lpv, sky, x0 + i, y, p, s_lut);
        out[2 * i] = r.x;
        out[2 * i + 1] = r.y;
    }
    uint8_t* dst = const_cast<uint8_t*>(a.lit.ptr) + (size_t)y * a.lit.pitch + (size_t)x0 * 8;
    if constexpr (PPT == 1) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(out[0], out[1]);
    } else {
        store_words<2 * PPT>(dst, out);
    }
}

// ---- fix-up kernel: general restatement over the deferred pixels -------------------------------------------------------------------
// The fast kernel leaves, per wave, a segment of byte codes (lane * PPT + pixel) and their count (params.hpp: FastArgs).  One
// workgroup takes kFixupSegs consecutive segments, scans their counts in LDS and deals the listed pixels to its threads, so lanes
// stay busy whether a segment holds one stray pixel or is all sky.  No atomics, no list clearing: every wave of the fast kernel
// rewrites its count.
constexpr uint32_t kFixupSegs = 16;
// (at most this many workgroups, striding over the groups of segments: 512 leave an empty list 0.7 us sooner than 2,048, but take
//  3.7 % longer over a frame that lists 29 K pixels — tools/experiments/r5/ab_fixup_grid.sh)
#ifndef SAH_FIXUP_MAX_WGS
#define SAH_FIXUP_MAX_WGS 2048
#endif
constexpr uint32_t kFixupMaxWorkgroups = SAH_FIXUP_MAX_WGS;
template <int SUN, int GI, int PPT>
__global__ void __launch_bounds__(256) k_lighting_fixup(const LightingArgs a, const CsmArgs csm, const LpvArgs lpv, const SkyArgs sky,
                                                        const FastArgs f) {
    // nothing listed by any wave (FrameState::deferred_hint): every workgroup leaves here
    if (__hip_atomic_load(&f.state->deferred_hint[f.hint_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    __shared__ uint32_t s_pref[kFixupSegs + 1];
    __shared__ float s_lut[512];
    s_lut[threadIdx.x] = a.luts[threadIdx.x];
    s_lut[threadIdx.x + 256] = a.luts[threadIdx.x + 256];
    const uint32_t ngroups = (f.num_segments + kFixupSegs - 1u) / kFixupSegs;
    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    __syncthreads();  // (s_pref of the previous group is read no more; the LUT is in place)
    const uint32_t seg0 = grp * kFixupSegs;
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        const uint32_t c = (lane < kFixupSegs && seg0 + lane < f.num_segments) ? (uint32_t)f.seg_count[seg0 + lane] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < (int)kFixupSegs; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += t;
        }
        if (lane < kFixupSegs) s_pref[lane + 1] = incl;
        if (lane == 0) s_pref[0] = 0;
    }
    __syncthreads();
    const uint32_t total = s_pref[kFixupSegs];
    if (total == 0) continue;
    const uint32_t groups_per_row = a.width / PPT;
    for (uint32_t i = threadIdx.x; i < total; i += 256u) {
        uint32_t s = 0;  // last segment whose prefix is <= i (counts are small: a 4-step search over 16 LDS words)
#pragma unroll
        for (uint32_t step = kFixupSegs / 2; step >= 1; step >>= 1)
            if (s_pref[s + step] <= i) s += step;
        const uint32_t code = f.seg_list[(size_t)(seg0 + s) * f.seg_stride + (i - s_pref[s])];
        const uint32_t g = (seg0 + s) * 64u + code / PPT;
        const uint32_t ry = g / groups_per_row;
        const uint32_t y = a.row_begin + ry, x = (g - ry * groups_per_row) * PPT + code % PPT;
        Px p;
        p.color = *reinterpret_cast<const uint32_t*>(a.color.ptr + (size_t)y * a.color.pitch + (size_t)x * 4);
        p.data = *reinterpret_cast<const uint32_t*>(a.data.ptr + (size_t)y * a.data.pitch + (size_t)x * 4);
        p.emission = *reinterpret_cast<const uint32_t*>(a.emission.ptr + (size_t)y * a.emission.pitch + (size_t)x * 4);
        p.depth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x * 4);
        const uint2 n = *reinterpret_cast<const uint2*>(a.normals.ptr + (size_t)y * a.normals.pitch + (size_t)x * 8);
        p.n01 = n.x;
        p.n23 = n.y;
        p.ao = (GI == SAH_GI_LPV && a.has_ao) ? *reinterpret_cast<const float*>(a.ao.ptr + (size_t)y * a.ao.pitch + (size_t)x * 4) : 1.0f;
        p.mask = (SUN == SAH_SHADOW_MODE_RT && a.has_mask)
                     ? *reinterpret_cast<const float*>(a.shadow_mask.ptr + (size_t)y * a.shadow_mask.pitch + (size_t)x * 4)
                     : 1.0f;
        const uint2 r = shade_pixel_general<SUN, GI>(a, csm, lpv, sky, x, y, p, s_lut);
        *reinterpret_cast<uint2*>(const_cast<uint8_t*>(a.lit.ptr) + (size_t)y * a.lit.pitch + (size_t)x * 8) = r;
    }
    }
}

// ---- sky kernel: the deferred depth == 0 pixels (back half of the segments) ---------------------------------------------------------
// ---- LPV gather copy + finiteness scan -----------------------------------------------------------------------------------
// Once per Lighting pass (3 MiB in, 4 MiB out, L2 resident): interleaves the three RGBA16F volumes into 24-byte texels
// {R[4], G[4], B[4]} surrounded by a two-texel border of zeros, and flags inf / NaN texels (feeds the "specular quirk is inert"
// proof).  The fast kernel then gathers one trilinear footprint as 4 (y,z) rows x 48 contiguous bytes = 12 dwordx4 loads touching
// 4-5 cache lines, instead of 24 dwordx2 loads over three allocations, and CLAMP_TO_BORDER needs no per-tap masking: border
// texels are real zeros.
__global__ void __launch_bounds__(256) k_lpv_pack(const VolumeArg r, const VolumeArg g, const VolumeArg b, uint8_t* packed, uint32_t pk_row_pitch,
                                                  uint32_t pk_slice_pitch, FrameState* state) {
    const uint32_t prow = blockIdx.x;  // one block per padded (z, y) row
    const uint32_t ph = r.height + 2 * kLpvPackBorder;
    const uint32_t pz = prow / ph, py = prow - pz * ph;
    const int z = (int)pz - (int)kLpvPackBorder, y = (int)py - (int)kLpvPackBorder;
    const bool row_inside = z >= 0 && z < (int)r.depth && y >= 0 && y < (int)r.height;
    uint32_t bad = 0;
    uint8_t* dst_row = packed + (size_t)pz * pk_slice_pitch + (size_t)py * pk_row_pitch;
    for (uint32_t px = threadIdx.x; px < r.width + 2 * kLpvPackBorder; px += 256) {
        const int x = (int)px - (int)kLpvPackBorder;
        uint2 t[3] = {make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
        if (row_inside && x >= 0 && x < (int)r.width) {
            t[0] = *reinterpret_cast<const uint2*>(r.ptr + (size_t)z * r.slice_pitch + (size_t)y * r.row_pitch + (size_t)x * 8);
            t[1] = *reinterpret_cast<const uint2*>(g.ptr + (size_t)z * g.slice_pitch + (size_t)y * g.row_pitch + (size_t)x * 8);
            t[2] = *reinterpret_cast<const uint2*>(b.ptr + (size_t)z * b.slice_pitch + (size_t)y * b.row_pitch + (size_t)x * 8);
#pragma unroll
            for (int c = 0; c < 3; c++) {  // fp16 exponent all ones <=> inf or NaN
                bad |= ((t[c].x & 0x7c00u) == 0x7c00u) | ((t[c].x & 0x7c000000u) == 0x7c000000u) | ((t[c].y & 0x7c00u) == 0x7c00u) |
                       ((t[c].y & 0x7c000000u) == 0x7c000000u);
            }
        }
        uint2* dst = reinterpret_cast<uint2*>(dst_row + (size_t)px * kLpvPackTexel);
        dst[0] = t[0];
        dst[1] = t[1];
        dst[2] = t[2];
    }
    if (wave_any(bad != 0) && (threadIdx.x & 63) == 0) atomicMax(&state->nonfinite, 1u);
}

// per-column numerator of the view-space x (inverse_projection separable: vs.x = p0 * ndc.x + p12), with the two texcoord conventions:
// GLSL ((x + 0.5) + 0.5) / W (gl_FragCoord already carries the half), Slang (x + 0.5) / W
SAH_DEV float colx_glsl_of(const LightingArgs& a, const FastArgs& f, uint32_t x) {
    const Fn tx = (Fn((float)x + 0.5f) + Fn(0.5f)) / Fn(a.res[0]);
    return (Fn(f.p0) * (tx * Fn(2.0f) - Fn(1.0f)) + Fn(f.p12)).v;
}
SAH_DEV float colx_slang_of(const LightingArgs& a, const FastArgs& f, uint32_t x) {
    const Fn tx = (Fn((float)x) + Fn(0.5f)) / Fn(a.res[0]);
    return (Fn(f.p0) * (tx * Fn(2.0f) - Fn(1.0f)) + Fn(f.p12)).v;
}
// the same values for every column, once per (width, render resolution, p0, p12): the kernel then loads PPT of them instead of dividing
// ... and the per-row numerators of the view-space y (vs.y = p5 * ndc.y + p13), rows [0, height): at out + 2 * stride (GLSL) and
// out + 2 * stride + row_stride (Slang) — two IEEE divides per THREAD otherwise, which at four pixels per thread is 8 instructions per pixel
SAH_DEV float rowy_glsl_of(const LightingArgs& a, const FastArgs& f, uint32_t y) {
    const Fn ty = (Fn((float)y + 0.5f) + Fn(0.5f)) / Fn(a.res[1]);
    return (Fn(f.p5) * (ty * Fn(2.0f) - Fn(1.0f)) + Fn(f.p13)).v;
}
SAH_DEV float rowy_slang_of(const LightingArgs& a, const FastArgs& f, uint32_t y) {
    const Fn ty = (Fn((float)y) + Fn(0.5f)) / Fn(a.res[1]);
    return (Fn(f.p5) * (ty * Fn(2.0f) - Fn(1.0f)) + Fn(f.p13)).v;
}
__global__ void __launch_bounds__(256) k_colx_table(const LightingArgs a, const FastArgs f, float* out, uint32_t stride, uint32_t row_stride) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < a.width) {
        out[i] = colx_glsl_of(a, f, i);
        out[stride + i] = colx_slang_of(a, f, i);
    }
    if (i < a.height) {
        out[2u * stride + i] = rowy_glsl_of(a, f, i);
        out[2u * stride + row_stride + i] = rowy_slang_of(a, f, i);
    }
}

// surface workgroups per sky workgroup: FastArgs::sky_ratio, chosen by the host with the launch (api.cpp: a whole 4K frame 4 — 1, 2, 8, 16 measured
// worse: DESIGN.md §5 "Deferred pixels ... and the sky" —; small launches fewer, so that a thread of a sky workgroup over an all-sky stretch walks
// fewer pixels one after the other: that walk is the launch's critical path when the launch is short)

// ---- fast kernel (per-pixel body: lighting_fast.hpp) ----------------------------------------------------------------------
template <int SUN, int GI, int PPT, bool SKY>
// (106 VGPRs at 4 px/thread = 4 waves per SIMD.  Forcing 5 or 6 with amdgpu_waves_per_eu spills 36-44 bytes per lane and is 12-20 %
// slower, measured.)
// (The sky path — two fp64 transcendentals per pixel — wants 126-137 VGPRs whatever stands next to it, and a kernel has ONE register count: left
// alone the allocator takes 133-137 = 3 waves per SIMD for every wave of the 4- and 2-pixel bodies.  The bound holds every SKY body at four waves
// per SIMD: the 4-pixel body spills 12 bytes of the sky path, the others nothing.  Holding the SKY bodies at their sky-less twins' occupancy (5-8
// waves: 72-96 VGPRs) was built and measured in round 6: the sky path then spills 160-236 bytes per lane and its waves become the launch's tail —
// 1280 x 720 0.0222 -> 0.0380 ms, 4K CSM only 0.118 -> 0.151: tools/experiments/r6/README.md §2.)
__global__ void __launch_bounds__(256, SKY ? 4 : 1) k_lighting_fast(const LightingArgs a, const CsmArgs csm, const LpvArgs lpv, const SkyArgs sky,
                                                                                 const FastArgs f) {
    // Sky.  ProceduralSky::render_sky overwrites lit_scene where depth == 0 (sky_unified.slang:185-206): those pixels need their
    // coordinates and two fp64 transcendentals each, nothing of the surface code.  With a sky bound, one workgroup in sky_ratio + 1
    // is a sky workgroup: it reads the depth of the pixels of the sky_ratio surface workgroups before it and shades the sky among them;
    // surface workgroups leave those pixels alone.  Interleaved like this the sky's arithmetic fills issue slots the surface waves leave
    // idle (as a kernel of its own behind this one it cost 20 us on the 4 % of sky in the atrium frame: one busy wave per SIMD, in series).
    uint32_t block_id = blockIdx.x;
    if (SKY) {
        const uint32_t ratio = f.sky_ratio, period = ratio + 1u, turn = f.sky_first ? blockIdx.x : blockIdx.x / period;
        if (f.sky_first ? blockIdx.x < f.sky_first : blockIdx.x - turn * period == ratio) {
            const uint32_t groups_per_row = a.width / PPT, total = groups_per_row * (a.row_end - a.row_begin);
#pragma unroll 1
            for (uint32_t k = 0; k < ratio; k++) {
                const uint32_t gid = (turn * ratio + k) * 256u + threadIdx.x;
                if (gid >= total) break;
                const uint32_t ry = gid / groups_per_row, y = a.row_begin + ry, x0 = (gid - ry * groups_per_row) * PPT;
                uint32_t wz[PPT];
                load_words<PPT>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x0 * 4, wz);
#pragma unroll 1
                for (int i = 0; i < PPT; i++) {
                    if (__uint_as_float(wz[i]) != 0.f) continue;
                    Hn lit[4];
                    sky_frag(a, sky, x0 + (uint32_t)i, y, lit);
                    *reinterpret_cast<uint2*>(const_cast<uint8_t*>(a.lit.ptr) + (size_t)y * a.lit.pitch + (size_t)(x0 + i) * 8) = pack_lit(lit);
                }
            }
            return;
        }
        block_id = f.sky_first ? blockIdx.x - f.sky_first : blockIdx.x - turn;
    }
    // the thread's plane loads are requested first: the LUT staging and its barrier below then overlap their latency
    const uint32_t groups_per_row = a.width / PPT;
    const uint32_t gid = block_id * 256u + threadIdx.x;
    const uint32_t rows = a.row_end - a.row_begin;
    const bool active = gid < groups_per_row * rows;
    const uint32_t ry = active ? (f.row_magic ? __umulhi(gid, f.row_magic) : gid / groups_per_row) : 0u;
    const uint32_t y = a.row_begin + ry;
    const uint32_t x0 = active ? (gid - ry * groups_per_row) * PPT : 0u;
    const bool lpv_bad = (GI == SAH_GI_LPV) ? (f.state->nonfinite != 0u) : false;

    uint32_t wc[PPT], wd[PPT], we[PPT], wz[PPT], wn[2 * PPT], wao[PPT], wm[PPT];
    float colx_g[PPT], colx_s[PPT];
    if (active) {
        load_words<PPT>(a.color.ptr + (size_t)y * a.color.pitch + (size_t)x0 * 4, wc);
        load_words<PPT>(a.data.ptr + (size_t)y * a.data.pitch + (size_t)x0 * 4, wd);
        load_words<PPT>(a.emission.ptr + (size_t)y * a.emission.pitch + (size_t)x0 * 4, we);
        load_words<PPT>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x0 * 4, wz);
        load_words<2 * PPT>(a.normals.ptr + (size_t)y * a.normals.pitch + (size_t)x0 * 8, wn);
        if (GI == SAH_GI_LPV && a.has_ao) load_words<PPT>(a.ao.ptr + (size_t)y * a.ao.pitch + (size_t)x0 * 4, wao);
        if (SUN == SAH_SHADOW_MODE_RT && a.has_mask) load_words<PPT>(a.shadow_mask.ptr + (size_t)y * a.shadow_mask.pitch + (size_t)x0 * 4, wm);
        if (f.colx_tab) {  // (uniform) the per-column numerators of the view-space x: one load instead of PPT IEEE divides
            uint32_t t[PPT];
            if (SUN == SAH_SHADOW_MODE_CSM || GI == SAH_GI_LPV) {
                load_words<PPT>(reinterpret_cast<const uint8_t*>(f.colx_tab + x0), t);
#pragma unroll
                for (int i = 0; i < PPT; i++) colx_g[i] = __uint_as_float(t[i]);
            }
            if (SUN == SAH_SHADOW_MODE_RT) {
                load_words<PPT>(reinterpret_cast<const uint8_t*>(f.colx_tab + f.colx_stride + x0), t);
#pragma unroll
                for (int i = 0; i < PPT; i++) colx_s[i] = __uint_as_float(t[i]);
            }
        }
    }
    __shared__ __attribute__((aligned(16))) float s_lut[TAB_SIZE];
    s_lut[threadIdx.x] = a.luts[threadIdx.x];
    s_lut[threadIdx.x + 256] = a.luts[threadIdx.x + 256];
    if (SUN == SAH_SHADOW_MODE_CSM && threadIdx.x < 48) {  // [cascade][row x,y,z][col 0..3] of biasMat * cascade_matrices
        const uint32_t c = threadIdx.x / 12u, j = threadIdx.x % 12u;
        s_lut[TAB_CSM + threadIdx.x] = csm.biased[c][(j & 3u) * 4u + (j >> 2)];
    }
    if (GI == SAH_GI_LPV && threadIdx.x >= 64 && threadIdx.x < 96) {  // [cascade][sx sy sz - tx ty tz -]
        const uint32_t t = threadIdx.x - 64u, c = t >> 3, j = t & 7u;
        s_lut[TAB_LPV + t] = (j & 3u) == 3u ? 0.f : (j < 4u ? f.lpv_s[c][j] : f.lpv_t[c][j - 4u]);
    }
    if (threadIdx.x >= 128 && threadIdx.x < 140) {  // rows x,y,z of the (affine) inverse view matrix: (m[i], m[4+i], m[8+i], m[12+i])
        const uint32_t t = threadIdx.x - 128u;
        s_lut[TAB_VIEW + t] = a.inv_view[(t & 3u) * 4u + (t >> 2)];
    }
    __syncthreads();

    // per-row / per-column terms of the view-space position (two texcoord conventions, see lighting_common.hpp): from the table where
    // the host has built one (uniform branch), else computed
    float rowy_glsl = 0.f, rowy_slang = 0.f;
    if (f.colx_tab) {
        if (SUN == SAH_SHADOW_MODE_CSM || GI == SAH_GI_LPV) rowy_glsl = f.colx_tab[2u * f.colx_stride + y];
        if (SUN == SAH_SHADOW_MODE_RT) rowy_slang = f.colx_tab[2u * f.colx_stride + f.rowy_stride + y];
    } else {
        if (SUN == SAH_SHADOW_MODE_CSM || GI == SAH_GI_LPV) rowy_glsl = rowy_glsl_of(a, f, y);
        if (SUN == SAH_SHADOW_MODE_RT) rowy_slang = rowy_slang_of(a, f, y);
    }

    uint32_t out[2 * PPT];
    uint32_t deferred_mask = 0, sky_mask = 0;
    bool emissive_wave;  // (one vote for the thread's PPT pixels)
    {
        uint32_t any_e = 0;
#pragma unroll
        for (int i = 0; i < PPT; i++) any_e |= we[i];
        emissive_wave = wave_any(active && (any_e & 0xffffffu) != 0u);
    }
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        if (!active) break;
        Px p;
        p.color = wc[i];
        p.data = wd[i];
        p.emission = we[i];
        p.n01 = wn[2 * i];
        p.n23 = wn[2 * i + 1];
        p.depth = __uint_as_float(wz[i]);
        p.ao = (GI == SAH_GI_LPV && a.has_ao) ? __uint_as_float(wao[i]) : 1.0f;
        p.mask = (SUN == SAH_SHADOW_MODE_RT && a.has_mask) ? __uint_as_float(wm[i]) : 1.0f;
        const uint32_t x = x0 + i;
        float colx_glsl = 0.f, colx_slang = 0.f;
        if (f.colx_tab) {
            if (SUN == SAH_SHADOW_MODE_CSM || GI == SAH_GI_LPV) colx_glsl = colx_g[i];
            if (SUN == SAH_SHADOW_MODE_RT) colx_slang = colx_s[i];
        } else {
            if (SUN == SAH_SHADOW_MODE_CSM || GI == SAH_GI_LPV) colx_glsl = colx_glsl_of(a, f, x);
            if (SUN == SAH_SHADOW_MODE_RT) colx_slang = colx_slang_of(a, f, x);
        }
        const FastPixelOut r = shade_pixel_fast_sl<SUN, GI>(a, csm, lpv, f, colx_glsl, rowy_glsl, colx_slang, rowy_slang, p, s_lut, lpv_bad, emissive_wave);
        out[2 * i] = r.lit.x;
        out[2 * i + 1] = r.lit.y;
        if (r.deferred) deferred_mask |= 1u << i;
        if (p.depth == 0.f) sky_mask |= 1u << i;
    }
    if (active) {
        uint8_t* dst = const_cast<uint8_t*>(a.lit.ptr) + (size_t)y * a.lit.pitch + (size_t)x0 * 8;
        if (SKY && sky_mask != 0u) {  // the sky workgroup owns these pixels: per-pixel stores around them
#pragma unroll
            for (int i = 0; i < PPT; i++)
                if (!((sky_mask >> i) & 1u)) *reinterpret_cast<uint2*>(dst + 8 * i) = make_uint2(out[2 * i], out[2 * i + 1]);
        } else if constexpr (PPT == 1) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(out[0], out[1]);
        } else {
            store_words<2 * PPT>(dst, out);
        }
    }
    // deferred pixels -> this wave's segment (no atomics: the wave owns it).  Slots by ballot + mbcnt, one bit plane per pixel of
    // the thread; the order inside a segment is irrelevant.  (Sky pixels are not listed: the sky workgroups find them by their depth.)
    const uint32_t seg = gid >> 6, lane = threadIdx.x & 63u;
    uint32_t front = 0;
    const uint32_t listed = SKY ? deferred_mask & ~sky_mask : deferred_mask;
    if (wave_any(listed != 0u)) {  // (a wave of a coherent frame lists nothing: one vote instead of PPT ballots)
        uint8_t* seg_codes = f.seg_list + (size_t)seg * f.seg_stride;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const bool mine = (listed >> i) & 1u;
            const uint64_t m = __ballot(mine);
            if (m) {
                const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (mine) seg_codes[front + before] = (uint8_t)(lane * PPT + (uint32_t)i);
                front += (uint32_t)__builtin_popcountll(m);
            }
        }
    }
    if (lane == 0 && seg < f.num_segments) {
        f.seg_count[seg] = (uint16_t)front;
        if (front != 0u) f.state->deferred_hint[f.hint_slot] = 1u;  // (see FrameState: the fix-up kernel's workgroups leave at once without it)
    }
    if (gid == 0u) f.state->deferred_hint[f.hint_slot ^ 1u] = 0u;  // the next call's word
}

// ---- launchers ----------------------------------------------------------------------------------------------------------
// (re)builds the gather copy of the LPV volumes in front of a Lighting kernel that gathers from it (the fast kernel; the tiled one with f.lpv_fast)
static hipError_t launch_lpv_pack(const LpvArgs& lpv, const FastArgs& f, hipStream_t st) {
    const uint32_t prows = (lpv.red.height + 2 * kLpvPackBorder) * (lpv.red.depth + 2 * kLpvPackBorder);
    const hipError_t me = hipMemsetAsync(&f.state->nonfinite, 0, sizeof(uint32_t), st);  // the new copy's verdict starts at "finite"
    if (me != hipSuccess) return me;
    hipLaunchKernelGGL(k_lpv_pack, dim3(prows), dim3(256), 0, st, lpv.red, lpv.green, lpv.blue, const_cast<uint8_t*>(f.lpv_packed), f.pk_row_pitch,
                       f.pk_slice_pitch, f.state);
    return hipGetLastError();
}
template <int SUN, int GI>
static hipError_t launch_general_ppt(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const SkyArgs& sky, int ppt, hipStream_t st) {
    const uint32_t rows = a.row_end - a.row_begin;
    const uint64_t groups = (uint64_t)(a.width / ppt) * rows;
    if (groups == 0) return hipSuccess;
    const dim3 grid((uint32_t)((groups + 255) / 256)), block(256);
    if (ppt == 4) hipLaunchKernelGGL((k_lighting_general<SUN, GI, 4>), grid, block, 0, st, a, csm, lpv, sky);
    else if (ppt == 2) hipLaunchKernelGGL((k_lighting_general<SUN, GI, 2>), grid, block, 0, st, a, csm, lpv, sky);
    else hipLaunchKernelGGL((k_lighting_general<SUN, GI, 1>), grid, block, 0, st, a, csm, lpv, sky);
    return hipGetLastError();
}

template <int SUN, int GI>
static hipError_t launch_fast_ppt(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const SkyArgs& sky, const FastArgs& f, int ppt,
                                  hipStream_t st) {
    const uint32_t rows = a.row_end - a.row_begin;
    const uint64_t groups = (uint64_t)(a.width / ppt) * rows;
    if (groups == 0) return hipSuccess;
    if (GI == SAH_GI_LPV && f.repack) {
        const hipError_t pe = launch_lpv_pack(lpv, f, st);
        if (pe != hipSuccess) return pe;
    }
    const uint32_t blocks = (uint32_t)((groups + 255) / 256);
    const dim3 block(256);
    auto launch = [&](auto ppt_c) {
        constexpr int P = decltype(ppt_c)::value;
        if (sky.enabled && f.sky_first) {
            FastArgs g = f;
            g.sky_first = (blocks + f.sky_ratio - 1u) / f.sky_ratio;
            hipLaunchKernelGGL((k_lighting_fast<SUN, GI, P, true>), dim3(g.sky_first + blocks), block, 0, st, a, csm, lpv, sky, g);
        } else if (sky.enabled) hipLaunchKernelGGL((k_lighting_fast<SUN, GI, P, true>), dim3((f.sky_ratio + 1u) * ((blocks + f.sky_ratio - 1u) / f.sky_ratio)), block, 0, st, a, csm, lpv, sky, f);
        else hipLaunchKernelGGL((k_lighting_fast<SUN, GI, P, false>), dim3(blocks), block, 0, st, a, csm, lpv, sky, f);
    };
    if (ppt == 4) launch(std::integral_constant<int, 4>{});
    else if (ppt == 2) launch(std::integral_constant<int, 2>{});
    else launch(std::integral_constant<int, 1>{});
    // (a grid-stride loop over the groups of segments)
    const uint32_t fgroups = (f.num_segments + kFixupSegs - 1) / kFixupSegs;
    const dim3 fgrid(fgroups < kFixupMaxWorkgroups ? fgroups : kFixupMaxWorkgroups);
    if (ppt == 4) hipLaunchKernelGGL((k_lighting_fixup<SUN, GI, 4>), fgrid, block, 0, st, a, csm, lpv, sky, f);
    else if (ppt == 2) hipLaunchKernelGGL((k_lighting_fixup<SUN, GI, 2>), fgrid, block, 0, st, a, csm, lpv, sky, f);
    else hipLaunchKernelGGL((k_lighting_fixup<SUN, GI, 1>), fgrid, block, 0, st, a, csm, lpv, sky, f);
    return hipGetLastError();
}

template <int SUN>
static hipError_t launch_gi(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const SkyArgs& sky, const FastArgs* f, int gi, int ppt,
                            hipStream_t st) {
    switch (gi) {
        case SAH_GI_NONE:
            return f ? launch_fast_ppt<SUN, SAH_GI_NONE>(a, csm, lpv, sky, *f, ppt, st) : launch_general_ppt<SUN, SAH_GI_NONE>(a, csm, lpv, sky, ppt, st);
        case SAH_GI_LPV:
            return f ? launch_fast_ppt<SUN, SAH_GI_LPV>(a, csm, lpv, sky, *f, ppt, st) : launch_general_ppt<SUN, SAH_GI_LPV>(a, csm, lpv, sky, ppt, st);
        default: return hipErrorNotSupported;
    }
}

hipError_t launch_colx_table(const LightingArgs& a, const FastArgs& f, float* out, uint32_t stride, uint32_t row_stride, hipStream_t st) {
    hipLaunchKernelGGL(k_colx_table, dim3((max(a.width, a.height) + 255u) / 256u), dim3(256), 0, st, a, f, out, stride, row_stride);
    return hipGetLastError();
}

hipError_t launch_lighting_tiled(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const CacheArgs& cache, const RtgiArgs& rtgi,
                                 const SkyArgs& sky, int sun_mode, int gi, bool brute_force_lights, const FastArgs* fast, hipStream_t st);

hipError_t launch_lighting(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const CacheArgs& cache, const RtgiArgs& rtgi,
                           const SkyArgs& sky, const FastArgs* fast, int sun_mode, int gi, int ppt, bool brute_force_lights, hipStream_t st) {
    // point lights and the GI overlays without a fast path run in the 16x16-tile kernel (lighting_tiled.hip)
    if (a.num_lights || gi == SAH_GI_CACHE || gi == SAH_GI_RTGI) {
        if (gi == SAH_GI_LPV && fast && fast->lpv_fast && fast->repack && a.row_end > a.row_begin) {
            const hipError_t pe = launch_lpv_pack(lpv, *fast, st);
            if (pe != hipSuccess) return pe;
        }
        return launch_lighting_tiled(a, csm, lpv, cache, rtgi, sky, sun_mode, gi, brute_force_lights, fast, st);
    }
    switch (sun_mode) {
        case SAH_SHADOW_MODE_OFF: return launch_gi<SAH_SHADOW_MODE_OFF>(a, csm, lpv, sky, fast, gi, ppt, st);
        case SAH_SHADOW_MODE_CSM: return launch_gi<SAH_SHADOW_MODE_CSM>(a, csm, lpv, sky, fast, gi, ppt, st);
        case SAH_SHADOW_MODE_RT: return launch_gi<SAH_SHADOW_MODE_RT>(a, csm, lpv, sky, fast, gi, ppt, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sah
