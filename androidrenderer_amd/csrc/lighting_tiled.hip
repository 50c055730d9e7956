// Tiled Lighting kernel for gfx950: one 256-thread workgroup shades one 16x16 screen tile.
//
// Used for (a) the point-light extension (BASELINE configs 2/3/5): per-tile light culling with the light
// list in LDS, built with wavefront ballot + prefix so that the list keeps light-index order (the sum over the culled
// list is then bit-identical to brute force), and (b) the GI overlays that have no fast path yet (irradiance cache a4,
// RTGI reconstruction a5).  The per-pixel arithmetic is the general restatement of lighting_common.hpp /
// lighting_gi_ext.hpp.
//
// Culling: the tile's bound is the axis-aligned box of the WORLD-SPACE positions the shading itself reconstructs for the
// tile's surface pixels (min/max by wave shuffles, then across the four waves through LDS); a light survives if its
// sphere, inflated by 1e-4 relative, touches the box.  A culled light has d >= r for every pixel of the tile, hence
// attenuation exactly 0 and contribution exactly +-0 (or NaN -> 0 by the guard): skipping it cannot change the sum.
#include <hip/hip_runtime.h>

#include "../../include/sah_hip.h"
#include "lighting_common.hpp"
#include "lighting_fast.hpp"
#include "lighting_gi_ext.hpp"
#include "numerics.hpp"
#include "params.hpp"

namespace sah {

constexpr uint32_t kMaxTileLights = 1024;

SAH_DEV float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = __builtin_fminf(v, __shfl_xor(v, d, 64));
    return v;
}
SAH_DEV float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

#ifndef SAH_EXP_CACHE_WAVES
#define SAH_EXP_CACHE_WAVES 1  // experiments (tools/experiments/r6): waves per SIMD the cache-GI body without a light list is held to (1: the allocator's own 4)
#endif
template <int SUN, int GI, bool LIGHTS>
__global__ void __launch_bounds__(256, ((GI == SAH_GI_CACHE && !LIGHTS) ? SAH_EXP_CACHE_WAVES : 1)) k_lighting_tiled(const LightingArgs a, const CsmArgs csm, const LpvArgs lpv, const CacheArgs cache,
                                                        const RtgiArgs rtgi, const SkyArgs sky, const uint32_t brute_force, const FastArgs f,
                                                        const uint32_t fast_geom) {
    // `fast_geom`: the uniform blocks have the structure the fast kernel's geometry and CSM sun rely on (api.cpp: detect_fast_path);
    // then the fp32 geometry (normal, position, view vector) and the CSM sun come from lighting_fast.hpp — the same bits at a third of
    // the instructions — and only pixels outside their domains take the general restatement.
    __shared__ __attribute__((aligned(16))) float s_lut[TAB_SIZE];
    if (SUN == SAH_SHADOW_MODE_CSM && threadIdx.x < 48) {  // [cascade][row x,y,z][col 0..3] of biasMat * cascade_matrices
        const uint32_t c = threadIdx.x / 12u, j = threadIdx.x % 12u;
        s_lut[TAB_CSM + threadIdx.x] = csm.biased[c][(j & 3u) * 4u + (j >> 2)];
    }
    if (threadIdx.x >= 128 && threadIdx.x < 140) {  // rows x,y,z of the (affine) inverse view matrix
        const uint32_t t = threadIdx.x - 128u;
        s_lut[TAB_VIEW + t] = a.inv_view[(t & 3u) * 4u + (t >> 2)];
    }
    if (GI == SAH_GI_LPV && threadIdx.x >= 64 && threadIdx.x < 96) {  // [cascade][sx sy sz - tx ty tz -] (meaningful when f.lpv_fast)
        const uint32_t t = threadIdx.x - 64u, c = t >> 3, j = t & 7u;
        s_lut[TAB_LPV + t] = (j & 3u) == 3u ? 0.f : (j < 4u ? f.lpv_s[c][j] : f.lpv_t[c][j - 4u]);
    }
    __shared__ float s_box[4][6];
    __shared__ uint32_t s_wave_count[4];
    __shared__ uint16_t s_list[kMaxTileLights];
    __shared__ float2 s_lconst[LIGHTS ? kMaxTileLights : 1u];  // per kept light: far2, refined 1 / radius (see the shading loop)
    s_lut[threadIdx.x] = a.luts[threadIdx.x];
    s_lut[threadIdx.x + 256] = a.luts[threadIdx.x + 256];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // a wave is an 8 x 8 pixel square of the 16 x 16 tile (not four rows of it): the light loop skips a light for the whole wave when no
    // lane is in range, and a compact footprint is in range of fewer lights
    // Without a light list nothing ties the workgroup to a tile: it is then 32 x 8 pixels, a wave two rows of 32 — plane loads of 128 / 256
    // contiguous bytes per row instead of eight 32-byte pieces (cache GI in the frame 0.3557 -> 0.3538 ms, the RTGI overlay 0.1290 -> 0.1232;
    // 64 x 1 and 16 x 4 waves measured within 0.5 % of it: tools/experiments/r4/README.md)
    const uint32_t x = LIGHTS ? blockIdx.x * 16u + (threadIdx.x & 7u) + ((threadIdx.x >> 3) & 8u) : blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t y = a.row_begin + (LIGHTS ? blockIdx.y * 16u + ((threadIdx.x >> 3) & 7u) + ((threadIdx.x >> 4) & 8u) : blockIdx.y * 8u + (threadIdx.x >> 5));
    const bool inside = x < a.width && y < a.row_end;

    Px p;
    p.color = p.data = p.emission = p.n01 = p.n23 = 0u;
    p.depth = 0.f;
    p.ao = 1.f;
    p.mask = 1.f;
    if (inside) {
        p.color = *reinterpret_cast<const uint32_t*>(a.color.ptr + (size_t)y * a.color.pitch + (size_t)x * 4);
        p.data = *reinterpret_cast<const uint32_t*>(a.data.ptr + (size_t)y * a.data.pitch + (size_t)x * 4);
        p.emission = *reinterpret_cast<const uint32_t*>(a.emission.ptr + (size_t)y * a.emission.pitch + (size_t)x * 4);
        p.depth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x * 4);
        const uint2 n = *reinterpret_cast<const uint2*>(a.normals.ptr + (size_t)y * a.normals.pitch + (size_t)x * 8);
        p.n01 = n.x;
        p.n23 = n.y;
        if (GI == SAH_GI_LPV && a.has_ao) p.ao = *reinterpret_cast<const float*>(a.ao.ptr + (size_t)y * a.ao.pitch + (size_t)x * 4);
        if (SUN == SAH_SHADOW_MODE_RT && a.has_mask)
            p.mask = *reinterpret_cast<const float*>(a.shadow_mask.ptr + (size_t)y * a.shadow_mask.pitch + (size_t)x * 4);
    }
    const bool surface = inside && p.depth != 0.f;
    SurfIn si = unpack_surface(p, s_lut);

    Hn lit[4] = {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};

    // fp32 geometry of the GLSL passes (CSM sun, point lights): fast form where its preconditions hold
    constexpr bool kNeedsGeom = SUN == SAH_SHADOW_MODE_CSM || LIGHTS;
    bool geom_ok = false;
    FastGeom g;
    Surface<Fn> s;
    if constexpr (kNeedsGeom) {
        if (fast_geom) {
            const float dn = (si.normal[0] * si.normal[0] + si.normal[1] * si.normal[1]) + si.normal[2] * si.normal[2];
            geom_ok = surface && finite_f(p.depth) && dn > 0.f && finite_f(dn);
            // (the per-column / per-row numerators come from the context's tables — api.cpp builds them for this kernel too; clamped
            //  indices: the threads of a partial tile read a valid entry)
            const uint32_t xt = min(x, a.width - 1u), yt = min(y, a.height - 1u);
            float colx_glsl, rowy_glsl;
            if (f.colx_tab) {
                colx_glsl = f.colx_tab[xt];
                rowy_glsl = f.colx_tab[2u * f.colx_stride + yt];
            } else {
                const Fn tx = (Fn((float)x + 0.5f) + Fn(0.5f)) / Fn(a.res[0]), ty = (Fn((float)y + 0.5f) + Fn(0.5f)) / Fn(a.res[1]);
                colx_glsl = (Fn(f.p0) * (tx * Fn(2.0f) - Fn(1.0f)) + Fn(f.p12)).v;
                rowy_glsl = (Fn(f.p5) * (ty * Fn(2.0f) - Fn(1.0f)) + Fn(f.p13)).v;
            }
            g = fast_geometry(a, f, colx_glsl, rowy_glsl, p.depth, si, dn, s_lut, geom_ok);
        }
        s.base_color = {Fn(si.color[0]), Fn(si.color[1]), Fn(si.color[2])};
        s.normal = g.N;
        s.roughness = Fn(si.rough);
        s.metalness = Fn(si.metal);
        if (wave_any(surface && !geom_ok)) {  // general restatement of the same quantities (IEEE operators) for the pixels that need it
            if (surface && !geom_ok) {
                s.normal = normalize(F3{Fn(si.normal[0]), Fn(si.normal[1]), Fn(si.normal[2])});
                const F3 vs = viewspace_position_glsl(a, x, y, p.depth);
                const F4 ws4 = mul44(a.inv_view, F4{vs.x, vs.y, vs.z, Fn(1.0f)});
                g.N = s.normal;
                g.ws = {ws4.x, ws4.y, ws4.z};
                g.vsz = vs.z;
                g.V = normalize(g.ws - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])});
            }
        }
    }

    // (2) sun, CSM mode
    if constexpr (SUN == SAH_SHADOW_MODE_CSM) {
        Fn sc[4] = {Fn(0.f), Fn(0.f), Fn(0.f), Fn(1.0f)};
        bool sun_ok = geom_ok;
        if (fast_geom) {
            const F3 L = {Fn(a.sun_L[0]), Fn(a.sun_L[1]), Fn(a.sun_L[2])};
            Fn sc3[3];
            fast_csm_sun(a, csm, s_lut, g.N, g.ws, g.vsz, g.V, L, s, si, lanes(sun_ok && surface), sun_ok, sc3);
            sc[0] = sc3[0]; sc[1] = sc3[1]; sc[2] = sc3[2];
        }
        if (wave_any(surface && !sun_ok)) {
            if (surface && !sun_ok) sun_frag(a, csm, x, y, p, si, sc);
        }
        if (surface) {
            if (a.flags & SAH_LIGHTING_QUIRK_SUN_BLEND) {
#pragma unroll
                for (int i = 0; i < 3; i++) lit[i] = Hn((sc[i] * sc[i] + Fn(tof(lit[i])) * Fn(tof(lit[i]))).v);
                lit[3] = Hn((sc[3] * Fn(0.f) + Fn(tof(lit[3])) * Fn(0.f)).v);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) lit[i] = Hn(tof(lit[i]) + sc[i].v);
            }
        }
    }

    // (2b) point lights
    if constexpr (LIGHTS) {
        // shading inputs as the a9 spec builds them: the sun fragment's fp32 surface, N, V and position (above)
        const F3 ws = g.ws, V = g.V;
        const PointLightWords* light_words = reinterpret_cast<const PointLightWords*>(a.lights);

        // tile bound: box of the positions of the surface pixels (non-finite positions get 0 from every light anyway)
        const float inf = __builtin_inff();
        const bool bounded = surface && __builtin_fabsf(ws.x.v) < inf && __builtin_fabsf(ws.y.v) < inf && __builtin_fabsf(ws.z.v) < inf;
        float lo[3] = {bounded ? ws.x.v : inf, bounded ? ws.y.v : inf, bounded ? ws.z.v : inf};
        float hi[3] = {bounded ? ws.x.v : -inf, bounded ? ws.y.v : -inf, bounded ? ws.z.v : -inf};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            lo[i] = wave_min(lo[i]);
            hi[i] = wave_max(hi[i]);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                s_box[wave][i] = lo[i];
                s_box[wave][3 + i] = hi[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; i++) {
            lo[i] = __builtin_fminf(__builtin_fminf(s_box[0][i], s_box[1][i]), __builtin_fminf(s_box[2][i], s_box[3][i]));
            hi[i] = __builtin_fmaxf(__builtin_fmaxf(s_box[0][3 + i], s_box[1][3 + i]), __builtin_fmaxf(s_box[2][3 + i], s_box[3][3 + i]));
        }

        F3 sum = F3(Fn(0.f));
        const lanemask surface_m = lanes(surface);
        const BrdfPixel bp = brdf_fast_pixel(s, V);  // the light-independent half of the BRDF, once per pixel
        for (uint32_t batch = 0; batch < a.num_lights; batch += kMaxTileLights) {
            const uint32_t batch_n = min(kMaxTileLights, a.num_lights - batch);
            uint32_t count = 0;  // lights kept so far in this batch (uniform)
            if (brute_force) {
                count = batch_n;
            } else {
                for (uint32_t base = 0; base < batch_n; base += 256u) {
                    const uint32_t i = base + threadIdx.x;
                    bool keep = false;
                    float radius = 0.f;
                    if (i < batch_n) {
                        const float4 pr = *reinterpret_cast<const float4*>(&light_words[batch + i]);  // position, radius
                        radius = pr.w;
                        const float c[3] = {pr.x, pr.y, pr.z};
                        float d2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const float d = __builtin_fmaxf(__builtin_fmaxf(lo[k] - c[k], c[k] - hi[k]), 0.f);
                            d2 += d * d;
                        }
                        const float rr = radius * 1.0001f + 1e-6f;
                        keep = d2 <= rr * rr;  // empty box (lo = +inf) gives d2 = inf: nothing kept; NaN light data: kept
                        keep = keep || !(d2 == d2) || !(rr == rr);
                    }
                    // order-preserving compaction: ballot + prefix inside the wave, wave totals through LDS
                    const unsigned long long ballot = __ballot(keep);
                    const uint32_t before = __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
                    if (lane == 0) s_wave_count[wave] = (uint32_t)__builtin_popcountll(ballot);
                    __syncthreads();
                    uint32_t offset = count, total = 0;
#pragma unroll
                    for (uint32_t w = 0; w < 4; w++) {
                        const uint32_t c = s_wave_count[w];
                        offset += w < wave ? c : 0u;
                        total += c;
                    }
                    if (keep) {  // (few lights of a batch: what the shading loop needs of the light alone is worked out for those only)
                        const PointLightWords plw = light_words[batch + i];
                        const float far_r = radius * 1.001f;
                        s_list[offset + before] = (uint16_t)(i | (point_light_hot_ok(plw) ? 0x8000u : 0u));  // index in the batch | hot form allowed << 15
                        s_lconst[offset + before] = {far_r * far_r, div_nr_refine(radius)};
                    }
                    count += total;
                    __syncthreads();
                }
            }
            // shade the (ordered) list
            for (uint32_t j = 0; j < count; j++) {
                // the light and what depends on it alone (uniform): from the culling pass where there was one — computed here they are VALU
                // instructions of every wave and light.  A pixel farther than 1.001 r from the light has xr >= 1.0009, so
                // w = clamp(1 - xr^4, 0, 1) = 0 and its term is +-0 (or NaN -> 0): adding it cannot change `sum`; waves in which no pixel is
                // nearer skip the light (uniform branch).  `light_ok`: the preconditions of the hot form; anything else takes the general one.
                const uint32_t entry = brute_force ? j : (uint32_t)__builtin_amdgcn_readfirstlane((int)s_list[j]);  // (scalar: its flag is tested on the scalar unit)
                const PointLightWords plw = light_words[batch + (entry & 0x7fffu)];
                const PointLightDev pl = point_light_values(plw);
                float far2, inv_radius;
                bool light_ok;
                if (brute_force) {
                    const float far_r = pl.radius * 1.001f;
                    far2 = far_r * far_r;
                    inv_radius = div_nr_refine(pl.radius);
                    light_ok = point_light_hot_ok(plw);
                } else {
                    const float2 lc = s_lconst[j];
                    far2 = lc.x;
                    inv_radius = lc.y;
                    light_ok = (entry & 0x8000u) != 0u;
                }
                const F3 lv = F3{Fn(pl.px), Fn(pl.py), Fn(pl.pz)} - ws;
                const Fn d2 = dot(lv, lv);
                const bool near = surface && d2.v <= far2;
                const lanemask near_m = lanes(d2.v <= far2) & surface_m;  // (votes as scalar mask algebra: numerics.hpp, lanes())
                if (!near_m) continue;
                // A light behind the surface (N.L <= 0) contributes ndotl * (...) with ndotl = +0: +-0, or NaN -> 0 — nothing, in the hot
                // form and in the general one alike (both evaluate L = lv / |lv| with the same bits while d2 is inside the domain of
                // the restricted-range root and reciprocal).  Waves whose reachable pixels all face away skip the light.
                const lanemask d2_out = lanes(!(d2.v >= 0x1p-80f)) | lanes(!(d2.v <= 0x1p+40f));
                const F3 Le = lv * Fn(rcp_nr(sqrt_nr(d2.v)));
                const lanemask away = lanes(dot(s.normal, Le).v <= 0.f);
                if (!(near_m & (d2_out | ~away))) continue;
                F3 c = F3(Fn(0.f));
                lanemask redo_m = ~0ull;
                if (light_ok) c = point_light_contribution_fast(s, bp, lv, d2, V, pl, inv_radius, d2_out, away, redo_m);
                if (near_m & redo_m) {
                    const bool redo = (redo_m >> lane) & 1ull;
                    const F3 cg = point_light_contribution(s, ws, V, pl);
                    // (component by component: `c = redo ? cg : c` on the struct goes through scratch memory, 28 bytes a lane, in every iteration)
                    c.x = redo ? cg.x : c.x;
                    c.y = redo ? cg.y : c.y;
                    c.z = redo ? cg.z : c.z;
                }
                if (near) sum = sum + c;
            }
            __syncthreads();  // s_list is rewritten by the next batch
        }
        if (surface) {
            const Fn exposure = Fn(0.00031415927f);
            lit[0] = Hn(tof(lit[0]) + (sum.x * exposure).v);
            lit[1] = Hn(tof(lit[1]) + (sum.y * exposure).v);
            lit[2] = Hn(tof(lit[2]) + (sum.z * exposure).v);
            lit[3] = lit[3] + Hn::lit(1.0f);  // (1 is an fp16 value: the blend is an fp16 add, as for the Slang overlays below)
        }
    }

    // the half-precision surface, location and view vector that the Slang passes of this pixel share (lighting_gi_ext.hpp: SlangGeom)
    constexpr bool kNeedsSlang = SUN == SAH_SHADOW_MODE_RT || GI == SAH_GI_CACHE || GI == SAH_GI_RTGI;
    SlangGeom sg;
    if constexpr (kNeedsSlang) {
        // hot form: fast_geometry() (lighting_fast.hpp) with the Slang texcoords — the three view-space quotients through one refined
        // reciprocal, the rows of the affine inverse view from the LDS table, restricted-range root and reciprocal for the view vector:
        // the bits of worldspace_location_slang() / normalize() inside the domains its `ok` reports (depth and distance from the camera
        // within 2^+-40), the general form for the pixels outside them
        bool hot = false;
        if (fast_geom && f.pos_div_nr) {
            const uint32_t xt = min(x, a.width - 1u), yt = min(y, a.height - 1u);
            float colx, rowy;
            if (f.colx_tab) {
                colx = f.colx_tab[f.colx_stride + xt];
                rowy = f.colx_tab[2u * f.colx_stride + f.rowy_stride + yt];
            } else {
                const Fn tx = (Fn((float)x) + Fn(0.5f)) / Fn(a.res[0]), ty = (Fn((float)y) + Fn(0.5f)) / Fn(a.res[1]);
                colx = (Fn(f.p0) * (tx * Fn(2.0f) - Fn(1.0f)) + Fn(f.p12)).v;
                rowy = (Fn(f.p5) * (ty * Fn(2.0f) - Fn(1.0f)) + Fn(f.p13)).v;
            }
            hot = surface && finite_f(p.depth);
            const FastGeom fg = fast_geometry(a, f, colx, rowy, p.depth, si, 1.0f, s_lut, hot);  // (its fp32 normal is not used: dn = 1)
            sg.s.base_color = {Hn(si.color[0]), Hn(si.color[1]), Hn(si.color[2])};
            sg.s.normal = normalize(H3{Hn(si.normal[0]), Hn(si.normal[1]), Hn(si.normal[2])});
            sg.s.roughness = Hn(si.rough);
            sg.s.metalness = Hn(si.metal);
            sg.location = fg.ws;
            sg.V = to_h(fg.V);
            sg.bv = brdf_sl_view(sg.s, sg.V);
        }
        if (wave_any(surface && !hot)) {
            if (surface && !hot) sg = slang_geometry(a, x, y, p, si);
        }
    }
    // (3) GI overlay
    if (surface) {
        if constexpr (GI == SAH_GI_LPV) {
            // round 6: the fast kernel's overlay (gather from the packed copy, three of the nine trilinear fetches) where its proofs hold — the
            // uniform blocks (f.lpv_fast: api.cpp), fast_geometry()'s domain, a non-zero roughness byte, finite volumes —, the general
            // restatement for every other pixel.  (Inside `if (surface)`: the votes below are on masks that include `surface`.)
            bool lpv_ok = false;
            float add[3] = {0.f, 0.f, 0.f};
            if constexpr (kNeedsGeom) {
                if (f.lpv_fast) {
                    lpv_ok = geom_ok && ((p.data >> 8) & 0xffu) != 0u && f.state->nonfinite == 0u;
                    fast_lpv_overlay(lpv, f, s_lut, g.N, g.ws, si, p.ao, lanes(lpv_ok), lpv_ok, add);
                }
            }
            if (lpv_ok) {
#pragma unroll
                for (int i = 0; i < 3; i++) lit[i] = Hn(tof(lit[i]) + add[i]);
                lit[3] = lit[3] + Hn::lit(1.0f);  // (the fp16 sum, as in the fast kernel)
            } else {
                Fn s[4];
                gi_lpv_frag(a, lpv, x, y, p, si, s);
#pragma unroll
                for (int i = 0; i < 4; i++) lit[i] = Hn(tof(lit[i]) + s[i].v);
            }
        } else if constexpr (GI == SAH_GI_CACHE || GI == SAH_GI_RTGI) {
            Hn s[4];
            if constexpr (GI == SAH_GI_CACHE) {
                bool redo = false;  // hot form first; a pixel outside its preconditions (rare) is re-evaluated with the general form
                if (cache.hot_ok) gi_cache_frag(a, cache, sg, s, s_lut, &redo);
                if (redo || !cache.hot_ok) gi_cache_frag(a, cache, sg, s);
            } else {
                gi_rtgi_frag(a, rtgi, x, y, sg, s_lut, s);
            }
            // the blend RN16(RN32(lit + s)) of two fp16 values is their fp16 sum: rounding a sum of 11-bit operands to 24 bits first is
            // innocuous (24 >= 2 * 11 + 2), so one v_add_f16 per channel stands for two conversions, the fp32 add and the conversion back
#pragma unroll
            for (int i = 0; i < 4; i++) lit[i] = lit[i] + s[i];
        }
    }
    // (4) emissive.  A wave without an emissive texel adds +0 to every colour (table entry 0 is +0: api.cpp's format tables decode byte 0 to
    // 0).  RN16(RN32(lit + 0)) is lit for every fp16 lit but -0, which becomes +0 (a tiny negative GI term rounds to -0 in the blend before) —
    // exactly what one fp16 add of +0 does (fp16 -> fp32 is exact, a NaN keeps its bits), so the look-ups, products and fp32 blends become
    // three v_add_f16
    {
        if (wave_any((p.emission & 0xffffffu) != 0u)) {
            const Fn e = Fn(3.1415927f);
            lit[0] = Hn(tof(lit[0]) + (Fn(s_lut[p.emission & 0xffu]) * e).v);
            lit[1] = Hn(tof(lit[1]) + (Fn(s_lut[(p.emission >> 8) & 0xffu]) * e).v);
            lit[2] = Hn(tof(lit[2]) + (Fn(s_lut[(p.emission >> 16) & 0xffu]) * e).v);
        } else {
#pragma unroll
            for (int i = 0; i < 3; i++) lit[i] = lit[i] + Hn::lit(0.f);
        }
        lit[3] = lit[3] + Hn::lit(1.0f);
    }
    // (5) sky
    if (sky.enabled && inside && !surface) sky_frag(a, sky, x, y, lit);
    // (6) sun, RT mode
    if constexpr (SUN == SAH_SHADOW_MODE_RT) {
        if (surface) {
            float add[3];
            sun_rt_shared(a, p, sg, add);
#pragma unroll
            for (int i = 0; i < 3; i++) lit[i] = Hn(tof(lit[i]) + add[i]);
        }
    }
    if (inside) *reinterpret_cast<uint2*>(const_cast<uint8_t*>(a.lit.ptr) + (size_t)y * a.lit.pitch + (size_t)x * 8) = pack_lit(lit);
}

template <int SUN, int GI>
static hipError_t launch_tiled_lights(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const CacheArgs& cache, const RtgiArgs& rtgi,
                                      const SkyArgs& sky, bool brute, const FastArgs* fast, hipStream_t st) {
    const FastArgs f = fast ? *fast : FastArgs{};
    const uint32_t fast_geom = fast ? 1u : 0u;
    const uint32_t rows = a.row_end - a.row_begin;
    if (rows == 0 || a.width == 0) return hipSuccess;
    const dim3 grid((a.width + 15) / 16, (rows + 15) / 16), block(256);
    if (a.num_lights) hipLaunchKernelGGL((k_lighting_tiled<SUN, GI, true>), grid, block, 0, st, a, csm, lpv, cache, rtgi, sky, brute ? 1u : 0u, f, fast_geom);
    else hipLaunchKernelGGL((k_lighting_tiled<SUN, GI, false>), dim3((a.width + 31) / 32, (rows + 7) / 8), block, 0, st, a, csm, lpv, cache, rtgi, sky, 0u, f, fast_geom);
    return hipGetLastError();
}

template <int SUN>
static hipError_t launch_tiled_gi(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const CacheArgs& cache, const RtgiArgs& rtgi,
                                  const SkyArgs& sky, int gi, bool brute, const FastArgs* fast, hipStream_t st) {
    switch (gi) {
        case SAH_GI_NONE: return launch_tiled_lights<SUN, SAH_GI_NONE>(a, csm, lpv, cache, rtgi, sky, brute, fast, st);
        case SAH_GI_LPV: return launch_tiled_lights<SUN, SAH_GI_LPV>(a, csm, lpv, cache, rtgi, sky, brute, fast, st);
        case SAH_GI_CACHE: return launch_tiled_lights<SUN, SAH_GI_CACHE>(a, csm, lpv, cache, rtgi, sky, brute, fast, st);
        case SAH_GI_RTGI: return launch_tiled_lights<SUN, SAH_GI_RTGI>(a, csm, lpv, cache, rtgi, sky, brute, fast, st);
        default: return hipErrorInvalidValue;
    }
}

// R11G11B10 atlas -> float4 per texel.  uf11 / uf10 share fp16's exponent width and bias, so the fp16 value of a channel is its bits
// shifted into place (sample_cascade_fast), and fp16 -> fp32 is exact: inf and NaN stay what they are.
__global__ void __launch_bounds__(256) k_probe_irr_unpack(const VolumeArg src, uint8_t* dst) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= src.width) return;
    const uint32_t off = z * src.slice_pitch + y * src.row_pitch + x * 4u;
    const uint32_t w = *reinterpret_cast<const uint32_t*>(src.ptr + off);
    float4 t;
    t.x = (float)__builtin_bit_cast(_Float16, (uint16_t)((w << 4) & 0x7ff0u));
    t.y = (float)__builtin_bit_cast(_Float16, (uint16_t)((w >> 7) & 0x7ff0u));
    t.z = (float)__builtin_bit_cast(_Float16, (uint16_t)((w >> 17) & 0x7fe0u));
    t.w = 0.f;
    *reinterpret_cast<float4*>(dst + 4u * (size_t)off) = t;
}
// The same for the blocks of a list of probes: what sah_probe_update runs behind its dispatches when the context keeps the copy current
// (SAH_GENERATION_TRACKED).  An update stores to the cells -2 .. R of a probe's (R + 2)-wide block (probes.hip: ordered_stores — the odd
// block sizes reach into the neighbours'), so cells [-2, block) per axis are widened again: 90 texels per probe of the 7 x 8 atlas.
__global__ void __launch_bounds__(128) k_probe_irr_unpack_probes(const VolumeArg src, uint8_t* dst, const uint32_t* probes, uint32_t bw, uint32_t bh) {
    const uint32_t px = probes[3u * blockIdx.x], py = probes[3u * blockIdx.x + 1u], pz = probes[3u * blockIdx.x + 2u];
    if (px >= 32768u || py >= 32768u || pz >= src.depth) return;  // (not a probe of this atlas: sah_probe_update stored nothing for it either)
    const uint32_t w = bw + 2u, h = bh + 2u;
    for (uint32_t t = threadIdx.x; t < w * h; t += 128u) {
        const int x = (int)(px * bw) - 2 + (int)(t % w), y = (int)(py * bh) - 2 + (int)(t / w);
        if (x < 0 || y < 0 || x >= (int)src.width || y >= (int)src.height) continue;
        const uint32_t off = pz * src.slice_pitch + (uint32_t)y * src.row_pitch + (uint32_t)x * 4u;
        const uint32_t wd = *reinterpret_cast<const uint32_t*>(src.ptr + off);
        float4 v;
        v.x = (float)__builtin_bit_cast(_Float16, (uint16_t)((wd << 4) & 0x7ff0u));
        v.y = (float)__builtin_bit_cast(_Float16, (uint16_t)((wd >> 7) & 0x7ff0u));
        v.z = (float)__builtin_bit_cast(_Float16, (uint16_t)((wd >> 17) & 0x7fe0u));
        v.w = 0.f;
        *reinterpret_cast<float4*>(dst + 4u * (size_t)off) = v;
    }
}
hipError_t launch_probe_irr_unpack_probes(const VolumeArg& src, uint8_t* dst, const uint32_t* probes, uint32_t num_probes, hipStream_t st) {
    if (num_probes == 0 || src.width < 32 || src.height < 32) return hipSuccess;
    hipLaunchKernelGGL(k_probe_irr_unpack_probes, dim3(num_probes), dim3(128), 0, st, src, dst, probes, src.width / 32u, src.height / 32u);
    return hipGetLastError();
}
hipError_t launch_probe_irr_unpack(const VolumeArg& src, uint8_t* dst, hipStream_t st) {
    if (src.width == 0 || src.height == 0 || src.depth == 0) return hipSuccess;
    hipLaunchKernelGGL(k_probe_irr_unpack, dim3((src.width + 255u) / 256u, src.height, src.depth), dim3(256), 0, st, src, dst);
    return hipGetLastError();
}

hipError_t launch_lighting_tiled(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const CacheArgs& cache, const RtgiArgs& rtgi,
                                 const SkyArgs& sky, int sun_mode, int gi, bool brute_force_lights, const FastArgs* fast, hipStream_t st) {
    switch (sun_mode) {
        case SAH_SHADOW_MODE_OFF: return launch_tiled_gi<SAH_SHADOW_MODE_OFF>(a, csm, lpv, cache, rtgi, sky, gi, brute_force_lights, fast, st);
        case SAH_SHADOW_MODE_CSM: return launch_tiled_gi<SAH_SHADOW_MODE_CSM>(a, csm, lpv, cache, rtgi, sky, gi, brute_force_lights, fast, st);
        case SAH_SHADOW_MODE_RT: return launch_tiled_gi<SAH_SHADOW_MODE_RT>(a, csm, lpv, cache, rtgi, sky, gi, brute_force_lights, fast, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sah
