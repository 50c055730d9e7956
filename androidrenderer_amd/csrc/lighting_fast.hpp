// Straight-line (branch-free) per-pixel body of the fast Lighting kernel.
//
// Everything is evaluated unconditionally and merged with selects so that the whole PPT-unrolled kernel body is ONE basic
// block: the scheduler can then issue the shadow-map and LPV gathers early and overlap their latency with the BRDF
// arithmetic (with divergent branches every region waited for its own loads: 43 % of wave cycles in s_waitcnt).
// Pixels whose inputs fall outside what the proofs cover are flagged `deferred`; whatever was computed for them is
// discarded and the fix-up kernel shades them with the general restatement.  All gathers use clamped (in-bounds)
// addresses, so evaluating garbage for a deferred pixel is harmless.
#pragma once
#include "lighting_common.hpp"

namespace sah {

SAH_DEV bool finite_f(float x) { return __builtin_fabsf(x) < __builtin_inff(); }

// LDS table of the fast kernel: [0,512) format LUTs, then the per-cascade rows that are indexed per lane
enum : uint32_t { TAB_CSM = 512, TAB_LPV = 512 + 48, TAB_VIEW = 512 + 48 + 32, TAB_SIZE = 512 + 48 + 32 + 12 };

struct FastPixelOut {
    uint2 lit;
    bool deferred;
};

// fp32 ("GLSL") geometry shared by the CSM sun, the LPV overlay and the point lights: normalised normal, world-space position with
// the shader's (x + 1) / W texcoord quirk, view-space depth, and the view vector.  `ok` is cleared when an operand leaves the domain of a
// restricted-range operator (the caller then evaluates the general restatement for that pixel).
struct FastGeom {
    F3 N, ws, V;
    Fn vsz;
};
SAH_DEV FastGeom fast_geometry(const LightingArgs& a, const FastArgs& f, float colx_glsl, float rowy_glsl, float D, const SurfIn& si, float dn,
                               const float* tab, bool& ok) {
    FastGeom g;
    // dn is a sum of squares of fp16 values: 2^-48 <= dn < 2^35 whenever it is positive and finite (checked above), so the
    // restricted-range sqrt / reciprocal apply (numerics.hpp)
    const Fn inv = Fn(rcp_nr(sqrt_nr(dn)));
    g.N = F3{Fn(si.normal[0]) * inv, Fn(si.normal[1]) * inv, Fn(si.normal[2]) * inv};
    // inverse_projection separable: vs = ((p0*X)+p12, (p5*Y)+p13, (p10*D)+p14, (p11*D)+p15)
    const Fn vw = Fn(f.p11) * Fn(D) + Fn(f.p15);
    const Fn vzn = Fn(f.p10) * Fn(D) + Fn(f.p14);
    Fn vx, vy;
    if (f.pos_div_nr) {
        // The three quotients share one refined reciprocal (div_nr with the y1 steps hoisted): 8 + 4 + 3 * 10 cycles instead of
        // 3 * 34.  Domain: |vw| in [2^-40, 2^40] (checked here), numerators +0 or in [2^-40, 2^40] in magnitude (host:
        // detect_fast_path bounds p0, p5, p14 and requires p10 == p12 == p13 == +0, so x / y numerators are products of a
        // bounded coefficient with a multiple of 2^-24 and the z numerator is the constant p14).
        const float aw = __builtin_fabsf(vw.v);
        ok = ok && aw >= kDivLo && aw <= kDivHi;
        const float y0 = __builtin_amdgcn_rcpf(vw.v);
        const float y1 = __builtin_fmaf(__builtin_fmaf(-vw.v, y0, 1.0f), y0, y0);
        auto quot = [&](float num) {
            const float q0 = num * y1;
            const float q1 = __builtin_fmaf(__builtin_fmaf(-vw.v, q0, num), y1, q0);
            return Fn(__builtin_fmaf(__builtin_fmaf(-vw.v, q1, num), y1, q1));
        };
        vx = quot(colx_glsl);
        vy = quot(rowy_glsl);
        g.vsz = quot(vzn.v);
    } else {
        vx = Fn(colx_glsl) / vw;
        vy = Fn(rowy_glsl) / vw;
        g.vsz = vzn / vw;
    }
    // inverse_view affine: ws_i = ((v0i*x + v1i*y) + v2i*z) + v3i
    const float4 mx = *reinterpret_cast<const float4*>(tab + TAB_VIEW), my = *reinterpret_cast<const float4*>(tab + TAB_VIEW + 4u),
                 mz = *reinterpret_cast<const float4*>(tab + TAB_VIEW + 8u);  // rows of inverse_view (LDS broadcast reads)
    g.ws.x = Fn(mx.x) * vx + Fn(mx.y) * vy + Fn(mx.z) * g.vsz + Fn(mx.w);
    g.ws.y = Fn(my.x) * vx + Fn(my.y) * vy + Fn(my.z) * g.vsz + Fn(my.w);
    g.ws.z = Fn(mz.x) * vx + Fn(mz.y) * vy + Fn(mz.z) * g.vsz + Fn(mz.w);
    const F3 d = g.ws - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])};
    const Fn d2 = dot(d, d);
    // 2^-80 <= d2 <= 2^80 implies finite ws (view_pos is finite: host check) and puts sqrt / reciprocal in their domain
    ok = ok && d2.v >= 0x1p-80f && d2.v <= 0x1p+80f;
    g.V = d * Fn(rcp_nr(sqrt_nr(d2.v)));
    return g;
}

// a1: the CSM-mode sun term of one pixel, sc = direct * exposure per channel (0 when unlit or shadowed).  direct = ((ndotl * brdf) *
// colour) * shadow is exactly 0 (or NaN, which the shader's guard turns into 0) whenever ndotl == 0 or shadow == 0, so both the PCF
// lookup and the BRDF are skipped when no lane of the wave needs them (wave-uniform votes: no divergent branches).
SAH_DEV void fast_csm_sun(const LightingArgs& a, const CsmArgs& csm, const float* tab, const F3& N, const F3& ws, const Fn vsz, const F3& V, const F3& L,
                          const Surface<Fn>& s, const SurfIn& si, lanemask act, bool& ok, Fn (&sc)[3]) {
    const Fn ndotl_sun = nclamp(dot(N, L), Fn(0.f), Fn(1.f));
    sc[0] = sc[1] = sc[2] = Fn(0.f);
    const lanemask lit_m = lanes(ndotl_sun.v > 0.f) & act;  // (`act`: the lanes whose sun term is used — surface pixels inside the hot form's domain)
    if (lit_m) {
        uint32_t cascade = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) cascade = (vsz.v < csm.splits[i]) ? i + 1u : cascade;
        const uint32_t cc = cascade > 3u ? 3u : cascade;  // cascade 4 means "no shadow map": keep the address legal
        // 1 - ndotl^2 is 0 or >= 2^-24 (ndotl in [0,1]); 0.0005 * sqrt(.) is +0 or in [2^-24, 2^-10]; ndotl needs a lower bound
        const Fn bias_num = Fn(0.0005f) * Fn(sqrt_nr0((Fn(1.0f) - ndotl_sun * ndotl_sun).v));
        const Fn bias = Fn(div_nr(bias_num.v, ndotl_sun.v));
        ok = ok && !(ndotl_sun.v > 0.f && ndotl_sun.v < kDivLo);
        // affine shadow matrix: sp.w == 1, so the perspective divide is the identity; rows come from the LDS table
        const float4 rx = *reinterpret_cast<const float4*>(tab + TAB_CSM + cc * 12u);
        const float4 ry = *reinterpret_cast<const float4*>(tab + TAB_CSM + cc * 12u + 4u);
        const float4 rz = *reinterpret_cast<const float4*>(tab + TAB_CSM + cc * 12u + 8u);
        const Fn spx = Fn(rx.x) * ws.x + Fn(rx.y) * ws.y + Fn(rx.z) * ws.z + Fn(rx.w);
        const Fn spy = Fn(ry.x) * ws.x + Fn(ry.y) * ws.y + Fn(ry.z) * ws.z + Fn(ry.w);
        const Fn spz = Fn(rz.x) * ws.x + Fn(rz.y) * ws.y + Fn(rz.z) * ws.z + Fn(rz.w);
        // !(any(sp < 0) || any(sp > 1)) via min3 / max3; sp is finite for every pixel that is not deferred (see inside())
        const float sp_mn = __builtin_fminf(__builtin_fminf(spx.v, spy.v), spz.v), sp_mx = __builtin_fmaxf(__builtin_fmaxf(spx.v, spy.v), spz.v);
        const bool sp_inside = !(sp_mn < 0.f || sp_mx > 1.f);
        float pcf_ref = (spz - bias).v;
        pcf_ref = pcf_ref < 0.f ? 0.f : (pcf_ref > 1.f ? 1.f : pcf_ref);  // D16: D_ref clamped to [0,1]
        const VolumeArg& sm = csm.shadowmap;
        const float px = spx.v * (float)sm.width - 0.5f, py = spy.v * (float)sm.height - 0.5f;
        const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
        const float pcf_fx = px - fx0, pcf_fy = py - fy0;
        // texel indices clamped into the map (CLAMP_TO_EDGE; a coordinate far outside — its taps are not used: sp_inside — or not finite
        // still addresses the map): clamp_index(f, n) = min(max(int(f), 0), n), and fx0 + 1 is exact below 2^24
        const int wm1 = (int)sm.width - 1, hm1 = (int)sm.height - 1;
        const uint32_t xa = (uint32_t)clamp_index(fx0, wm1) * 2u, xb = (uint32_t)clamp_index(fx0 + 1.0f, wm1) * 2u;
        const uint32_t ra = (uint32_t)clamp_index(fy0, hm1) * sm.row_pitch, rb = (uint32_t)clamp_index(fy0 + 1.0f, hm1) * sm.row_pitch;
        const uint32_t lo = cc * sm.slice_pitch;  // host guarantees the shadow map is < 4 GiB for the fast path
        const uint32_t pcf_off[4] = {lo + ra + xa, lo + ra + xb, lo + rb + xa, lo + rb + xb};
        // PCF taps (compare LESS, then filter); the fast path is D16_UNORM only (anything else: general kernel)
        float dtap[4];
        uint16_t raw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) raw[k] = *reinterpret_cast<const uint16_t*>(sm.ptr + pcf_off[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float v = (float)raw[k];
            const float q = v * csm.d16_recip;  // host-verified 3-flop v / 65535 (see shadow_pcf)
            dtap[k] = __builtin_fmaf(__builtin_fmaf(-q, 65535.0f, v), csm.d16_recip, q);
        }
        const float wx0 = 1.0f - pcf_fx, wy0 = 1.0f - pcf_fy;
        float pcf = __builtin_fmaf(wx0 * wy0, (pcf_ref < dtap[0]) ? 1.0f : 0.0f, 0.0f);
        pcf = __builtin_fmaf(pcf_fx * wy0, (pcf_ref < dtap[1]) ? 1.0f : 0.0f, pcf);
        pcf = __builtin_fmaf(wx0 * pcf_fy, (pcf_ref < dtap[2]) ? 1.0f : 0.0f, pcf);
        pcf = __builtin_fmaf(pcf_fx * pcf_fy, (pcf_ref < dtap[3]) ? 1.0f : 0.0f, pcf);
        // ndotl > 0 ? (cascade > 3 ? 0 : (outside ? 1 : pcf)) : 1
        float shadow = sp_inside ? pcf : 1.0f;
        shadow = cascade > 3u ? 0.0f : shadow;
        shadow = ndotl_sun.v > 0.f ? shadow : 1.0f;
        if (lit_m & lanes(shadow != 0.0f)) {
            bool brdf_out_of_domain;
            const F3 b = brdf_fast(s, L, V, brdf_out_of_domain);
            const F3 direct = ndotl_sun * b * F3{Fn(a.sun_color[0]), Fn(a.sun_color[1]), Fn(a.sun_color[2])} * Fn(shadow);
            // `if (any(isnan(direct))) direct = 0`: a NaN component (inf * 0 after an overflow; rare) sends the pixel to the
            // fix-up kernel instead of paying three compare + select pairs here.  x + y + z is NaN iff a component is NaN
            // or two are opposite infinities (then the pixel is deferred needlessly, which is harmless).
            const float nan_probe = (direct.x + direct.y + direct.z).v;
            ok = ok && !brdf_out_of_domain && nan_probe == nan_probe;
            const Fn exposure = Fn(0.00031415927f);
            sc[0] = direct.x * exposure;
            sc[1] = direct.y * exposure;
            sc[2] = direct.z * exposure;
        }
    }
}

// The LPV overlay of shade_pixel_fast_sl() below as a function of its own, for the tiled kernel (a light list with an LPV: configs[4]) — the same
// operators in the same order (cascade selection on the scale + translate rows of the LDS table, the gather from the packed copy, Fd == diffuse
// colour / pi, the specular quirk's +-0 dropped: DESIGN.md "Fast path proofs"), kept apart from the headline kernel's body, whose schedule it
// must not disturb.  Preconditions (the caller's `ok`): finite depth, a finite non-zero normal, fast_geometry()'s domain, a non-zero roughness
// byte, finite volumes.  `add`: what is blended into lit.rgb (total * exposure; lit.a gets + 1); `ok` is cleared when a term is NaN (the general
// form then zeroes the pixel's overlay: the caller re-evaluates it).
SAH_DEV void fast_lpv_overlay(const LpvArgs& lpv, const FastArgs& f, const float* tab, F3 N, F3 ws, const SurfIn& si, float ao_px, lanemask act, bool& ok,
                              float (&add)[3]) {
    lanemask out_m = 0;
    auto inside = [&](uint32_t i) {
        const float4 cs_ = *reinterpret_cast<const float4*>(tab + TAB_LPV + i * 8u);
        const float4 ct_ = *reinterpret_cast<const float4*>(tab + TAB_LPV + i * 8u + 4u);
        const Fn cx = Fn(cs_.x) * ws.x + Fn(ct_.x);
        const Fn cy = Fn(cs_.y) * ws.y + Fn(ct_.y);
        const Fn cz = Fn(cs_.z) * ws.z + Fn(ct_.z);
        const float mn = __builtin_fminf(__builtin_fminf(cx.v, cy.v), cz.v), mxv = __builtin_fmaxf(__builtin_fmaxf(cx.v, cy.v), cz.v);
        out_m = lanes(!(mn > 0.f)) | lanes(!(mxv < 1.f));
        return mn > 0.f && mxv < 1.f;
    };
    uint32_t selected = 0;
    const bool in0 = inside(0u);
    if (out_m & act) {
#pragma unroll
        for (int i = 3; i >= 1; i--) {
            if (i < (int)lpv.num_cascades) selected = inside((uint32_t)i) ? (uint32_t)i : selected;
        }
        selected = in0 ? 0u : selected;
    }
    F3 lpv_normal = -N;
    lpv_normal.x = lpv_normal.x * Fn(-1.0f);
    Fn nc[4];
    dir_to_sh(lpv_normal, nc);
    const float4 cs = *reinterpret_cast<const float4*>(tab + TAB_LPV + selected * 8u);
    const float4 ct = *reinterpret_cast<const float4*>(tab + TAB_LPV + selected * 8u + 4u);
    Fn cpx = Fn(cs.x) * (ws.x + N.x) + Fn(ct.x);
    const Fn cpy = Fn(cs.y) * (ws.y + N.y) + Fn(ct.y);
    const Fn cpz = Fn(cs.z) * (ws.z + N.z) + Fn(ct.z);
    cpx = cpx + Fn((float)selected);
    cpx = f.ncasc_pow2 ? cpx * Fn(f.inv_ncasc) : cpx / Fn(lpv.num_cascades_f);
    Fn indirect[3];
    lpv_fetch_packed(lpv, f.lpv_packed, f.pk_row_pitch, f.pk_slice_pitch, cpx.v, cpy.v, cpz.v, nc, indirect);
    const Fn dielectric_f0 = Fn(0.04f);
    const F3 base_color = {Fn(si.color[0]), Fn(si.color[1]), Fn(si.color[2])};
    const F3 diffuse_color = base_color * (Fn(1.0f) - dielectric_f0) * (Fn(1.0f) - Fn(si.metal));
    const Fn inv_pi = (Fn(1.0f) * Fn(1.0f)) * (Fn(1.0f) / Fn(3.1415927f));
    const F3 diffuse_factor = diffuse_color * inv_pi;
    const Fn ao = Fn(ao_px);
    const F3 total = {indirect[0] * diffuse_factor.x * ao, indirect[1] * diffuse_factor.y * ao, indirect[2] * diffuse_factor.z * ao};
    const float nan_probe = (total.x + total.y + total.z).v;
    ok = ok && nan_probe == nan_probe;
    const Fn exposure = Fn(lpv.exposure);
    add[0] = (total.x * exposure).v;
    add[1] = (total.y * exposure).v;
    add[2] = (total.z * exposure).v;
}

template <int SUN, int GI>
SAH_DEV FastPixelOut shade_pixel_fast_sl(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const FastArgs& f, float colx_glsl,
                                         float rowy_glsl, float colx_slang, float rowy_slang, const Px& p, const float* tab,
                                         bool lpv_has_nonfinite, bool emissive_wave) {
    const float* lut = tab;
    const float D = p.depth;
    const bool sky_px = D == 0.f;
    const uint32_t rough_byte = (p.data >> 8) & 0xffu;
    const SurfIn si = unpack_surface(p, lut);
    const float dn = (si.normal[0] * si.normal[0] + si.normal[1] * si.normal[1]) + si.normal[2] * si.normal[2];
    bool ok = finite_f(D) && dn > 0.f && finite_f(dn);
    if (GI == SAH_GI_LPV) ok = ok && rough_byte != 0u && !lpv_has_nonfinite;

    Hn lit[4] = {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};

    // ---------------- fp32 ("GLSL") geometry shared by the CSM sun and the LPV overlay ----------------
    F3 N, ws, V;
    Fn vsz;
    if (SUN == SAH_SHADOW_MODE_CSM || GI == SAH_GI_LPV) {
        const FastGeom g = fast_geometry(a, f, colx_glsl, rowy_glsl, D, si, dn, tab, ok);
        N = g.N;
        ws = g.ws;
        V = g.V;
        vsz = g.vsz;
    }

    Surface<Fn> s;
    s.base_color = {Fn(si.color[0]), Fn(si.color[1]), Fn(si.color[2])};
    s.normal = N;
    s.roughness = Fn(si.rough);
    s.metalness = Fn(si.metal);
    // lanes whose surface terms are used: the votes below are scalar algebra on this mask and the masks of single compares (numerics.hpp,
    // lanes()).  `ok` only shrinks from here on, so a vote on this mask never skips what a pixel needs.
    const lanemask act = lanes(ok && !sky_px);

    const F3 L = {Fn(a.sun_L[0]), Fn(a.sun_L[1]), Fn(a.sun_L[2])};
    // LPV: cascade selection and the gather coordinate
    Fn nc[4];
    float lpv_u = 0.f, lpv_v = 0.f, lpv_w = 0.f;
    if constexpr (GI == SAH_GI_LPV) {
        // selected = smallest i whose unit box contains the point (overlay.frag:97-103 scans down and overwrites; 0 if none).
        // Cascade 0 is tested first: when every lane of the wave is inside it (coherent near-field pixels) the other
        // cascades cannot change the answer and are skipped.  Scale / translate rows come from the LDS table (uniform
        // addresses: broadcast reads) so that they do not occupy SGPRs.
        lanemask out_m = 0;  // lanes outside the box last tested
        auto inside = [&](uint32_t i) {
            const float4 cs_ = *reinterpret_cast<const float4*>(tab + TAB_LPV + i * 8u);
            const float4 ct_ = *reinterpret_cast<const float4*>(tab + TAB_LPV + i * 8u + 4u);
            const Fn cx = Fn(cs_.x) * ws.x + Fn(ct_.x);
            const Fn cy = Fn(cs_.y) * ws.y + Fn(ct_.y);
            const Fn cz = Fn(cs_.z) * ws.z + Fn(ct_.z);
            // all(c > 0) && all(c < 1) as min3 / max3 + two compares (compares cost as much as min3: 4 cycles).  cx, cy, cz are
            // finite for every pixel that is not deferred (finite ws, |scale|, |translate| <= 2^40: host check), so the
            // NaN-dropping behaviour of v_min3 / v_max3 cannot matter
            const float mn = __builtin_fminf(__builtin_fminf(cx.v, cy.v), cz.v), mxv = __builtin_fmaxf(__builtin_fmaxf(cx.v, cy.v), cz.v);
            out_m = lanes(!(mn > 0.f)) | lanes(!(mxv < 1.f));
            return mn > 0.f && mxv < 1.f;
        };
        uint32_t selected = 0;
        const bool in0 = inside(0u);
        if (out_m & act) {
#pragma unroll
            for (int i = 3; i >= 1; i--) {
                if (i < (int)lpv.num_cascades) selected = inside((uint32_t)i) ? (uint32_t)i : selected;
            }
            selected = in0 ? 0u : selected;
        }
        F3 lpv_normal = -N;
        lpv_normal.x = lpv_normal.x * Fn(-1.0f);
        dir_to_sh(lpv_normal, nc);
        // scale+translate cascade: cp_i = (s_i * (ws_i + N_i)) + t_i   (pos.w == 1)
        const float4 cs = *reinterpret_cast<const float4*>(tab + TAB_LPV + selected * 8u);
        const float4 ct = *reinterpret_cast<const float4*>(tab + TAB_LPV + selected * 8u + 4u);
        Fn cpx = Fn(cs.x) * (ws.x + N.x) + Fn(ct.x);
        const Fn cpy = Fn(cs.y) * (ws.y + N.y) + Fn(ct.y);
        const Fn cpz = Fn(cs.z) * (ws.z + N.z) + Fn(ct.z);
        cpx = cpx + Fn((float)selected);
        cpx = f.ncasc_pow2 ? cpx * Fn(f.inv_ncasc) : cpx / Fn(lpv.num_cascades_f);
        lpv_u = cpx.v;
        lpv_v = cpy.v;
        lpv_w = cpz.v;
    }
#ifdef SAH_EXP_LPV_TOUCH
    // experiment (tools/experiments/r6): one dword of each of the four (y, z) rows of the LPV footprint requested BEFORE the sun's arithmetic, consumed
    // behind it: the twelve 16-byte loads of lpv_fetch_packed() then find their lines in the vector cache
    uint32_t lpv_touch[4] = {0u, 0u, 0u, 0u};
    if constexpr (GI == SAH_GI_LPV && SUN == SAH_SHADOW_MODE_CSM) {
        const int W = (int)lpv.red.width, H = (int)lpv.red.height, Dd = (int)lpv.red.depth;
        const float px = lpv_u * (float)W - 0.5f, py = lpv_v * (float)H - 0.5f, pz = lpv_w * (float)Dd - 0.5f;
        const int b = (int)kLpvPackBorder;
        const uint32_t x0 = (uint32_t)(min(max(clamp_to_int(__builtin_floorf(px)), -b), W) + b), y0 = (uint32_t)(min(max(clamp_to_int(__builtin_floorf(py)), -b), H) + b),
                       z0 = (uint32_t)(min(max(clamp_to_int(__builtin_floorf(pz)), -b), Dd) + b);
        const uint32_t base = z0 * f.pk_slice_pitch + y0 * f.pk_row_pitch + x0 * kLpvPackTexel;
        lpv_touch[0] = *reinterpret_cast<const uint32_t*>(f.lpv_packed + base);
        lpv_touch[1] = *reinterpret_cast<const uint32_t*>(f.lpv_packed + base + f.pk_row_pitch);
        lpv_touch[2] = *reinterpret_cast<const uint32_t*>(f.lpv_packed + base + f.pk_slice_pitch);
        lpv_touch[3] = *reinterpret_cast<const uint32_t*>(f.lpv_packed + base + f.pk_slice_pitch + f.pk_row_pitch);
    }
#endif

    // ---------------- a1: sun, CSM mode ----------------
    // direct = ((ndotl * brdf) * colour) * shadow is exactly 0 (or NaN, which the shader's guard turns into 0) whenever
    // ndotl == 0 or shadow == 0, so both the PCF lookup and the BRDF are skipped when no lane of the wave needs them.
    // The votes are wave-uniform: the body stays free of divergent branches.
    if constexpr (SUN == SAH_SHADOW_MODE_CSM) {
        Fn sc[3];
        fast_csm_sun(a, csm, tab, N, ws, vsz, V, L, s, si, act, ok, sc);
        const bool quirk = (a.flags & SAH_LIGHTING_QUIRK_SUN_BLEND) != 0;
        // quirk: dst is the cleared target, s*s + 0*0 == s*s, alpha 1*0 + 0*0 == 0; otherwise plain additive
        float x1[3];
#pragma unroll
        for (int i = 0; i < 3; i++) x1[i] = quirk ? (sc[i] * sc[i]).v : (Fn(0.f) + sc[i]).v;  // 0 + s: a -0 source blends to +0
#pragma unroll
        for (int i = 0; i < 3; i++) lit[i] = Hn(x1[i]);
        lit[3] = quirk ? Hn::lit(0.0f) : Hn::lit(1.0f);
    }

    // ---------------- a3: LPV overlay ----------------
    if constexpr (GI == SAH_GI_LPV) {
#ifdef SAH_EXP_LPV_TOUCH
        asm volatile("" ::"v"(lpv_touch[0]), "v"(lpv_touch[1]), "v"(lpv_touch[2]), "v"(lpv_touch[3]));
#endif
        Fn indirect[3];
        lpv_fetch_packed(lpv, f.lpv_packed, f.pk_row_pitch, f.pk_slice_pitch, lpv_u, lpv_v, lpv_w, nc, indirect);
        // Fd(surface, N, N) == diffuse_color * (1/pi) exactly when N is a finite normalised vector, and the specular term
        // is (finite) * (finite * 0) == +-0 when roughness > 0 and the volumes are finite (DESIGN.md "Fast path proofs").
        const Fn dielectric_f0 = Fn(0.04f);
        const F3 diffuse_color = s.base_color * (Fn(1.0f) - dielectric_f0) * (Fn(1.0f) - s.metalness);
        const Fn inv_pi = (Fn(1.0f) * Fn(1.0f)) * (Fn(1.0f) / Fn(3.1415927f));
        const F3 diffuse_factor = diffuse_color * inv_pi;
        const Fn ao = Fn(p.ao);
        const F3 total = {indirect[0] * diffuse_factor.x * ao, indirect[1] * diffuse_factor.y * ao, indirect[2] * diffuse_factor.z * ao};
        // `if (any(isnan(total))) total = 0` (a NaN AO texel, inf * 0): deferred rather than selected, as for the sun term
        const float nan_probe = (total.x + total.y + total.z).v;
        ok = ok && nan_probe == nan_probe;
        const Fn exposure = Fn(lpv.exposure);
        lit[0] = Hn(tof(lit[0]) + (total.x * exposure).v);
        lit[1] = Hn(tof(lit[1]) + (total.y * exposure).v);
        lit[2] = Hn(tof(lit[2]) + (total.z * exposure).v);
        lit[3] = lit[3] + Hn::lit(1.0f);  // (1 is an fp16 value: RN16(RN32(a + 1)) of an fp16 a is the fp16 sum, 24 >= 2 * 11 + 2)
    }

    // ---------------- a2: emissive (also the only pass that touches depth == 0 pixels when there is no sky) --------------
    // `emissive_wave` (wave-uniform): some pixel of the wave has a non-zero emission texel.  Without one every colour gets +0 added (table
    // entry 0 is +0).  RN16(RN32(lit + 0)) is lit for every fp16 lit but -0, which becomes +0 (a tiny negative LPV term rounds to -0 in the
    // blend before) — exactly what one fp16 add of +0 does (fp16 -> fp32 is exact, a NaN keeps its bits): the look-ups, products and fp32
    // blends become three v_add_f16
    float er = 0.f, eg = 0.f, eb = 0.f;
    if (emissive_wave) {
        const Fn e = Fn(3.1415927f);
        er = (Fn(lut[p.emission & 0xffu]) * e).v;
        eg = (Fn(lut[(p.emission >> 8) & 0xffu]) * e).v;
        eb = (Fn(lut[(p.emission >> 16) & 0xffu]) * e).v;
        lit[0] = Hn(tof(lit[0]) + er);
        lit[1] = Hn(tof(lit[1]) + eg);
        lit[2] = Hn(tof(lit[2]) + eb);
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) lit[i] = lit[i] + Hn::lit(0.f);
    }
    lit[3] = lit[3] + Hn::lit(1.0f);

    // ---------------- a1b: sun, RT mode (Slang half flavour) ----------------
    if constexpr (SUN == SAH_SHADOW_MODE_RT) {
        Surface<Hn> sh;
        sh.base_color = {Hn(si.color[0]), Hn(si.color[1]), Hn(si.color[2])};
        sh.normal = normalize(H3{Hn(si.normal[0]), Hn(si.normal[1]), Hn(si.normal[2])});
        sh.roughness = Hn(si.rough);
        sh.metalness = Hn(si.metal);
        const Fn vw = Fn(f.p11) * Fn(D) + Fn(f.p15);
        const Fn vx = Fn(colx_slang) / vw, vy = Fn(rowy_slang) / vw, vz = (Fn(f.p10) * Fn(D) + Fn(f.p14)) / vw;
        const Fn vww = vw / vw;  // == 1 for finite non-zero vw; NaN otherwise, which then poisons the location as in the shader
        const float* m = a.inv_view;
        F3 loc;
        loc.x = Fn(m[0]) * vx + Fn(m[4]) * vy + Fn(m[8]) * vz + Fn(m[12]) * vww;
        loc.y = Fn(m[1]) * vx + Fn(m[5]) * vy + Fn(m[9]) * vz + Fn(m[13]) * vww;
        loc.z = Fn(m[2]) * vx + Fn(m[6]) * vy + Fn(m[10]) * vz + Fn(m[14]) * vww;
        const Hn ndotl = Hn(nclamp(dot(L, to_f(sh.normal)), Fn(0.f), Fn(1.f)).v);
        const H3 Vh = to_h(normalize(loc - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])}));
        const H3 Lh = to_h(L);
        const H3 b = brdf_sl(sh, Lh, Vh);
        const H3 nb = ndotl * b;
        F3 radiance = to_f(nb) * F3{Fn(a.sun_color[0]), Fn(a.sun_color[1]), Fn(a.sun_color[2])};
        const F3 masked = radiance * Fn(p.mask);
        const bool lit_side = tof(ndotl) > 0.f;
        radiance = {lit_side ? masked.x : radiance.x, lit_side ? masked.y : radiance.y, lit_side ? masked.z : radiance.z};
        const Fn exposure = Fn(0.00031415927f);
        lit[0] = Hn(tof(lit[0]) + (radiance.x * exposure).v);
        lit[1] = Hn(tof(lit[1]) + (radiance.y * exposure).v);
        lit[2] = Hn(tof(lit[2]) + (radiance.z * exposure).v);
    }

    FastPixelOut o;
    // depth == 0: only the emissive term exists (sky pixels are shaded by the fix-up kernel when a sky is bound)
    Hn sky_lit[4] = {Hn(er), Hn(eg), Hn(eb), Hn::lit(1.0f)};
    const uint2 surf = pack_lit(lit), bare = pack_lit(sky_lit);
    o.lit.x = sky_px ? bare.x : surf.x;
    o.lit.y = sky_px ? bare.y : surf.y;
    o.deferred = sky_px ? (f.sky_enabled != 0u) : !ok;
    return o;
}

}  // namespace sah
