// Device code shared by the Lighting kernels: packed-texel helpers, the emulated samplers and the *general*
// per-pixel restatement of every sub-pass (valid for arbitrary matrices and arbitrary, even non-finite, inputs).
// The fast kernel (lighting.hip) only takes pixels for which it can prove it computes the same bits; everything else
// is shaded by shade_pixel_general() in the fix-up kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sah_hip.h"
#include "numerics.hpp"
#include "params.hpp"

namespace sah {

// (non-temporal forms of these loads / stores were measured and lose: off/none 0.0614 -> 0.0686 ms, headline 0.1728 -> 0.1795 ms —
//  profiles/r3_lpv_pack32_experiment.txt, second table)
template <int NW> SAH_DEV void load_words(const uint8_t* p, uint32_t (&w)[NW]) {
    if constexpr (NW == 1) {
        w[0] = *reinterpret_cast<const uint32_t*>(p);
    } else if constexpr (NW == 2) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        w[0] = v.x; w[1] = v.y;
    } else if constexpr (NW == 4) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        static_assert(NW == 8, "unsupported width");
        uint4 v0 = *reinterpret_cast<const uint4*>(p);
        uint4 v1 = *reinterpret_cast<const uint4*>(p + 16);
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
        w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
    }
}
template <int NW> SAH_DEV void store_words(uint8_t* p, const uint32_t (&w)[NW]) {
    if constexpr (NW == 2) {
        *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
    } else if constexpr (NW == 4) {
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        static_assert(NW == 8, "unsupported width");
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(p + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

struct Px {  // one pixel's G-buffer texels, still packed
    uint32_t color, data, emission, n01, n23;
    float depth, ao, mask;
};

SAH_DEV int clamp_to_int(float f) { return (int)__builtin_fminf(__builtin_fmaxf(f, -1.0e9f), 1.0e9f); }
// min(max(clamp_to_int(f), 0), hi) for a wave-uniform hi in [0, 10^9] (an extent from the kernel arguments: it is bound to a scalar
// register) as two instructions for four: v_cvt_i32_f32 saturates by itself (NaN -> 0, which the clamp above also ends at), and the
// two-sided clamp is one v_med3_i32.
SAH_DEV int clamp_index(float f, int hi) {
    int i, r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(f));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "s"(hi));
    return r;
}
SAH_DEV _Float16 hbits(uint32_t b) { return __builtin_bit_cast(_Float16, (uint16_t)b); }

// ---- samplers: Vulkan weighted-sum formula, fma chain in tap order (DESIGN.md "Sampling") -----------------
// `fmaf((float)half, w, acc)` selects v_fma_mix_f32: the fp16 -> fp32 conversion is free.

// 3D, linear, CLAMP_TO_BORDER (transparent black): accumulates the 4 channels of one RGBA16F volume.
struct TriSetup {
    float w[8];
    int x0, y0, z0;
    bool nan;
};
SAH_DEV TriSetup tri_setup(uint32_t W, uint32_t H, uint32_t D, float u, float v, float ww) {
    TriSetup s;
    const float px = u * (float)W - 0.5f, py = v * (float)H - 0.5f, pz = ww * (float)D - 0.5f;
    s.nan = isnan_f(px) || isnan_f(py) || isnan_f(pz);
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py), fz0 = __builtin_floorf(pz);
    const float fx = px - fx0, fy = py - fy0, fz = pz - fz0;
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy, wz0 = 1.0f - fz;
    s.x0 = clamp_to_int(fx0);
    s.y0 = clamp_to_int(fy0);
    s.z0 = clamp_to_int(fz0);
    const float wxy[4] = {wx0 * wy0, fx * wy0, wx0 * fy, fx * fy};
#pragma unroll
    for (int k = 0; k < 8; k++) s.w[k] = wxy[k & 3] * ((k >> 2) ? fz : wz0);
    return s;
}
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

SAH_DEV half4_t load_texel8_border(const VolumeArg& v, int x, int y, int z) {
    if ((unsigned)x < v.width && (unsigned)y < v.height && (unsigned)z < v.depth) {
        const uint32_t off = (uint32_t)z * v.slice_pitch + (uint32_t)y * v.row_pitch + (uint32_t)x * 8u;  // volumes are < 4 GiB
        return *reinterpret_cast<const half4_t*>(v.ptr + off);
    }
    return half4_t{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
}
// acc_c = fma(w_k, t_k.c, acc_c) from +0, taps in order; (float)half inside an fma selects v_fma_mix_f32 (free conversion)
SAH_DEV void tri_accumulate(const VolumeArg& v, const TriSetup& s, float (&out)[4]) {
    if (s.nan) {
        out[0] = out[1] = out[2] = out[3] = __builtin_nanf("");
        return;
    }
    half4_t t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = load_texel8_border(v, s.x0 + (k & 1), s.y0 + ((k >> 1) & 1), s.z0 + (k >> 2));
    out[0] = out[1] = out[2] = out[3] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int c = 0; c < 4; c++) out[c] = __builtin_fmaf(s.w[k], (float)t[k][c], out[c]);
    }
}

// 2D RGBA16F, linear, REPEAT (sky LUTs: RenderCore/render/procedural_sky.cpp:62-68)
SAH_DEV void sample_bilinear_repeat_rgba16f(const PlaneArg& p, uint32_t W, uint32_t H, float u, float v, float (&out)[4]) {
    const float px = u * (float)W - 0.5f, py = v * (float)H - 0.5f;
    if (isnan_f(px) || isnan_f(py)) {
        out[0] = out[1] = out[2] = out[3] = __builtin_nanf("");
        return;
    }
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    const float fx = px - fx0, fy = py - fy0;
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    int x0 = clamp_to_int(fx0) % (int)W, y0 = clamp_to_int(fy0) % (int)H;
    if (x0 < 0) x0 += (int)W;
    if (y0 < 0) y0 += (int)H;
    const int x1 = x0 + 1 == (int)W ? 0 : x0 + 1, y1 = y0 + 1 == (int)H ? 0 : y0 + 1;
    const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    const float w[4] = {wx0 * wy0, fx * wy0, wx0 * fy, fx * fy};
    half4_t q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = *reinterpret_cast<const half4_t*>(p.ptr + (size_t)ys[k] * p.pitch + (size_t)xs[k] * 8);
    out[0] = out[1] = out[2] = out[3] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int c = 0; c < 4; c++) out[c] = __builtin_fmaf(w[k], (float)q[k][c], out[c]);
    }
}

// directional_light.frag:45-53 / gi/lpv/overlay.frag:43-51: texcoord = (gl_FragCoord + 0.5) / res = (x + 1) / W (quirk)
SAH_DEV F3 viewspace_position_glsl(const LightingArgs& a, uint32_t x, uint32_t y, float depth) {
    const Fn fx = Fn((float)x + 0.5f) + Fn(0.5f), fy = Fn((float)y + 0.5f) + Fn(0.5f);
    const Fn tx = fx / Fn(a.res[0]), ty = fy / Fn(a.res[1]);
    const F4 ndc = {tx * Fn(2.0f) - Fn(1.0f), ty * Fn(2.0f) - Fn(1.0f), Fn(depth), Fn(1.0f)};
    const F4 vs = mul44(a.inv_proj, ndc);
    return {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w};
}

// directional_light.rt.slang:39-48 / gi/cache/overlay.frag.slang:35-44 (correct pixel centre)
SAH_DEV F3 worldspace_location_slang(const LightingArgs& a, float px, float py, float depth) {
    const Fn tx = (Fn(px) + Fn(0.5f)) / Fn(a.res[0]);
    const Fn ty = (Fn(py) + Fn(0.5f)) / Fn(a.res[1]);
    const F4 ndc = {tx * Fn(2.0f) - Fn(1.0f), ty * Fn(2.0f) - Fn(1.0f), Fn(depth), Fn(1.0f)};
    F4 vs = mul44(a.inv_proj, ndc);
    vs = {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w, vs.w / vs.w};
    const F4 ws = mul44(a.inv_view, vs);
    return {ws.x, ws.y, ws.z};
}

struct SurfIn {  // unpacked texels
    float color[3];
    float normal[3];
    float rough, metal;
};

SAH_DEV SurfIn unpack_surface(const Px& p, const float* lut) {
    SurfIn s;
    s.color[0] = lut[p.color & 0xffu];
    s.color[1] = lut[(p.color >> 8) & 0xffu];
    s.color[2] = lut[(p.color >> 16) & 0xffu];
    s.normal[0] = (float)hbits(p.n01 & 0xffffu);
    s.normal[1] = (float)hbits(p.n01 >> 16);
    s.normal[2] = (float)hbits(p.n23 & 0xffffu);
    s.rough = lut[256 + ((p.data >> 8) & 0xffu)];
    s.metal = lut[256 + ((p.data >> 16) & 0xffu)];
    return s;
}

// ---- a1: directional_light.frag:96-149 (CSM-mode sun) -------------------------------------------------
SAH_DEV float shadow_pcf(const CsmArgs& c, float u, float v, uint32_t layer, float ref) {
    const VolumeArg& sm = c.shadowmap;
    const float px = u * (float)sm.width - 0.5f, py = v * (float)sm.height - 0.5f;
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    const float fx = px - fx0, fy = py - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    float cmp[4];
    const uint8_t* layer_base = sm.ptr + (size_t)layer * sm.slice_pitch;
    const int xa = x0 < 0 ? 0 : (x0 > (int)sm.width - 1 ? (int)sm.width - 1 : x0);  // CLAMP_TO_EDGE per tap
    const int xb = x0 + 1 < 0 ? 0 : (x0 + 1 > (int)sm.width - 1 ? (int)sm.width - 1 : x0 + 1);
    const int ya = y0 < 0 ? 0 : (y0 > (int)sm.height - 1 ? (int)sm.height - 1 : y0);
    const int yb = y0 + 1 < 0 ? 0 : (y0 + 1 > (int)sm.height - 1 ? (int)sm.height - 1 : y0 + 1);
    const uint32_t ra = (uint32_t)ya * sm.row_pitch, rb = (uint32_t)yb * sm.row_pitch;
    const int xs[4] = {xa, xb, xa, xb};
    const uint32_t rs[4] = {ra, ra, rb, rb};
    // (uniform format branches stay OUTSIDE the tap loops so that the four loads issue back to back)
    float d[4];
    if (c.is_d16) {
        uint16_t raw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) raw[k] = *reinterpret_cast<const uint16_t*>(layer_base + rs[k] + (uint32_t)xs[k] * 2u);
        if (c.d16_recip_ok) {
            // v / 65535 correctly rounded with three flops: q = v*y; q' = fma(fma(-q, 65535, v), y, q), y = RN(1/65535).
            // The host checks all 65536 inputs against the true quotient before setting d16_recip_ok (api.cpp).
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float v = (float)raw[k];
                const float q = v * c.d16_recip;
                d[k] = __builtin_fmaf(__builtin_fmaf(-q, 65535.0f, v), c.d16_recip, q);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = (float)raw[k] / 65535.0f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = *reinterpret_cast<const float*>(layer_base + rs[k] + (uint32_t)xs[k] * 4u);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cmp[k] = (ref < d[k]) ? 1.0f : 0.0f;  // compare op LESS, then filter
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    float r = __builtin_fmaf(wx0 * wy0, cmp[0], 0.0f);
    r = __builtin_fmaf(fx * wy0, cmp[1], r);
    r = __builtin_fmaf(wx0 * fy, cmp[2], r);
    r = __builtin_fmaf(fx * fy, cmp[3], r);
    return r;
}

SAH_DEV Fn sample_csm(const CsmArgs& c, F3 ws, Fn viewspace_depth, Fn ndotl) {
    uint32_t cascade = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        if (viewspace_depth.v < c.splits[i]) cascade = i + 1;
    }
    if (cascade > 3) return Fn(0.0f);
    const Fn bias = Fn(0.0005f) * nsqrt(Fn(1.0f) - ndotl * ndotl) / ndotl;
    F4 sp = mul44(c.biased[cascade], F4{ws.x, ws.y, ws.z, Fn(1.0f)});
    sp = {sp.x / sp.w, sp.y / sp.w, sp.z / sp.w, sp.w / sp.w};
    if (sp.x.v < 0.f || sp.y.v < 0.f || sp.z.v < 0.f || sp.x.v > 1.f || sp.y.v > 1.f || sp.z.v > 1.f) return Fn(1.0f);
    if (c.shadowmap.ptr == nullptr) return Fn(1.0f);
    if (isnan_f(sp.x.v) || isnan_f(sp.y.v)) return Fn(__builtin_nanf(""));
    float ref = (sp.z - bias).v;
    if (c.is_d16) ref = ref < 0.f ? 0.f : (ref > 1.f ? 1.f : ref);
    return Fn(shadow_pcf(c, sp.x.v, sp.y.v, cascade, ref));
}

SAH_DEV void sun_frag(const LightingArgs& a, const CsmArgs& csm, uint32_t x, uint32_t y, const Px& p, const SurfIn& si, Fn (&out)[4]) {
    Surface<Fn> s;
    s.base_color = {Fn(si.color[0]), Fn(si.color[1]), Fn(si.color[2])};
    s.normal = normalize(F3{Fn(si.normal[0]), Fn(si.normal[1]), Fn(si.normal[2])});
    s.roughness = Fn(si.rough);
    s.metalness = Fn(si.metal);
    const F3 vs = viewspace_position_glsl(a, x, y, p.depth);
    const F4 ws4 = mul44(a.inv_view, F4{vs.x, vs.y, vs.z, Fn(1.0f)});
    const F3 ws = {ws4.x, ws4.y, ws4.z};
    const F3 V = normalize(ws - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])});
    const F3 L = {Fn(a.sun_L[0]), Fn(a.sun_L[1]), Fn(a.sun_L[2])};
    const Fn ndotl = nclamp(dot(s.normal, L), Fn(0.f), Fn(1.f));
    Fn shadow = Fn(1.0f);
    if (ndotl.v > 0.f) shadow = sample_csm(csm, ws, vs.z, ndotl);
    const F3 b = brdf_sl(s, L, V);  // == Fd(s, L, V) + Fr(s, L, V)
    F3 direct = ndotl * b * F3{Fn(a.sun_color[0]), Fn(a.sun_color[1]), Fn(a.sun_color[2])} * shadow;
    if (any_nan(direct)) direct = F3(Fn(0.f));
    const Fn exposure = Fn(0.00031415927f);
    out[0] = direct.x * exposure;
    out[1] = direct.y * exposure;
    out[2] = direct.z * exposure;
    out[3] = Fn(1.0f);
}

// ---- a1b: directional_light.rt.slang:58-139 with the ray query replaced by the shadow-mask plane --------
SAH_DEV void sun_rt(const LightingArgs& a, uint32_t x, uint32_t y, const Px& p, const SurfIn& si, float (&add)[3]) {
    Surface<Hn> s;
    s.base_color = {Hn(si.color[0]), Hn(si.color[1]), Hn(si.color[2])};
    s.normal = normalize(H3{Hn(si.normal[0]), Hn(si.normal[1]), Hn(si.normal[2])});
    s.roughness = Hn(si.rough);
    s.metalness = Hn(si.metal);
    const F3 location = worldspace_location_slang(a, (float)x, (float)y, p.depth);
    const F3 L = {Fn(a.sun_L[0]), Fn(a.sun_L[1]), Fn(a.sun_L[2])};
    const Hn ndotl = Hn(nclamp(dot(L, to_f(s.normal)), Fn(0.f), Fn(1.f)).v);
    const H3 V = to_h(normalize(location - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])}));
    const H3 Lh = to_h(L);
    const H3 b = brdf_sl(s, Lh, V);  // == Fd(s, Lh, V) + Fr(s, Lh, V)
    const H3 nb = ndotl * b;
    F3 radiance = to_f(nb) * F3{Fn(a.sun_color[0]), Fn(a.sun_color[1]), Fn(a.sun_color[2])};
    if (tof(ndotl) > 0.f) radiance = radiance * Fn(p.mask);
    const Fn exposure = Fn(0.00031415927f);
    add[0] = (radiance.x * exposure).v;
    add[1] = (radiance.y * exposure).v;
    add[2] = (radiance.z * exposure).v;
}

// ---- a3: gi/lpv/overlay.frag:70-164 ---------------------------------------------------------------------
SAH_DEV void dir_to_sh(F3 d, Fn (&o)[4]) {
    const Fn c0 = Fn(0.282094792f), c1 = Fn(0.488602512f);
    o[0] = c0;
    o[1] = -c1 * d.y;
    o[2] = c1 * d.z;
    o[3] = -c1 * d.x;
}
SAH_DEV Fn dot4(const float (&t)[4], const Fn (&n)[4]) { return Fn(t[0]) * n[0] + Fn(t[1]) * n[1] + Fn(t[2]) * n[2] + Fn(t[3]) * n[3]; }

SAH_DEV void lpv_fetch(const LpvArgs& L, F4 p, const Fn (&n)[4], Fn (&out)[3]) {
    const TriSetup s = tri_setup(L.red.width, L.red.height, L.red.depth, p.x.v, p.y.v, p.z.v);
    float r[4], g[4], b[4];
    tri_accumulate(L.red, s, r);
    tri_accumulate(L.green, s, g);
    tri_accumulate(L.blue, s, b);
    out[0] = dot4(r, n);
    out[1] = dot4(g, n);
    out[2] = dot4(b, n);
}

// Fast-path LPV gather from the per-frame packed copy (lighting.hip: k_lpv_pack; params.hpp: FastArgs::lpv_packed) for FINITE
// coordinates and FINITE volume contents.  The copy holds the three volumes interleaved (24-byte texels) inside a two-texel
// border of zeros, so a trilinear footprint is 4 (y,z) rows of 48 contiguous bytes — 12 dwordx4 loads from 4 addresses — and
// CLAMP_TO_BORDER is literal: an outside tap reads a zero texel with its true weight, exactly the sampler formula.  The base
// index is clamped to [-2, size] per axis: both taps of an axis are then border texels whenever both true taps are outside.
SAH_DEV void lpv_fetch_packed(const LpvArgs& L, const uint8_t* packed, uint32_t row_pitch, uint32_t slice_pitch, float u, float v, float w,
                              const Fn (&n)[4], Fn (&out)[3]) {
    const int W = (int)L.red.width, H = (int)L.red.height, D = (int)L.red.depth;
    const float px = u * (float)W - 0.5f, py = v * (float)H - 0.5f, pz = w * (float)D - 0.5f;
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py), fz0 = __builtin_floorf(pz);
    const float fx = px - fx0, fy = py - fy0, fz = pz - fz0;
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    const int b = (int)kLpvPackBorder;
    const uint32_t x0 = (uint32_t)(min(max(clamp_to_int(fx0), -b), W) + b), y0 = (uint32_t)(min(max(clamp_to_int(fy0), -b), H) + b),
                   z0 = (uint32_t)(min(max(clamp_to_int(fz0), -b), D) + b);
    const uint32_t base = z0 * slice_pitch + y0 * row_pitch + x0 * kLpvPackTexel;
    const uint32_t ro[4] = {base, base + row_pitch, base + slice_pitch, base + slice_pitch + row_pitch};  // (y0,z0) (y1,z0) (y0,z1) (y1,z1)
    uint32_t d[4][12];  // per row: R(x0) G(x0) B(x0) R(x1) G(x1) B(x1), two dwords (four halves) each
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint4 q = *reinterpret_cast<const uint4*>(packed + ro[r] + 16u * (uint32_t)j);
            d[r][4 * j] = q.x;
            d[r][4 * j + 1] = q.y;
            d[r][4 * j + 2] = q.z;
            d[r][4 * j + 3] = q.w;
        }
    }
    const float wxy[4] = {gx * gy, fx * gy, gx * fy, fx * fy};
    float wt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) wt[k] = wxy[k & 3] * ((k >> 2) ? fz : gz);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; k++) {  // tap order: x fastest, then y, then z
            const int r = k >> 1, xs = k & 1;
            const uint32_t lo = d[r][xs * 6 + c * 2], hi = d[r][xs * 6 + c * 2 + 1];
            a[0] = fma_mix_lo(wt[k], lo, a[0]);
            a[1] = fma_mix_hi(wt[k], lo, a[1]);
            a[2] = fma_mix_lo(wt[k], hi, a[2]);
            a[3] = fma_mix_hi(wt[k], hi, a[3]);
        }
        out[c] = dot4(a, n);
    }
}

SAH_DEV void gi_lpv_frag(const LightingArgs& a, const LpvArgs& L, uint32_t x, uint32_t y, const Px& p, const SurfIn& si, Fn (&out)[4]) {
    Surface<Fn> s;
    s.base_color = {Fn(si.color[0]), Fn(si.color[1]), Fn(si.color[2])};
    s.normal = normalize(F3{Fn(si.normal[0]), Fn(si.normal[1]), Fn(si.normal[2])});
    s.roughness = Fn(si.rough);
    s.metalness = Fn(si.metal);
    const F3 vs = viewspace_position_glsl(a, x, y, p.depth);
    const F4 ws = mul44(a.inv_view, F4{vs.x, vs.y, vs.z, Fn(1.0f)});
    const F3 V = normalize(F3{ws.x, ws.y, ws.z} - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])});

    uint32_t selected = 0;
    for (int i = (int)L.num_cascades - 1; i >= 0; i--) {
        const F4 cp = mul44(L.world_to_cascade[i], ws);
        if (cp.x.v > 0.f && cp.y.v > 0.f && cp.z.v > 0.f && cp.x.v < 1.f && cp.y.v < 1.f && cp.z.v < 1.f) selected = (uint32_t)i;
    }
    F3 lpv_normal = -s.normal;
    lpv_normal.x = lpv_normal.x * Fn(-1.0f);
    Fn nc[4];
    dir_to_sh(lpv_normal, nc);

    Fn indirect[3];
    {
        const F4 pos = {ws.x + s.normal.x, ws.y + s.normal.y, ws.z + s.normal.z, ws.w + Fn(0.f)};
        F4 cp = mul44(L.world_to_cascade[selected], pos);
        cp.x = cp.x + Fn((float)selected);
        cp.x = cp.x / Fn(L.num_cascades_f);
        lpv_fetch(L, cp, nc, indirect);
    }
    const F3 I = -V;
    const F3 refl = I - s.normal * (Fn(2.0f) * dot(s.normal, I));
    Fn spec[3] = {Fn(0.f), Fn(0.f), Fn(0.f)};
    if (selected == 0) {
        const F4 cp = mul44(L.world_to_cascade[0], ws);
        Fn rc[4];
        dir_to_sh(refl, rc);
        lpv_fetch(L, cp, rc, spec);
        const F3 loc = F3{ws.x, ws.y, ws.z} + refl * Fn(1.0f);
        const F4 cp1 = mul44(L.world_to_cascade[0], F4{loc.x, loc.y, loc.z, Fn(1.f)});
        Fn more[3];
        lpv_fetch(L, cp1, rc, more);
#pragma unroll
        for (int i = 0; i < 3; i++) spec[i] = (spec[i] + more[i]) / Fn(2.0f);
    }
    const F3 diffuse_factor = Fd(s, s.normal, s.normal);
    const F3 fr = Fr(s, s.normal, refl);
    const F3 specular_factor = {fr.x * Fn(0.f), fr.y * Fn(0.f), fr.z * Fn(0.f)};
    const Fn ao = Fn(p.ao);
    F3 total = {indirect[0] * diffuse_factor.x * ao + spec[0] * specular_factor.x,
                indirect[1] * diffuse_factor.y * ao + spec[1] * specular_factor.y,
                indirect[2] * diffuse_factor.z * ao + spec[2] * specular_factor.z};
    if (any_nan(total)) total = F3(Fn(0.f));
    const Fn exposure = Fn(L.exposure);
    out[0] = total.x * exposure;
    out[1] = total.y * exposure;
    out[2] = total.z * exposure;
    out[3] = Fn(1.0f);
}

// ---- a6: sky/sky_unified.slang:54-206 ---------------------------------------------------------------------
// acos / atan / exp are evaluated in fp64 and rounded to fp32 (the oracle defines them as correctly rounded).
SAH_DEV Fn cr_acos(Fn x) { return Fn((float)acos((double)x.v)); }
SAH_DEV Fn cr_atan(Fn x) { return Fn((float)atan((double)x.v)); }
SAH_DEV Fn cr_exp(Fn x) { return Fn((float)exp((double)x.v)); }
SAH_DEV Fn fsign(Fn x) { return Fn(x.v > 0.f ? 1.0f : (x.v < 0.f ? -1.0f : 0.0f)); }

SAH_DEV Fn ray_intersect_sphere(F3 ro, F3 rd, Fn rad) {
    const Fn b = dot(ro, rd);
    const Fn c = dot(ro, ro) - rad * rad;
    if (c.v > 0.0f && b.v > 0.0f) return Fn(-1.0f);
    const Fn discr = b * b - c;
    if (discr.v < 0.0f) return Fn(-1.0f);
    if (discr.v > (b * b).v) return (-b + nsqrt(discr));
    return -b - nsqrt(discr);
}

// get_sky_color(view_vector_worldspace, sunDir, ...) of sky_unified.slang:137-166, sunDir and its uniform sub-expressions in `k`
SAH_DEV F3 sky_color(const SkyArgs& k, F3 rayDir) {
    const Fn sky_pi = Fn(3.14159265358f);
    const F3 sunDir = {Fn(k.sun_dir[0]), Fn(k.sun_dir[1]), Fn(k.sun_dir[2])};
    const F3 up = {Fn(0.0f) / Fn(k.height), Fn(k.up_y), Fn(0.0f) / Fn(k.height)};
    const F3 view_pos = {Fn(0.f), Fn(k.view_pos_y), Fn(0.f)};

    // getValFromSkyLUT :80-109
    const Fn altitudeAngle = Fn(k.horizon_angle) - cr_acos(dot(rayDir, up));
    Fn azimuthAngle;
    if (__builtin_fabsf(altitudeAngle.v) > k.azimuth_limit) {
        azimuthAngle = Fn(0.0f);
    } else {
        const F3 right = {Fn(k.right[0]), Fn(k.right[1]), Fn(k.right[2])};
        const F3 forward = {Fn(k.forward[0]), Fn(k.forward[1]), Fn(k.forward[2])};
        const F3 projectedDir = normalize(rayDir - up * (dot(rayDir, up)));
        const Fn sinTheta = dot(projectedDir, right);
        const Fn cosTheta = dot(projectedDir, forward);
        azimuthAngle = cr_atan(cosTheta / sinTheta) + sky_pi;
    }
    const Fn v = Fn(0.5f) + Fn(0.5f) * fsign(altitudeAngle) * nsqrt(nabs(altitudeAngle) * Fn(2.0f) / sky_pi);
    const Fn u = azimuthAngle / (Fn(2.0f) * sky_pi);
    float lut[4];
    sample_bilinear_repeat_rgba16f(k.sky_view, k.s_w, k.s_h, u.v, v.v, lut);
    F3 lum = {Fn(lut[0]), Fn(lut[1]), Fn(lut[2])};

    // sunWithBloom :120-135
    Fn sun;
    {
        const Fn cosTheta = dot(rayDir, sunDir);
        if (cosTheta.v >= k.min_sun_cos) {
            sun = Fn(1.f);
        } else {
            const Fn offset = Fn(k.min_sun_cos) - cosTheta;
            // exp(x) rounds to +0 in fp32 for x <= -104 (below half the smallest denormal), which is every pixel more than a few degrees
            // from the sun: skip the fp64 exp there (the product with 0.5 is then +0 as well)
            const Fn ex = -offset * Fn(50000.0f);
            const Fn gaussianBloom = ex.v <= -110.0f ? Fn(0.0f) : cr_exp(ex) * Fn(0.5f);
            const Fn invBloom = Fn(1.0f) / (Fn(0.02f) + offset * Fn(300.0f)) * Fn(0.01f);
            sun = gaussianBloom + invBloom;
        }
    }
    // smoothstep(0.002h, 1.0h, sunLum)
    const Fn t = nclamp((sun - Fn(k.smooth_e0)) / (Fn(1.0f) - Fn(k.smooth_e0)), Fn(0.0f), Fn(1.0f));
    const Fn s = t * t * (Fn(3.0f) - Fn(2.0f) * t);
    F3 sunLum = F3(s);
    if (length(sunLum).v > 0.0f) {
        if (ray_intersect_sphere(view_pos, rayDir, Fn(6.360f)).v >= 0.0f) {
            sunLum = F3(Fn(0.f));
        } else {
            // getValFromTLUT(transmittance_lut, viewPos, sunDir) :111-118
            const Fn sunCosZenithAngle = dot(sunDir, up);
            const Fn tu = nclamp(Fn(0.5f) + Fn(0.5f) * sunCosZenithAngle, Fn(0.0f), Fn(1.0f));
            const Fn tv = nmax(Fn(0.0f), nmin(Fn(1.0f), (Fn(k.height) - Fn(6.360f)) / (Fn(6.460f) - Fn(6.360f))));
            float tl[4];
            sample_bilinear_repeat_rgba16f(k.transmittance, k.t_w, k.t_h, tu.v, tv.v, tl);
            sunLum = sunLum * F3{Fn(tl[0]), Fn(tl[1]), Fn(tl[2])};
        }
    }
    lum = lum + sunLum;
    lum = lum * Fn(20.0f);
    lum = lum * Fn(1.0f);
    return lum;
}

// main_fs of sky_unified.slang:185-206: the view vector of the pixel (SV_Position carries the +0.5: (x + 1) / W; clip xy in [0, 1]: both quirks)
SAH_DEV void sky_frag(const LightingArgs& a, const SkyArgs& k, uint32_t x, uint32_t y, Hn (&out)[4]) {
    const Fn sx = (Fn((float)x + 0.5f) + Fn(0.5f)) / Fn(a.res[0]);
    const Fn sy = (Fn((float)y + 0.5f) + Fn(0.5f)) / Fn(a.res[1]);
    F4 vs = mul44(a.inv_proj, F4{sx, sy, Fn(1.f), Fn(1.f)});
    vs = {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w, vs.w / vs.w};
    const F4 wv = mul44(a.inv_view, F4{vs.x, vs.y, vs.z, Fn(0.f)});
    F3 rayDir = -normalize(F3{wv.x, wv.y, wv.z});
    rayDir.y = rayDir.y * Fn(-1.0f);
    const F3 lum = sky_color(k, rayDir);
    out[0] = Hn(lum.x.v);
    out[1] = Hn(lum.y.v);
    out[2] = Hn(lum.z.v);
    out[3] = Hn(1.0f);
}

SAH_DEV uint2 pack_lit(const Hn (&lit)[4]) {
    uint2 r;
    r.x = (uint32_t)__builtin_bit_cast(uint16_t, lit[0].v) | ((uint32_t)__builtin_bit_cast(uint16_t, lit[1].v) << 16);
    r.y = (uint32_t)__builtin_bit_cast(uint16_t, lit[2].v) | ((uint32_t)__builtin_bit_cast(uint16_t, lit[3].v) << 16);
    return r;
}

// ---- a0: per-pixel composition, general restatement ---------------------------------------------------------
template <int SUN, int GI>
SAH_DEV uint2 shade_pixel_general(const LightingArgs& a, const CsmArgs& csm, const LpvArgs& lpv, const SkyArgs& sky, uint32_t x, uint32_t y,
                                  const Px& p, const float* lut) {
    Hn lit[4] = {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};
    const bool surface = p.depth != 0.f;  // every lighting shader discards on depth == 0
    SurfIn si;
    if (surface) si = unpack_surface(p, lut);

    if constexpr (SUN == SAH_SHADOW_MODE_CSM) {
        if (surface) {
            Fn s[4];
            sun_frag(a, csm, x, y, p, si, s);
            if (a.flags & SAH_LIGHTING_QUIRK_SUN_BLEND) {
#pragma unroll
                for (int i = 0; i < 3; i++) lit[i] = Hn((s[i] * s[i] + Fn(tof(lit[i])) * Fn(tof(lit[i]))).v);
                lit[3] = Hn((s[3] * Fn(0.f) + Fn(tof(lit[3])) * Fn(0.f)).v);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) lit[i] = Hn(tof(lit[i]) + s[i].v);
            }
        }
    }
    if constexpr (GI == SAH_GI_LPV) {
        if (surface) {
            Fn s[4];
            gi_lpv_frag(a, lpv, x, y, p, si, s);
#pragma unroll
            for (int i = 0; i < 4; i++) lit[i] = Hn(tof(lit[i]) + s[i].v);
        }
    }
    {  // emissive.frag:15-22 — no discard
        const Fn e = Fn(3.1415927f);
        lit[0] = Hn(tof(lit[0]) + (Fn(lut[p.emission & 0xffu]) * e).v);
        lit[1] = Hn(tof(lit[1]) + (Fn(lut[(p.emission >> 8) & 0xffu]) * e).v);
        lit[2] = Hn(tof(lit[2]) + (Fn(lut[(p.emission >> 16) & 0xffu]) * e).v);
        lit[3] = Hn(tof(lit[3]) + 1.0f);
    }
    if (sky.enabled && !surface) sky_frag(a, sky, x, y, lit);
    if constexpr (SUN == SAH_SHADOW_MODE_RT) {
        if (surface) {
            float add[3];
            sun_rt(a, x, y, p, si, add);
#pragma unroll
            for (int i = 0; i < 3; i++) lit[i] = Hn(tof(lit[i]) + add[i]);
        }
    }
    return pack_lit(lit);
}

}  // namespace sah
