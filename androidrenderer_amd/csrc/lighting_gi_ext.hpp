// Device restatement of the irradiance-cache overlay (a4) and the RTGI reconstruction overlay (a5).
//   a4: RenderCore/shaders/gi/cache/overlay.frag.slang:46-118, probe_sampling.slangi:6-106, common/octahedral.slangi:56-74
//   a5: RenderCore/shaders/gi/rtgi/overlay.frag.slang:68-117
// Operator order and rounding per DESIGN.md §3; used by the general / tiled kernels.
#pragma once
#include "lighting_common.hpp"

namespace sah {

// ---- format decode ---------------------------------------------------------------------------------------------
SAH_DEV float uf11_to_f32(uint32_t v) {
    const uint32_t e = (v >> 6) & 0x1fu, m = v & 0x3fu;
    if (e == 0) return (float)m * (1.0f / 64.0f) * 6.103515625e-5f;
    if (e == 31) return m ? __builtin_nanf("") : __builtin_inff();
    return __uint_as_float(((e + 112u) << 23) | (m << 17));
}
SAH_DEV float uf10_to_f32(uint32_t v) {
    const uint32_t e = (v >> 5) & 0x1fu, m = v & 0x1fu;
    if (e == 0) return (float)m * (1.0f / 32.0f) * 6.103515625e-5f;
    if (e == 31) return m ? __builtin_nanf("") : __builtin_inff();
    return __uint_as_float(((e + 112u) << 23) | (m << 18));
}

SAH_DEV int wrap_repeat(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}
SAH_DEV int array_layer(float l, uint32_t layers) {  // round to nearest even, clamp to [0, layers-1]
    // (min / max, not early returns: those compile to exec-mask regions around one conversion; fmaxf(NaN, 0) is 0)
    return (int)__builtin_fminf(__builtin_fmaxf(__builtin_rintf(l), 0.f), (float)(layers - 1));
}

// 2D-array bilinear, REPEAT (irradiance_cache.cpp:205-217), weighted-sum fma chain; NCH channels decoded by `fetch`
template <int NCH, class Fetch>
SAH_DEV void bilinear_repeat_array(const VolumeArg& v, float u, float vv, int layer, float (&out)[NCH], Fetch fetch) {
    const float px = u * (float)v.width - 0.5f, py = vv * (float)v.height - 0.5f;
    if (isnan_f(px) || isnan_f(py)) {
#pragma unroll
        for (int c = 0; c < NCH; c++) out[c] = __builtin_nanf("");
        return;
    }
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    const float fx = px - fx0, fy = py - fy0;
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    const int x0 = wrap_repeat(clamp_to_int(fx0), (int)v.width), y0 = wrap_repeat(clamp_to_int(fy0), (int)v.height);
    const int x1 = x0 + 1 == (int)v.width ? 0 : x0 + 1, y1 = y0 + 1 == (int)v.height ? 0 : y0 + 1;
    const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    const float w[4] = {wx0 * wy0, fx * wy0, wx0 * fy, fx * fy};
    const uint8_t* base = v.ptr + (size_t)layer * v.slice_pitch;
    float t[4][NCH];
#pragma unroll
    for (int k = 0; k < 4; k++) fetch(*reinterpret_cast<const uint32_t*>(base + (size_t)ys[k] * v.row_pitch + (size_t)xs[k] * 4), t[k]);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) a = __builtin_fmaf(w[k], t[k][c], a);
        out[c] = a;
    }
}

struct F2 {
    Fn x, y;
};

// octahedral.slangi:56-63
SAH_DEV F2 octahedral_coordinates(F3 dir) {
    const Fn l1 = nabs(dir.x) + nabs(dir.y) + nabs(dir.z);
    const Fn inv = Fn(1.f) / l1;
    F2 uv = {dir.x * inv, dir.y * inv};
    if (dir.z.v < 0.f) {
        const Fn sx = Fn(uv.x.v >= 0.f ? 1.f : -1.f), sy = Fn(uv.y.v >= 0.f ? 1.f : -1.f);
        const F2 r = {(Fn(1.f) - nabs(uv.y)) * sx, (Fn(1.f) - nabs(uv.x)) * sy};
        uv = r;
    }
    return uv;
}
// octahedral.slangi:65-74
SAH_DEV void probe_uv(const uint32_t (&idx)[3], F2 oct, uint32_t n0, uint32_t n1, Fn (&uv)[2]) {
    const uint32_t n[2] = {n0, n1};
    const Fn o[2] = {oct.x, oct.y};
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const Fn total = Fn((float)n[i]) + Fn(2.f);
        const Fn tex_size = total * Fn(32.f);
        Fn u = Fn((float)idx[i]) * total + total * Fn(0.5f);
        u = u + o[i] * (Fn((float)n[i]) * Fn(0.5f));
        uv[i] = u / tex_size;
    }
}
SAH_DEV uint32_t f2uint(float f) { return cvt_u32_sat(f); }  // hardware float -> uint: NaN / negatives -> 0, saturating

// probe_sampling.slangi:6-106
SAH_DEV F3 sample_cascade(const CacheArgs& c, F3 location, F3 direction, uint32_t cascade_index) {
    const Fn spacing = Fn(c.spacing[cascade_index]);
    const F3 rel = location - F3{Fn(c.cascade_min[cascade_index][0]), Fn(c.cascade_min[cascade_index][1]), Fn(c.cascade_min[cascade_index][2])};
    const F3 ps = rel / spacing;
    const F3 min_probe = {Fn(__builtin_floorf(ps.x.v)), Fn(__builtin_floorf(ps.y.v)), Fn(__builtin_floorf(ps.z.v))};
    const F3 alpha = {nclamp(ps.x - min_probe.x, Fn(0.f), Fn(1.f)), nclamp(ps.y - min_probe.y, Fn(0.f), Fn(1.f)),
                      nclamp(ps.z - min_probe.z, Fn(0.f), Fn(1.f))};
    F3 irradiance = F3(Fn(0.f));
    Fn weight = Fn(0.f);
    for (uint32_t i = 0; i < 8; i++) {
        const F3 off = {Fn((float)(i & 1u)), Fn((float)((i >> 1) & 1u)), Fn((float)((i >> 2) & 1u))};
        const F3 probe_location = min_probe + off;
        const F3 dir_to_probe = probe_location - ps;
        const Fn dist = length(dir_to_probe) * spacing;
        const F3 pidx_f = probe_location + F3{Fn(0.f), Fn((float)cascade_index) * Fn(8.f), Fn(0.f)};
        const uint32_t pidx[3] = {f2uint(pidx_f.x.v), f2uint(pidx_f.y.v), f2uint(pidx_f.z.v)};
        float validity = 0.f;  // Texture2DArray<half>[uint3]: out-of-range loads return 0
        if (pidx[0] < c.validity.width && pidx[1] < c.validity.height && pidx[2] < c.validity.depth) {
            const uint8_t b = c.validity.ptr[(size_t)pidx[2] * c.validity.slice_pitch + (size_t)pidx[1] * c.validity.row_pitch + pidx[0]];
            validity = rh((float)b / 255.0f);
        }
        if (validity == 0.f) continue;
        const F3 tri = {nmax(Fn(0.001f), mix(Fn(1.f) - alpha.x, alpha.x, off.x)), nmax(Fn(0.001f), mix(Fn(1.f) - alpha.y, alpha.y, off.y)),
                        nmax(Fn(0.001f), mix(Fn(1.f) - alpha.z, alpha.z, off.z))};
        const Fn trilinear_weight = tri.x * tri.y * tri.z;
        Fn probe_weight = Fn(1.f);

        const F2 depth_oct = octahedral_coordinates(-dir_to_probe);
        Fn duv[2];
        probe_uv(pidx, depth_oct, 10u, 10u, duv);
        float dt[2];
        bilinear_repeat_array<2>(c.depth, duv[0].v, duv[1].v, array_layer((float)pidx[2], c.depth.depth), dt, [](uint32_t w, float (&o)[2]) {
            o[0] = (float)hbits(w & 0xffffu);
            o[1] = (float)hbits(w >> 16);
        });
        const Hn dx = Hn(dt[0]), dy = Hn(dt[1]);  // Sampler2DArray<half2>
        const Fn variance = Fn(tof(nabs(dx * dx - dy)));
        Fn cheb = Fn(1.f);
        if (dist.v > tof(dx)) {
            const Fn v = dist - Fn(tof(dx));
            cheb = variance / (variance + (v * v));
            cheb = nmax(cheb * cheb * cheb, Fn(0.f));
        }
        probe_weight = probe_weight * nmax(Fn(0.05f), cheb);
        probe_weight = nmax(Fn(0.000001f), probe_weight);
        const Fn crush = Fn(0.2f);
        if (probe_weight.v < crush.v) probe_weight = probe_weight * ((probe_weight * probe_weight) * (Fn(1.f) / (crush * crush)));
        probe_weight = probe_weight * trilinear_weight;

        const F2 irr_oct = octahedral_coordinates(direction);
        Fn iuv[2];
        probe_uv(pidx, irr_oct, c.probe_size[0], c.probe_size[1], iuv);
        float it[3];
        bilinear_repeat_array<3>(c.irradiance, iuv[0].v, iuv[1].v, array_layer((float)pidx[2], c.irradiance.depth), it,
                                 [](uint32_t w, float (&o)[3]) {
                                     o[0] = uf11_to_f32(w & 0x7ffu);
                                     o[1] = uf11_to_f32((w >> 11) & 0x7ffu);
                                     o[2] = uf10_to_f32((w >> 22) & 0x3ffu);
                                 });
        const H3 pi = {Hn(it[0]), Hn(it[1]), Hn(it[2])};  // Sampler2DArray<half3>
        irradiance = irradiance + to_f(pi) * probe_weight;
        weight = weight + probe_weight;
    }
    if (weight.v == 0.f) return F3(Fn(0.f));
    irradiance = irradiance / weight;
    return irradiance * Fn(2.f) * Fn(rh(3.1415927f));  // PI = 3.1415927h (brdf.slangi wins the #ifndef race)
}

// ---- a4, hot form ----------------------------------------------------------------------------------------------------------
// sample_cascade() with the per-probe work reduced to what the inputs allow; every shortcut is an identity on the bits:
//  * R11G11B10 -> fp16 is a bit shuffle (uf11 = fp16 >> 4, uf10 = fp16 >> 5: same 5-bit exponent and bias, denormals, inf, NaN),
//    so the bilinear taps are v_fma_mix_f32 on the shuffled words instead of a branchy decode per channel;
//  * probe texcoords lie strictly inside (0,1) (u in [1, tex_size - 1] texels), so REPEAT needs one wrap of -1 and no modulo,
//    and u / tex_size has both operands in the restricted-range divide's domain by construction;
//  * validity bytes go through the UNORM8 table; the octahedral direction of the shading normal is probe independent;
//  * sqrt / reciprocal / divide use the restricted-range twins where a compare establishes the domain.  `bad` is returned
//    set when such a compare fails: the caller then evaluates sample_cascade() for the pixel.
SAH_DEV void probe_uv_nr(const uint32_t (&idx)[3], F2 oct, uint32_t n0, uint32_t n1, float (&uv)[2]) {
    const uint32_t n[2] = {n0, n1};
    const Fn o[2] = {oct.x, oct.y};
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const Fn total = Fn((float)n[i]) + Fn(2.f);
        const Fn tex_size = total * Fn(32.f);
        Fn u = Fn((float)idx[i]) * total + total * Fn(0.5f);
        u = u + o[i] * (Fn((float)n[i]) * Fn(0.5f));
        uv[i] = div_nr(u.v, tex_size.v);
    }
}
struct BilinearTaps {
    uint32_t row0, row1;  // byte offsets of texel (x0, y0) and (x0, y0 + 1); texel (x0 + 1, .) follows at +4
    float w[4];
};
// 2D-array bilinear set-up for probe texcoords: u * width lies in [1, width - 1] (probe_uv_nr on an atlas that is exactly
// 32 blocks wide: CacheArgs::hot_ok), so floor(u * width - 0.5) is in [0, width - 2] and REPEAT never wraps: the two taps of a row
// are adjacent in memory and are fetched as one 8-byte load.  The clamp only matters for garbage coordinates of a pixel that is
// being re-evaluated anyway (`bad`): it keeps every address inside the atlas, so the loads need no predicate.
SAH_DEV BilinearTaps bilinear_setup_probe(const VolumeArg& v, float u, float vv, int layer) {
    const float px = u * (float)v.width - 0.5f, py = vv * (float)v.height - 0.5f;
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    const float fx = px - fx0, fy = py - fy0;
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    const int x0 = clamp_index(fx0, (int)v.width - 2), y0 = clamp_index(fy0, (int)v.height - 2);
    BilinearTaps t;
    t.row0 = (uint32_t)layer * v.slice_pitch + (uint32_t)y0 * v.row_pitch + (uint32_t)x0 * 4u;  // atlases are < 4 GiB (host check)
    t.row1 = t.row0 + v.row_pitch;
    t.w[0] = wx0 * wy0;
    t.w[1] = fx * wy0;
    t.w[2] = wx0 * fy;
    t.w[3] = fx * fy;
    return t;
}
SAH_DEV uint2 load_pair(const uint8_t* base, uint32_t off) {  // 4-byte aligned, 8 bytes
    uint2 q;
    __builtin_memcpy(&q, base + off, 8);
    return q;
}
// octahedral_coordinates() with the reciprocal of the L1 norm from rcp_nr; `bad` when the norm is outside its domain
SAH_DEV F2 octahedral_coordinates_nr(F3 dir, bool& bad) {
    const Fn l1 = nabs(dir.x) + nabs(dir.y) + nabs(dir.z);
    bad = bad | !((l1.v >= kDivLo) & (l1.v <= kDivHi));  // (bitwise on purpose, here and below: `||` / `&&` chains of compares compile to
                                                          //  nested exec-mask regions of one instruction each)
    const Fn inv = Fn(rcp_nr(l1.v));
    F2 uv = {dir.x * inv, dir.y * inv};
    const Fn sx = Fn(uv.x.v >= 0.f ? 1.f : -1.f), sy = Fn(uv.y.v >= 0.f ? 1.f : -1.f);
    const F2 r = {(Fn(1.f) - nabs(uv.y)) * sx, (Fn(1.f) - nabs(uv.x)) * sy};
    const bool fold = dir.z.v < 0.f;
    return {fold ? r.x : uv.x, fold ? r.y : uv.y};
}

// a / b for a divisor known on the host: z = RN(1 / b).  One correction step is exact for every b = 32 (n + 2), n = 1..30, and every
// fp32 a in [0.5, b] (tools/microbench/div_const_check.c, exhaustive) — the range probe texcoords live in; other operands belong to
// pixels that are re-evaluated anyway.
SAH_DEV float div_const(float a, float b, float z) {
    const float q0 = a * z;
    const float r0 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r0, z, q0);
}
// One axis of bilinear_setup_probe(): texel offset of the first tap (clamped as there) and the two weights.
struct ProbeAxis {
    uint32_t off;  // x0 * 4 or y0 * row_pitch
    float w0, w1;
};
SAH_DEV ProbeAxis probe_axis(float u, uint32_t size, uint32_t stride) {
    const float p = u * (float)size - 0.5f;
    const float f0 = __builtin_floorf(p);
    const float f = p - f0;
    const int i0 = clamp_index(f0, (int)size - 2);
    return {(uint32_t)i0 * stride, 1.0f - f, f};
}

// Everything that depends on one axis only (probe_location, its distance term, index, trilinear weight, the irradiance texcoord
// and its bilinear set-up: the shading normal's octahedral direction is probe independent) is evaluated for the two probes of that
// axis instead of for all eight: the same operators on the same operands, so the same bits.  mix(1 - a, a, offset) with offset 0 / 1
// is (1 - a) * 1 + a * 0 and (1 - a) * 0 + a * 1: 1 - a and a for the finite alpha in [0, 1] the clamp leaves.
SAH_DEV F3 sample_cascade_fast(const CacheArgs& c, F3 location, F3 direction, uint32_t cascade_index, const float* lut, bool& bad) {
    const Fn spacing = Fn(c.spacing[cascade_index]);
    const F3 rel = location - F3{Fn(c.cascade_min[cascade_index][0]), Fn(c.cascade_min[cascade_index][1]), Fn(c.cascade_min[cascade_index][2])};
    // (a power-of-two spacing — the reference's 0.5 m doubled per cascade — divides exactly: the product with its reciprocal is the same
    // real number rounded the same way; uniform branch)
    const F3 ps = c.spacing_pow2 ? rel * Fn(c.inv_spacing[cascade_index]) : rel / spacing;
    const Fn psa[3] = {ps.x, ps.y, ps.z};
    const F2 irr_oct = octahedral_coordinates(direction);  // probe independent (IEEE form: once per pixel)
    const bool irr_oct_ok = (__builtin_fabsf(irr_oct.x.v) <= 1.0f) & (__builtin_fabsf(irr_oct.y.v) <= 1.0f);  // false for NaN
    bad = bad | !irr_oct_ok;

    Fn dp[3][2], sq[3][2], tri[3][2];
    uint32_t pidx[3][2];
    const Fn cascade_row = Fn((float)cascade_index) * Fn(8.f);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const Fn mp = Fn(__builtin_floorf(psa[k].v));
        const Fn alpha = nclamp(psa[k] - mp, Fn(0.f), Fn(1.f));
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const Fn pl = mp + Fn((float)j);
            dp[k][j] = pl - psa[k];
            sq[k][j] = dp[k][j] * dp[k][j];
            pidx[k][j] = f2uint((k == 1 ? pl + cascade_row : pl + Fn(0.f)).v);
            tri[k][j] = nmax(Fn(0.001f), j ? alpha : Fn(1.f) - alpha);
        }
        // signs of the direction from a probe to the point, per axis: -dp[k][0] = ps - floor(ps) >= 0 and -dp[k][1] = ps - floor(ps) - 1 < 0.
        // The octahedral fold below takes them as known, and the restricted-range root / reciprocal / divide of the probe loop take from
        // here that every component of that direction is 0 or at least 2^-40 in magnitude and at most 1: the squared distance is then 0
        // or in [2^-80, 3], the L1 norm 0 (corner 0 of all three axes only) or in [2^-40, 3].  A coordinate within 2^-40 cells of a
        // cell boundary (or a non-finite one) sends the pixel to sample_cascade()
        bad = bad | !((dp[k][1].v >= 0x1p-40f) & (dp[k][1].v <= 1.0f)) | !((dp[k][0].v == 0.f) | ((dp[k][0].v <= -0x1p-40f) & (dp[k][0].v >= -1.0f)));
    }
    // texcoord terms per axis: idx * total + total / 2 (depth atlas: 10 + 2 texels), and the whole irradiance axis
    Fn dbase[2][2];
    ProbeAxis iax[2][2];
    const Fn oct[2] = {irr_oct.x, irr_oct.y};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const Fn total = Fn((float)c.probe_size[k]) + Fn(2.f);
        const Fn tex_size = total * Fn(32.f);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            dbase[k][j] = Fn((float)pidx[k][j]) * Fn(12.f) + Fn(12.f) * Fn(0.5f);
            Fn u = Fn((float)pidx[k][j]) * total + total * Fn(0.5f);
            u = u + oct[k] * (Fn((float)c.probe_size[k]) * Fn(0.5f));
            const float uv = div_const(u.v, tex_size.v, c.inv_tex[k]);
            iax[k][j] = k == 0 ? probe_axis(uv, c.irradiance.width, 4u) : probe_axis(uv, c.irradiance.height, c.irradiance.row_pitch);
        }
    }
    uint32_t dlayer[2], ilayer[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        dlayer[j] = (uint32_t)array_layer((float)pidx[2][j], c.depth.depth) * c.depth.slice_pitch;
        ilayer[j] = (uint32_t)array_layer((float)pidx[2][j], c.irradiance.depth) * c.irradiance.slice_pitch;
    }
    // the eight probes' validity bytes, fetched together before the loop (Texture2DArray<half>[uint3]: out-of-range loads return 0, so
    // the address is clamped into the atlas and the range test applied to the result): read where they are used, each probe's byte is a
    // memory round trip the rest of its iteration waits for — eight in a row per pixel.  Only `validity == 0` is used, and
    // (half)(b / 255) is zero for b = 0 alone
    // The range test as a byte mask per axis corner (all ones / zero), so that "byte != 0 and all three indices in range" is one
    // three-input AND and one compare per probe (as bools the compiler builds 0 / 1 words and combines those: eight instructions)
    uint32_t voff[3][2], vmask[3][2];
    {
        const uint32_t vext[3] = {c.validity.width, c.validity.height, c.validity.depth}, vpitch[3] = {1u, c.validity.row_pitch, c.validity.slice_pitch};
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                vmask[k][j] = pidx[k][j] < vext[k] ? 0xffu : 0u;
                voff[k][j] = min(pidx[k][j], vext[k] - 1u) * vpitch[k];
            }
    }
    const uint32_t vmask_yz[2][2] = {{vmask[1][0] & vmask[2][0], vmask[1][1] & vmask[2][0]}, {vmask[1][0] & vmask[2][1], vmask[1][1] & vmask[2][1]}};  // [jz][jy]
    uint32_t vbyte[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) vbyte[i] = c.validity.ptr[voff[2][(i >> 2) & 1u] + voff[1][(i >> 1) & 1u] + voff[0][i & 1u]];

    F3 irradiance = F3(Fn(0.f));
    Fn weight = Fn(0.f);
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        const int jx = i & 1u, jy = (i >> 1) & 1u, jz = (i >> 2) & 1u;
        if ((vbyte[i] & vmask[0][jx] & vmask_yz[jz][jy]) == 0u) continue;
        // a valid probe index is < 32 per axis (validity atlas extent, host check <= 64), which bounds every texcoord below
        const F3 dir_to_probe = {dp[0][jx], dp[1][jy], dp[2][jz]};
        const Fn d2 = sq[0][jx] + sq[1][jy] + sq[2][jz];
        bool pbad = false;  // (d2 is 0 or in [2^-80, 3]: the per-axis check above)
        const Fn dist = Fn(sqrt_nr0(d2.v)) * spacing;
        const Fn trilinear_weight = tri[0][jx] * tri[1][jy] * tri[2][jz];
        Fn probe_weight = Fn(1.f);

        // octahedral_coordinates(-dir_to_probe) with the signs above: uv = -dp.xy / L1 is >= 0 for corner 0 and < 0 for corner 1 of its
        // axis; the fold (direction.z < 0) happens for the z corner 1 and nowhere else, and multiplying by sign_not_zero is a negation or
        // nothing
        F2 depth_oct;
        {
            const Fn l1 = nabs(dir_to_probe.x) + nabs(dir_to_probe.y) + nabs(dir_to_probe.z);
            if (i == 0) pbad = !(l1.v > 0.f);  // the point ON the probe: no direction; every other corner has a component >= 2^-40
            const Fn inv = Fn(rcp_nr(l1.v));
            const F2 uv = {-dir_to_probe.x * inv, -dir_to_probe.y * inv};
            if (jz == 0) {
                depth_oct = uv;
            } else {
                const Fn rx = Fn(1.f) - nabs(uv.y), ry = Fn(1.f) - nabs(uv.x);
                depth_oct = {jx == 0 ? rx : -rx, jy == 0 ? ry : -ry};
            }
        }
        const float du = div_const((dbase[0][jx] + depth_oct.x * (Fn(10.f) * Fn(0.5f))).v, 384.0f, 1.0f / 384.0f);
        const float dv = div_const((dbase[1][jy] + depth_oct.y * (Fn(10.f) * Fn(0.5f))).v, 384.0f, 1.0f / 384.0f);
        // (the depth atlas is 384 x 384 texels in the hot form — CacheArgs::hot_ok —, as the 384 of the two quotients above says already)
        const ProbeAxis dxa = probe_axis(du, 384u, 4u), dya = probe_axis(dv, 384u, c.depth.row_pitch);
        const uint32_t drow0 = dlayer[jz] + dya.off + dxa.off;
        const uint2 d0 = load_pair(c.depth.ptr, drow0), d1 = load_pair(c.depth.ptr, drow0 + c.depth.row_pitch);
        const uint32_t dw[4] = {d0.x, d0.y, d1.x, d1.y};  // tap order (x0,y0) (x1,y0) (x0,y1) (x1,y1)
        const float dwt[4] = {dxa.w0 * dya.w0, dxa.w1 * dya.w0, dxa.w0 * dya.w1, dxa.w1 * dya.w1};
        float dt0 = 0.f, dt1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dt0 = fma_mix_lo(dwt[k], dw[k], dt0);
            dt1 = fma_mix_hi(dwt[k], dw[k], dt1);
        }
        const Hn dx = Hn(dt0), dy = Hn(dt1);  // Sampler2DArray<half2>
        const Fn variance = Fn(tof(nabs(dx * dx - dy)));
        const Fn v = dist - Fn(tof(dx));
        const Fn cden = variance + (v * v);
        const bool behind = dist.v > tof(dx);
        // div_nr's domain: variance is a half value widened (0, or in [2^-24, 65504], or not finite — and then so is cden), and
        // cden >= variance: both are in range iff cden is (one unsigned compare on the bits: negative, NaN and inf fall outside)
        pbad = pbad | (behind & !(__builtin_bit_cast(uint32_t, cden.v) - __builtin_bit_cast(uint32_t, kDivLo) <=
                                  __builtin_bit_cast(uint32_t, kDivHi) - __builtin_bit_cast(uint32_t, kDivLo)));
        Fn cheb = Fn(div_nr(variance.v, cden.v));
        cheb = nmax(cheb * cheb * cheb, Fn(0.f));
        cheb = behind ? cheb : Fn(1.f);
        probe_weight = probe_weight * nmax(Fn(0.05f), cheb);
        probe_weight = nmax(Fn(0.000001f), probe_weight);
        const Fn crush = Fn(0.2f);
        const Fn crushed = probe_weight * ((probe_weight * probe_weight) * (Fn(1.f) / (crush * crush)));
        probe_weight = probe_weight.v < crush.v ? crushed : probe_weight;
        probe_weight = probe_weight * trilinear_weight;

        const ProbeAxis ixa = iax[0][jx], iya = iax[1][jy];
        const uint32_t irow0 = ilayer[jz] + iya.off + ixa.off;
        // the four taps from the widened copy (R11G11B10 -> fp16 is a bit shuffle, fp16 -> fp32 exact: the products below are the
        // fused multiply-adds v_fma_mix_f32 would do on the shuffled words).  The copy's pitches are 4 x the atlas's.
        const float4* irow = reinterpret_cast<const float4*>(c.irr32 + 4u * irow0);
        const float4* irow_next = reinterpret_cast<const float4*>(c.irr32 + 4u * (irow0 + c.irradiance.row_pitch));
        const float4 it[4] = {irow[0], irow[1], irow_next[0], irow_next[1]};
        const float iwt[4] = {ixa.w0 * iya.w0, ixa.w1 * iya.w0, ixa.w0 * iya.w1, ixa.w1 * iya.w1};
        float ir = 0.f, ig = 0.f, ib = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ir = __builtin_fmaf(iwt[k], it[k].x, ir);
            ig = __builtin_fmaf(iwt[k], it[k].y, ig);
            ib = __builtin_fmaf(iwt[k], it[k].z, ib);
        }
        const H3 pi = {Hn(ir), Hn(ig), Hn(ib)};  // Sampler2DArray<half3>
        irradiance = irradiance + F3{Fn(mul_mix(probe_weight.v, pi.x)), Fn(mul_mix(probe_weight.v, pi.y)), Fn(mul_mix(probe_weight.v, pi.z))};  // to_f(pi) * probe_weight
        weight = weight + probe_weight;
        bad = bad | pbad;
    }
    if (weight.v == 0.f) return F3(Fn(0.f));
    // the three quotients through one refined reciprocal (div_nr with its y1 steps shared): the sums are +0 or at least 2^-40 and the weight
    // at least 2^-40 wherever the probes carry light and one of them counts; a pixel outside that domain is re-evaluated by sample_cascade()
    {
        const float w = weight.v;
        // (on the bit patterns, unsigned: the sums are >= +0, so "0 or >= 2^-40" is bits - 1 >= bits(2^-40) - 1 with 0 - 1 wrapping to the
        // top, and anything negative, infinite or NaN lies above bits(2^40))
        const uint32_t bx = __builtin_bit_cast(uint32_t, irradiance.x.v), by = __builtin_bit_cast(uint32_t, irradiance.y.v),
                       bz = __builtin_bit_cast(uint32_t, irradiance.z.v), bw = __builtin_bit_cast(uint32_t, w);
        const uint32_t mn = min(min(bx - 1u, by - 1u), min(bz - 1u, bw - 1u)), mx = max(max(bx, by), max(bz, bw));
        bad = bad | !((mn >= __builtin_bit_cast(uint32_t, kDivLo) - 1u) & (mx <= __builtin_bit_cast(uint32_t, kDivHi)));
        const float y0 = __builtin_amdgcn_rcpf(w);
        const float y1 = __builtin_fmaf(__builtin_fmaf(-w, y0, 1.0f), y0, y0);
        auto quot = [&](float a) {
            const float q0 = a * y1;
            const float q1 = __builtin_fmaf(__builtin_fmaf(-w, q0, a), y1, q0);
            return Fn(__builtin_fmaf(__builtin_fmaf(-w, q1, a), y1, q1));
        };
        irradiance = {quot(irradiance.x.v), quot(irradiance.y.v), quot(irradiance.z.v)};
    }
    return irradiance * Fn(2.f) * Fn(rh(3.1415927f));  // PI = 3.1415927h (brdf.slangi wins the #ifndef race)
}

// What the Slang passes of one pixel — the RT-mode sun (directional_light.rt.slang:58-89), the cache overlay (overlay.frag.slang:46-66) and
// the RTGI overlay (rtgi/overlay.frag.slang:68-80) — all begin with: the half-precision surface, the world-space location from
// (pixel + 0.5) / resolution, and the half view vector.  The same operators on the same inputs in all three, so the tiled kernel, which runs
// two of them per pixel, evaluates them once (the Hn constructors are opaque to the optimiser, which therefore cannot merge them itself).
struct SlangGeom {
    Surface<Hn> s;
    F3 location;
    H3 V;
    BrdfView<Hn> bv;  // brdf_sl_view(s, V): the half BRDF's view-only terms, shared by every brdf of the pixel (set by the kernel once s and V stand)
};
SAH_DEV SlangGeom slang_geometry(const LightingArgs& a, uint32_t x, uint32_t y, const Px& p, const SurfIn& si) {
    SlangGeom g;
    g.s.base_color = {Hn(si.color[0]), Hn(si.color[1]), Hn(si.color[2])};
    g.s.normal = normalize(H3{Hn(si.normal[0]), Hn(si.normal[1]), Hn(si.normal[2])});
    g.s.roughness = Hn(si.rough);
    g.s.metalness = Hn(si.metal);
    g.location = worldspace_location_slang(a, (float)x, (float)y, p.depth);
    g.V = to_h(normalize(g.location - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])}));
    g.bv = brdf_sl_view(g.s, g.V);
    return g;
}
// sun_rt() (lighting_common.hpp) on the shared geometry
SAH_DEV void sun_rt_shared(const LightingArgs& a, const Px& p, const SlangGeom& g, float (&add)[3]) {
    const F3 L = {Fn(a.sun_L[0]), Fn(a.sun_L[1]), Fn(a.sun_L[2])};
    const Hn ndotl = Hn(nclamp(dot(L, to_f(g.s.normal)), Fn(0.f), Fn(1.f)).v);
    const H3 Lh = to_h(L);
    const H3 b = brdf_sl_light(g.s, g.bv, Lh, g.V);  // == Fd(s, Lh, V) + Fr(s, Lh, V)
    const H3 nb = ndotl * b;
    F3 radiance = to_f(nb) * F3{Fn(a.sun_color[0]), Fn(a.sun_color[1]), Fn(a.sun_color[2])};
    if (tof(ndotl) > 0.f) radiance = radiance * Fn(p.mask);
    const Fn exposure = Fn(0.00031415927f);
    add[0] = (radiance.x * exposure).v;
    add[1] = (radiance.y * exposure).v;
    add[2] = (radiance.z * exposure).v;
}

// `lut` != nullptr selects sample_cascade_fast(); *bad is then set when the pixel has to be re-evaluated with lut == nullptr.
// (the Slang overlays return half4: `out` is what the fragment hands to the blender)
SAH_DEV void gi_cache_frag(const LightingArgs& a, const CacheArgs& c, const SlangGeom& geom, Hn (&out)[4], const float* lut = nullptr, bool* bad = nullptr) {
    const Surface<Hn>& s = geom.s;
    const F3 location = geom.location;
    const H3 V = geom.V;
    // the first cascade whose box holds the point (5: none).  Cascade 0 is tested first: where it holds every lane of the wave (near-field
    // pixels) the other three boxes cannot change the answer and are not tested (six compares each)
    auto inside = [&](uint32_t i) {
        return location.x.v > c.cascade_min[i][0] && location.y.v > c.cascade_min[i][1] && location.z.v > c.cascade_min[i][2] &&
               location.x.v < c.cascade_max[i][0] && location.y.v < c.cascade_max[i][1] && location.z.v < c.cascade_max[i][2];
    };
    uint32_t cascade_index = 0;
    const bool in0 = inside(0u);
    if (!wave_all(in0)) {
        cascade_index = 5;
#pragma unroll
        for (int i = 3; i >= 1; i--) cascade_index = inside((uint32_t)i) ? (uint32_t)i : cascade_index;
        cascade_index = in0 ? 0u : cascade_index;
    }
    if (cascade_index > 3) {  // returns (half4)0 and is still blended (overlay.frag.slang:79-81)
        out[0] = out[1] = out[2] = out[3] = Hn::lit(0.f);
        return;
    }
    const H3 irradiance = to_h(lut ? sample_cascade_fast(c, location, to_f(s.normal), cascade_index, lut, *bad)
                                   : sample_cascade(c, location, to_f(s.normal), cascade_index));
    const H3 b = brdf_sl_light(s, geom.bv, s.normal, V);  // == Fd(s, N, V) + Fr(s, N, V)
    const Hn exposure = Hn::lit(0.314159f);
    H3 col = b * irradiance * exposure;
    if (c.debug_mode == 1) {  // red, green, blue, yellow by cascade (three selects: a table indexed per lane costs a compare chain per entry)
        const bool r = cascade_index == 0u || cascade_index == 3u, g = cascade_index == 1u || cascade_index == 3u, bl = cascade_index == 2u;
        col = {r ? Hn::lit(1.f) : Hn::lit(0.f), g ? Hn::lit(1.f) : Hn::lit(0.f), bl ? Hn::lit(1.f) : Hn::lit(0.f)};
    }
    if (any_nan(col)) col = H3(Hn::lit(0.f));
    out[0] = col.x;
    out[1] = col.y;
    out[2] = col.z;
    out[3] = Hn::lit(1.f);
}

// ---- a5 ------------------------------------------------------------------------------------------------------------------
SAH_DEV H3 rtgi_contribution(const Surface<Hn>& s, const BrdfView<Hn>& bv, H3 V, H3 dir, H3 irr) {
    const H3 b = brdf_sl_light(s, bv, dir, V);  // == Fd(s, dir, V) + Fr(s, dir, V)
    const Hn ndotl = Hn(nclamp(Fn(tof(dot(dir, s.normal))), Fn(0.f), Fn(1.f)).v);
    return b * irr * ndotl;
}
SAH_DEV void load_path(const LightingArgs& a, const RtgiArgs& r, uint32_t px, uint32_t py, H3& dir, H3& irr) {
    if (px < a.width && py < a.height) {  // out-of-range image loads return 0
        const half4_t d = *reinterpret_cast<const half4_t*>(r.ray_buffer.ptr + (size_t)py * r.ray_buffer.pitch + (size_t)px * 8);
        const half4_t i = *reinterpret_cast<const half4_t*>(r.ray_irradiance.ptr + (size_t)py * r.ray_irradiance.pitch + (size_t)px * 8);
        dir = {Hn::raw(d[0]), Hn::raw(d[1]), Hn::raw(d[2])};
        irr = {Hn::raw(i[0]), Hn::raw(i[1]), Hn::raw(i[2])};
    } else {
        dir = H3(Hn::lit(0.f));
        irr = H3(Hn::lit(0.f));
    }
}

SAH_DEV void gi_rtgi_frag(const LightingArgs& a, const RtgiArgs& r, uint32_t x, uint32_t y, const SlangGeom& geom, const float* lut, Hn (&out)[4]) {
    const Surface<Hn>& s = geom.s;
    const F3 location = geom.location;
    const H3 V = geom.V;
    H3 dir, irr;
    load_path(a, r, x, y, dir, irr);
    H3 radiance = rtgi_contribution(s, geom.bv, V, dir, irr);
    uint32_t num_samples = 1;
    for (uint32_t ray = 0; ray < r.num_extra_rays; ray++) {
        const Fn phi = Fn(1.618033988749895f);  // r1(n): overlay.frag.slang:30-36
        const Fn q = Fn((float)ray) / phi;
        Fn r1x = Fn(2.f) + q, r1y = Fn(3.f) + q;
        r1x = r1x - Fn(__builtin_floorf(r1x.v));
        r1y = r1y - Fn(__builtin_floorf(r1y.v));
        const uint32_t nox = f2uint((r1x * Fn(128.f)).v), noy = f2uint((r1y * Fn(128.f)).v);
        const uint32_t nxp = (x + nox) % 128u, nyp = (y + noy) % 128u;
        float n0 = 0.f, n1 = 0.f;
        if (r.noise.ptr && nxp < r.noise_w && nyp < r.noise_h) {
            const uint32_t nw = *reinterpret_cast<const uint32_t*>(r.noise.ptr + (size_t)nyp * r.noise.pitch + (size_t)nxp * 4);
            n0 = lut[256 + (nw & 0xffu)];
            n1 = lut[256 + ((nw >> 8) & 0xffu)];
        }
        const Hn nsx = Hn(n0) * Hn::lit(2.f) - Hn::lit(1.f), nsy = Hn(n1) * Hn::lit(2.f) - Hn::lit(1.f);
        const Fn ox = Fn((float)x) + Fn(tof(nsx)) * Fn(r.extra_ray_radius), oy = Fn((float)y) + Fn(tof(nsy)) * Fn(r.extra_ray_radius);
        const uint32_t opx = f2uint(__builtin_rintf(ox.v)), opy = f2uint(__builtin_rintf(oy.v));
        float odepth = 0.f;
        if (opx < a.width && opy < a.height) odepth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)opy * a.depth.pitch + (size_t)opx * 4);
        const F3 other = worldspace_location_slang(a, (float)opx, (float)opy, odepth);
        if (length(location - other).v > 2.f) continue;  // NaN compares false: not skipped, as in the shader
        H3 d2, i2;
        load_path(a, r, opx, opy, d2, i2);
        radiance = radiance + rtgi_contribution(s, geom.bv, V, d2, i2);
        num_samples++;
    }
    if (any_nan(radiance)) radiance = H3(Hn::lit(0.f));
    const Hn n = Hn((float)num_samples);
    out[0] = radiance.x / n;
    out[1] = radiance.y / n;
    out[2] = radiance.z / n;
    out[3] = Hn::lit(1.f);
}

// ---- a9 (extension): one point light, spec in DESIGN.md §5b and include/sah_hip.h ---------------------------------------
struct PointLightDev {
    float px, py, pz, radius, cr, cg, cb, intensity;
};
// The light as the kernel loads it (wave-uniform: eight scalar registers): the bit patterns beside the values, so that the
// preconditions of the hot form are integer compares of scalars — they run on the scalar unit, where float compares were six VALU
// instructions per light and wave.  (Tests written on a bit_cast of the FLOAT are recognised by the compiler and turned back into
// v_cmp_class_f32; a word that was loaded as an integer is not.)
struct PointLightWords {
    uint32_t w[8];
};
SAH_DEV PointLightDev point_light_values(const PointLightWords& b) {
    auto f = [](uint32_t u) { return __builtin_bit_cast(float, u); };
    return {f(b.w[0]), f(b.w[1]), f(b.w[2]), f(b.w[3]), f(b.w[4]), f(b.w[5]), f(b.w[6]), f(b.w[7])};
}
SAH_DEV bool point_light_hot_ok(const PointLightWords& b) {
    auto bits = [](float f) { return __builtin_bit_cast(uint32_t, f); };
    auto finite = [&](uint32_t u) { return (~u & 0x7f800000u) != 0u; };
    // kDivLo <= radius <= kDivHi: unsigned distance from the lower bound (negative, NaN and infinite patterns wrap past the upper one)
    return b.w[3] - bits(kDivLo) <= bits(kDivHi) - bits(kDivLo) && finite(b.w[4]) && finite(b.w[5]) && finite(b.w[6]) && finite(b.w[7]);
}
SAH_DEV F3 point_light_contribution(const Surface<Fn>& s, F3 ws, F3 V, const PointLightDev& pl) {
    const F3 lv = F3{Fn(pl.px), Fn(pl.py), Fn(pl.pz)} - ws;
    const Fn d2 = dot(lv, lv);
    const F3 L = lv * inversesqrt(d2);
    const Fn dist = nsqrt(d2);
    const Fn ndotl = nclamp(dot(s.normal, L), Fn(0.f), Fn(1.f));
    const Fn xr = dist / Fn(pl.radius);
    const Fn x2 = xr * xr;
    const Fn x4 = x2 * x2;
    const Fn w = nclamp(Fn(1.f) - x4, Fn(0.f), Fn(1.f));
    const Fn att = (w * w) / nmax(d2, Fn(1e-4f));
    const F3 b = brdf_sl(s, L, V);  // == Fd(s, L, V) + Fr(s, L, V)
    F3 c = ndotl * b * F3{Fn(pl.cr), Fn(pl.cg), Fn(pl.cb)} * (Fn(pl.intensity) * att);
    if (any_nan(c)) c = F3(Fn(0.f));
    return c;
}

// Hot form of point_light_contribution() for a light with 2^-40 <= radius <= 2^40 and finite fields (point_light_hot_ok(), per
// light, uniformly): restricted-range sqrt / reciprocal / divide (numerics.hpp) and the shared-subexpression BRDF of the fast
// path.  `bad` (a lane mask: numerics.hpp, lanes()) is set when an operand leaves a domain or the result holds a NaN (the spec turns that into 0): the caller then
// evaluates the general form for the pixel.
// `inv_radius` = div_nr_refine(pl.radius): the part of dist / radius that depends on the light only.  `d2_out`, `away`: the masks of
// "d2 outside [2^-80, 2^40]" and "N.L <= 0" the caller's votes have built (a ballot of a compare from another basic block is not free).
SAH_DEV F3 point_light_contribution_fast(const Surface<Fn>& s, const BrdfPixel& bp, F3 lv, Fn d2, F3 V, const PointLightDev& pl, float inv_radius, lanemask d2_out,
                                         lanemask away, lanemask& bad) {
    const Fn dist = Fn(sqrt_nr(d2.v));
    const F3 L = lv * Fn(rcp_nr(dist.v));
    const Fn ndotl = nclamp(dot(s.normal, L), Fn(0.f), Fn(1.f));
    const Fn xr = Fn(div_nr_y1(dist.v, pl.radius, inv_radius));
    const Fn x2 = xr * xr;
    const Fn x4 = x2 * x2;
    const Fn w = nclamp(Fn(1.f) - x4, Fn(0.f), Fn(1.f));
    const Fn ww = w * w;
    const Fn att = Fn(div_nr(ww.v, vmax_f32(d2.v, 1e-4f)));  // (d2 is a sum of products: never a signalling NaN)
    lanemask brdf_bad;
    const F3 b = brdf_fast_light(s, bp, L, V, away, brdf_bad);
    const F3 c = ndotl * b * F3{Fn(pl.cr), Fn(pl.cg), Fn(pl.cb)} * (Fn(pl.intensity) * att);
    const float nan_probe = (c.x + c.y + c.z).v;
    bad = brdf_bad | d2_out | (lanes(ww.v != 0.f) & lanes(!(ww.v >= kDivLo))) | lanes(nan_probe != nan_probe);
    return c;
}

}  // namespace sah
