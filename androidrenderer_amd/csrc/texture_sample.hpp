// Material texture sampling shared by the scene rasteriser (raster.hip: SampleBias with quad derivatives) and the any-hit stage of the
// ray tracer (rt.hip: SampleLevel 0).  The rules are the ones include/sah_hip.h lists under sah_texture.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sah_hip.h"
#include "numerics.hpp"

namespace sah {

// ---- material textures: textures[index].SampleBias(texcoord, mip_bias), gltf_basic_pbr.slang:177-226 ------------------------------------
// The rules include/sah_hip.h lists under sah_texture (level of detail from fine quad derivatives, bias and clamps, level selection,
// filters, address modes, decode).
SAH_DEV int wrap_texel(int i, int n, uint32_t mode) {
    if (mode == SAH_ADDRESS_CLAMP_TO_EDGE) return min(max(i, 0), n - 1);
    if (mode == SAH_ADDRESS_MIRRORED_REPEAT) {
        int m = i % (2 * n);
        if (m < 0) m += 2 * n;
        return m < n ? m : 2 * n - 1 - m;
    }
    const int m = i % n;
    return m < 0 ? m + n : m;
}
SAH_DEV void fetch_rgba8(const float* luts, const sah_plane& p, int x, int y, float out[4]) {
    uint32_t w;
    __builtin_memcpy(&w, (const uint8_t*)p.ptr + (size_t)y * p.row_pitch_bytes + (size_t)x * 4, 4);
    const uint32_t rgb_table = p.format == SAH_FORMAT_R8G8B8A8_SRGB ? 0u : 256u;  // luts: 256 sRGB8 -> linear, 256 UNORM8 -> float
    for (int c = 0; c < 3; c++) out[c] = luts[rgb_table + ((w >> (8 * c)) & 0xffu)];
    out[3] = luts[256u + (w >> 24)];
}
SAH_DEV int floor_to_int(float f) { return (int)__builtin_fminf(__builtin_fmaxf(f, -1.0e9f), 1.0e9f); }  // f is integral
SAH_DEV void sample_level(const float* luts, const sah_plane& p, const sah_sampler& s, uint32_t filter, float u, float v, float out[4]) {
    const int w = (int)p.width, h = (int)p.height;
    if (filter == SAH_FILTER_NEAREST) {
        const float px = u * (float)w, py = v * (float)h;
        if (px != px || py != py) { for (int c = 0; c < 4; c++) out[c] = __builtin_nanf(""); return; }
        fetch_rgba8(luts, p, wrap_texel(floor_to_int(__builtin_floorf(px)), w, s.address_u), wrap_texel(floor_to_int(__builtin_floorf(py)), h, s.address_v), out);
        return;
    }
    const float px = u * (float)w - 0.5f, py = v * (float)h - 0.5f;
    if (px != px || py != py) { for (int c = 0; c < 4; c++) out[c] = __builtin_nanf(""); return; }
    const float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
    const float fx = px - fx0, fy = py - fy0;
    const int x0 = floor_to_int(fx0), y0 = floor_to_int(fy0);
    const int xa = wrap_texel(x0, w, s.address_u), xb = wrap_texel(x0 + 1, w, s.address_u);
    const int ya = wrap_texel(y0, h, s.address_v), yb = wrap_texel(y0 + 1, h, s.address_v);
    float t00[4], t10[4], t01[4], t11[4];
    fetch_rgba8(luts, p, xa, ya, t00); fetch_rgba8(luts, p, xb, ya, t10); fetch_rgba8(luts, p, xa, yb, t01); fetch_rgba8(luts, p, xb, yb, t11);
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    const float w00 = wx0 * wy0, w10 = fx * wy0, w01 = wx0 * fy, w11 = fx * fy;
    for (int c = 0; c < 4; c++) {
        float acc = __builtin_fmaf(w00, t00[c], 0.0f);
        acc = __builtin_fmaf(w10, t10[c], acc);
        acc = __builtin_fmaf(w01, t01[c], acc);
        acc = __builtin_fmaf(w11, t11[c], acc);
        out[c] = acc;
    }
}
// tau at level of detail lambda_base (+ sampler bias + shader bias, then the sampler's clamp): the part of the sampling operation behind
// the LOD computation, shared by SampleBias (lambda_base from the derivatives) and SampleLevel (lambda_base = the explicit level)
SAH_DEV void sample_texture_lod(const float* luts, const sah_texture& T, const float uv[2], float lambda_base, float shader_bias, float out[4]) {
    const sah_sampler s = T.sampler;
    float lambda = lambda_base + (s.mip_lod_bias + shader_bias);
    lambda = __builtin_fminf(__builtin_fmaxf(lambda, s.min_lod), s.max_lod);
    const uint32_t filter = lambda <= 0.0f ? s.mag_filter : s.min_filter;
    const int q = (int)T.num_mips - 1;
    if (s.mipmap_mode == SAH_FILTER_NEAREST) {
        int level = 0;
        if (!(lambda <= 0.5f)) level = !(lambda < (float)q) ? q : min((int)__builtin_ceilf(lambda + 0.5f) - 1, q);
        sample_level(luts, T.mips[level], s, filter, uv[0], uv[1], out);
        return;
    }
    const float d = __builtin_fminf(__builtin_fmaxf(lambda, 0.0f), (float)q);
    const float hi_f = __builtin_floorf(d);
    const int hi = (int)hi_f, lo = min(hi + 1, q);
    const float delta = d - hi_f;
    float ta[4], tb[4];
    sample_level(luts, T.mips[hi], s, filter, uv[0], uv[1], ta);
    sample_level(luts, T.mips[lo], s, filter, uv[0], uv[1], tb);
    const float one_minus = 1.0f - delta;
    for (int c = 0; c < 4; c++) out[c] = one_minus * ta[c] + delta * tb[c];
}

SAH_DEV void sample_texture(const float* luts, const sah_texture& T, const float uv[2], const float ddx[2], const float ddy[2], float shader_bias, float out[4]) {
    const float W0 = (float)T.mips[0].width, H0 = (float)T.mips[0].height;
    const float mxx = ddx[0] * W0, mxy = ddx[1] * H0, myx = ddy[0] * W0, myy = ddy[1] * H0;
    const float rx = mxx * mxx + mxy * mxy, ry = myx * myx + myy * myy;
    const float rho2 = __builtin_fmaxf(rx, ry);
    const float lambda = rho2 > 0.0f ? 0.5f * (float)log2((double)rho2) : -__builtin_inff();
    const float A = T.sampler.max_anisotropy;
    if (!(A > 1.0f)) {
        sample_texture_lod(luts, T, uv, lambda, shader_bias, out);
        return;
    }
    // anisotropic footprint (sah_hip.h "anisotropy"): N taps along the major axis
    const float rmin2 = __builtin_fminf(rx, ry);
    float eta = 1.0f;
    if (rho2 > 0.0f) eta = rmin2 > 0.0f ? __builtin_fminf(__builtin_sqrtf(rho2 / rmin2), A) : A;
    const int N = (int)__builtin_ceilf(eta);
    const float lambda_a = rho2 > 0.0f ? lambda - (float)log2((double)eta) : lambda;
    if (N <= 1) {
        sample_texture_lod(luts, T, uv, lambda_a, shader_bias, out);
        return;
    }
    const float* d = rx > ry ? ddx : ddy;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 1; i <= N; i++) {
        const float a = (float)i / (float)(N + 1) - 0.5f;
        const float p[2] = {uv[0] + a * d[0], uv[1] + a * d[1]};
        float t[4];
        sample_texture_lod(luts, T, p, lambda_a, shader_bias, t);
        for (int c = 0; c < 4; c++) acc[c] = acc[c] + t[c];
    }
    for (int c = 0; c < 4; c++) out[c] = acc[c] / (float)N;
}

}  // namespace sah
