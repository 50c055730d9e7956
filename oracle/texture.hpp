// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// Material texture sampling of the G-buffer / shadow / RSM fragment stages:
//   RenderCore/shaders/materials/gltf_basic_pbr.slang:177-226   `textures[index].SampleBias(vertex.texcoord, mip_bias)`
//   RenderCore/model_import/gltf_model.cpp:229-279, 520-586     the samplers glTF materials bind
//   RenderCore/render/backend/render_backend.cpp:1129-1134      the default sampler (NEAREST / REPEAT)
// Vulkan 1.4 "Texel Filtering" / "Scale Factor Operation, LOD Operation and Image Level(s) Selection" leave the LOD arithmetic, the
// derivative quads and anisotropy to the implementation; include/sah_hip.h (sah_texture) lists what is fixed here.
#pragma once
#include <cmath>
#include <cstdint>

#include "../include/sah_hip.h"
#include "codec.hpp"

namespace orc {

inline int wrap_texel(int i, int n, uint32_t mode) {
    if (mode == SAH_ADDRESS_CLAMP_TO_EDGE) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    if (mode == SAH_ADDRESS_MIRRORED_REPEAT) {
        int m = i % (2 * n);
        if (m < 0) m += 2 * n;
        return m < n ? m : 2 * n - 1 - m;
    }
    int m = i % n;  // REPEAT
    return m < 0 ? m + n : m;
}

inline void fetch_rgba8(const sah_plane& p, int x, int y, float out[4]) {
    const uint8_t* t = (const uint8_t*)p.ptr + (size_t)y * p.row_pitch_bytes + (size_t)x * 4;
    const bool srgb = p.format == SAH_FORMAT_R8G8B8A8_SRGB;
    for (int c = 0; c < 3; c++) out[c] = srgb ? srgb8_to_linear(t[c]) : unorm8_to_float(t[c]);
    out[3] = unorm8_to_float(t[3]);
}

// tau of one level at (u, v)
inline void sample_level(const sah_plane& p, const sah_sampler& s, uint32_t filter, float u, float v, float out[4]) {
    const int w = (int)p.width, h = (int)p.height;
    auto toi = [](float f) { return f < -1e9f ? -1000000000 : (f > 1e9f ? 1000000000 : (int)f); };
    if (filter == SAH_FILTER_NEAREST) {
        const float px = u * (float)w, py = v * (float)h;
        if (std::isnan(px) || std::isnan(py)) { for (int c = 0; c < 4; c++) out[c] = NAN; return; }
        fetch_rgba8(p, wrap_texel(toi(std::floor(px)), w, s.address_u), wrap_texel(toi(std::floor(py)), h, s.address_v), out);
        return;
    }
    const float px = u * (float)w - 0.5f, py = v * (float)h - 0.5f;
    if (std::isnan(px) || std::isnan(py)) { for (int c = 0; c < 4; c++) out[c] = NAN; return; }
    const float fx0 = std::floor(px), fy0 = std::floor(py);
    const float fx = px - fx0, fy = py - fy0;
    const int x0 = toi(fx0), y0 = toi(fy0);
    const int xa = wrap_texel(x0, w, s.address_u), xb = wrap_texel(x0 + 1, w, s.address_u);
    const int ya = wrap_texel(y0, h, s.address_v), yb = wrap_texel(y0 + 1, h, s.address_v);
    float t00[4], t10[4], t01[4], t11[4];
    fetch_rgba8(p, xa, ya, t00); fetch_rgba8(p, xb, ya, t10); fetch_rgba8(p, xa, yb, t01); fetch_rgba8(p, xb, yb, t11);
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    const float w00 = wx0 * wy0, w10 = fx * wy0, w01 = wx0 * fy, w11 = fx * fy;
    for (int c = 0; c < 4; c++) {
        float a = std::fmaf(w00, t00[c], 0.0f);
        a = std::fmaf(w10, t10[c], a);
        a = std::fmaf(w01, t01[c], a);
        a = std::fmaf(w11, t11[c], a);
        out[c] = a;
    }
}

// The sampling operation behind the LOD computation: lambda_base is what the derivatives give (SampleBias) or the explicit level
// (SampleLevel, gltf_basic_pbr.slang:309: the any-hit stage of the ray tracer)
inline void sample_texture_lod(const sah_texture& T, const float uv[2], float lambda_base, float shader_bias, float out[4]) {
    float lambda = lambda_base + (T.sampler.mip_lod_bias + shader_bias);
    lambda = std::fmin(std::fmax(lambda, T.sampler.min_lod), T.sampler.max_lod);
    const uint32_t filter = lambda <= 0.0f ? T.sampler.mag_filter : T.sampler.min_filter;
    const int q = (int)T.num_mips - 1;
    if (T.sampler.mipmap_mode == SAH_FILTER_NEAREST) {
        int level = 0;
        if (!(lambda <= 0.5f)) level = !(lambda < (float)q) ? q : std::min((int)std::ceil(lambda + 0.5f) - 1, q);  // (a NaN lambda -> q)
        sample_level(T.mips[level], T.sampler, filter, uv[0], uv[1], out);
        return;
    }
    const float d = std::fmin(std::fmax(lambda, 0.0f), (float)q);
    const float hi_f = std::floor(d);
    const int hi = (int)hi_f, lo = std::min(hi + 1, q);
    const float delta = d - hi_f;
    float a[4], b[4];
    sample_level(T.mips[hi], T.sampler, filter, uv[0], uv[1], a);
    sample_level(T.mips[lo], T.sampler, filter, uv[0], uv[1], b);
    const float one_minus = 1.0f - delta;
    for (int c = 0; c < 4; c++) out[c] = one_minus * a[c] + delta * b[c];
}

// SampleBias: uv and its quad derivatives in, four fp32 channels out
inline void sample_texture(const sah_texture& T, const float uv[2], const float ddx[2], const float ddy[2], float shader_bias, float out[4]) {
    const float W0 = (float)T.mips[0].width, H0 = (float)T.mips[0].height;
    const float mxx = ddx[0] * W0, mxy = ddx[1] * H0, myx = ddy[0] * W0, myy = ddy[1] * H0;
    const float rx = mxx * mxx + mxy * mxy, ry = myx * myx + myy * myy;
    const float rho2 = std::fmax(rx, ry);
    const float lambda = rho2 > 0.0f ? 0.5f * (float)std::log2((double)rho2) : -INFINITY;
    const float A = T.sampler.max_anisotropy;
    if (!(A > 1.0f)) {
        sample_texture_lod(T, uv, lambda, shader_bias, out);
        return;
    }
    // Vulkan specification, "Texel Anisotropic Filtering" (the example formulation), as sah_hip.h fixes its arithmetic; the sampler
    // asks for it in gltf_model.cpp:581-584
    const float rmin2 = std::fmin(rx, ry);
    float eta = 1.0f;
    if (rho2 > 0.0f) eta = rmin2 > 0.0f ? std::fmin(std::sqrt(rho2 / rmin2), A) : A;
    const int N = (int)std::ceil(eta);
    const float lambda_a = rho2 > 0.0f ? lambda - (float)std::log2((double)eta) : lambda;
    if (N <= 1) {
        sample_texture_lod(T, uv, lambda_a, shader_bias, out);
        return;
    }
    const float* d = rx > ry ? ddx : ddy;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 1; i <= N; i++) {
        const float a = (float)i / (float)(N + 1) - 0.5f;
        const float p[2] = {uv[0] + a * d[0], uv[1] + a * d[1]};
        float t[4];
        sample_texture_lod(T, p, lambda_a, shader_bias, t);
        for (int c = 0; c < 4; c++) acc[c] = acc[c] + t[c];
    }
    for (int c = 0; c < 4; c++) out[c] = acc[c] / (float)N;
}

inline bool texture_ok(const sah_texture& T) {
    if (T.num_mips < 1 || T.num_mips > SAH_MAX_TEXTURE_MIPS) return false;
    if (T.sampler.mag_filter > 1 || T.sampler.min_filter > 1 || T.sampler.mipmap_mode > 1 || T.sampler.address_u > 2 || T.sampler.address_v > 2 ||
        T.sampler.max_anisotropy > 16.0f)
        return false;
    for (uint32_t i = 0; i < T.num_mips; i++) {
        const sah_plane& p = T.mips[i];
        if (!p.ptr || p.width == 0 || p.height == 0 || p.width > 16384 || p.height > 16384 || p.format != T.mips[0].format) return false;
        if (p.format != SAH_FORMAT_R8G8B8A8_UNORM && p.format != SAH_FORMAT_R8G8B8A8_SRGB) return false;
    }
    return true;
}

}  // namespace orc
