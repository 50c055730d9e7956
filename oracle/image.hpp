// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// Linear row-major images + the Vulkan sampler semantics the reference's passes rely on
// (SURVEY.md Appendix A "Sampling conventions"): unnormalised coordinate = uv*size - 0.5, i0 = floor,
// f = p - i0, weights in full fp32 (no 8-bit sub-texel quantisation), address mode applied per tap.
#pragma once
#include <cmath>
#include <cstdint>

#include "codec.hpp"

namespace orc {

// VkFormat numeric values (Vulkan 1.4 core), passed through unchanged.
enum : uint32_t {
    FMT_R8_UNORM = 9,
    FMT_R8G8B8A8_UNORM = 37,
    FMT_R8G8B8A8_SRGB = 43,
    FMT_R16_SFLOAT = 76,
    FMT_R16G16_SFLOAT = 83,
    FMT_R16G16B16A16_SFLOAT = 97,
    FMT_R32_SFLOAT = 100,
    FMT_B10G11R11_UFLOAT = 122,
    FMT_D16_UNORM = 124,
    FMT_D32_SFLOAT = 126,
};

enum AddressMode { ADDR_REPEAT = 0, ADDR_CLAMP_TO_EDGE = 2, ADDR_CLAMP_TO_BORDER = 3 };

struct Image {
    const uint8_t* ptr;
    uint32_t width, height, depth;  // depth = array layers for 2D arrays
    uint32_t row_pitch, slice_pitch;
    uint32_t format;
};

struct Texel {
    float c[4];
};

static inline uint32_t format_bpp(uint32_t f) {
    switch (f) {
        case FMT_R8_UNORM: return 1;
        case FMT_R16_SFLOAT: case FMT_D16_UNORM: return 2;
        case FMT_R8G8B8A8_UNORM: case FMT_R8G8B8A8_SRGB: case FMT_R16G16_SFLOAT: case FMT_R32_SFLOAT:
        case FMT_B10G11R11_UFLOAT: case FMT_D32_SFLOAT: return 4;
        case FMT_R16G16B16A16_SFLOAT: return 8;
    }
    return 0;
}

// Integer texel load (texelFetch / Texture[pixel]); coordinates must be in range.
static inline Texel load_texel(const Image& im, int x, int y, int z) {
    const uint8_t* p = im.ptr + (size_t)z * im.slice_pitch + (size_t)y * im.row_pitch + (size_t)x * format_bpp(im.format);
    Texel t = {{0.f, 0.f, 0.f, 1.f}};
    switch (im.format) {
        case FMT_R8_UNORM: t.c[0] = unorm8_to_float(p[0]); break;
        case FMT_R8G8B8A8_UNORM:
            for (int i = 0; i < 4; i++) t.c[i] = unorm8_to_float(p[i]);
            break;
        case FMT_R8G8B8A8_SRGB:
            for (int i = 0; i < 3; i++) t.c[i] = srgb8_to_linear(p[i]);
            t.c[3] = unorm8_to_float(p[3]);
            break;
        case FMT_R16_SFLOAT: { uint16_t h; std::memcpy(&h, p, 2); t.c[0] = f16_to_f32(h); } break;
        case FMT_R16G16_SFLOAT: { uint16_t h[2]; std::memcpy(h, p, 4); t.c[0] = f16_to_f32(h[0]); t.c[1] = f16_to_f32(h[1]); } break;
        case FMT_R16G16B16A16_SFLOAT: { uint16_t h[4]; std::memcpy(h, p, 8); for (int i = 0; i < 4; i++) t.c[i] = f16_to_f32(h[i]); } break;
        case FMT_R32_SFLOAT: case FMT_D32_SFLOAT: { float f; std::memcpy(&f, p, 4); t.c[0] = f; } break;
        case FMT_D16_UNORM: { uint16_t h; std::memcpy(&h, p, 2); t.c[0] = unorm16_to_float(h); } break;
        case FMT_B10G11R11_UFLOAT: { uint32_t u; std::memcpy(&u, p, 4); r11g11b10_decode(u, t.c); } break;
    }
    return t;
}

// Wrap one tap index; returns false if the tap reads the (transparent black) border.
static inline bool wrap_index(int& i, int size, AddressMode mode) {
    if (mode == ADDR_REPEAT) {
        i %= size;
        if (i < 0) i += size;
        return true;
    }
    if (mode == ADDR_CLAMP_TO_EDGE) {
        if (i < 0) i = 0;
        if (i > size - 1) i = size - 1;
        return true;
    }
    return i >= 0 && i < size;
}

static inline Texel load_texel_addr(const Image& im, int x, int y, int z, AddressMode mode, bool wrap_z) {
    bool ok = wrap_index(x, (int)im.width, mode) & wrap_index(y, (int)im.height, mode);
    if (wrap_z) ok = ok & wrap_index(z, (int)im.depth, mode);
    if (!ok) return Texel{{0.f, 0.f, 0.f, 0.f}};
    return load_texel(im, x, y, z);
}

// Linear filtering follows the Vulkan spec's weighted-sum formula (Vulkan 1.4 "Texel Filtering"):
//   tau_2D = (1-a)(1-b) t_ij + a(1-b) t_i1j + (1-a) b t_ij1 + a b t_i1j1
//   tau_3D = the same with (1-g) / g on the third axis, taps ordered x fastest, then y, then z.
// Weights are fp32 products formed left to right; the sum is an fma chain in the listed tap order starting from
// +0.0f (acc = fma(w_k, t_k, acc)).  DESIGN.md "Sampling".
static inline Texel sample_bilinear(const Image& im, float u, float v, int layer, AddressMode mode) {
    float px = u * (float)im.width - 0.5f;
    float py = v * (float)im.height - 0.5f;
    Texel r;
    if (std::isnan(px) || std::isnan(py)) {
        for (int i = 0; i < 4; i++) r.c[i] = NAN;
        return r;
    }
    float fx0 = std::floor(px), fy0 = std::floor(py);
    float fx = px - fx0, fy = py - fy0;
    auto toi = [](float f) { return f < -1e9f ? -1000000000 : (f > 1e9f ? 1000000000 : (int)f); };
    int x0 = toi(fx0), y0 = toi(fy0);
    Texel t00 = load_texel_addr(im, x0, y0, layer, mode, false);
    Texel t10 = load_texel_addr(im, x0 + 1, y0, layer, mode, false);
    Texel t01 = load_texel_addr(im, x0, y0 + 1, layer, mode, false);
    Texel t11 = load_texel_addr(im, x0 + 1, y0 + 1, layer, mode, false);
    float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    float w00 = wx0 * wy0, w10 = fx * wy0, w01 = wx0 * fy, w11 = fx * fy;
    for (int i = 0; i < 4; i++) {
        float a = std::fmaf(w00, t00.c[i], 0.0f);
        a = std::fmaf(w10, t10.c[i], a);
        a = std::fmaf(w01, t01.c[i], a);
        a = std::fmaf(w11, t11.c[i], a);
        r.c[i] = a;
    }
    return r;
}

// Trilinear sample of a 3D image at normalised (u, v, w): the 2D rule on three axes.
static inline Texel sample_trilinear(const Image& im, float u, float v, float w, AddressMode mode) {
    float px = u * (float)im.width - 0.5f;
    float py = v * (float)im.height - 0.5f;
    float pz = w * (float)im.depth - 0.5f;
    float fx0 = std::floor(px), fy0 = std::floor(py), fz0 = std::floor(pz);
    float fx = px - fx0, fy = py - fy0, fz = pz - fz0;
    float wx0 = 1.0f - fx, wy0 = 1.0f - fy, wz0 = 1.0f - fz;
    Texel r;
    Texel t[8];
    if (std::isnan(px) || std::isnan(py) || std::isnan(pz)) {
        // NaN coordinates: every tap is treated as out of range (border); weights are NaN -> NaN result
        for (int i = 0; i < 4; i++) r.c[i] = NAN;
        return r;
    }
    // clamp the float index before the int conversion so that huge coordinates stay "out of range"
    auto toi = [](float f) { return f < -1e9f ? -1000000000 : (f > 1e9f ? 1000000000 : (int)f); };
    int x0 = toi(fx0), y0 = toi(fy0), z0 = toi(fz0);
    for (int k = 0; k < 8; k++) t[k] = load_texel_addr(im, x0 + (k & 1), y0 + ((k >> 1) & 1), z0 + (k >> 2), mode, true);
    const float wxy[4] = {wx0 * wy0, fx * wy0, wx0 * fy, fx * fy};
    float wt[8];
    for (int k = 0; k < 8; k++) wt[k] = wxy[k & 3] * ((k >> 2) ? fz : wz0);
    for (int i = 0; i < 4; i++) {
        float a = 0.0f;
        for (int k = 0; k < 8; k++) a = std::fmaf(wt[k], t[k].c[i], a);
        r.c[i] = a;
    }
    return r;
}

// Depth-compare sampler (compare op LESS): compare each tap, then bilinear-blend the 0/1 results.
static inline float sample_shadow_pcf(const Image& im, float u, float v, int layer, float ref, AddressMode mode) {
    float px = u * (float)im.width - 0.5f;
    float py = v * (float)im.height - 0.5f;
    float fx0 = std::floor(px), fy0 = std::floor(py);
    float fx = px - fx0, fy = py - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    float c[4];
    for (int k = 0; k < 4; k++) {
        Texel t = load_texel_addr(im, x0 + (k & 1), y0 + (k >> 1), layer, mode, false);
        c[k] = (ref < t.c[0]) ? 1.0f : 0.0f;
    }
    float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    float a = std::fmaf(wx0 * wy0, c[0], 0.0f);
    a = std::fmaf(fx * wy0, c[1], a);
    a = std::fmaf(wx0 * fy, c[2], a);
    a = std::fmaf(fx * fy, c[3], a);
    return a;
}

}  // namespace orc
