// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's fixed-function format
// semantics (Vulkan 1.4 spec behaviour the reference relies on).
// PARITY UNPINNED: the reference ships no tests / golden vectors for this path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// fp32 -> fp16 bits, round-to-nearest-even, overflow -> inf, denormals preserved, NaN stays NaN.
static inline uint16_t f32_to_f16(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) {  // inf / nan
        if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((ax >> 13) & 0x3ffu));
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax >= 0x477ff000u) {  // >= 65520 rounds to inf
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x38800000u) {  // < 2^-14: half denormal (or zero)
        if (ax < 0x33000000u) {  // < 2^-25: rounds to zero (2^-25 exactly ties to even = 0)
            return (uint16_t)sign;
        }
        uint32_t e = ax >> 23;  // biased exponent, 102..112
        uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126 - e;  // 14..24: result = m >> shift with rounding
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    uint32_t e = (ax >> 23) - 112;  // 1..30
    uint32_t m = ax & 0x7fffffu;
    uint32_t q = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;  // carry may bump exponent: still correct
    return (uint16_t)(sign | q);
}

static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        // denormal: m * 2^-24
        float v = (float)m * 5.9604644775390625e-8f;
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112) << 23) | (m << 13));
}

// Round an fp32 value to the nearest fp16-representable fp32 value.
static inline float rh(float f) { return f16_to_f32(f32_to_f16(f)); }

// sRGB8 -> linear fp32, 256-entry table computed in double and rounded to fp32 (Appendix A).
struct SrgbTable {
    float lin[256];
    SrgbTable() {
        for (int i = 0; i < 256; i++) {
            double c = (double)i / 255.0;
            double l = (c <= 0.04045) ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
            lin[i] = (float)l;
        }
    }
};
static inline const SrgbTable& srgb_table() { static SrgbTable t; return t; }
static inline float srgb8_to_linear(uint8_t v) { return srgb_table().lin[v]; }
static inline float unorm8_to_float(uint8_t v) { return (float)v / 255.0f; }
static inline float unorm16_to_float(uint16_t v) { return (float)v / 65535.0f; }

// linear -> sRGB OETF in fp32 semantics: correctly rounded result of the real-valued formula.
static inline float linear_to_srgb_f(float c) {
    if (std::isnan(c)) return 0.0f;
    if (c <= 0.0f) return 0.0f;
    if (c >= 1.0f) return 1.0f;
    double d = (double)c;
    double s = (d <= 0.0031308) ? 12.92 * d : 1.055 * std::pow(d, 1.0 / 2.4) - 0.055;
    return (float)s;
}
// float [0,1] -> UNORM8, round to nearest (Vulkan: implementation may round either way at exact .5;
// we use floor(x*255 + 0.5) evaluated in fp32).
static inline uint8_t float_to_unorm8(float c) {
    if (std::isnan(c)) return 0;
    if (c <= 0.0f) return 0;
    if (c >= 1.0f) return 255;
    float s = c * 255.0f + 0.5f;
    return (uint8_t)s;
}

// B10G11R11_UFLOAT_PACK32 decode: bits 0-10 R (5e6m), 11-21 G (5e6m), 22-31 B (5e5m).
static inline float uf11_to_f32(uint32_t v) {
    uint32_t e = (v >> 6) & 0x1fu, m = v & 0x3fu;
    if (e == 0) return (float)m * (1.0f / 64.0f) * 6.103515625e-5f;  // m/64 * 2^-14
    if (e == 31) return m ? u2f(0x7fc00000u) : u2f(0x7f800000u);
    return u2f(((e + 112) << 23) | (m << 17));
}
static inline float uf10_to_f32(uint32_t v) {
    uint32_t e = (v >> 5) & 0x1fu, m = v & 0x1fu;
    if (e == 0) return (float)m * (1.0f / 32.0f) * 6.103515625e-5f;
    if (e == 31) return m ? u2f(0x7fc00000u) : u2f(0x7f800000u);
    return u2f(((e + 112) << 23) | (m << 18));
}
// fp32 -> UF11 / UF10, round toward zero (documented choice, Appendix A), negatives clamp to 0.
static inline uint32_t f32_to_uf11(float f) {
    uint32_t x = f2u(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return 0x7c0u | 0x20u;  // NaN
    if (x & 0x80000000u) return 0;                                // negative (incl. -inf) -> 0
    if (x >= 0x7f800000u) return 0x7c0u;                          // +inf
    if (x >= 0x477e0000u) return 0x7bfu;                          // >= 65024: clamp to max finite (RTZ)
    if (x < 0x38800000u) {                                        // denormal range
        if (x < 0x35800000u) return 0;                            // < 2^-20
        uint32_t e = x >> 23;
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        uint32_t shift = 130 - e;  // value = m * 2^(e-150); units of 2^-20 => m >> (130 - e)
        return m >> shift;
    }
    return (((x >> 23) - 112) << 6) | ((x & 0x7fffffu) >> 17);
}
static inline uint32_t f32_to_uf10(float f) {
    uint32_t x = f2u(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return 0x3e0u | 0x10u;
    if (x & 0x80000000u) return 0;
    if (x >= 0x7f800000u) return 0x3e0u;
    if (x >= 0x477c0000u) return 0x3dfu;  // >= 64512
    if (x < 0x38800000u) {
        if (x < 0x36000000u) return 0;  // < 2^-19
        uint32_t e = x >> 23;
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        uint32_t shift = 131 - e;
        return m >> shift;
    }
    return (((x >> 23) - 112) << 5) | ((x & 0x7fffffu) >> 18);
}
static inline void r11g11b10_decode(uint32_t p, float out[3]) {
    out[0] = uf11_to_f32(p & 0x7ffu);
    out[1] = uf11_to_f32((p >> 11) & 0x7ffu);
    out[2] = uf10_to_f32((p >> 22) & 0x3ffu);
}
static inline uint32_t r11g11b10_encode(const float in[3]) {
    return f32_to_uf11(in[0]) | (f32_to_uf11(in[1]) << 11) | (f32_to_uf10(in[2]) << 22);
}

}  // namespace orc
