// ORACLE — TEST INFRASTRUCTURE ONLY.  Scalar arithmetic model used by the CPU restatement.
// PARITY UNPINNED.
//
// Numerics contract (DESIGN.md "Numerics"):
//   * GLSL passes (mediump == RelaxedPrecision) evaluate in IEEE fp32, every operator individually
//     rounded (no FMA contraction), sqrt and divide correctly rounded.
//   * Slang passes evaluate `half` expressions in fp16: every operator rounds its exact result to fp16
//     (float op + round is exact for + - * / sqrt because 24 >= 2*11+2).
//   * normalize(v) = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1 / sqrt(x)  (Vulkan precision table:
//     normalize inherits from x * inversesqrt(dot(x,x))).
//   * dot() sums left to right; mix(x,y,a) = x*(1-a) + y*a; pow(x,5) is the correctly rounded fifth power.
#pragma once
#include <cmath>

#include "codec.hpp"

namespace orc {

// ---- number models -------------------------------------------------------------------------------
struct F {  // fp32
    float v;
    F() : v(0.f) {}
    F(float x) : v(x) {}
    static F lit(double x) { return F((float)x); }
    static F from_f32(float x) { return F(x); }
};
static inline F operator+(F a, F b) { return F(a.v + b.v); }
static inline F operator-(F a, F b) { return F(a.v - b.v); }
static inline F operator*(F a, F b) { return F(a.v * b.v); }
static inline F operator/(F a, F b) { return F(a.v / b.v); }
static inline F operator-(F a) { return F(-a.v); }
static inline F nsqrt(F a) { return F(std::sqrt(a.v)); }
static inline F npow5(F a) {
    double d = (double)a.v;
    return F((float)(d * d * d * d * d));
}

struct H {  // fp16 value held in an fp32 that is always fp16-representable
    float v;
    H() : v(0.f) {}
    explicit H(float x) : v(rh(x)) {}
    static H lit(double x) { return H((float)x); }  // literal with `h` suffix: nearest fp16 (via fp32)
    static H from_f32(float x) { return H(x); }
    static H raw(float already_half) { H h; h.v = already_half; return h; }
};
static inline H operator+(H a, H b) { return H(a.v + b.v); }
static inline H operator-(H a, H b) { return H(a.v - b.v); }
static inline H operator*(H a, H b) { return H(a.v * b.v); }
static inline H operator/(H a, H b) { return H(a.v / b.v); }
static inline H operator-(H a) { return H::raw(-a.v); }
static inline H nsqrt(H a) { return H(std::sqrt(a.v)); }
static inline H npow5(H a) {
    double d = (double)a.v;
    return H((float)(d * d * d * d * d));
}

template <class T> static inline bool lt(T a, T b) { return a.v < b.v; }
template <class T> static inline T nabs(T a) { T r = a; r.v = std::fabs(a.v); return r; }
// GLSL clamp = min(max(x, lo), hi); NaN handling: max(NaN, lo) -> lo on this model (fmax semantics).
template <class T> static inline T nmax(T a, T b) { T r = a; r.v = (a.v < b.v || std::isnan(a.v)) ? b.v : a.v; return r; }
template <class T> static inline T nmin(T a, T b) { T r = a; r.v = (b.v < a.v || std::isnan(a.v)) ? b.v : a.v; return r; }
template <class T> static inline T nclamp(T x, T lo, T hi) { return nmin(nmax(x, lo), hi); }

// ---- small vectors -------------------------------------------------------------------------------
template <class T> struct V3 {
    T x, y, z;
    V3() {}
    V3(T a, T b, T c) : x(a), y(b), z(c) {}
    explicit V3(T a) : x(a), y(a), z(a) {}
};
template <class T> static inline V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> static inline V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> static inline V3<T> operator*(V3<T> a, V3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class T> static inline V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> static inline V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> static inline V3<T> operator/(V3<T> a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <class T> static inline V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <class T> static inline T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> static inline T inversesqrt(T x) { return T::lit(1.0) / nsqrt(x); }
template <class T> static inline V3<T> normalize(V3<T> a) { return a * inversesqrt(dot(a, a)); }
template <class T> static inline T length(V3<T> a) { return nsqrt(dot(a, a)); }
template <class T> static inline V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
template <class T> static inline T mix(T x, T y, T a) { return x * (T::lit(1.0) - a) + y * a; }
template <class T> static inline V3<T> mix(V3<T> x, V3<T> y, T a) { return {mix(x.x, y.x, a), mix(x.y, y.y, a), mix(x.z, y.z, a)}; }
template <class T> static inline bool any_nan(V3<T> a) { return std::isnan(a.x.v) || std::isnan(a.y.v) || std::isnan(a.z.v); }

using F3 = V3<F>;
using H3 = V3<H>;
static inline H3 to_h(F3 a) { return {H(a.x.v), H(a.y.v), H(a.z.v)}; }
static inline F3 to_f(H3 a) { return {F(a.x.v), F(a.y.v), F(a.z.v)}; }

struct F4 {
    F x, y, z, w;
};

// column-major 4x4 (glm / GLSL layout): m[col*4 + row]
struct M4 {
    float m[16];
};
// GLSL `M * v` / Slang `mul(M, v)`: r_i = ((M[0][i]*v.x + M[1][i]*v.y) + M[2][i]*v.z) + M[3][i]*v.w
static inline F4 mul(const M4& M, F4 v) {
    F4 r;
    F* o[4] = {&r.x, &r.y, &r.z, &r.w};
    for (int i = 0; i < 4; i++) {
        *o[i] = F(M.m[0 + i]) * v.x + F(M.m[4 + i]) * v.y + F(M.m[8 + i]) * v.z + F(M.m[12 + i]) * v.w;
    }
    return r;
}

}  // namespace orc
