// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// Restates RenderCore/shaders/common/brdf.glsl:29-121 (T = F, fp32 "mediump") and
// RenderCore/shaders/common/brdf.slangi:22-114 (T = H, real fp16) — the two files are the same
// maths in two precisions, so one template follows both line by line.
#pragma once
#include "math.hpp"

namespace orc {

template <class T> struct Surface {
    V3<T> base_color;
    V3<T> normal;
    T metalness;
    T roughness;
};

// brdf.glsl:3-5 `#define PI 3.1415927`; brdf.slangi:4-6 `#define PI 3.1415927h`
template <class T> static inline T brdf_pi() { return T::lit(3.1415927); }

// brdf.glsl:29-32 / brdf.slangi:22-25 — note `roughness` is used un-squared.
template <class T> static inline T D_GGX(T NoH, T roughness) {
    T k = roughness / (T::lit(1.0) - NoH * NoH + roughness * roughness);
    return k * k * (T::lit(1.0) / brdf_pi<T>());
}

// brdf.glsl:34 / brdf.slangi:27
template <class T> static inline V3<T> F_Schlick(T u, V3<T> f0, T f90) {
    T p = npow5(nclamp(T::lit(1.0) - u, T::lit(0.0), T::lit(1.0)));
    return {f0.x + (f90 - f0.x) * p, f0.y + (f90 - f0.y) * p, f0.z + (f90 - f0.z) * p};
}

// brdf.glsl:36-42 / brdf.slangi:29-35
template <class T> static inline T V_SmithGGXCorrelated(T NoV, T NoL, T a) {
    T a2 = a * a;
    T GGXL = NoV * nsqrt((-NoL * a2 + NoL) * NoL + a2);
    T GGXV = NoL * nsqrt((-NoV * a2 + NoV) * NoV + a2);
    return T::lit(0.5) / (GGXV + GGXL);
}

// brdf.glsl:46-52 / brdf.slangi:39-45
template <class T> static inline V3<T> Fd_Burley(T NoV, T NoL, T LoH, T roughness) {
    T f90 = T::lit(0.5) + T::lit(2.0) * roughness * LoH * LoH;
    V3<T> one(T::lit(1.0));
    V3<T> lightScatter = F_Schlick(NoL, one, f90);
    V3<T> viewScatter = F_Schlick(NoV, one, f90);
    return lightScatter * viewScatter * (T::lit(1.0) / brdf_pi<T>());
}

// brdf.glsl:65-89 / brdf.slangi:58-82
template <class T> static inline V3<T> Fd(const Surface<T>& s, V3<T> l, V3<T> v) {
    const T dielectric_f0 = T::lit(0.04);
    const V3<T> diffuse_color = s.base_color * (T::lit(1.0) - dielectric_f0) * (T::lit(1.0) - s.metalness);
    const V3<T> h = normalize(v + l);
    T NoV = dot(s.normal, v) + T::lit(1e-5);
    T NoL = dot(s.normal, l);
    if (NoL.v <= 0.f) return V3<T>(T::lit(0.0));
    NoV = nabs(NoV);
    NoL = nclamp(NoL, T::lit(0.0), T::lit(1.0));
    const T LoH = nclamp(dot(l, h), T::lit(0.0), T::lit(1.0));
    return diffuse_color * Fd_Burley(NoV, NoL, LoH, s.roughness);
}

// brdf.glsl:91-116 / brdf.slangi:84-109
template <class T> static inline V3<T> Fr(const Surface<T>& s, V3<T> l, V3<T> v) {
    const T dielectric_f0 = T::lit(0.04);
    const V3<T> f0 = mix(V3<T>(dielectric_f0), s.base_color, s.metalness);
    const V3<T> h = normalize(v + l);
    T NoV = dot(s.normal, v) + T::lit(1e-5);
    T NoL = dot(s.normal, l);
    const T NoH = nclamp(dot(s.normal, h), T::lit(0.0), T::lit(1.0));
    const T VoH = nclamp(dot(v, h), T::lit(0.0), T::lit(1.0));
    if (NoL.v <= 0.f) return V3<T>(T::lit(0.0));
    NoV = nabs(NoV);
    NoL = nclamp(NoL, T::lit(0.0), T::lit(1.0));
    const T D = D_GGX(NoH, s.roughness);
    const V3<T> Fv = F_Schlick(VoH, f0, T::lit(1.0));
    const T V = V_SmithGGXCorrelated(NoV, NoL, s.roughness);
    return (D * V) * Fv;
}

// brdf.glsl:118-121 / brdf.slangi:111-114
template <class T> static inline V3<T> brdf(const Surface<T>& s, V3<T> l, V3<T> v) { return Fd(s, l, v) + Fr(s, l, v); }

}  // namespace orc
