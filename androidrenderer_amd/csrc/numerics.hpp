// Device-side arithmetic for the gfx950 kernels.
//
// Numerics contract (DESIGN.md "Numerics"): the library is compiled with -ffp-contract=off, so every fp32
// operator below is one individually rounded IEEE operation (v_add/v_mul/v_sub, correctly rounded v_div
// expansion and v_sqrt); fp16 ("Slang half") expressions round after every operator.  Nothing here may be
// built with -ffast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sah {

#define SAH_DEV __device__ __forceinline__

// ---- wave votes ---------------------------------------------------------------------------------
// A ballot of a COMPARE is the compare itself, written to a scalar register pair; a ballot of anything else (a conjunction, a bool that
// crossed a basic block) — and HIP's __any / __all, which take an int — first materialises the predicate in a vector register
// (v_cndmask 0 / 1) and compares it again: two VALU instructions per vote in kernels that are bound by VALU issue.  Hot loops therefore
// combine the masks of their elementary compares with scalar and / or (lanes(a) & lanes(b)) instead of voting on a && b.
using lanemask = unsigned long long;
SAH_DEV lanemask lanes(bool p) { return __builtin_amdgcn_ballot_w64(p); }
SAH_DEV bool wave_any(bool p) { return lanes(p) != 0ull; }
SAH_DEV bool wave_all(bool p) { return lanes(!p) == 0ull; }

// ---- fp16 storage <-> fp32 ----------------------------------------------------------------------
SAH_DEV float h2f(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
SAH_DEV uint16_t f2h(float f) {  // v_cvt_f16_f32, RNE; the asm keeps LLVM from fusing the producer into v_fma_mixlo_f16
    asm volatile("" : "+v"(f));
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// fma(w, (float)half, acc) with the fp16 operand read straight from the low / high half of a packed dword:
// v_fma_mix_f32 (VOP3P; op_sel_hi marks src1 as f16, op_sel picks its half).  One instruction per tap and channel —
// hipcc otherwise emits v_cvt_f32_f16 pairs feeding v_pk_fma_f32 (1.5 issue slots per fma).  Same arithmetic as
// __builtin_fmaf(w, (float)h, acc): the conversion is exact and the fma is IEEE fp32.
SAH_DEV float fma_mix_lo(float w, uint32_t packed, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(acc));
    return r;
}
SAH_DEV float fma_mix_hi(float w, uint32_t packed, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(acc));
    return r;
}

// ---- number models ------------------------------------------------------------------------------
// fp32: plain float.  fp16: value kept in a _Float16; + - * are native v_*_f16 (exact result, one rounding),
// divide and sqrt go through fp32 (correctly rounded there; the second rounding to fp16 is innocuous because
// 24 >= 2*11 + 2).
struct Fn {
    float v;
    SAH_DEV Fn() : v(0.f) {}
    SAH_DEV Fn(float x) : v(x) {}
    static SAH_DEV Fn lit(float x) { return Fn(x); }
};
SAH_DEV Fn operator+(Fn a, Fn b) { return Fn(a.v + b.v); }
SAH_DEV Fn operator-(Fn a, Fn b) { return Fn(a.v - b.v); }
SAH_DEV Fn operator*(Fn a, Fn b) { return Fn(a.v * b.v); }
SAH_DEV Fn operator/(Fn a, Fn b) { return Fn(a.v / b.v); }
SAH_DEV Fn operator-(Fn a) { return Fn(-a.v); }
SAH_DEV Fn nsqrt(Fn a) { return Fn(__builtin_sqrtf(a.v)); }
SAH_DEV Fn npow5(Fn a) {
    double d = (double)a.v;
    return Fn((float)(d * d * d * d * d));
}
SAH_DEV float tof(Fn a) { return a.v; }

// ---- correctly rounded sqrt / reciprocal / divide for operands of KNOWN range ----------------------------------------------
// hipcc's IEEE expansions carry range handling the fast kernel does not need once the operand range is established (by
// construction or by a compare that sends the pixel to the fix-up kernel): sqrt = 15 instructions (denormal pre/post scaling, a
// +-1 ulp probe with two compare/select pairs, inf/zero select), divide = 11 (two v_div_scale, v_div_fmas, v_div_fixup).  On
// MI355X v_{mul,add,fma}_f32 issue in 2 cycles, compares / selects / v_div_* in 4, v_rcp/v_rsq/v_sqrt in 8
// (profiles/r1_valu_issue_cost.txt), so the range handling is most of the cost (sqrt 66 -> 22 cycles, 1/x 34 -> 16, a/b 34 -> 22).
// The sequences below are the classical Newton / Markstein refinements; `tools/microbench/exact_math_check.hip` compares
// them with the IEEE operators for EVERY fp32 input of the stated domain (sqrt, reciprocal) and for 2^36 operand pairs
// including all-ones / all-zeros mantissa neighbourhoods (divide).  Outside the domain the results are unspecified.
constexpr float kNrLo = 0x1p-100f, kNrHi = 0x1p+100f;  // domain of sqrt_nr / rcp_nr (magnitudes)
constexpr float kDivLo = 0x1p-40f, kDivHi = 0x1p+40f;  // domain of div_nr (|a|, |b|; a may also be +0)

// x in [2^-100, 2^100] -> RN(sqrt(x))
SAH_DEV float sqrt_nr(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    const float g1 = __builtin_fmaf(g, r, g), h1 = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g1, g1, x);
    return __builtin_fmaf(d, h1, g1);
}
// as sqrt_nr, and +0 -> +0 (rsq(0) = +inf is clamped so that g = 0 * y stays 0)
SAH_DEV float sqrt_nr0(float x) {
    const float y = __builtin_fminf(__builtin_amdgcn_rsqf(x), 0x1p+100f);
    const float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    const float g1 = __builtin_fmaf(g, r, g), h1 = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g1, g1, x);
    return __builtin_fmaf(d, h1, g1);
}
// |x| in [2^-100, 2^100] -> RN(1 / x)
SAH_DEV float rcp_nr(float x) {
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float r = __builtin_fmaf(-x, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
// |a| in {+0} U [2^-40, 2^40], |b| in [2^-40, 2^40] -> RN(a / b): hipcc's own expansion minus v_div_scale / v_div_fmas scaling /
// v_div_fixup, none of which acts on such operands
SAH_DEV float div_nr(float a, float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}

// |b| in [2^-40, 2^40] -> RN(0.5 / b) = RN(1 / b) / 2: halving is exact (no result below 2^-41), and so is the order of the two roundings
SAH_DEV float half_over_nr(float b) { return 0.5f * rcp_nr(b); }
// a / b as div_nr(a, b) with the refined reciprocal y1 of b — div_nr's first three instructions — supplied by the caller (a divisor that is
// uniform over a loop's lanes or iterations): the same operations on the same operands, so the same bits
SAH_DEV float div_nr_y1(float a, float b, float y1) {
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}
SAH_DEV float div_nr_refine(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
}
// max(a, b) as one v_max_f32 for an operand a that is not a signalling NaN (a result of arithmetic) and a uniform b (a constant: scalar
// register): __builtin_fmaxf on a value from another basic block is preceded by a canonicalising v_max_f32 x, x
SAH_DEV float vmax_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(b), "v"(a));
    return r;
}

// float -> uint as the hardware converts (GLSL / Slang uint(float) on this path): truncation, negatives and NaN -> 0, 2^32 and above ->
// 0xffffffff.  Written in C++ ("f > 0 ? (f >= 2^32 ? ~0 : (uint32_t)f) : 0") it compiles to two nested exec-mask regions around the one
// instruction that does all of it.
SAH_DEV uint32_t cvt_u32_sat(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// An fp32 value is hidden from the optimiser before it is rounded to fp16.  Without this LLVM (a) narrows
// fptrunc(fdiv(fpext, fpext)) to a half fdiv whose v_rcp_f16 expansion is not correctly rounded, and (b) fuses
// fptrunc(fmul/fadd) into v_fma_mixlo_f16, i.e. ONE rounding of the exact result, where the contract (and the
// RGBA16F blend emulation) is "round to fp32, then to fp16".
SAH_DEV float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
SAH_DEV float rh(float f) { return (float)(_Float16)opaque(f); }  // round to the nearest fp16-representable value
struct Hn {
    _Float16 v;
    SAH_DEV Hn() : v((_Float16)0.f) {}
    SAH_DEV explicit Hn(float x) : v((_Float16)opaque(x)) {}
    static SAH_DEV Hn lit(float x) { Hn r; r.v = (_Float16)x; return r; }
    static SAH_DEV Hn raw(_Float16 h) { Hn r; r.v = h; return r; }
};
SAH_DEV Hn operator+(Hn a, Hn b) { return Hn::raw(a.v + b.v); }
SAH_DEV Hn operator-(Hn a, Hn b) { return Hn::raw(a.v - b.v); }
SAH_DEV Hn operator*(Hn a, Hn b) { return Hn::raw(a.v * b.v); }
// fp16 divide and root: the contract is the IEEE fp32 operator on the widened operands, rounded to fp16 (the second rounding is innocuous).
// For operands that ARE fp16 values the fp32 result only has to be accurate enough to round to the same fp16 — and the operand space is
// small enough to check every case: tools/microbench/half_math_check.hip compares, for ALL 2^32 (a, b) pairs / all 2^16 inputs (zeros,
// denormals, infinities, NaNs included), the fp16 bits of the sequences below with those of hipcc's IEEE expansions: 0 mismatches on
// gfx950 (profiles/r4_half_math_check.txt).  Divide: v_rcp_f32, one Newton step on the quotient, v_div_fixup_f32 for the special cases
// — 5 instructions, 19 issue cycles, for 11 / 34 (no v_div_scale / v_div_fmas: fp16 operands never need the scaling).  Root: v_sqrt_f32
// alone (1 ulp in fp32) — 1 instruction for 15; a NaN result may carry another sign / payload than the expansion's.
SAH_DEV float hdiv_f32(float a, float b) {
    const float y = __builtin_amdgcn_rcpf(b);
    const float q0 = a * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
    return __builtin_amdgcn_div_fixupf(q1, b, a);
}
SAH_DEV Hn operator/(Hn a, Hn b) { return Hn(hdiv_f32((float)a.v, (float)b.v)); }
SAH_DEV Hn operator-(Hn a) { return Hn::raw(-a.v); }
SAH_DEV Hn nsqrt(Hn a) { return Hn(__builtin_amdgcn_sqrtf((float)a.v)); }
// pow(x, 5) of an fp16 value in [0, 1] (F_Schlick's clamped argument, the only caller): the contract's fp64 product chain, rounded to fp32
// and then to fp16, has the fp16 bits of ((x * x) * (x * x)) * x in fp32 — x * x is exact (22 bits) — for every one of the 15 361 inputs
// (tools/microbench/half_math_check.hip, profiles/r4_half_math_check.txt): three fp32 multiplies for two conversions and four fp64 ones
SAH_DEV Hn npow5(Hn a) {
    const float x = (float)a.v, x2 = x * x;
    return Hn((x2 * x2) * x);
}
SAH_DEV float tof(Hn a) { return (float)a.v; }
// w * (float)h as one v_fma_mix_f32 with the addend -0 (x + -0 == x for every x, signed zeros included): the widening is exact and the
// product is rounded once, as v_cvt_f32_f16 + v_mul_f32 round it — one instruction for two.
SAH_DEV float mul_mix(float w, Hn h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(h.v), "s"(-0.0f));
    return r;
}

template <class T> SAH_DEV T nabs(T a) { T r = a; r.v = a.v < 0 ? -a.v : a.v; return r; }
template <> SAH_DEV Fn nabs<Fn>(Fn a) { return Fn(__builtin_fabsf(a.v)); }
template <> SAH_DEV Hn nabs<Hn>(Hn a) { return Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(__builtin_bit_cast(uint16_t, a.v) & 0x7fffu))); }
// max/min with fmax/fmin NaN semantics (the non-NaN operand wins), as the oracle defines clamp().
SAH_DEV Fn nmax(Fn a, Fn b) { return Fn(__builtin_fmaxf(a.v, b.v)); }
SAH_DEV Fn nmin(Fn a, Fn b) { return Fn(__builtin_fminf(a.v, b.v)); }
// v_max_f16 / v_min_f16: the operands are fp16 already, so this is the fp32 fmax / fmin of the (exactly) widened values
SAH_DEV Hn nmax(Hn a, Hn b) { return Hn::raw(__builtin_fmaxf16(a.v, b.v)); }
SAH_DEV Hn nmin(Hn a, Hn b) { return Hn::raw(__builtin_fminf16(a.v, b.v)); }
template <class T> SAH_DEV T nclamp(T x, T lo, T hi) { return nmin(nmax(x, lo), hi); }
SAH_DEV bool isnan_f(float x) { return x != x; }

// ---- small vectors ------------------------------------------------------------------------------
template <class T> struct V3 {
    T x, y, z;
    SAH_DEV V3() {}
    SAH_DEV V3(T a, T b, T c) : x(a), y(b), z(c) {}
    SAH_DEV explicit V3(T a) : x(a), y(a), z(a) {}
};
template <class T> SAH_DEV V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> SAH_DEV V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> SAH_DEV V3<T> operator*(V3<T> a, V3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class T> SAH_DEV V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> SAH_DEV V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> SAH_DEV V3<T> operator/(V3<T> a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <class T> SAH_DEV V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <class T> SAH_DEV T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> SAH_DEV T inversesqrt(T x) { return T::lit(1.0f) / nsqrt(x); }
template <class T> SAH_DEV V3<T> normalize(V3<T> a) { return a * inversesqrt(dot(a, a)); }
template <class T> SAH_DEV T length(V3<T> a) { return nsqrt(dot(a, a)); }
template <class T> SAH_DEV V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
template <class T> SAH_DEV T mix(T x, T y, T a) { return x * (T::lit(1.0f) - a) + y * a; }
template <class T> SAH_DEV V3<T> mix(V3<T> x, V3<T> y, T a) { return {mix(x.x, y.x, a), mix(x.y, y.y, a), mix(x.z, y.z, a)}; }
template <class T> SAH_DEV bool any_nan(V3<T> a) { return isnan_f(tof(a.x)) || isnan_f(tof(a.y)) || isnan_f(tof(a.z)); }

using F3 = V3<Fn>;
using H3 = V3<Hn>;
SAH_DEV H3 to_h(F3 a) { return {Hn(a.x.v), Hn(a.y.v), Hn(a.z.v)}; }
SAH_DEV F3 to_f(H3 a) { return {Fn((float)a.x.v), Fn((float)a.y.v), Fn((float)a.z.v)}; }

struct F4 {
    Fn x, y, z, w;
};
// column-major mat4 * vec4, summed left to right (GLSL `M * v`, Slang `mul(M, v)`)
SAH_DEV F4 mul44(const float* m, F4 v) {
    F4 r;
    r.x = Fn(m[0]) * v.x + Fn(m[4]) * v.y + Fn(m[8]) * v.z + Fn(m[12]) * v.w;
    r.y = Fn(m[1]) * v.x + Fn(m[5]) * v.y + Fn(m[9]) * v.z + Fn(m[13]) * v.w;
    r.z = Fn(m[2]) * v.x + Fn(m[6]) * v.y + Fn(m[10]) * v.z + Fn(m[14]) * v.w;
    r.w = Fn(m[3]) * v.x + Fn(m[7]) * v.y + Fn(m[11]) * v.z + Fn(m[15]) * v.w;
    return r;
}

// ---- BRDF: RenderCore/shaders/common/brdf.glsl:29-121 (T = Fn) / brdf.slangi:22-114 (T = Hn) -------
template <class T> struct Surface {
    V3<T> base_color;
    V3<T> normal;
    T metalness;
    T roughness;
};

template <class T> SAH_DEV T brdf_pi() { return T::lit(3.1415927f); }
// 1.0 / PI of the two BRDF flavours.  The fp16 one as a constant: Hn's operators are opaque to the optimiser, which would otherwise divide
// 1.0h by 3.140625h at run time in every D_GGX and Fd_Burley (the same value: IEEE fp32 quotient of the two fp16 numbers, rounded to fp16)
template <class T> SAH_DEV T inv_pi();
template <> SAH_DEV Fn inv_pi<Fn>() { return Fn(1.0f) / Fn(3.1415927f); }
template <> SAH_DEV Hn inv_pi<Hn>() {
    constexpr _Float16 v = (_Float16)(1.0f / (float)(_Float16)3.1415927f);
    return Hn::raw(v);
}

template <class T> SAH_DEV T D_GGX(T NoH, T roughness) {
    T k = roughness / (T::lit(1.0f) - NoH * NoH + roughness * roughness);
    return k * k * inv_pi<T>();
}
template <class T> SAH_DEV V3<T> F_Schlick(T u, V3<T> f0, T f90) {
    T p = npow5(nclamp(T::lit(1.0f) - u, T::lit(0.0f), T::lit(1.0f)));
    return {f0.x + (f90 - f0.x) * p, f0.y + (f90 - f0.y) * p, f0.z + (f90 - f0.z) * p};
}
template <class T> SAH_DEV T V_SmithGGXCorrelated(T NoV, T NoL, T a) {
    T a2 = a * a;
    T GGXL = NoV * nsqrt((-NoL * a2 + NoL) * NoL + a2);
    T GGXV = NoL * nsqrt((-NoV * a2 + NoV) * NoV + a2);
    return T::lit(0.5f) / (GGXV + GGXL);
}
template <class T> SAH_DEV V3<T> Fd_Burley(T NoV, T NoL, T LoH, T roughness) {
    T f90 = T::lit(0.5f) + T::lit(2.0f) * roughness * LoH * LoH;
    V3<T> one(T::lit(1.0f));
    V3<T> lightScatter = F_Schlick(NoL, one, f90);
    V3<T> viewScatter = F_Schlick(NoV, one, f90);
    return lightScatter * viewScatter * inv_pi<T>();
}
template <class T> SAH_DEV V3<T> Fd(const Surface<T>& s, V3<T> l, V3<T> v) {
    const T dielectric_f0 = T::lit(0.04f);
    const V3<T> diffuse_color = s.base_color * (T::lit(1.0f) - dielectric_f0) * (T::lit(1.0f) - s.metalness);
    const V3<T> h = normalize(v + l);
    T NoV = dot(s.normal, v) + T::lit(1e-5f);
    T NoL = dot(s.normal, l);
    if (tof(NoL) <= 0.f) return V3<T>(T::lit(0.0f));  // false for NaN: NaN flows on, as in the shader
    NoV = nabs(NoV);
    NoL = nclamp(NoL, T::lit(0.0f), T::lit(1.0f));
    const T LoH = nclamp(dot(l, h), T::lit(0.0f), T::lit(1.0f));
    return diffuse_color * Fd_Burley(NoV, NoL, LoH, s.roughness);
}
template <class T> SAH_DEV V3<T> Fr(const Surface<T>& s, V3<T> l, V3<T> v) {
    const T dielectric_f0 = T::lit(0.04f);
    const V3<T> f0 = mix(V3<T>(dielectric_f0), s.base_color, s.metalness);
    const V3<T> h = normalize(v + l);
    T NoV = dot(s.normal, v) + T::lit(1e-5f);
    T NoL = dot(s.normal, l);
    const T NoH = nclamp(dot(s.normal, h), T::lit(0.0f), T::lit(1.0f));
    const T VoH = nclamp(dot(v, h), T::lit(0.0f), T::lit(1.0f));
    if (tof(NoL) <= 0.f) return V3<T>(T::lit(0.0f));
    NoV = nabs(NoV);
    NoL = nclamp(NoL, T::lit(0.0f), T::lit(1.0f));
    const T D = D_GGX(NoH, s.roughness);
    const V3<T> Fv = F_Schlick(VoH, f0, T::lit(1.0f));
    const T V = V_SmithGGXCorrelated(NoV, NoL, s.roughness);
    return (D * V) * Fv;
}

// brdf() = Fd() + Fr() (brdf.glsl:118-121 / brdf.slangi:111-114) with the shared sub-expressions written once and the
// `NoL <= 0 -> 0` early-outs turned into one select (both halves return 0 together, and 0 + 0 == +0): the same operations on
// the same operands, so the same bits as Fd(s,l,v) + Fr(s,l,v), at two thirds of the work and without divergent returns.
template <class T> SAH_DEV V3<T> brdf_sl(const Surface<T>& s, V3<T> l, V3<T> v) {
    const T one = T::lit(1.0f), zero = T::lit(0.0f);
    const T dielectric_f0 = T::lit(0.04f);
    const V3<T> f0 = mix(V3<T>(dielectric_f0), s.base_color, s.metalness);
    const V3<T> diffuse_color = s.base_color * (one - dielectric_f0) * (one - s.metalness);
    const V3<T> h = normalize(v + l);
    T NoV = dot(s.normal, v) + T::lit(1e-5f);
    T NoL = dot(s.normal, l);
    const T NoH = nclamp(dot(s.normal, h), zero, one);
    const T VoH = nclamp(dot(v, h), zero, one);
    const bool dark = tof(NoL) <= 0.f;
    NoV = nabs(NoV);
    NoL = nclamp(NoL, zero, one);
    const T LoH = nclamp(dot(l, h), zero, one);
    const V3<T> fd = diffuse_color * Fd_Burley(NoV, NoL, LoH, s.roughness);
    const T D = D_GGX(NoH, s.roughness);
    const V3<T> Fv = F_Schlick(VoH, f0, one);
    const T Vis = V_SmithGGXCorrelated(NoV, NoL, s.roughness);
    const V3<T> fr = (D * Vis) * Fv;
    const V3<T> sum = fd + fr;
    return {dark ? zero : sum.x, dark ? zero : sum.y, dark ? zero : sum.z};
}

// brdf_sl() split for pixels that evaluate it more than once with the same surface and view vector (the tiled kernel's Slang passes: the
// RT-mode sun with L = the sun, the cache overlay with L = N, the RTGI overlay with one L per sample): what depends on the surface and V
// only — f0, the diffuse colour, |N.V + 1e-5|, a^2, the view half of the Smith term, pow5(clamp(1 - NoV)) — is evaluated once.  The
// per-light part applies the same operators to the same operands in the same order as brdf_sl(), so the same bits (the three channels
// of Fd_Burley's two Schlick terms are one value: f0 = 1 in all of them).
template <class T> struct BrdfView {
    V3<T> f0, diffuse_color;
    T NoV, a2, sqrtV, powV;  // |N.V + 1e-5|, a^2, sqrt((-NoV * a2 + NoV) * NoV + a2), pow5(clamp(1 - NoV, 0, 1))
};
template <class T> SAH_DEV BrdfView<T> brdf_sl_view(const Surface<T>& s, V3<T> v) {
    const T one = T::lit(1.0f), zero = T::lit(0.0f);
    const T dielectric_f0 = T::lit(0.04f);
    BrdfView<T> p;
    p.f0 = mix(V3<T>(dielectric_f0), s.base_color, s.metalness);
    p.diffuse_color = s.base_color * (one - dielectric_f0) * (one - s.metalness);
    p.NoV = nabs(dot(s.normal, v) + T::lit(1e-5f));
    p.a2 = s.roughness * s.roughness;
    p.sqrtV = nsqrt((-p.NoV * p.a2 + p.NoV) * p.NoV + p.a2);
    p.powV = npow5(nclamp(one - p.NoV, zero, one));
    return p;
}
template <class T> SAH_DEV V3<T> brdf_sl_light(const Surface<T>& s, const BrdfView<T>& p, V3<T> l, V3<T> v) {
    const T one = T::lit(1.0f), zero = T::lit(0.0f);
    const V3<T> h = normalize(v + l);
    T NoL = dot(s.normal, l);
    const T NoH = nclamp(dot(s.normal, h), zero, one);
    const T VoH = nclamp(dot(v, h), zero, one);
    const bool dark = tof(NoL) <= 0.f;
    NoL = nclamp(NoL, zero, one);
    const T LoH = nclamp(dot(l, h), zero, one);
    // Fd_Burley(NoV, NoL, LoH, roughness)
    const T f90 = T::lit(0.5f) + T::lit(2.0f) * s.roughness * LoH * LoH;
    const T light_scatter = one + (f90 - one) * npow5(nclamp(one - NoL, zero, one));
    const T view_scatter = one + (f90 - one) * p.powV;
    const T burley = light_scatter * view_scatter * inv_pi<T>();
    const V3<T> fd = p.diffuse_color * V3<T>(burley);
    const T D = D_GGX(NoH, s.roughness);
    const V3<T> Fv = F_Schlick(VoH, p.f0, one);
    // V_SmithGGXCorrelated(NoV, NoL, roughness)
    const T GGXL = p.NoV * nsqrt((-NoL * p.a2 + NoL) * NoL + p.a2);
    const T GGXV = NoL * p.sqrtV;
    const T Vis = T::lit(0.5f) / (GGXV + GGXL);
    const V3<T> fr = (D * Vis) * Fv;
    const V3<T> sum = fd + fr;
    return {dark ? zero : sum.x, dark ? zero : sum.y, dark ? zero : sum.z};
}

// fp32 brdf() = Fd() + Fr() for the hot paths: the shared sub-expressions written once, the `NoL <= 0 -> 0` early-outs as one
// select (both halves return 0 together and 0 + 0 == +0), and every sqrt / divide replaced by its restricted-range twin
// (sqrt_nr, rcp_nr, div_nr: same bits inside the domain).  `out_of_domain` is set when an operand leaves the domain for a pixel
// whose BRDF value is used (NoL > 0); the caller then evaluates the general form (fix-up kernel / inline fallback).  Ranges by
// construction for unit N, V, L: roughness is a UNORM8 value and the clamped dots are in [0,1], so only lower bounds need a compare.
SAH_DEV F3 brdf_fast(const Surface<Fn>& s, F3 l, F3 v, bool& out_of_domain) {
    const Fn one = Fn(1.0f), zero = Fn(0.0f);
    const Fn dielectric_f0 = Fn(0.04f);
    const F3 f0 = mix(F3(dielectric_f0), s.base_color, s.metalness);
    const F3 diffuse_color = s.base_color * (one - dielectric_f0) * (one - s.metalness);
    const F3 vl = v + l;
    const Fn dh = dot(vl, vl);  // <= 4 + eps
    const F3 h = vl * Fn(rcp_nr(sqrt_nr(dh.v)));
    Fn NoV = dot(s.normal, v) + Fn(1e-5f);
    Fn NoL = dot(s.normal, l);
    const Fn NoH = nclamp(dot(s.normal, h), zero, one);
    const Fn VoH = nclamp(dot(v, h), zero, one);
    const bool dark = NoL.v <= 0.f;
    NoV = nabs(NoV);
    NoL = nclamp(NoL, zero, one);
    const Fn LoH = nclamp(dot(l, h), zero, one);
    const F3 fd = diffuse_color * Fd_Burley(NoV, NoL, LoH, s.roughness);
    // D_GGX
    const Fn a = s.roughness;
    const Fn dden = one - NoH * NoH + a * a;
    const Fn k = Fn(div_nr(a.v, dden.v));
    const Fn D = k * k * (one / brdf_pi<Fn>());
    const F3 Fv = F_Schlick(VoH, f0, one);
    // V_SmithGGXCorrelated
    const Fn a2 = a * a;
    const Fn argL = (-NoL * a2 + NoL) * NoL + a2, argV = (-NoV * a2 + NoV) * NoV + a2;
    const Fn GGXL = NoV * Fn(sqrt_nr(argL.v));
    const Fn GGXV = NoL * Fn(sqrt_nr(argV.v));
    const Fn vden = GGXV + GGXL;
    const Fn Vis = Fn(half_over_nr(vden.v));
    const F3 fr = (D * Vis) * Fv;
    const F3 sum = fd + fr;
    // lower bounds: one min3 for the roots, one compare per divisor (a min of values that come from another basic block costs a
    // canonicalising v_max per operand).  Upper bounds: dh <= 4, NoV <= 1.00002, a <= 1, so every operand above is <= 4.
    const float lo_sqrt = __builtin_fminf(__builtin_fminf(dh.v, argL.v), argV.v);
    out_of_domain = !dark && !((lo_sqrt >= 0x1p-80f) & (dden.v >= kDivLo) & (vden.v >= kDivLo));
    return {dark ? zero : sum.x, dark ? zero : sum.y, dark ? zero : sum.z};
}

// brdf_fast() split for loops over lights: everything that depends on the surface and the view vector only (f0, the diffuse colour,
// NoV and its Schlick power, the view half of the Smith term) is evaluated once per pixel; the per-light part applies the same
// operators to the same operands as brdf_fast(), so the same bits.
struct BrdfPixel {
    F3 f0, diffuse_color;
    Fn NoV, a2, argV, sqrtV, powV;  // |N.V + 1e-5|, a^2, the view-side Smith argument and its root, pow5(clamp(1 - NoV))
};
SAH_DEV BrdfPixel brdf_fast_pixel(const Surface<Fn>& s, F3 v) {
    const Fn one = Fn(1.0f), zero = Fn(0.0f);
    const Fn dielectric_f0 = Fn(0.04f);
    BrdfPixel p;
    p.f0 = mix(F3(dielectric_f0), s.base_color, s.metalness);
    p.diffuse_color = s.base_color * (one - dielectric_f0) * (one - s.metalness);
    p.NoV = nabs(dot(s.normal, v) + Fn(1e-5f));
    const Fn a = s.roughness;
    p.a2 = a * a;
    p.argV = (-p.NoV * p.a2 + p.NoV) * p.NoV + p.a2;
    p.sqrtV = Fn(sqrt_nr(p.argV.v));  // (garbage outside the domain: brdf_fast_light() reports it)
    p.powV = npow5(nclamp(one - p.NoV, zero, one));
    return p;
}
// `dark_m`: lanes(dot(s.normal, l) <= 0), which the caller has voted on already (a ballot of a compare from another basic block is not free).
SAH_DEV F3 brdf_fast_light(const Surface<Fn>& s, const BrdfPixel& p, F3 l, F3 v, lanemask dark_m, lanemask& out_of_domain) {
    const Fn one = Fn(1.0f), zero = Fn(0.0f);
    const F3 vl = v + l;
    const Fn dh = dot(vl, vl);  // <= 4 + eps
    const F3 h = vl * Fn(rcp_nr(sqrt_nr(dh.v)));
    Fn NoL = dot(s.normal, l);
    const Fn NoH = nclamp(dot(s.normal, h), zero, one);
    const Fn VoH = nclamp(dot(v, h), zero, one);
    const bool dark = NoL.v <= 0.f;
    NoL = nclamp(NoL, zero, one);
    const Fn LoH = nclamp(dot(l, h), zero, one);
    // Fd_Burley(NoV, NoL, LoH, roughness): F_Schlick(u, 1, f90) = 1 + (f90 - 1) * pow5(clamp(1 - u)), the same in every channel
    const Fn f90 = Fn(0.5f) + Fn(2.0f) * s.roughness * LoH * LoH;
    const Fn light_scatter = one + (f90 - one) * npow5(nclamp(one - NoL, zero, one));
    const Fn view_scatter = one + (f90 - one) * p.powV;
    const Fn burley = light_scatter * view_scatter * (one / brdf_pi<Fn>());
    const F3 fd = p.diffuse_color * F3(burley);
    // D_GGX
    const Fn a = s.roughness;
    const Fn dden = one - NoH * NoH + a * a;
    const Fn k = Fn(div_nr(a.v, dden.v));
    const Fn D = k * k * (one / brdf_pi<Fn>());
    const F3 Fv = F_Schlick(VoH, p.f0, one);
    // V_SmithGGXCorrelated
    const Fn argL = (-NoL * p.a2 + NoL) * NoL + p.a2;
    const Fn GGXL = p.NoV * Fn(sqrt_nr(argL.v));
    const Fn GGXV = NoL * p.sqrtV;
    const Fn vden = GGXV + GGXL;
    const Fn Vis = Fn(half_over_nr(vden.v));
    const F3 fr = (D * Vis) * Fv;
    const F3 sum = fd + fr;
    const float lo_sqrt = __builtin_fminf(__builtin_fminf(dh.v, argL.v), p.argV.v);
    // (as a lane mask, from the masks of the elementary compares: the caller votes on it — see lanes())
    out_of_domain = ~dark_m & (lanes(!(lo_sqrt >= 0x1p-80f)) | lanes(!(dden.v >= kDivLo)) | lanes(!(vden.v >= kDivLo)));
    return {dark ? zero : sum.x, dark ? zero : sum.y, dark ? zero : sum.z};
}

}  // namespace sah
