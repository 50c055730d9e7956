// Exhaustive check of the fp16 ("Slang half") divide and square root of csrc/numerics.hpp against the contract's definition — the IEEE fp32
// operator on the widened operands, rounded to fp16 — for EVERY operand: 2^32 (a, b) pairs for the divide, 2^16 inputs for the
// root, NaN / inf / zero / denormal bit patterns included.  MI355X, about a second:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         tools/microbench/half_math_check.hip -o tools/microbench/half_math_check && tools/microbench/half_math_check
// Candidates (whichever passes with 0 mismatches may replace the IEEE expansion in Hn's operator/ and nsqrt):
//   div A: q = a * rcp(b)                                     div B: one Newton step on q            div C: B + v_div_fixup_f32 (specials)
//   sqrt A: v_sqrt_f32                                        sqrt B: sqrt_nr-style refinement of v_rsq_f32
// A result counts as equal when the fp16 bit patterns are equal, or both are NaN (reported separately: NaN payloads / signs that differ).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../androidrenderer_amd/csrc/numerics.hpp"

using namespace sah;

struct Report {
    unsigned long long mismatches[8], nan_bits_differ[8];
    uint32_t first[8][2];
    uint32_t nan_got, nan_want;  // sqrt A: the fp16 bits of one NaN result whose bits differ from the expansion's
};

__device__ inline uint16_t to_h(float x) { return __builtin_bit_cast(uint16_t, (_Float16)opaque(x)); }
__device__ inline bool is_nan_h(uint16_t h) { return (h & 0x7fffu) > 0x7c00u; }

__device__ inline void check(Report* r, int which, uint16_t got, uint16_t want, uint32_t a, uint32_t b) {
    if (got == want) return;
    if (is_nan_h(got) && is_nan_h(want)) {
        atomicAdd(&r->nan_bits_differ[which], 1ull);
        return;
    }
    if (atomicAdd(&r->mismatches[which], 1ull) == 0ull) {
        r->first[which][0] = a;
        r->first[which][1] = b;
    }
}

__device__ inline float div_a(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ inline float div_b(float a, float b) {
    const float y = __builtin_amdgcn_rcpf(b);
    const float q0 = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
}
__device__ inline float div_c(float a, float b) { return __builtin_amdgcn_div_fixupf(div_b(a, b), b, a); }
__device__ inline float div_d(float a, float b) { return __builtin_amdgcn_div_fixupf(div_a(a, b), b, a); }
__device__ inline float sqrt_b(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    return __builtin_fmaf(g, r, g);
}

__global__ void __launch_bounds__(256) k_div(Report* rep) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * 256ull) {
        const uint32_t ab = (uint32_t)(i >> 16), bb = (uint32_t)(i & 0xffffu);
        const float a = (float)__builtin_bit_cast(_Float16, (uint16_t)ab), b = (float)__builtin_bit_cast(_Float16, (uint16_t)bb);
        const uint16_t want = to_h(a / b);
        check(rep, 0, to_h(div_a(a, b)), want, ab, bb);
        check(rep, 1, to_h(div_b(a, b)), want, ab, bb);
        check(rep, 2, to_h(div_c(a, b)), want, ab, bb);
        check(rep, 3, to_h(div_d(a, b)), want, ab, bb);
    }
}
__global__ void __launch_bounds__(256) k_sqrt(Report* rep) {
    const uint32_t hb = blockIdx.x * 256u + threadIdx.x;
    if (hb >= 65536u) return;
    const float x = (float)__builtin_bit_cast(_Float16, (uint16_t)hb);
    const uint16_t want = to_h(__builtin_sqrtf(x));
    check(rep, 4, to_h(__builtin_amdgcn_sqrtf(x)), want, hb, 0);
    check(rep, 5, to_h(sqrt_b(x)), want, hb, 0);
    if (hb == 0xbc00u) {  // sqrt(-1)
        rep->nan_got = to_h(__builtin_amdgcn_sqrtf(x));
        rep->nan_want = want;
    }
    // pow(x, 5) of a clamped fp16 value (F_Schlick's argument, [0, 1]): the fp64 product chain rounded to fp32, then to fp16 — against fp32 forms
    if (hb <= 0x3c00u) {
        const double d = (double)x;
        const uint16_t w5 = to_h((float)(d * d * d * d * d));
        const float a2 = x * x;  // exact: 22 bits
        const float a4 = a2 * a2;
        check(rep, 6, to_h(a4 * x), w5, hb, 0);
        const float lo = __builtin_fmaf(a2, a2, -a4);  // a2 * a2 = a4 + lo exactly
        check(rep, 7, to_h(__builtin_fmaf(a4, x, lo * x)), w5, hb, 0);
    }
}

int main() {
    Report* rep;
    hipMalloc(&rep, sizeof(Report));
    hipMemset(rep, 0, sizeof(Report));
    hipLaunchKernelGGL(k_div, dim3(256 * 16), dim3(256), 0, 0, rep);
    hipLaunchKernelGGL(k_sqrt, dim3(256), dim3(256), 0, 0, rep);
    Report h;
    if (hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
        printf("hip error\n");
        return 2;
    }
    const char* names[8] = {"div A  a * rcp(b)", "div B  + one Newton step", "div C  B + v_div_fixup_f32", "div D  A + v_div_fixup_f32", "sqrt A v_sqrt_f32", "sqrt B rsq + one step",
                            "pow5 A (a2 * a2) * a in fp32", "pow5 B fma(a4, a, lo * a)"};
    for (int i = 0; i < 8; i++)
        printf("%-28s mismatches %12llu   NaN with other bits %10llu   first (a, b) = (0x%04x, 0x%04x)\n", names[i], h.mismatches[i], h.nan_bits_differ[i], h.first[i][0],
               h.first[i][1]);
    printf("sqrt(-1.0h): v_sqrt_f32 gives fp16 bits 0x%04x, the IEEE expansion 0x%04x\n", h.nan_got, h.nan_want);
    return 0;
}
