// Exhaustive / massive check of the restricted-range correctly-rounded primitives of csrc/numerics.hpp against hipcc's IEEE
// operators (built with the library's flags).  MI355X, a few seconds:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         tools/microbench/exact_math_check.hip -o tools/microbench/exact_math_check && tools/microbench/exact_math_check
// sqrt_nr / sqrt_nr0 / rcp_nr: EVERY fp32 bit pattern whose magnitude lies in [2^-100, 2^100] (and +0 for sqrt_nr0).
// div_nr: 2^36 pseudo-random operand pairs with |a|, |b| in [2^-40, 2^40], a quarter of them with mantissas within a few ulps of
// all-zeros / all-ones (the hard cases of Newton division), plus a = +0.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../androidrenderer_amd/csrc/numerics.hpp"

using namespace sah;

struct Report {
    unsigned long long tested, mismatches;
    uint32_t first_a, first_b;
};

__device__ inline bool same(float x, float y) { return __float_as_uint(x) == __float_as_uint(y); }

__device__ inline void fail(Report* r, uint32_t a, uint32_t b) {
    if (atomicAdd(&r->mismatches, 1ull) == 0ull) {
        r->first_a = a;
        r->first_b = b;
    }
}

// which: 0 sqrt_nr, 1 sqrt_nr0, 2 rcp_nr
__global__ void __launch_bounds__(256) k_unary(Report* rep, int which) {
    unsigned long long tested = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * 256ull) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float m = __builtin_fabsf(x);
        const bool in_domain = m >= kNrLo && m <= kNrHi;
        if (which == 0) {
            if (!(in_domain && x > 0.f)) continue;
            if (!same(sqrt_nr(x), __builtin_sqrtf(x))) fail(rep, bits, 0);
        } else if (which == 1) {
            if (!((in_domain && x > 0.f) || bits == 0u)) continue;
            if (!same(sqrt_nr0(x), __builtin_sqrtf(x))) fail(rep, bits, 0);
        } else {
            if (!in_domain) continue;
            if (!same(rcp_nr(x), 1.0f / x)) fail(rep, bits, 0);
        }
        tested++;
    }
    atomicAdd(&rep->tested, tested);
}

__device__ inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ inline float make_operand(uint32_t r, uint32_t r2, bool hard) {
    uint32_t mant = r & 0x7fffffu;
    if (hard) {  // mantissa within 8 ulps of 0x000000 or 0x7fffff
        const uint32_t d = r2 & 7u;
        mant = (r2 & 8u) ? 0x7fffffu - d : d;
    }
    const uint32_t exp = 127u - 40u + ((r >> 23) % 80u);  // [2^-40, 2^40)
    const uint32_t sign = (r2 >> 31) << 31;
    return __uint_as_float(sign | (exp << 23) | mant);
}
__global__ void __launch_bounds__(256) k_div(Report* rep, uint32_t pairs_per_thread, uint64_t seed) {
    const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
    unsigned long long tested = 0;
    for (uint32_t j = 0; j < pairs_per_thread; j++) {
        const uint64_t h = mix64(seed + tid * pairs_per_thread + j), h2 = mix64(h);
        const bool hard_a = ((h2 >> 40) & 3u) == 0u, hard_b = ((h2 >> 42) & 1u) == 0u && (((h2 >> 43) & 1u) == 0u);
        float a = make_operand((uint32_t)h, (uint32_t)(h2 >> 8), hard_a);
        const float b = make_operand((uint32_t)(h >> 32), (uint32_t)h2, hard_b);
        if (((h2 >> 50) & 1023u) == 0u) a = 0.0f;  // +0 numerator
        if (!same(div_nr(a, b), a / b)) fail(rep, __float_as_uint(a), __float_as_uint(b));
        tested++;
    }
    atomicAdd(&rep->tested, tested);
}

static int report(const char* name, Report* d_rep) {
    (void)hipDeviceSynchronize();
    Report r;
    (void)hipMemcpy(&r, d_rep, sizeof(r), hipMemcpyDeviceToHost);
    printf("%-10s tested %14llu  mismatches %llu", name, r.tested, r.mismatches);
    if (r.mismatches) printf("  first: a=0x%08x b=0x%08x", r.first_a, r.first_b);
    printf("\n");
    (void)hipMemset(d_rep, 0, sizeof(Report));
    return r.mismatches != 0;
}

int main() {
    Report* d_rep;
    (void)hipMalloc(&d_rep, sizeof(Report));
    (void)hipMemset(d_rep, 0, sizeof(Report));
    int bad = 0;
    hipLaunchKernelGGL(k_unary, dim3(256 * 32), dim3(256), 0, 0, d_rep, 0);
    bad |= report("sqrt_nr", d_rep);
    hipLaunchKernelGGL(k_unary, dim3(256 * 32), dim3(256), 0, 0, d_rep, 1);
    bad |= report("sqrt_nr0", d_rep);
    hipLaunchKernelGGL(k_unary, dim3(256 * 32), dim3(256), 0, 0, d_rep, 2);
    bad |= report("rcp_nr", d_rep);
    for (int pass = 0; pass < 16; pass++) {  // 16 x 2^32 pairs
        hipLaunchKernelGGL(k_div, dim3(1 << 16), dim3(256), 0, 0, d_rep, 256u, 0x1234567ull + (uint64_t)pass * (1ull << 40));
    }
    bad |= report("div_nr", d_rep);
    (void)hipFree(d_rep);
    return bad;
}
