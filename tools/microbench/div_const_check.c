// Exhaustive check of the constant-divisor divide used by the irradiance-cache gather (lighting_gi_ext.hpp: div_const):
//   q0 = a * z; r0 = fma(-b, q0, a); q1 = fma(r0, z, q0)      with z = RN(1 / b)
// against a / b (IEEE) for every divisor b = (n + 2) * 32, n = 1..30 (atlas widths the hot path accepts) and EVERY fp32 a in [0.5, b].
//   gcc -O2 -fopenmp -ffp-contract=off tools/microbench/div_const_check.c -lm -o /tmp/div_const_check && /tmp/div_const_check
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void) {
    long total_bad1 = 0, total_bad2 = 0;
    for (int n = 1; n <= 30; n++) {
        const float b = (float)((n + 2) * 32);
        const float z = 1.0f / b;
        const uint32_t lo = f2u(0.5f), hi = f2u(b);
        long bad1 = 0, bad2 = 0;
#pragma omp parallel for reduction(+ : bad1, bad2)
        for (uint32_t u = lo; u <= hi; u++) {
            const float a = u2f(u);
            const float want = a / b;
            const float q0 = a * z;
            const float r0 = fmaf(-b, q0, a);
            const float q1 = fmaf(r0, z, q0);
            const float r1 = fmaf(-b, q1, a);
            const float q2 = fmaf(r1, z, q1);
            bad1 += f2u(q1) != f2u(want);
            bad2 += f2u(q2) != f2u(want);
        }
        printf("b = %4.0f: one correction %ld wrong, two corrections %ld wrong (of %u)\n", b, bad1, bad2, hi - lo + 1);
        total_bad1 += bad1;
        total_bad2 += bad2;
    }
    printf("total: one correction %ld, two corrections %ld\n", total_bad1, total_bad2);
    return total_bad1 ? 1 : 0;
}
