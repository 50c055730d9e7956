// Issue-cost microbenchmark for the VALU instructions the lighting kernels are made of (MI355X, gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate && tools/microbench/valu_rate
// Every kernel runs ITERS x 16 instructions (16 independent dependency chains) per thread with 8 waves resident per SIMD; the
// figure printed is shader-clock cycles per wave-instruction per SIMD (clock measured with s_memtime against the 100 MHz
// wall clock), i.e. the issue cost of one wave64 instruction when the VALU is the only bottleneck.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;

#define CHAIN16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

// One kernel per instruction template.  OPS is the asm text with %0 = the chain register (in/out), %1 and %2 = two more VGPRs.
#define DEF_KERNEL_F32(NAME, OPS)                                                                 \
    __global__ void __launch_bounds__(256) NAME(float* out, float a, float b, long long* clk) {  \
        float r[16];                                                                              \
        for (int i = 0; i < 16; i++) r[i] = (float)threadIdx.x + i + 1.0f;                        \
        const long long t0 = clock64(), w0 = wall_clock64();                                      \
        for (int it = 0; it < ITERS; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(OPS : "+v"(r[i]) : "v"(a), "v"(b) : "vcc", "s20", "s21"); \
        }                                                                                         \
        const long long t1 = clock64(), w1 = wall_clock64();                                      \
        float s = 0;                                                                              \
        for (int i = 0; i < 16; i++) s += r[i];                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }           \
    }

#define DEF_KERNEL_PAIR(NAME, OPS)                                                                \
    __global__ void __launch_bounds__(256) NAME(float* out, float a, float b, long long* clk) {  \
        f2v r[16];                                                                                \
        const f2v av = {a, a}, bv = {b, b};                                                       \
        for (int i = 0; i < 16; i++) r[i] = f2v{(float)threadIdx.x + i + 1.0f, (float)i + 1.0f}; \
        const long long t0 = clock64(), w0 = wall_clock64();                                      \
        for (int it = 0; it < ITERS; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(OPS : "+v"(r[i]) : "v"(av), "v"(bv) : "vcc", "s20", "s21"); \
        }                                                                                         \
        const long long t1 = clock64(), w1 = wall_clock64();                                      \
        float s = 0;                                                                              \
        for (int i = 0; i < 16; i++) s += r[i].x + r[i].y;                                        \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }           \
    }

DEF_KERNEL_F32(k_mul_e32, "v_mul_f32_e32 %0, %0, %1")
DEF_KERNEL_F32(k_add_e32, "v_add_f32_e32 %0, %0, %1")
DEF_KERNEL_F32(k_fmac_e32, "v_fmac_f32_e32 %0, %1, %2")
DEF_KERNEL_F32(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_F32(k_fma_neg, "v_fma_f32 %0, -%0, %1, %2")
DEF_KERNEL_F32(k_mul_e64, "v_mul_f32_e64 %0, %0, |%1|")
DEF_KERNEL_F32(k_fma_mix, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]")
DEF_KERNEL_F32(k_mov, "v_mov_b32_e32 %0, %1")
DEF_KERNEL_F32(k_add_u32, "v_add_u32_e32 %0, %0, %1")
DEF_KERNEL_F32(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
DEF_KERNEL_F32(k_and, "v_and_b32_e32 %0, %0, %1")
DEF_KERNEL_F32(k_max_f32, "v_max_f32_e32 %0, %0, %1")
DEF_KERNEL_F32(k_med3, "v_med3_f32 %0, %0, %1, %2")
DEF_KERNEL_F32(k_min_i32, "v_min_i32_e32 %0, %0, %1")
DEF_KERNEL_F32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL_F32(k_floor, "v_floor_f32_e32 %0, %0")
DEF_KERNEL_F32(k_cvt_f16, "v_cvt_f16_f32_e32 %0, %0")
DEF_KERNEL_F32(k_cvt_f32_f16, "v_cvt_f32_f16_e32 %0, %0")
DEF_KERNEL_F32(k_cvt_i32, "v_cvt_i32_f32_e32 %0, %0")
DEF_KERNEL_F32(k_cvt_f32_u32, "v_cvt_f32_u32_e32 %0, %0")
DEF_KERNEL_F32(k_rcp, "v_rcp_f32_e32 %0, %0")
DEF_KERNEL_F32(k_sqrt, "v_sqrt_f32_e32 %0, %0")
DEF_KERNEL_F32(k_rsq, "v_rsq_f32_e32 %0, %0")
DEF_KERNEL_F32(k_div_scale, "v_div_scale_f32 %0, s[20:21], %0, %1, %2")
DEF_KERNEL_F32(k_div_fmas, "v_div_fmas_f32 %0, %0, %1, %2")
DEF_KERNEL_F32(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
DEF_KERNEL_F32(k_cmp_vcc, "v_cmp_lt_f32_e32 vcc, %0, %1")
DEF_KERNEL_F32(k_cmp_sgpr, "v_cmp_lt_f32_e64 s[20:21], %0, %1")
DEF_KERNEL_F32(k_cmp_class, "v_cmp_class_f32_e64 s[20:21], %0, %1")
DEF_KERNEL_F32(k_cndmask_vcc, "v_cndmask_b32_e32 %0, %0, %1, vcc")
DEF_KERNEL_F32(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEF_KERNEL_F32(k_cmp_cnd, "v_cmp_lt_f32_e32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %2, vcc")
DEF_KERNEL_PAIR(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_PAIR(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
DEF_KERNEL_PAIR(k_pk_add, "v_pk_add_f32 %0, %0, %1")
DEF_KERNEL_PAIR(k_mul_f64, "v_mul_f64 %0, %0, %1")
DEF_KERNEL_PAIR(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
DEF_KERNEL_PAIR(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
DEF_KERNEL_F32(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
DEF_KERNEL_F32(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
DEF_KERNEL_F32(k_add_f16, "v_add_f16_e32 %0, %0, %1")

typedef void (*kern_t)(float*, float, float, long long*);

static void run(const char* name, kern_t kern, float* d_out, long long* d_clk, int per_iter = 16) {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, d_clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, d_clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long clk[2];
    (void)hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost);
    const double mhz = clk[1] > 0 ? (double)clk[0] / ((double)clk[1] / 100.0) : 0.0;  // wall clock ticks at 100 MHz
    const double wave_instr_per_simd = (double)blocks * 4 * ITERS * per_iter / 1024.0;
    const double cycles = ms * 1e-3 * mhz * 1e6 / wave_instr_per_simd;
    printf("%-18s %8.3f ms  shader clock %6.0f MHz  %6.2f cycles / wave-instruction / SIMD\n", name, ms, mhz, cycles);
}

int main() {
    float* d;
    long long* clk;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    (void)hipMalloc(&clk, 2 * sizeof(long long));
#define RUN(K) run(#K, K, d, clk)
    RUN(k_mul_e32); RUN(k_add_e32); RUN(k_fmac_e32); RUN(k_fma); RUN(k_fma_neg); RUN(k_mul_e64); RUN(k_fma_mix); RUN(k_mov); RUN(k_add_u32);
    RUN(k_lshl_add); RUN(k_and); RUN(k_max_f32); RUN(k_med3); RUN(k_min_i32); RUN(k_mul_lo); RUN(k_floor); RUN(k_cvt_f16); RUN(k_cvt_f32_f16);
    RUN(k_cvt_i32); RUN(k_cvt_f32_u32); RUN(k_rcp); RUN(k_sqrt); RUN(k_rsq); RUN(k_div_scale); RUN(k_div_fmas); RUN(k_div_fixup); RUN(k_cmp_vcc);
    RUN(k_cmp_sgpr); RUN(k_cmp_class); RUN(k_cndmask_vcc); RUN(k_cndmask_sgpr);
    run("k_cmp_cnd (pair)", k_cmp_cnd, d, clk, 32);
    RUN(k_pk_fma); RUN(k_pk_mul); RUN(k_pk_add); RUN(k_mul_f64); RUN(k_fma_f64); RUN(k_lshl_add_u64); RUN(k_pk_add_f16); RUN(k_pk_mul_f16); RUN(k_add_f16);
    (void)hipFree(d);
    (void)hipFree(clk);
    return 0;
}
