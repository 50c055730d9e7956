// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h notes the flag as "not supported on AMD GFX9xx boards" for
// the module-launch entry point.)  Two spinning kernels of one workgroup each, ~200 us apiece: back to back they take ~400 us, overlapped ~200.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/any_order.hip -o tools/microbench/any_order && tools/microbench/any_order
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void spin(long long ticks, unsigned* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicAdd(out, 1u);
}

static float run(hipStream_t st, unsigned flags, unsigned* d, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < n; i++) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, (i == 0) ? 0u : flags, 20000ll, d);  // 100 MHz: 200 us
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 100ll, d);  // an ordinary launch behind them (a barrier packet)
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned* d;
    hipMalloc(&d, 4);
    hipMemset(d, 0, 4);
    run(st, 0, d, 2);
    for (int rep = 0; rep < 3; rep++) {
        printf("4 spinning kernels, ordinary launches:        %.3f ms\n", run(st, 0, d, 4));
        printf("4 spinning kernels, hipExtAnyOrderLaunch 2-4: %.3f ms\n", run(st, hipExtAnyOrderLaunch, d, 4));
    }
    unsigned h = 0;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("kernels run: %u, last error: %s\n", h, hipGetErrorString(hipGetLastError()));
    return 0;
}
