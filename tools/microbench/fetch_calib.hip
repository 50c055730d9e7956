// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ on gfx950 for the access patterns of the Lighting pass: the guide
// (MI355X_MICROARCH.md "HBM") states the x2 correction for WIDE coalesced streaming reads only and calls every other width uncalibrated.
// Each kernel reads a KNOWN number of bytes / distinct cache lines from a buffer larger than the 256 MiB Infinity Cache, once:
//   stream16 / stream4 / stream2   coalesced reads of 16 / 4 / 2 bytes per lane over the whole buffer
//   gather2_lines                  one 2-byte load per lane from a DISTINCT random 128-byte line each (a D16 texel gather that never shares a line)
//   gather2_pcf                    the PCF footprint: 2 x 2 texels of a 4096-wide D16 layer at a random position per lane (two rows: two lines)
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib ; run under rocprofv3 --pmc (tools/experiments/r5/r5_pmc_gather.sh)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void stream16(const uint4* p, size_t n, uint32_t* sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 v = p[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1;
}
__global__ void stream4(const uint32_t* p, size_t n, uint32_t* sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (p[i] == 0x12345678u) *sink = 1;
}
__global__ void stream2(const uint16_t* p, size_t n, uint32_t* sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (p[i] == 0x1234u) *sink = 1;
}
__global__ void gather2_lines(const uint16_t* p, const uint32_t* line_of, size_t n, uint32_t* sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (p[(size_t)line_of[i] * 64u + (i & 63u)] == 0x1234u) *sink = 1;  // 64 texels of 2 bytes per 128-byte line
}
__global__ void gather2_pcf(const uint16_t* p, const uint32_t* pos, size_t n, uint32_t width, uint32_t* sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = pos[i];
    const uint32_t s = (uint32_t)p[o] + p[o + 1] + p[o + width] + p[o + width + 1];
    if (s == 0x12345u) *sink = 1;
}

int main() {
    const size_t bytes = 512ull << 20;  // twice the Infinity Cache
    uint8_t* buf = nullptr;
    uint32_t *sink = nullptr, *idx = nullptr;
    CHECK(hipMalloc((void**)&buf, bytes));
    CHECK(hipMemset(buf, 0x5a, bytes));
    CHECK(hipMalloc((void**)&sink, 4));
    const size_t lines = bytes / 128, n_gather = 1u << 20;
    std::vector<uint32_t> h(n_gather);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    // distinct random lines: a stride permutation over the line count (odd multiplier modulo a power of two)
    for (size_t i = 0; i < n_gather; i++) h[i] = (uint32_t)(((i * 2654435761ull) + 12345ull) & (lines - 1));
    CHECK(hipMalloc((void**)&idx, n_gather * 4));
    CHECK(hipMemcpy(idx, h.data(), n_gather * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> pos(n_gather);
    const uint32_t width = 4096, layer_texels = 4096u * 4096u * 4u;  // 4 x 4096^2 D16 = 128 MiB of the buffer
    for (size_t i = 0; i < n_gather; i++) {
        const uint32_t x = (uint32_t)(rnd() % (width - 1)), y = (uint32_t)(rnd() % (4096u * 4u - 1));
        pos[i] = y * width + x;
    }
    (void)layer_texels;
    uint32_t* dpos = nullptr;
    CHECK(hipMalloc((void**)&dpos, n_gather * 4));
    CHECK(hipMemcpy(dpos, pos.data(), n_gather * 4, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    const int T = 256;
    // flush between kernels: a 512 MB memset evicts what the previous kernel left in the Infinity Cache
    hipLaunchKernelGGL(stream16, dim3((unsigned)((bytes / 16 + T - 1) / T)), dim3(T), 0, 0, (const uint4*)buf, bytes / 16, sink);
    CHECK(hipMemset(buf, 0x5a, bytes));
    hipLaunchKernelGGL(stream4, dim3((unsigned)((bytes / 4 / 4 + T - 1) / T)), dim3(T), 0, 0, (const uint32_t*)buf, bytes / 4 / 4, sink);  // a quarter of the buffer: 128 MiB
    CHECK(hipMemset(buf, 0x5a, bytes));
    hipLaunchKernelGGL(stream2, dim3((unsigned)((bytes / 2 / 8 + T - 1) / T)), dim3(T), 0, 0, (const uint16_t*)buf, bytes / 2 / 8, sink);  // an eighth: 64 MiB
    CHECK(hipMemset(buf, 0x5a, bytes));
    hipLaunchKernelGGL(gather2_lines, dim3((unsigned)((n_gather + T - 1) / T)), dim3(T), 0, 0, (const uint16_t*)buf, idx, n_gather, sink);
    CHECK(hipMemset(buf, 0x5a, bytes));
    hipLaunchKernelGGL(gather2_pcf, dim3((unsigned)((n_gather + T - 1) / T)), dim3(T), 0, 0, (const uint16_t*)buf, dpos, n_gather, width, sink);
    CHECK(hipDeviceSynchronize());
    printf("known bytes: stream16 %zu  stream4 %zu  stream2 %zu ; gather2_lines: %zu loads of 2 B from %zu distinct 128-B lines (+ %zu B of indices, coalesced 4 B/lane) ; "
           "gather2_pcf: %zu footprints of 2 x 2 texels, two rows each (+ %zu B of positions)\n",
           bytes, bytes / 4, bytes / 8, n_gather, n_gather, n_gather * 4, n_gather, n_gather * 4);
    return 0;
}
