// Probe (round 3): which cross-process primitives work on this pool for the direct exchange backend — hipIpcGetMemHandle /
// hipIpcOpenMemHandle on hipMalloc'ed and fine-grained memory (interior pointers through hipMemGetAddressRange), kernel-side flag
// signalling across two processes that share one GPU, hipMemcpyAsync into the peer's buffer.
//   ipc_probe <rank 0|1> <dir>      (two processes, started before either touches the GPU; handles travel through files in <dir>)
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); fflush(stdout); exit(3); } } while (0)
static int g_rank;

__global__ void k_signal(uint32_t* flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_wait(const uint32_t* flag, uint32_t v, uint32_t* status) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 200000000ll) { *status = 1; return; }  // 2 s at 100 MHz
    }
    *status = 0;
}
__global__ void k_fill(uint32_t* p, uint32_t n, uint32_t v) { const uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v + i; }

static void put(const std::string& path, const void* p, size_t n) {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb"); fwrite(p, 1, n, f); fclose(f); rename(tmp.c_str(), path.c_str());
}
static void get(const std::string& path, void* p, size_t n) {
    for (int i = 0; i < 3000; i++) { FILE* f = fopen(path.c_str(), "rb"); if (f) { size_t r = fread(p, 1, n, f); fclose(f); if (r == n) return; } usleep(10000); }
    printf("rank %d: timeout waiting for %s\n", g_rank, path.c_str()); exit(4);
}

int main(int argc, char** argv) {
    g_rank = atoi(argv[1]);
    const std::string dir = argv[2];
    const int peer = 1 - g_rank;
    CK(hipSetDevice(0));
    const uint32_t N = 1 << 20;
    // a sub-allocation inside a larger block, as a caching allocator would hand out
    char* block; CK(hipMalloc((void**)&block, 64 << 20));
    uint32_t* buf = (uint32_t*)(block + (16 << 20));
    void* base; size_t size; CK(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, buf));
    printf("rank %d: range base offset %zu size %zu\n", g_rank, (size_t)((char*)buf - (char*)base), size);
    uint32_t* flags = nullptr;
    hipError_t fe = hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocFinegrained);
    printf("rank %d: fine-grained alloc: %s\n", g_rank, hipGetErrorString(fe));
    if (fe != hipSuccess) CK(hipMalloc((void**)&flags, 4096));
    CK(hipMemset(flags, 0, 4096));
    hipIpcMemHandle_t hb, hf;
    CK(hipIpcGetMemHandle(&hb, base));
    hipError_t e = hipIpcGetMemHandle(&hf, flags);
    printf("rank %d: ipc handle of flag memory: %s\n", g_rank, hipGetErrorString(e));
    if (e != hipSuccess) exit(5);
    struct { hipIpcMemHandle_t hb, hf; size_t off; } mine = {hb, hf, (size_t)((char*)buf - (char*)base)}, theirs;
    put(dir + "/h" + std::to_string(g_rank), &mine, sizeof(mine));
    get(dir + "/h" + std::to_string(peer), &theirs, sizeof(theirs));
    void *pbase, *pflags_v;
    CK(hipIpcOpenMemHandle(&pbase, theirs.hb, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle(&pflags_v, theirs.hf, hipIpcMemLazyEnablePeerAccess));
    uint32_t* pbuf = (uint32_t*)((char*)pbase + theirs.off);
    uint32_t* pflags = (uint32_t*)pflags_v;
    uint32_t* status; CK(hipHostMalloc((void**)&status, 64)); status[0] = 99;
    hipStream_t st; CK(hipStreamCreate(&st));
    for (uint32_t round = 1; round <= 3; round++) {
        // my half of my buffer, then push it into the peer's buffer, then signal; wait for the peer's signal; check
        const uint32_t half = N / 2, mine_off = g_rank * half, peer_off = peer * half;
        hipLaunchKernelGGL(k_fill, dim3(half / 256), dim3(256), 0, st, buf + mine_off, half, round * 1000000u + g_rank * 100u);
        CK(hipMemcpyAsync(pbuf + mine_off, buf + mine_off, half * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, pflags + g_rank, round);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, flags + peer, round, status);
        CK(hipStreamSynchronize(st));
        uint32_t first, last;
        CK(hipMemcpy(&first, buf + peer_off, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&last, buf + peer_off + half - 1, 4, hipMemcpyDeviceToHost));
        const uint32_t want = round * 1000000u + peer * 100u;
        printf("rank %d round %u: wait status %u, peer half [%u .. %u] want [%u .. %u] %s\n", g_rank, round, status[0], first, last, want, want + half - 1,
               (status[0] == 0 && first == want && last == want + half - 1) ? "OK" : "BAD");
        // second handshake so that nobody overwrites before the other has checked
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, pflags + 8 + g_rank, round);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, flags + 8 + peer, round, status);
        CK(hipStreamSynchronize(st));
    }
    CK(hipIpcCloseMemHandle(pbase));
    CK(hipIpcCloseMemHandle(pflags_v));
    printf("rank %d: done\n", g_rank);
    return 0;
}
