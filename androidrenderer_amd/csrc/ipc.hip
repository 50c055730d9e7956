// Direct exchange over peer-mapped memory (include/sah_hip.h "direct exchange"): HIP IPC handles, a mailbox of monotonic arrival
// counters per rank in fine-grained device memory, and two tiny kernels — one that stores a counter value into the peers' mailboxes
// (system-scope release behind everything enqueued before it) and one wave that polls the own mailbox (system-scope acquire, s_sleep
// between polls, bounded: it gives up after two seconds and raises a host-visible flag, so that no kernel of this library can spin forever).
// No reference counterpart: the reference drives one device (RenderCore/render/backend/render_backend.cpp:135-153).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sah_hip.h"

namespace sah {

struct IpcPeers {
    uint32_t* slot[SAH_IPC_MAX_WORLD];  // where to store (signal) / what to poll (wait); null: skipped
};

__global__ void __launch_bounds__(64) k_ipc_signal(const IpcPeers peers, uint32_t value) {
    const uint32_t p = threadIdx.x;
    if (p < SAH_IPC_MAX_WORLD && peers.slot[p]) __hip_atomic_store(peers.slot[p], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// every polled counter >= value, or 2 s (wall_clock64 ticks at 100 MHz) have passed: then *timed_out (pinned host memory) is raised
__global__ void __launch_bounds__(64) k_ipc_wait(const IpcPeers own, uint32_t value, uint32_t* timed_out) {
    const uint32_t p = threadIdx.x;
    const bool mine = p < SAH_IPC_MAX_WORLD && own.slot[p];
    const long long t0 = wall_clock64();
    bool ok = !mine;
    for (;;) {
        if (!ok) ok = (int32_t)(__hip_atomic_load(own.slot[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) >= 0;
        if (__all(ok)) break;
        if (wall_clock64() - t0 > 200000000ll) {  // every lane leaves together: the wave drains whatever the peers do
            if (threadIdx.x == 0) __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

hipError_t launch_ipc_signal(const IpcPeers& peers, uint32_t value, hipStream_t st) {
    hipLaunchKernelGGL(k_ipc_signal, dim3(1), dim3(64), 0, st, peers, value);
    return hipGetLastError();
}
hipError_t launch_ipc_wait(const IpcPeers& own, uint32_t value, uint32_t* timed_out, hipStream_t st) {
    hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, st, own, value, timed_out);
    return hipGetLastError();
}

}  // namespace sah
