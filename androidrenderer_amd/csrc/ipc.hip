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
struct IpcCopies {
    uint8_t* dst[SAH_IPC_MAX_WORLD];  // the own slot inside every peer's buffer; null: skipped
};

// `abort`: a device word of this context that a wait which gave up has raised.  From then on the gather it belongs to — and every later
// one — neither signals nor copies: a peer that did not arrive may still be reading or writing its buffer.
__global__ void __launch_bounds__(64) k_ipc_signal(const IpcPeers peers, uint32_t value, const uint32_t* abort) {
    if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const uint32_t p = threadIdx.x;
    if (p < SAH_IPC_MAX_WORLD && peers.slot[p]) __hip_atomic_store(peers.slot[p], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// every polled counter >= value, or 2 s (wall_clock64 ticks at 100 MHz) have passed: then *abort (device) and *timed_out (pinned host
// memory) are raised
// `notes`: this rank's "gave up" word in every peer's mailbox, raised together with them: a peer that arrives late still finds its own
// ready-wait satisfied (this rank's counter is there) and would copy its rows into a buffer this rank may be about to give back — its copy
// kernel reads the note first (k_ipc_copy).
__global__ void __launch_bounds__(64) k_ipc_wait(const IpcPeers own, uint32_t value, uint32_t* abort, uint32_t* timed_out, const IpcPeers notes) {
    if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const uint32_t p = threadIdx.x;
    const bool mine = p < SAH_IPC_MAX_WORLD && own.slot[p];
    const long long t0 = wall_clock64();
    bool ok = !mine;
    for (;;) {
        if (!ok) ok = (int32_t)(__hip_atomic_load(own.slot[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) >= 0;
        if (__all(ok)) break;
        if (wall_clock64() - t0 > 200000000ll) {  // every lane leaves together: the wave drains whatever the peers do
            if (threadIdx.x == 0) {
                __hip_atomic_store(abort, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (p < SAH_IPC_MAX_WORLD && notes.slot[p]) __hip_atomic_store(notes.slot[p], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

// The own slot into every peer's copy of the buffer: blockIdx.y = peer, 16 bytes per lane and step.  (A kernel instead of one
// hipMemcpyAsync per peer so that the abort word can stop it: copies the host has already enqueued cannot be taken back.)
// `gave_up`: the peers' notes in the own mailbox (k_ipc_wait): nothing is copied into a peer that has given up.
__global__ void __launch_bounds__(256) k_ipc_copy(const IpcCopies c, const uint8_t* src, uint64_t bytes, const uint32_t* abort, const uint32_t* gave_up) {
    if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    uint8_t* dst = c.dst[blockIdx.y];
    if (!dst) return;
    if (__hip_atomic_load(gave_up + blockIdx.y, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0u;
    const uint64_t whole = aligned ? (bytes & ~(uint64_t)15) : 0u;
    // four 16-byte loads in flight per lane and step (the stores over the link are posted; the loads are what a step waits for)
    constexpr uint64_t kStep = 256u * 16u;
    const uint64_t stride = (uint64_t)gridDim.x * kStep * 4u;
    for (uint64_t base = (uint64_t)blockIdx.x * kStep * 4u + (uint64_t)threadIdx.x * 16u; base < whole; base += stride) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + k * kStep < whole) v[k] = *reinterpret_cast<const uint4*>(src + base + k * kStep);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + k * kStep < whole) *reinterpret_cast<uint4*>(dst + base + k * kStep) = v[k];
    }
    for (uint64_t o = whole + (uint64_t)blockIdx.x * 256u + threadIdx.x; o < bytes; o += (uint64_t)gridDim.x * 256u) dst[o] = src[o];
}

hipError_t launch_ipc_signal(const IpcPeers& peers, uint32_t value, const uint32_t* abort, hipStream_t st) {
    hipLaunchKernelGGL(k_ipc_signal, dim3(1), dim3(64), 0, st, peers, value, abort);
    return hipGetLastError();
}
hipError_t launch_ipc_wait(const IpcPeers& own, uint32_t value, uint32_t* abort, uint32_t* timed_out, const IpcPeers& notes, hipStream_t st) {
    hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, st, own, value, abort, timed_out, notes);
    return hipGetLastError();
}
hipError_t launch_ipc_copy(const IpcCopies& c, int world, const uint8_t* src, uint64_t bytes, const uint32_t* abort, const uint32_t* gave_up, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    // a few workgroups per peer keep a link busy without taking the chip from the frame that is being shaded beside the exchange
    const uint64_t chunks = (bytes + 4u * 256u * 16u - 1) / (4u * 256u * 16u);
    const uint32_t gx = (uint32_t)(chunks < 32u ? chunks : 32u);  // per peer; grid-stride beyond
    hipLaunchKernelGGL(k_ipc_copy, dim3(gx, (uint32_t)world), dim3(256), 0, st, c, src, bytes, abort, gave_up);
    return hipGetLastError();
}

}  // namespace sah
