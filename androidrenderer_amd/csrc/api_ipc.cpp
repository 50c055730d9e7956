// C ABI: the direct exchange backend (include/sah_hip.h "direct exchange"; kernels in ipc.hip).
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/sah_hip.h"
#include "ctx.hpp"

namespace sah {
struct IpcPeers {
    uint32_t* slot[SAH_IPC_MAX_WORLD];
};
struct IpcCopies {
    uint8_t* dst[SAH_IPC_MAX_WORLD];
};
hipError_t launch_ipc_signal(const IpcPeers& peers, uint32_t value, const uint32_t* abort, hipStream_t st);
hipError_t launch_ipc_wait(const IpcPeers& own, uint32_t value, uint32_t* abort, uint32_t* timed_out, const IpcPeers& notes, hipStream_t st);
hipError_t launch_ipc_copy(const IpcCopies& c, int world, const uint8_t* src, uint64_t bytes, const uint32_t* abort, const uint32_t* gave_up, hipStream_t st);
}  // namespace sah

namespace {
constexpr uint32_t kMagic = 0x53414849u;  // "SAHI"
// ready[buffer][rank], done[buffer][rank], then one "rank p has given up" note per peer (k_ipc_wait writes its own into every peer's mailbox)
constexpr uint32_t kNotesBase = 2 * SAH_IPC_MAX_BUFFERS * SAH_IPC_MAX_WORLD;
constexpr uint32_t kMailboxWords = kNotesBase + SAH_IPC_MAX_WORLD;
constexpr uint32_t kMailboxAlloc = kMailboxWords + 16;  // + the abort word (its own 64 bytes; peers never touch it)

struct Handle {  // SAH_IPC_HANDLE_BYTES
    hipIpcMemHandle_t mem;  // 64 bytes: the allocation the buffer lies in
    uint64_t offset, bytes;
    uint32_t magic, rank, device, pad;
    uint8_t reserved[128 - 64 - 16 - 16];
};
static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
static_assert(sizeof(Handle) == SAH_IPC_HANDLE_BYTES, "handle layout");

// The allocation kinds the exchange is defined for (include/sah_hip.h, "direct exchange"): ordinary device memory of the context's own device
// (hipMalloc, or a caching allocator's block of it).  Host-pinned and managed memory are refused: their pages may live on, or migrate to,
// another agent, and nothing in this protocol orders a peer's stores against that.
int check_exchange_memory(sah_ctx* ctx, const void* ptr) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "direct exchange: %p is not memory HIP knows (gathered buffers must be device memory of device %d)", ptr, ctx->device);
    }
    if (at.type != hipMemoryTypeDevice || at.isManaged)
        return fail(ctx, SAH_ERR_UNSUPPORTED, "direct exchange: the buffer is %s memory; the exchange is defined for device memory (hipMalloc) only",
                    at.isManaged ? "managed" : (at.type == hipMemoryTypeHost ? "host" : "not device"));
    if (at.device != ctx->device)
        return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "direct exchange: the buffer lives on device %d, the context on device %d", at.device, ctx->device);
    return SAH_OK;
}

int export_range(sah_ctx* ctx, const void* ptr, uint64_t bytes, Handle* h) {
    void* base = nullptr;
    size_t size = 0;
    HIP_TRY(ctx, hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)ptr));
    const uint64_t off = (uint64_t)((const uint8_t*)ptr - (const uint8_t*)base);
    if (off + bytes > size) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "the buffer leaves its allocation (%llu + %llu > %zu)", (unsigned long long)off, (unsigned long long)bytes, size);
    memset(h, 0, sizeof(*h));
    HIP_TRY(ctx, hipIpcGetMemHandle(&h->mem, base));
    h->offset = off;
    h->bytes = bytes;
    h->magic = kMagic;
    h->rank = (uint32_t)ctx->rank;
    h->device = (uint32_t)ctx->device;
    return SAH_OK;
}

// maps the allocation behind a peer's handle (once per allocation, counted per user) and returns the buffer's address in this process
int open_range(sah_ctx* ctx, const Handle& h, uint8_t** out) {
    for (auto& m : ctx->ipc.mappings)
        if (memcmp(m.handle, &h.mem, 64) == 0) {
            m.users++;
            *out = (uint8_t*)m.base + h.offset;
            return SAH_OK;
        }
    void* base = nullptr;
    HIP_TRY(ctx, hipIpcOpenMemHandle(&base, h.mem, hipIpcMemLazyEnablePeerAccess));
    sah_ctx::IpcState::Mapping m;
    memcpy(m.handle, &h.mem, 64);
    m.base = base;
    m.users = 1;
    ctx->ipc.mappings.push_back(m);
    *out = (uint8_t*)base + h.offset;
    return SAH_OK;
}
// gives back one use of the mapping that holds `ptr` (the mapping is closed with its last user)
void close_range(sah_ctx* ctx, const Handle& h) {
    auto& ms = ctx->ipc.mappings;
    for (size_t i = 0; i < ms.size(); i++)
        if (memcmp(ms[i].handle, &h.mem, 64) == 0) {
            if (--ms[i].users == 0) {
                (void)hipIpcCloseMemHandle(ms[i].base);
                ms.erase(ms.begin() + (long)i);
            }
            return;
        }
}
}  // namespace

extern "C" void sah_ipc_destroy(sah_ctx* ctx) {
    auto& s = ctx->ipc;
    for (const auto& m : s.mappings) (void)hipIpcCloseMemHandle(m.base);
    s.mappings.clear();
    if (s.mailbox) (void)hipFree(s.mailbox);
    if (s.timed_out) (void)hipHostFree(s.timed_out);
    s.mailbox = nullptr;
    s.timed_out = nullptr;
    s.open = s.connected = false;
    for (auto& b : s.buffers) b = {};
}

// a wait of an earlier gather gave up: every exchange entry point fails from then on (api.cpp: sah_sync, api_post.cpp: sah_comm_wait)
bool sah_ipc_timed_out(const sah_ctx* ctx) { return ctx->ipc.timed_out && *ctx->ipc.timed_out != 0; }

// the gather itself: called by allgather_bytes_impl (api_post.cpp) when `buffer` lies inside a registered buffer
int sah_ipc_gather(sah_ctx* ctx, uint32_t id, uint8_t* buffer, uint64_t bytes_per_rank, bool reversed, hipStream_t st) {
    using namespace sah;
    auto& s = ctx->ipc;
    auto& b = s.buffers[id];
    if (sah_ipc_timed_out(ctx)) return fail(ctx, SAH_ERR_COMM, "direct exchange: a peer did not arrive within 2 s (an earlier gather gave up)");
    const uint64_t off = (uint64_t)(buffer - b.local);
    if (off + (uint64_t)ctx->world * bytes_per_rank > b.bytes) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "the gather leaves the registered buffer");
    const uint32_t n = ++b.seq;
    const int slot = reversed ? ctx->world - 1 - ctx->rank : ctx->rank;
    const uint64_t slot_off = off + (uint64_t)slot * bytes_per_rank;
    IpcPeers ready_out{}, ready_in{}, done_out{}, done_in{}, notes{};
    IpcCopies copies{};
    for (int p = 0; p < ctx->world; p++) {
        if (p == ctx->rank) continue;
        ready_out.slot[p] = s.peer_mailbox[p] + id * SAH_IPC_MAX_WORLD + ctx->rank;
        ready_in.slot[p] = s.mailbox + id * SAH_IPC_MAX_WORLD + p;
        done_out.slot[p] = s.peer_mailbox[p] + (SAH_IPC_MAX_BUFFERS + id) * SAH_IPC_MAX_WORLD + ctx->rank;
        done_in.slot[p] = s.mailbox + (SAH_IPC_MAX_BUFFERS + id) * SAH_IPC_MAX_WORLD + p;
        copies.dst[p] = b.peer[p] + slot_off;
        notes.slot[p] = s.peer_mailbox[p] + kNotesBase + ctx->rank;
    }
    uint32_t* abort = s.mailbox + kMailboxWords;
    // 1. my rows are written (stream order) and my copy of the buffer may be overwritten; 2. so may every peer's
    HIP_TRY(ctx, launch_ipc_signal(ready_out, n, abort, st));
    HIP_TRY(ctx, launch_ipc_wait(ready_in, n, abort, s.timed_out, notes, st));
    // 3. one hop per peer (skipped, like everything behind it, once a wait has given up: a peer that did not arrive may still be using
    //    its copy of the buffer)
    HIP_TRY(ctx, launch_ipc_copy(copies, ctx->world, b.local + slot_off, bytes_per_rank, abort, s.mailbox + kNotesBase, st));
    // 4. my rows have landed everywhere; 5. so have everybody's here
    HIP_TRY(ctx, launch_ipc_signal(done_out, n, abort, st));
    HIP_TRY(ctx, launch_ipc_wait(done_in, n, abort, s.timed_out, notes, st));
    return SAH_OK;
}

// index of the registered buffer that holds [ptr, ptr + bytes), or -1 (registrations never overlap: sah_ipc_register)
int sah_ipc_find(const sah_ctx* ctx, const void* ptr, uint64_t bytes) {
    const auto& s = ctx->ipc;
    if (!s.connected) return -1;
    for (uint32_t i = 0; i < SAH_IPC_MAX_BUFFERS; i++) {
        const uint8_t* p = (const uint8_t*)ptr;
        if (s.buffers[i].in_use && p >= s.buffers[i].local && p + bytes <= s.buffers[i].local + s.buffers[i].bytes) return (int)i;
    }
    return -1;
}

extern "C" {

int sah_ipc_open(sah_ctx* ctx, void* out_handle) {
    if (!ctx || !out_handle) return SAH_ERR_INVALID_ARGUMENT;
    if (ctx->world > SAH_IPC_MAX_WORLD) return fail(ctx, SAH_ERR_UNSUPPORTED, "direct exchange: at most %d ranks", SAH_IPC_MAX_WORLD);
    auto& s = ctx->ipc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!s.open) {
        // fine-grained: peers' system-scope stores become visible to the polling wave without a kernel boundary.  No fall-back to
        // ordinary (coarse-grained) device memory: a polling wave might never see a peer's store there, and every gather would time out
        if (hipExtMallocWithFlags((void**)&s.mailbox, kMailboxAlloc * sizeof(uint32_t), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            s.mailbox = nullptr;
            return fail(ctx, SAH_ERR_UNSUPPORTED, "direct exchange: this device does not give fine-grained device memory (hipDeviceMallocFinegrained)");
        }
        HIP_TRY(ctx, hipMemset(s.mailbox, 0, kMailboxAlloc * sizeof(uint32_t)));
        HIP_TRY(ctx, hipHostMalloc((void**)&s.timed_out, 64));
        *s.timed_out = 0;
        s.open = true;
    }
    Handle h;
    if (int rc = export_range(ctx, s.mailbox, kMailboxWords * sizeof(uint32_t), &h); rc != SAH_OK) return rc;  // (the abort word stays private)
    memcpy(out_handle, &h, sizeof(h));
    return SAH_OK;
}

int sah_ipc_connect(sah_ctx* ctx, const void* all_handles) {
    if (!ctx || !all_handles) return SAH_ERR_INVALID_ARGUMENT;
    auto& s = ctx->ipc;
    if (!s.open) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sah_ipc_open first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const Handle* hs = (const Handle*)all_handles;
    for (int p = 0; p < ctx->world; p++) {
        if (hs[p].magic != kMagic || (int)hs[p].rank != p || hs[p].bytes != kMailboxWords * sizeof(uint32_t))
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "handle %d is not rank %d's mailbox", p, p);
        if (p == ctx->rank) {
            s.peer_mailbox[p] = s.mailbox;
            continue;
        }
        uint8_t* ptr = nullptr;
        if (int rc = open_range(ctx, hs[p], &ptr); rc != SAH_OK) return rc;
        s.peer_mailbox[p] = (uint32_t*)ptr;
    }
    s.connected = true;
    return SAH_OK;
}

int sah_ipc_export(sah_ctx* ctx, const void* buffer, uint64_t bytes, void* out_handle) {
    if (!ctx || !buffer || !bytes || !out_handle) return SAH_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = check_exchange_memory(ctx, buffer); rc != SAH_OK) return rc;
    Handle h;
    if (int rc = export_range(ctx, buffer, bytes, &h); rc != SAH_OK) return rc;
    memcpy(out_handle, &h, sizeof(h));
    return SAH_OK;
}

int sah_ipc_register(sah_ctx* ctx, void* buffer, uint64_t bytes, const void* all_handles) {
    if (!ctx || !buffer || !bytes || !all_handles) return SAH_ERR_INVALID_ARGUMENT;
    auto& s = ctx->ipc;
    if (!s.connected) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "sah_ipc_connect first");
    // A registration is looked up by address (sah_ipc_find), so no two may overlap: a buffer that was freed and whose address a caching
    // allocator has handed out again must be unregistered first — the stale entry carries the old peer addresses and counters
    int id = -1;
    for (uint32_t i = 0; i < SAH_IPC_MAX_BUFFERS; i++) {
        const auto& o = s.buffers[i];
        if (!o.in_use) {
            if (id < 0) id = (int)i;  // the lowest free index: the same on every rank when all register / unregister in the same order
            continue;
        }
        if ((uint8_t*)buffer < o.local + o.bytes && o.local < (uint8_t*)buffer + bytes)
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "the buffer overlaps registration %u (sah_ipc_unregister it first)", i);
    }
    if (id < 0) return fail(ctx, SAH_ERR_UNSUPPORTED, "at most %d registered buffers", SAH_IPC_MAX_BUFFERS);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = check_exchange_memory(ctx, buffer); rc != SAH_OK) return rc;
    const Handle* hs = (const Handle*)all_handles;
    for (int p = 0; p < ctx->world; p++)
        if (hs[p].magic != kMagic || (int)hs[p].rank != p || hs[p].bytes != bytes)
            return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "handle %d does not describe rank %d's copy of a %llu-byte buffer", p, p, (unsigned long long)bytes);
    auto& b = s.buffers[id];
    const uint32_t seq = b.seq;  // counters of a mailbox slot only grow between resets: a re-used index goes on counting where the last user stopped
    b = {};
    b.seq = seq;
    b.local = (uint8_t*)buffer;
    b.bytes = bytes;
    for (int p = 0; p < ctx->world; p++) {
        if (p == ctx->rank) {
            b.peer[p] = b.local;
            continue;
        }
        if (int rc = open_range(ctx, hs[p], &b.peer[p]); rc != SAH_OK) {
            for (int q = 0; q < p; q++)
                if (q != ctx->rank) close_range(ctx, hs[q]);
            return rc;
        }
        memcpy(b.peer_handle[p], &hs[p], sizeof(Handle));
    }
    b.in_use = true;
    return SAH_OK;
}

// After a gather has given up (SAH_ERR_COMM): the collective way back.  Every rank drains its streams (sah_sync, whatever it returns),
// all ranks meet (the caller's barrier), every rank calls this, all ranks meet again, and the exchange works as before.
// Between the two barriers nothing of anybody is in flight, so the exchange starts over from zero: every rank clears its WHOLE mailbox
// (arrival counters, give-up notes, abort word) and the sequence number of every registration slot.  (Rounds 4-5 raised each number to
// the highest counter a peer had stored here; a rank that went on enqueueing gathers behind the one that gave up — the ordinary pipelined
// case: the host notices *timed_out two seconds later — has counted them without ever signalling them, so no rank could see that number
// and the ranks came back apart: ADVICE r5.)
int sah_ipc_reset(sah_ctx* ctx) {
    if (!ctx) return SAH_ERR_INVALID_ARGUMENT;
    auto& s = ctx->ipc;
    if (!s.open) return SAH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    (void)hipStreamSynchronize(ctx->stream);
    HIP_TRY(ctx, hipMemset(s.mailbox, 0, kMailboxAlloc * sizeof(uint32_t)));
    for (auto& b : s.buffers) b.seq = 0;
    *s.timed_out = 0;
    ctx->comm_pending = false;
    return SAH_OK;
}

int sah_ipc_unregister(sah_ctx* ctx, const void* buffer) {
    if (!ctx || !buffer) return SAH_ERR_INVALID_ARGUMENT;
    auto& s = ctx->ipc;
    for (uint32_t i = 0; i < SAH_IPC_MAX_BUFFERS; i++) {
        auto& b = s.buffers[i];
        if (!b.in_use || b.local != (const uint8_t*)buffer) continue;
        // nothing of this context may still be copying into the peers' mappings
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int p = 0; p < ctx->world; p++) {
            if (p == ctx->rank) continue;
            Handle h;
            memcpy(&h, b.peer_handle[p], sizeof(Handle));
            close_range(ctx, h);
        }
        const uint32_t seq = b.seq;
        b = {};
        b.seq = seq;
        return SAH_OK;
    }
    return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "no registration starts at %p", buffer);
}

}  // extern "C"
