// Context object behind the C ABI and small shared helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/sah_hip.h"
#include "params.hpp"
#include "rt_args.hpp"

// Context-wide device state that entry points build once and re-use (gather copies, tables, scratch) is written and read on whatever
// stream is current.  A caller that moves a pass to another stream (sah_set_stream) must not see a table before its build kernel has
// finished, nor overwrite one the previous stream still reads: sah_set_stream records an event behind the old stream's work for every
// group of such state that was last touched there, and the next entry point that touches the group on a different stream waits for it.
// Single-stream use costs nothing; a caller that keeps each pass on its own stream (androidrenderer_amd/chain.py) never waits.
struct SahCacheGuard {
    hipEvent_t done = nullptr;   // behind the last use on `last`, recorded when the context left that stream
    hipStream_t last = nullptr;
    bool used = false, closed = false;
};

struct sah_ctx {
    int device = 0;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    float* luts = nullptr;  // device: 256 sRGB->linear + 256 UNORM8->float
    uint32_t* probe_slots = nullptr;  // device, 32^3 words (sah_probe_update)
    hipEvent_t probe_done = nullptr;  // behind the last sah_probe_update (its clear pass leaves probe_slots all zero again)
    hipStream_t probe_stream = nullptr;  // the stream that update ran on: an update on another stream waits for probe_done first
    bool lpv_tables_built = false;  // lpv.hip: c_prop_tables of this device filled (first sah_lpv_propagate)
    bool lpv_hot_structure = false;  // ... and, read back, they have the structure the hot form of the propagation relies on (lpv.hip)
    void* comm = nullptr;   // ncclComm_t
    void* comm_reversed = nullptr;  // ncclComm_t with rank world - 1 - rank (sah_allgather_rows_reversed), made on first use
    void* rccl = nullptr;   // dlopen handle
    hipStream_t comm_stream = nullptr;  // optional side stream of the exchange step (sah_comm_set_stream); not owned
    hipEvent_t comm_ready = nullptr, comm_done = nullptr;
    bool comm_pending = false;          // a gather on comm_stream has not been joined by sah_comm_wait yet
    int force_ppt = 0;      // tuning/testing hook: 0 = auto
    bool force_general = false;  // testing hook: always run the general kernel
    sah::FrameState* state = nullptr;  // device
    uint32_t* list = nullptr;          // device: deferred-pixel list
    size_t list_bytes = 0;
    uint8_t* irr32 = nullptr;          // device: fp32 copy of the probe irradiance atlas (lighting_tiled.hip: k_probe_irr_unpack)
    size_t irr32_bytes = 0;
    uint32_t irr32_generation = 0;     // sah_gi::probe_generation the copy was built for (0: not reusable)
    sah::VolumeArg irr32_source = {};
    uint8_t* lpv_packed = nullptr;     // device: per-frame interleaved, zero-bordered copy of the three LPV volumes (lighting.hip)
    size_t lpv_packed_bytes = 0;
    uint32_t lpv_pack_generation = 0;  // sah_gi::lpv_generation the gather copy was built for (0: not reusable)
    sah::VolumeArg lpv_pack_source[3] = {};
    // the volume extent whose two-texel border of the gather copy holds zeros ({0,0,0} + lpv_pack_all_zero: a new allocation, zero everywhere).
    // The buffer is grow-only and shared by every extent it is asked for: k_lpv_pack writes the border itself, the emitting propagation step
    // writes interior texels only and needs the border of ITS extent to be zero already (sah_lpv_pack_borders_for)
    uint32_t lpv_pack_extent[3] = {0, 0, 0};
    bool lpv_pack_all_zero = false;
    float* colx_table = nullptr;       // device: per-column view-space x numerators of the fast kernel, two flavours (lighting.hip: k_colx_table)
    uint32_t colx_capacity = 0, colx_width = 0;
    float colx_key[7] = {};            // render_resolution, p0, p12, p5, p13, height the tables were built for
    // Raised whenever something changes that a launch of sah_lighting / sah_tonemap_ex DEPENDS on beyond its arguments: a context buffer is
    // reallocated, a table is rebuilt for other extents, a gather copy that calls were re-using is dropped or has to be rebuilt.  While it
    // stands, the same call enqueues the same kernels with the same kernel arguments — what sah_chain's captured graphs rely on (api_chain.cpp).
    uint64_t cache_epoch = 0;
    uint32_t hint_slot = 0;  // FrameState::deferred_hint word of the next fast-path Lighting call (the calls of a context are serialised: guard_lighting)
    uint32_t dbg_lpv_packs = 0, dbg_irr_unpacks = 0;  // full rebuilds of the two gather copies by sah_lighting (debug hook sah_debug_copy_rebuilds)
    const uint16_t* last_seg_count = nullptr;  // debug hook (sah_debug_deferred_pixels)
    uint32_t last_num_segments = 0;
    float* tm_thresholds = nullptr;    // device: 256 tonemap code thresholds + the first-level bucket table (api_post.cpp)
    float* tm_code_table = nullptr;    // device: the same search as one float4 per bucket (TonemapArgs::code_table)
    uint32_t tm_bucket_base = 0, tm_bucket_count = 0;
    float tm_thr_lo = 0.f, tm_thr_hi = 0.f;
    void* tm_axis = nullptr;           // device: axis set-ups of the tolerance-mode composite (tonemap_tol.hip), rebuilt when the extents change
    size_t tm_axis_bytes = 0;
    uint32_t tm_axis_key[2 + 2 * 8 + 1] = {};  // output extent, mip extents, number of mips
    struct RasterScratch {             // device buffers of the scene rasteriser, grown on demand (api_raster.cpp)
        void* ptr[16] = {};
        size_t bytes[16] = {};
        uint8_t* half_to_srgb8 = nullptr;
        uint32_t* host_counters = nullptr;  // pinned, 16 words
    } raster;
    struct RtState {                   // acceleration structure of sah_rt_build (api_rt.cpp); buffers grow on demand
        void* ptr[7] = {};             // tri_base, build state, unsorted triangles, sorted triangles, keys, nodes, noise directions
        size_t bytes[7] = {};
        sah::RtBvh bvh = {};
        sah::RtScene scene = {};
        bool built = false;
        uint32_t row_begin = 0, row_end = 0;  // sah_rt_set_rows: the output rows the per-pixel ray generators write ((0, 0) = all)
        uint32_t num_bounces = 0;             // sah_rt_set_bounces: remaining_bounces of the GI generators' rays
    } rt;
    struct IpcState {                  // direct exchange (api_ipc.cpp): mailboxes and peer mappings
        bool open = false, connected = false;
        uint32_t* mailbox = nullptr;   // own: ready[SAH_IPC_MAX_BUFFERS][SAH_IPC_MAX_WORLD] then done[..][..], then the abort word; fine-grained device memory
        uint32_t* peer_mailbox[SAH_IPC_MAX_WORLD] = {};
        uint32_t* timed_out = nullptr; // pinned host word the wait kernel raises
        struct Mapping {               // one opened IPC handle (an allocation of a peer)
            unsigned char handle[64];
            void* base;
            uint32_t users;            // registered buffers (and the mailbox connection) that lie in it
        };
        std::vector<Mapping> mappings;
        struct Buffer {
            bool in_use = false;
            uint8_t* local = nullptr;
            uint64_t bytes = 0;
            uint8_t* peer[SAH_IPC_MAX_WORLD] = {};
            unsigned char peer_handle[SAH_IPC_MAX_WORLD][SAH_IPC_HANDLE_BYTES] = {};  // what sah_ipc_unregister gives back
            uint32_t seq = 0;          // gathers made through this index (mailbox counters only ever grow, whoever uses the index)
        } buffers[SAH_IPC_MAX_BUFFERS];
    } ipc;
    SahCacheGuard guard_lighting, guard_tonemap, guard_raster, guard_rt;  // see SahCacheGuard
    uint32_t raster_merge_cap = 2048;  // tiles whose bin list may be split (testing hook SAH_RASTER_MERGE_CAPACITY: 0 = every list whole)
    std::string last_error;
};

// roctx range around every C-ABI entry point (the equivalent of the Tracy zones the reference puts around every pass,
// RenderCore/render/backend/render_graph.cpp:102-103,188): `rocprofv3 --marker-trace` then shows the pass structure of a frame.
// Opt-in (SAH_ROCTX=1): the marker library is loaded at first use and nothing is linked against it.
struct SahRange {
    using push_fn = int (*)(const char*);
    using pop_fn = int (*)();
    static void resolve(push_fn& push, pop_fn& pop);
    explicit SahRange(const char* name) {
        static push_fn push = nullptr;
        static pop_fn pop_ = nullptr;
        static bool tried = false;
        if (!tried) {
            tried = true;
            resolve(push, pop_);
        }
        pop = pop_;
        if (push) push(name);
    }
    ~SahRange() {
        if (pop) pop();
    }
    pop_fn pop = nullptr;
};
#define SAH_RANGE() SahRange sah_range_(__func__)

inline int fail(sah_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return code;
}

// the entry point is about to touch the guarded state on ctx->stream
inline hipError_t sah_guard_touch(sah_ctx* ctx, SahCacheGuard& g) {
    hipError_t e = hipSuccess;
    if (g.used && g.closed && g.last != ctx->stream) e = hipStreamWaitEvent(ctx->stream, g.done, 0);
    g.used = true;
    g.closed = false;
    g.last = ctx->stream;
    return e;
}
// the context leaves ctx->stream (sah_set_stream); `drained`: that stream has been synchronised, nothing of it is pending
inline hipError_t sah_guard_leave(sah_ctx* ctx, SahCacheGuard& g, bool drained) {
    if (!g.used || g.closed || g.last != ctx->stream) return hipSuccess;
    if (drained) {
        g.used = false;
        return hipSuccess;
    }
    hipError_t e = hipSuccess;
    if (!g.done) e = hipEventCreateWithFlags(&g.done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(g.done, ctx->stream);
    g.closed = e == hipSuccess;
    return e;
}

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(ctx, SAH_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline void sah_drop_lpv_copy(sah_ctx* ctx) {  // the volumes change: the Lighting pass's gather copy of them is stale
    if (ctx->lpv_pack_generation != 0) ctx->cache_epoch++;
    ctx->lpv_pack_generation = 0;
}
inline void sah_drop_irr32_copy(sah_ctx* ctx) {  // the same for the fp32 copy of an irradiance atlas
    if (ctx->irr32_generation != 0) ctx->cache_epoch++;
    ctx->irr32_generation = 0;
}

// The gather copy of the LPV volumes (params.hpp: FastArgs::lpv_packed): its geometry for a volume extent, and the context's grow-only buffer.
// A NEW buffer is zeroed on ctx->stream: k_lpv_pack writes the border texels itself, the emitting propagation step (lpv.hip) relies on them
// being zero already.
struct SahLpvPackLayout {
    uint32_t row_pitch, slice_pitch;
    uint64_t total;
};
inline SahLpvPackLayout sah_lpv_pack_layout(uint32_t w, uint32_t h, uint32_t d) {
    const uint64_t row = (uint64_t)(w + 2 * sah::kLpvPackBorder) * sah::kLpvPackTexel;
    const uint64_t slice = row * (h + 2 * sah::kLpvPackBorder);
    return {(uint32_t)row, (uint32_t)slice, slice * (d + 2 * sah::kLpvPackBorder) + 64};  // + slack: the last x-pair is read as 48 bytes
}
inline hipError_t sah_lpv_pack_reserve(sah_ctx* ctx, uint64_t total) {
    if (ctx->lpv_packed_bytes >= total) return hipSuccess;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return e;
    if (ctx->lpv_packed) (void)hipFree(ctx->lpv_packed);
    ctx->lpv_packed = nullptr;
    ctx->lpv_packed_bytes = 0;
    ctx->lpv_pack_generation = 0;
    ctx->cache_epoch++;
    e = hipMalloc((void**)&ctx->lpv_packed, total);
    if (e != hipSuccess) return e;
    ctx->lpv_packed_bytes = total;
    ctx->lpv_pack_extent[0] = ctx->lpv_pack_extent[1] = ctx->lpv_pack_extent[2] = 0;
    ctx->lpv_pack_all_zero = true;
    return hipMemsetAsync(ctx->lpv_packed, 0, total, ctx->stream);
}
// k_lpv_pack has been enqueued for a w x h x d volume: interior and border of that layout are its own
inline void sah_lpv_pack_written_by_pack(sah_ctx* ctx, uint32_t w, uint32_t h, uint32_t d) {
    ctx->lpv_pack_extent[0] = w, ctx->lpv_pack_extent[1] = h, ctx->lpv_pack_extent[2] = d;
    ctx->lpv_pack_all_zero = false;
}
// The emitting propagation step is about to store the interior texels of a w x h x d layout: the border texels of THAT layout must hold zeros.
// They do in a new allocation and after any writer of the same extent; a buffer last laid out for another extent (a context that lit a
// 128 x 32 x 32 volume and now propagates three cascades: the grow-only buffer is not reallocated) holds old interior texels where the new
// border lies — cleared here, on ctx->stream, ahead of the steps.
inline hipError_t sah_lpv_pack_borders_for(sah_ctx* ctx, uint32_t w, uint32_t h, uint32_t d, uint64_t total) {
    const bool same = ctx->lpv_pack_extent[0] == w && ctx->lpv_pack_extent[1] == h && ctx->lpv_pack_extent[2] == d;
    hipError_t e = hipSuccess;
    if (!same && !ctx->lpv_pack_all_zero) e = hipMemsetAsync(ctx->lpv_packed, 0, total, ctx->stream);
    ctx->lpv_pack_extent[0] = w, ctx->lpv_pack_extent[1] = h, ctx->lpv_pack_extent[2] = d;
    ctx->lpv_pack_all_zero = false;
    return e;
}

// Uniform sub-expressions of sky_unified.slang:80-135 for a sun direction as get_sky_color() receives it (`sun_dir`): the Lighting pass's sky
// fill passes -normalize(direction) (:199), the GI miss shader the raw direction (:229).  Evaluated here in fp32, operator by operator (this
// header is compiled with -ffp-contract=off), exactly as the per-pixel code would.  Returns false when the LUTs are not RGBA16F.
inline bool fill_sky_args(const sah_sky_luts& s, const float sun_dir[3], sah::SkyArgs* sky) {
    if (!s.transmittance.ptr || !s.sky_view.ptr || s.transmittance.format != SAH_FORMAT_R16G16B16A16_SFLOAT || s.sky_view.format != SAH_FORMAT_R16G16B16A16_SFLOAT)
        return false;
    auto cross3 = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - b[1] * a[2];
        o[1] = a[2] * b[0] - b[2] * a[0];
        o[2] = a[0] * b[1] - b[0] * a[1];
    };
    sky->enabled = 1;
    sky->transmittance = sah::PlaneArg{(const uint8_t*)s.transmittance.ptr, s.transmittance.row_pitch_bytes};
    sky->sky_view = sah::PlaneArg{(const uint8_t*)s.sky_view.ptr, s.sky_view.row_pitch_bytes};
    sky->t_w = s.transmittance.width;
    sky->t_h = s.transmittance.height;
    sky->s_w = s.sky_view.width;
    sky->s_h = s.sky_view.height;
    const float sky_pi = 3.14159265358f;
    const float ground = 6.360f;
    for (int i = 0; i < 3; i++) sky->sun_dir[i] = sun_dir[i];
    sky->view_pos_y = 6.360f + 0.0002f;
    sky->height = __builtin_sqrtf((0.0f * 0.0f + sky->view_pos_y * sky->view_pos_y) + 0.0f * 0.0f);
    sky->up_y = sky->view_pos_y / sky->height;
    {
        float q = __builtin_sqrtf(sky->height * sky->height - ground * ground) / sky->height;
        q = __builtin_fminf(__builtin_fmaxf(q, -1.0f), 1.0f);
        sky->horizon_angle = (float)__builtin_acos((double)q);
    }
    sky->azimuth_limit = 0.5f * sky_pi - .0001f;
    sky->min_sun_cos = (float)__builtin_cos((double)(0.53f * sky_pi / 180.0f));
    const float up[3] = {0.0f / sky->height, sky->up_y, 0.0f / sky->height};
    cross3(sky->sun_dir, up, sky->right);
    cross3(up, sky->right, sky->forward);
    sky->smooth_e0 = (float)(_Float16)0.002f;
    return true;
}

inline uint32_t format_bpp(uint32_t f) {
    switch (f) {
        case SAH_FORMAT_R8_UNORM: return 1;
        case SAH_FORMAT_R16_SFLOAT: case SAH_FORMAT_D16_UNORM: return 2;
        case SAH_FORMAT_R8G8B8A8_UNORM: case SAH_FORMAT_R8G8B8A8_SRGB: case SAH_FORMAT_R16G16_SFLOAT: case SAH_FORMAT_R32_SFLOAT:
        case SAH_FORMAT_B10G11R11_UFLOAT_PACK32: case SAH_FORMAT_D32_SFLOAT: return 4;
        case SAH_FORMAT_R16G16B16A16_SFLOAT: return 8;
    }
    return 0;
}

inline bool plane_ok(const sah_plane* p, uint32_t fmt_a, uint32_t fmt_b, uint32_t w, uint32_t h) {
    if (!p || !p->ptr) return false;
    if (p->format != fmt_a && p->format != fmt_b) return false;
    if (p->width != w || p->height != h) return false;
    return (uint64_t)p->row_pitch_bytes >= (uint64_t)w * format_bpp(p->format);
}

// (field by field: VolumeArg has four bytes of padding behind its last member, which aggregate initialisation leaves unspecified — two
// descriptors of the same volume built on different paths need not be memcmp-equal)
inline bool same_volume(const sah::VolumeArg& a, const sah::VolumeArg& b) {
    return a.ptr == b.ptr && a.width == b.width && a.height == b.height && a.depth == b.depth && a.row_pitch == b.row_pitch && a.slice_pitch == b.slice_pitch;
}

inline sah::PlaneArg parg(const sah_plane* p) { return sah::PlaneArg{p ? (const uint8_t*)p->ptr : nullptr, p ? p->row_pitch_bytes : 0}; }
inline sah::VolumeArg varg(const sah_volume& v) {
    return sah::VolumeArg{(const uint8_t*)v.ptr, v.width, v.height, v.depth, v.row_pitch_bytes, v.slice_pitch_bytes};
}
