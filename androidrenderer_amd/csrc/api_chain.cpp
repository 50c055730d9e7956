// sah_chain_*: one rank's share of the row-sharded frame, two frames in flight, as a loop inside the library (include/sah_hip.h).
// The order of enqueues, waits and exchanges is the one androidrenderer_amd/chain.py: PipelinedChain defines (and tests/test_shard_chain.py
// checks against the unsharded chain); what moves here is the host side of it: eight to ten entry points, three stream switches and six
// event operations per frame cost 72 us when each crossed ctypes, against 0.12 ms of GPU work per rank at eight ranks.
// No reference counterpart: the reference has one queue and one device (RenderCore/render/backend/render_backend.cpp:135-153).
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

#include "../../include/sah_hip.h"
#include "ctx.hpp"

namespace {

// a sah_lighting_desc with everything it points to on the host, owned
struct OwnedLighting {
    bool used = false;
    sah_lighting_desc d = {};
    sah_gbuffer gbuffer = {};
    sah_plane ao = {}, lit = {}, shadow_mask = {};
    sah_view_data view = {};
    sah_sun_light_constants sun = {};
    sah_volume shadowmap = {};
    sah_light_list lights = {};
    sah_gi gi = {};
    sah_lpv_cascade_matrices cascades[4] = {};
    sah_sky_luts sky = {};

    bool take(const sah_lighting_desc* src) {
        if (!src) return true;
        if (!src->gbuffer || !src->lit || !src->view || !src->sun) return false;
        used = true;
        d = *src;
        gbuffer = *src->gbuffer;
        d.gbuffer = &gbuffer;
        lit = *src->lit;
        d.lit = &lit;
        view = *src->view;
        d.view = &view;
        sun = *src->sun;
        d.sun = &sun;
        if (src->ao) { ao = *src->ao; d.ao = &ao; }
        if (src->shadow_mask) { shadow_mask = *src->shadow_mask; d.shadow_mask = &shadow_mask; }
        if (src->shadowmap) { shadowmap = *src->shadowmap; d.shadowmap = &shadowmap; }
        if (src->lights) { lights = *src->lights; d.lights = &lights; }
        if (src->sky) { sky = *src->sky; d.sky = &sky; }
        if (src->gi) {
            gi = *src->gi;
            if (gi.lpv_cascades) {
                if (gi.lpv_num_cascades > 4) return false;
                for (uint32_t i = 0; i < gi.lpv_num_cascades; i++) cascades[i] = src->gi->lpv_cascades[i];
                gi.lpv_cascades = cascades;
            }
            d.gi = &gi;
        }
        return true;
    }
};

// One half of a frame (everything between two exchanges) as a captured HIP graph.  A half is enqueued call by call the first time its buffer
// set is used — the context's grow-only buffers, tables and gather copies come into being then —, captured the second time, and replayed
// from then on for as long as sah_ctx::cache_epoch stands (ctx.hpp: while it does, the same calls enqueue the same kernels with the same
// arguments).  Anything that moves the epoch — another user of the context, a dropped gather copy, other extents — sends the half back to
// direct calls and a fresh capture.  One replay costs the host what one launch costs it; a half is three to five launches.
struct HalfGraph {
    hipGraphExec_t exec = nullptr;
    uint64_t epoch = 0;
    uint32_t direct_runs = 0;
};

struct FrameSet {
    HalfGraph graph_l, graph_r, graph_b;
    OwnedLighting lighting[2];
    sah_plane lit = {}, antialiased = {}, mip1 = {}, out = {};
    sah_mipchain bloom = {};
    // behind the lighting / the reduction (copy, mip 0 and mip 1 rows) / the mip-1 gather / the final gather / the B part of the frame that last used the set
    hipEvent_t l_done = nullptr, r_done = nullptr, mip_done = nullptr, final_done = nullptr, b_done = nullptr;
    bool r_valid = false, mip_valid = false, final_valid = false, b_valid = false;
};

}  // namespace

struct sah_chain {
    sah_ctx* ctx = nullptr;
    sah_chain_plan plan = {};
    FrameSet sets[2];
    uint32_t tonemap_flags = 0;
    bool exchange = true;
    bool capture = false, capture_failed = false;  // SAH_CHAIN_CAPTURE; a capture that did not work out: direct calls from then on
    uint64_t replays = 0, captures = 0;
    std::string capture_note;  // why the capture failed (sah_debug_chain_graphs leaves it in the context's last error)
    hipStream_t work = nullptr, reduce = nullptr, post = nullptr;  // reduce == work / post == reduce: that part shares the stream of the one before it
    uint64_t submitted = 0, finished = 0;
};

#define CHAIN_TRY(expr)                  \
    do {                                 \
        const int rc_ = (expr);          \
        if (rc_ != SAH_OK) return rc_;   \
    } while (0)

static bool rows_nonempty(const uint32_t r[2]) { return r[1] > r[0]; }

// `body`: the direct calls of one half, enqueued on `st` (the context's current stream).  See HalfGraph.
// `guard`: the guarded context state the half's calls touch (ctx.hpp: SahCacheGuard), or null.  A replay enqueues the half's kernels without
// going through the entry points that would mark the guard, so it is marked here: a later user of that state on another stream is then
// ordered behind the replay by sah_set_stream's event like behind direct calls.
// An executable graph is destroyed only behind a synchronisation of the stream it was launched on: whether the runtime keeps an exec alive
// while a launch of it is in flight is not documented for HIP, and the two paths that retire one (a stale capture, a capture whose calls
// rebuilt something) are rare.
template <class Body> static int run_half(sah_chain* c, HalfGraph& g, hipStream_t st, bool may_replay, SahCacheGuard* guard, Body body) {
    sah_ctx* ctx = c->ctx;
    if (!c->capture || c->capture_failed || !may_replay || st == nullptr) return body();  // (the null stream cannot be captured)
    if (g.exec && g.epoch == ctx->cache_epoch) {
        if (guard) HIP_TRY(ctx, sah_guard_touch(ctx, *guard));
        HIP_TRY(ctx, hipGraphLaunch(g.exec, st));
        c->replays++;
        return SAH_OK;
    }
    if (g.exec) {  // stale: something the launches depend on has changed since the capture
        (void)hipStreamSynchronize(st);  // (earlier replays of it may still be running)
        (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        g.direct_runs = 0;
    }
    if (g.direct_runs == 0) {
        g.direct_runs++;
        return body();
    }
    const uint64_t epoch_before = ctx->cache_epoch;
    const uint32_t hint_before = ctx->hint_slot;  // (host state a captured-but-never-run Lighting call has advanced: put back before the half is run directly)
    if (const hipError_t eb = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed); eb != hipSuccess) {
        (void)hipGetLastError();
        c->capture_failed = true;
        c->capture_note = std::string("hipStreamBeginCapture: ") + hipGetErrorString(eb);
        return body();
    }
    const int rc = body();
    const std::string body_error = rc != SAH_OK ? ctx->last_error : std::string();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    hipGraphExec_t exec = nullptr;
    hipError_t ei = hipSuccess;
    if (rc != SAH_OK || e != hipSuccess || !graph || (ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess) {
        // (a call that cannot be captured — an allocation, a synchronous copy: nothing of the half has run; run it now, directly, and stay direct)
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        c->capture_failed = true;
        c->capture_note = rc != SAH_OK ? "a captured call failed: " + body_error
                                       : (e != hipSuccess ? std::string("hipStreamEndCapture: ") + hipGetErrorString(e) : std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
        ctx->hint_slot = hint_before;
        return body();
    }
    (void)hipGraphDestroy(graph);
    HIP_TRY(ctx, hipGraphLaunch(exec, st));
    c->captures++;
    if (ctx->cache_epoch != epoch_before) {  // the captured calls (re)built something: a replay would rebuild it every time.  Used once; captured again next time
        (void)hipStreamSynchronize(st);
        (void)hipGraphExecDestroy(exec);
        return SAH_OK;
    }
    g.exec = exec;
    g.epoch = ctx->cache_epoch;
    return SAH_OK;
}

// the stream the exchanges are enqueued on (allgather_bytes_impl, api_post.cpp): the side stream if there is one
static hipStream_t exchange_stream(const sah_ctx* ctx) { return (ctx->comm_stream && ctx->comm_stream != ctx->stream) ? ctx->comm_stream : ctx->stream; }

// A frame in three parts, each on its stream (streams may coincide):
//   L(i)  lighting of the rank's rows                                      work stream
//   R(i)  copy scene + bloom mip 0 over its band, its rows of mip 1        reduce stream      -> exchange of mip 1
//   B(i)  mips 2.., the composite of its rows                              post stream        -> exchange of the final image
// Everything a frame writes exists twice (sets[i & 1]); what a part overwrites was last read by a part of frame i - 2:
//   L(i) writes lit                 <- read by R(i - 2)
//   R(i) writes antialiased, mips   <- read by B(i - 2) (which had waited for the mip-1 gather of frame i - 2)
//   B(i) writes mips 2.., out       <- out read by the final gather of frame i - 2
// On ONE stream that is stream order; between streams it is the events below.  sah_chain_submit enqueues L(i), R(i), the mip-1 gather,
// then B(i - 1) and its gather: the gathers travel beside the parts enqueued behind them.  Three streams matter on a rank of eight: the
// lighting of frame i + 1 starts when that of frame i ends instead of behind its copy and mip rows (0.125 -> 0.106 ms per frame for one
// rank's share of the 4K chain, tools/experiments/chain_two_streams.py).

// B(j)
static int chain_finish(sah_chain* c, uint64_t j) {
    sah_ctx* ctx = c->ctx;
    FrameSet& s = c->sets[j & 1];
    hipStream_t st = c->post;
    CHAIN_TRY(sah_set_stream(ctx, (void*)st));  // the library enqueues B(j), and orders its gather, on the post stream
    if (c->post != c->reduce) HIP_TRY(ctx, hipStreamWaitEvent(st, s.r_done, 0));
    // mips 2.. read every rank's rows of mip 1 (the gather ran behind R(j)); the final gather of frame j - 2 still reads this set's image
    if (s.mip_valid) HIP_TRY(ctx, hipStreamWaitEvent(st, s.mip_done, 0));
    if (s.final_valid) HIP_TRY(ctx, hipStreamWaitEvent(st, s.final_done, 0));
    CHAIN_TRY(run_half(c, s.graph_b, st, true, &ctx->guard_tonemap, [&]() -> int {
        CHAIN_TRY(sah_bloom_from_mip(ctx, &s.antialiased, &s.bloom, 1));
        if (rows_nonempty(c->plan.out_rows)) CHAIN_TRY(sah_tonemap_ex(ctx, &s.antialiased, &s.bloom, &s.out, c->plan.out_rows[0], c->plan.out_rows[1], c->tonemap_flags));
        return SAH_OK;
    }));
    if (c->exchange) {
        CHAIN_TRY(sah_allgather_rows_reversed(ctx, &s.out, c->plan.rows_per_rank, c->plan.out_allocated_rows));
        HIP_TRY(ctx, hipEventRecord(s.final_done, exchange_stream(ctx)));
        s.final_valid = true;
    }
    if (c->post != c->reduce) {
        HIP_TRY(ctx, hipEventRecord(s.b_done, st));
        s.b_valid = true;
    }
    c->finished++;
    return SAH_OK;
}

extern "C" {

int sah_chain_create(sah_ctx* ctx, const sah_chain_plan* plan, const sah_chain_frame frames[2], uint32_t tonemap_flags, uint32_t chain_flags,
                     void* work_stream, void* reduce_stream, void* post_stream, sah_chain** out) {
    SAH_RANGE();
    if (!ctx || !plan || !frames || !out) return SAH_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const uint32_t* ranges[] = {plan->aa_rows, plan->mip0_rows, plan->mip1_rows, plan->out_rows};
    for (const uint32_t* r : ranges)
        if (r[1] < r[0]) return fail(ctx, SAH_ERR_INVALID_ARGUMENT, "chain plan: a row range ends before it begins");
    sah_chain* c = new (std::nothrow) sah_chain;
    if (!c) return fail(ctx, SAH_ERR_HIP, "out of host memory");
    c->ctx = ctx;
    c->plan = *plan;
    c->tonemap_flags = tonemap_flags;
    c->exchange = !(chain_flags & SAH_CHAIN_NO_EXCHANGE);
    c->capture = (chain_flags & SAH_CHAIN_CAPTURE) != 0;
    c->work = (hipStream_t)work_stream;
    c->reduce = reduce_stream ? (hipStream_t)reduce_stream : c->work;
    c->post = post_stream ? (hipStream_t)post_stream : c->reduce;
    auto bail = [&](int code, const char* msg) {
        sah_chain_destroy(c);
        return fail(ctx, code, "%s", msg);
    };
    if (hipSetDevice(ctx->device) != hipSuccess) return bail(SAH_ERR_HIP, "hipSetDevice failed");
    if (c->post == c->work && c->reduce != c->work) return bail(SAH_ERR_INVALID_ARGUMENT, "chain streams: the post stream is the reduce stream or a stream of its own");
    for (int k = 0; k < 2; k++) {
        const sah_chain_frame& f = frames[k];
        FrameSet& s = c->sets[k];
        if (!f.lighting[0] || !s.lighting[0].take(f.lighting[0]) || !s.lighting[1].take(f.lighting[1]))
            return bail(SAH_ERR_INVALID_ARGUMENT, "chain frame: lighting[0] (with gbuffer, lit, view, sun) is required; at most 4 LPV cascades");
        if (!f.lit.ptr || !f.antialiased.ptr || !f.out.ptr || f.bloom.num_mips < 2 || f.bloom.num_mips > SAH_MAX_BLOOM_MIPS)
            return bail(SAH_ERR_INVALID_ARGUMENT, "chain frame: lit, antialiased, out and a bloom chain of at least two mips are required");
        s.lit = f.lit;
        s.antialiased = f.antialiased;
        s.bloom = f.bloom;
        s.mip1 = f.bloom.mips[1];
        s.out = f.out;
        if (c->exchange && ((uint64_t)plan->mip1_rows_per_rank * ctx->world > plan->mip1_allocated_rows || (uint64_t)plan->rows_per_rank * ctx->world > plan->out_allocated_rows))
            return bail(SAH_ERR_INVALID_ARGUMENT, "chain plan: an allocation holds fewer rows than its gather's equal slots (slot rows * world)");
        for (hipEvent_t* e : {&s.l_done, &s.r_done, &s.mip_done, &s.final_done, &s.b_done})
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return bail(SAH_ERR_HIP, "hipEventCreateWithFlags failed");
    }
    const int rc = sah_set_stream(ctx, (void*)c->work);
    if (rc != SAH_OK) {
        sah_chain_destroy(c);
        return rc;
    }
    *out = c;
    return SAH_OK;
}

int sah_chain_submit(sah_chain* c, void* lighting_begin, void* lighting_end) {
    SAH_RANGE();
    if (!c) return SAH_ERR_INVALID_ARGUMENT;
    sah_ctx* ctx = c->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const sah_chain_plan& p = c->plan;
    FrameSet& s = c->sets[c->submitted & 1];
    // ---- L(i)
    CHAIN_TRY(sah_set_stream(ctx, (void*)c->work));
    if (c->reduce != c->work && s.r_valid) HIP_TRY(ctx, hipStreamWaitEvent(c->work, s.r_done, 0));  // lit of this set: last read by R(i - 2)
    // (a caller that wants the Lighting pass timed gets it call by call: its events are not part of a captured graph)
    CHAIN_TRY(run_half(c, s.graph_l, c->work, !lighting_begin && !lighting_end, &ctx->guard_lighting, [&]() -> int {
        if (lighting_begin) HIP_TRY(ctx, hipEventRecord((hipEvent_t)lighting_begin, c->work));
        for (OwnedLighting& l : s.lighting)
            if (l.used) CHAIN_TRY(sah_lighting(ctx, &l.d));
        if (lighting_end) HIP_TRY(ctx, hipEventRecord((hipEvent_t)lighting_end, c->work));
        return SAH_OK;
    }));
    // ---- R(i)
    if (c->reduce != c->work) {
        HIP_TRY(ctx, hipEventRecord(s.l_done, c->work));
        CHAIN_TRY(sah_set_stream(ctx, (void*)c->reduce));
        HIP_TRY(ctx, hipStreamWaitEvent(c->reduce, s.l_done, 0));
    }
    if (c->post != c->reduce && s.b_valid) HIP_TRY(ctx, hipStreamWaitEvent(c->reduce, s.b_done, 0));  // antialiased / mips of this set: last read by B(i - 2)
    CHAIN_TRY(run_half(c, s.graph_r, c->reduce, true, nullptr, [&]() -> int {
        if (rows_nonempty(p.aa_rows) && rows_nonempty(p.mip0_rows)) {  // one pass over lit: antialiased rows + mip 0 rows
            CHAIN_TRY(sah_copy_scene_bloom_mip0_rows(ctx, &s.lit, &s.antialiased, &s.bloom, p.aa_rows[0], p.aa_rows[1], p.mip0_rows[0], p.mip0_rows[1]));
        } else {
            if (rows_nonempty(p.aa_rows)) CHAIN_TRY(sah_copy_scene_rows(ctx, &s.lit, &s.antialiased, p.aa_rows[0], p.aa_rows[1]));
            if (rows_nonempty(p.mip0_rows)) CHAIN_TRY(sah_bloom_mip_rows(ctx, &s.antialiased, &s.bloom, 0, p.mip0_rows[0], p.mip0_rows[1]));
        }
        if (rows_nonempty(p.mip1_rows)) CHAIN_TRY(sah_bloom_mip_rows(ctx, &s.antialiased, &s.bloom, 1, p.mip1_rows[0], p.mip1_rows[1]));
        return SAH_OK;
    }));
    if (c->reduce != c->work || c->post != c->reduce) {  // (one rank without a communicator gathers nothing: B(i) is then ordered behind R(i) by this alone)
        HIP_TRY(ctx, hipEventRecord(s.r_done, c->reduce));
        s.r_valid = true;
    }
    if (c->exchange) {
        CHAIN_TRY(sah_allgather_rows(ctx, &s.mip1, p.mip1_rows_per_rank, p.mip1_allocated_rows));  // side stream, behind R(i)
        HIP_TRY(ctx, hipEventRecord(s.mip_done, exchange_stream(ctx)));
        s.mip_valid = true;
    }
    c->submitted++;
    // ---- B(i - 1) (done already if a flush came in between)
    while (c->finished + 1 < c->submitted) CHAIN_TRY(chain_finish(c, c->finished));
    CHAIN_TRY(sah_set_stream(ctx, (void*)c->work));
    return SAH_OK;
}

int sah_chain_flush(sah_chain* c) {
    SAH_RANGE();
    if (!c) return SAH_ERR_INVALID_ARGUMENT;
    sah_ctx* ctx = c->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    while (c->finished < c->submitted) CHAIN_TRY(chain_finish(c, c->finished));
    CHAIN_TRY(sah_set_stream(ctx, (void*)c->work));
    for (FrameSet& s : c->sets) {
        if (s.r_valid) HIP_TRY(ctx, hipStreamWaitEvent(c->work, s.r_done, 0));
        if (s.final_valid) HIP_TRY(ctx, hipStreamWaitEvent(c->work, s.final_done, 0));
        if (s.b_valid) HIP_TRY(ctx, hipStreamWaitEvent(c->work, s.b_done, 0));
    }
    return SAH_OK;
}

int sah_chain_counts(const sah_chain* c, uint64_t* submitted, uint64_t* finished) {
    if (!c) return SAH_ERR_INVALID_ARGUMENT;
    if (submitted) *submitted = c->submitted;
    if (finished) *finished = c->finished;
    return SAH_OK;
}

// test / analysis hook: out[0] = graph replays, out[1] = captures so far, out[2] = 1 when a capture failed and the chain went back to direct calls
int sah_debug_chain_graphs(const sah_chain* c, uint64_t out[3]) {
    if (!c || !out) return SAH_ERR_INVALID_ARGUMENT;
    out[0] = c->replays;
    out[1] = c->captures;
    out[2] = c->capture_failed ? 1u : 0u;
    if (c->capture_failed) c->ctx->last_error = "graph capture: " + c->capture_note;
    return SAH_OK;
}

void sah_chain_destroy(sah_chain* c) {
    if (!c) return;
    bool graphs = false;
    for (FrameSet& s : c->sets) graphs = graphs || s.graph_l.exec || s.graph_r.exec || s.graph_b.exec;
    if (graphs) {  // replays may still be running: an exec is destroyed behind its streams (see run_half)
        if (c->ctx) (void)hipSetDevice(c->ctx->device);
        for (hipStream_t st : {c->work, c->reduce, c->post})
            if (st) (void)hipStreamSynchronize(st);
    }
    for (FrameSet& s : c->sets) {
        for (hipEvent_t e : {s.l_done, s.r_done, s.mip_done, s.final_done, s.b_done})
            if (e) (void)hipEventDestroy(e);
        for (HalfGraph* g : {&s.graph_l, &s.graph_r, &s.graph_b})
            if (g->exec) (void)hipGraphExecDestroy(g->exec);
    }
    delete c;
}

}  // extern "C"
