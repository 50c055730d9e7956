"""One frame of the sharded full chain on one rank: lighting (own rows + halo) -> copy scene + bloom mip 0 rows (one pass) -> its bloom
mip 1 rows -> exchange of mip 1 -> bloom mips 2.. -> tonemap of the rank's final rows -> exchange of the RGBA8 image
(androidrenderer_amd/shard.py has the row arithmetic).  Plumbing shared by bench.py and tests/: buffers are torch tensors, every pass goes through the C ABI."""
from . import _abi, images, shard


def connect_direct_exchange(ctx, allgather):
    """Joins the context to the direct (IPC) exchange of its job.  allgather(b: bytes) -> [bytes of rank 0, bytes of rank 1, ...]: the
    caller's channel between the ranks (torch.distributed.all_gather_object, MPI, files ...)."""
    ctx.ipc_connect(allgather(ctx.ipc_open()))


class ShardedChain:
    def __init__(self, ctx, frame, device_arrays, rank, world, num_mips=6, tonemap_flags=0):
        import torch
        self.ctx, self.frame, self.dev = ctx, frame, device_arrays
        self.tonemap_flags = tonemap_flags  # 0 = strict, _abi.TONEMAP_TOLERANCE_1CODE = within one code of it (include/sah_hip.h)
        W, H = frame.width, frame.height
        self.plan = shard.chain_plan(H, world, rank)
        dev = device_arrays["depth"].device
        p = self.plan
        self.lit = torch.zeros((H, W, 4), dtype=torch.int16, device=dev)
        self.aa = torch.zeros((H, W, 4), dtype=torch.int16, device=dev)
        sizes = images.bloom_mip_sizes(W, H, num_mips)
        # mip 1 and the final image are gathered in place: equal slots, so their allocations are padded to slots * world rows
        self.mip1_alloc = torch.zeros((p.mip1_rows_per_rank * world, sizes[1][0], 4), dtype=torch.int16, device=dev)
        self.mips = [torch.zeros((sizes[0][1], sizes[0][0], 4), dtype=torch.int16, device=dev), self.mip1_alloc[:sizes[1][1]]] + \
                    [torch.zeros((mh, mw, 4), dtype=torch.int16, device=dev) for (mw, mh) in sizes[2:]]
        self.out_alloc = torch.zeros((p.rows_per_rank * world, W, 4), dtype=torch.uint8, device=dev)
        self.out = self.out_alloc[:H]
        self.lit_p = images.plane(self.lit, _abi.FORMAT_R16G16B16A16_SFLOAT)
        self.aa_p = images.plane(self.aa, _abi.FORMAT_R16G16B16A16_SFLOAT)
        self.mc = images.mipchain(self.mips)
        self.mip1_p = images.plane(self.mips[1], _abi.FORMAT_R16G16B16A16_SFLOAT)
        self.out_p = images.plane(self.out, _abi.FORMAT_R8G8B8A8_SRGB)
        self.descs = []
        for rows in ((p.lit_rows, p.lit_wrap_rows) if world > 1 else ((0, 0),)):
            if world > 1 and rows[1] <= rows[0]:
                continue
            frame.row_begin, frame.row_end = rows
            self.descs.append(frame.describe(device_arrays, self.lit))
        frame.row_begin = frame.row_end = 0
        self.world = world

    def register_direct_exchange(self, allgather):
        """Makes the two gathered buffers of this chain (bloom mip 1, final image) visible to the peers: their exchanges then go straight
        into every peer's copy (sah_ipc_register) instead of through RCCL.  Every rank must register its chains in the same order."""
        for t in (self.mip1_alloc, self.out_alloc):
            nbytes = t.numel() * t.element_size()
            self.ctx.ipc_register(t.data_ptr(), nbytes, allgather(self.ctx.ipc_export(t.data_ptr(), nbytes)))

    def unregister_direct_exchange(self):
        """Before the chain's buffers are dropped (every rank, same order): a caching allocator may hand their addresses out again, and a
        registration is found by address."""
        for t in (self.mip1_alloc, self.out_alloc):
            self.ctx.ipc_unregister(t.data_ptr())

    # the three local stages; `exchange_*` are the two gathers (replaceable: tests emulate several ranks on one device)
    def lighting(self):
        for desc, _keep in self.descs:
            self.ctx.lighting(desc)

    def reduce(self):
        p = self.plan
        if p.aa_rows[1] > p.aa_rows[0] and p.mip0_rows[1] > p.mip0_rows[0]:  # one pass over lit: antialiased rows + mip 0 rows
            self.ctx.copy_scene_bloom_mip0(self.lit_p, self.aa_p, self.mc, p.aa_rows, p.mip0_rows)
        else:
            if p.aa_rows[1] > p.aa_rows[0]:
                self.ctx.copy_scene(self.lit_p, self.aa_p, *p.aa_rows)
            if p.mip0_rows[1] > p.mip0_rows[0]:
                self.ctx.bloom_mip_rows(self.aa_p, self.mc, 0, *p.mip0_rows)
        if p.mip1_rows[1] > p.mip1_rows[0]:
            self.ctx.bloom_mip_rows(self.aa_p, self.mc, 1, *p.mip1_rows)

    def exchange_mip(self):
        self.ctx.allgather_rows(self.mip1_p, self.plan.mip1_rows_per_rank, self.plan.mip1_rows_per_rank * self.world)
        self.ctx.comm_wait()

    def composite(self):
        self.ctx.bloom_from_mip(self.aa_p, self.mc, 1)
        if self.plan.out_rows[1] > self.plan.out_rows[0]:
            self.ctx.tonemap(self.aa_p, self.mc, self.out_p, *self.plan.out_rows, flags=self.tonemap_flags)

    def exchange_final(self):
        self.ctx.allgather_rows_reversed(self.out_p, self.plan.rows_per_rank, self.plan.rows_per_rank * self.world)
        self.ctx.comm_wait()  # with a side stream set, readers of `out` on the work stream must come behind the gather

    def step(self, gather=True):
        self.lighting()
        self.reduce()
        if gather:
            self.exchange_mip()
        self.composite()
        if gather:
            self.exchange_final()


class PipelinedChain:
    """Two frames in flight on one rank, so that both exchanges run beside compute (bench.py, N > 1).  Frame i is split at its first
    exchange: A(i) = lighting + copy + bloom mip 0 and mip 1 rows, then the mip-1 gather is queued on the communicator's side stream; B(i) =
    bloom mips 2.. + tonemap, then the gather of the final rows.  Enqueue order: A(0) | A(1) B(0) | A(2) B(1) | ... : the mip-1 gather
    of frame i travels while A(i + 1) computes, the final gather of frame i while A(i + 2) and B(i + 1) do.  With `second_stream` the
    B halves run on that stream, beside the A half of the next frame: B is mostly the replicated small mips of the bloom pyramid —
    launches of a few hundred texels that leave the chip idle — and at 8 ranks a quarter of a rank's frame (one rank's compute of
    `4k_probe_gi_chain` at N = 8, emulated on one GPU: 0.176 -> 0.166 ms per frame, tools/experiments/chain_two_streams.py).
    Every buffer a frame writes exists twice (two ShardedChain sets); what orders a set's re-use is stated where the waits are."""

    def __init__(self, ctx, frame, device_arrays, rank, world, comm_stream, second_stream=None, tonemap_flags=0):
        import torch
        self.ctx, self.comm_stream, self.torch = ctx, comm_stream, torch
        self.work = torch.cuda.current_stream()
        self.work2 = second_stream
        self.sets = [ShardedChain(ctx, frame, device_arrays, rank, world, tonemap_flags=tonemap_flags) for _ in range(2)]
        self.plan = self.sets[0].plan
        self.a_done = [None, None]      # event behind the A half of the frame that last used the set (second stream only)
        self.mip_done = [None, None]    # event behind the mip-1 gather of the frame that last used the set
        self.final_done = [None, None]  # ... behind its final-image gather
        self.b_done = [None, None]      # ... behind its B half (second stream only)
        self.submitted = 0
        self.finished = 0
        ctx.set_stream(self.work.cuda_stream)
        ctx.comm_set_stream(comm_stream.cuda_stream)

    def register_direct_exchange(self, allgather):
        for s in self.sets:
            s.register_direct_exchange(allgather)

    def unregister_direct_exchange(self):
        for s in self.sets:
            s.unregister_direct_exchange()

    def submit(self, lighting_events=None):
        """Enqueue A(i) and the mip-1 exchange of the next frame, then B(i - 1) and the final exchange of the previous one."""
        i = self.submitted
        s = self.sets[i % 2]
        # A(i) overwrites the set's lit / antialiased / mip-0 / own mip-1 rows: their last readers were B(i - 2) — same stream and earlier,
        # or waited for here — and the mip-1 gather of frame i - 2 (B(i - 2) waited for it before it ran)
        if self.work2 is not None and self.b_done[i % 2] is not None:
            self.work.wait_event(self.b_done[i % 2])
        if lighting_events is not None:
            lighting_events[0].record(self.work)
        s.lighting()
        if lighting_events is not None:
            lighting_events[1].record(self.work)
        s.reduce()
        if self.work2 is not None:  # (without a communicator the gather below enqueues nothing: B(i) is then ordered behind A(i) by this alone)
            self.a_done[i % 2] = self.work.record_event()
        self.ctx.allgather_rows(s.mip1_p, s.plan.mip1_rows_per_rank, s.plan.mip1_rows_per_rank * s.world)  # side stream, behind A(i)
        self.mip_done[i % 2] = self.comm_stream.record_event()
        self.submitted += 1
        while self.finished < self.submitted - 1:  # B of the frame before this one (already done if flush() ran in between)
            self._finish(self.finished)

    def _finish(self, j):
        s = self.sets[j % 2]
        st = self.work2 if self.work2 is not None else self.work
        if self.work2 is not None:
            self.ctx.set_stream(st.cuda_stream)      # the library enqueues B(j), and orders its gather, on the second stream
        if self.work2 is not None:
            st.wait_event(self.a_done[j % 2])
        st.wait_event(self.mip_done[j % 2])          # mips 2.. read every rank's rows of mip 1 (and the gather ran behind A(j))
        if self.final_done[j % 2] is not None:       # the final gather of frame j - 2 still reads / writes this set's image
            st.wait_event(self.final_done[j % 2])
        s.composite()
        self.ctx.allgather_rows_reversed(s.out_p, s.plan.rows_per_rank, s.plan.rows_per_rank * s.world)
        self.final_done[j % 2] = self.comm_stream.record_event()
        if self.work2 is not None:
            self.b_done[j % 2] = st.record_event()
            self.ctx.set_stream(self.work.cuda_stream)
        self.finished += 1

    def flush(self):
        """Complete every submitted frame; the work stream then waits for the last gathers (and B halves)."""
        while self.finished < self.submitted:
            self._finish(self.finished)
        for ev in self.final_done + self.b_done:
            if ev is not None:
                self.work.wait_event(ev)

    def image(self, frame_index):
        return self.sets[frame_index % 2].out


class NativePipelinedChain:
    """PipelinedChain with the loop inside the library (sah_chain_create / _submit / _flush, csrc/api_chain.cpp): the same enqueue order,
    waits and exchanges, one ABI call per frame instead of ten plus the stream and event traffic — the host side of a rank's frame drops
    from 72 us to the cost of the launches themselves (tools/experiments/chain_two_streams.py).  PipelinedChain stays as the statement of the
    order the tests hold this one against (tests/test_shard_chain.py).  `exchange=False`: both gathers left out (SAH_CHAIN_NO_EXCHANGE: a
    rank's compute of an N-rank plan on a context of another world size).  `capture=True`: the two halves of a frame are replayed as HIP graphs
    (SAH_CHAIN_CAPTURE, include/sah_hip.h), one graph launch per half instead of three to five kernel launches."""

    def __init__(self, ctx, frame, device_arrays, rank, world, comm_stream=None, second_stream=None, tonemap_flags=0, exchange=True, capture=False, reduce_stream=None):
        import torch
        self.ctx, self.torch = ctx, torch
        self.work = torch.cuda.current_stream()
        self.work2 = second_stream
        self.reduce_stream = reduce_stream  # (kept alive: the library holds the raw handles)
        self.sets = [ShardedChain(ctx, frame, device_arrays, rank, world, tonemap_flags=tonemap_flags) for _ in range(2)]
        self.plan = p = self.sets[0].plan
        self.submitted = 0
        ctx.set_stream(self.work.cuda_stream)
        if comm_stream is not None:
            ctx.comm_set_stream(comm_stream.cuda_stream)
        cp = _abi.ChainPlan()
        cp.aa_rows[:], cp.mip0_rows[:], cp.mip1_rows[:], cp.out_rows[:] = p.aa_rows, p.mip0_rows, p.mip1_rows, p.out_rows
        cp.mip1_rows_per_rank, cp.mip1_allocated_rows = p.mip1_rows_per_rank, self.sets[0].mip1_alloc.shape[0]
        cp.rows_per_rank, cp.out_allocated_rows = p.rows_per_rank, self.sets[0].out_alloc.shape[0]
        frames = (_abi.ChainFrame * 2)()
        import ctypes as C
        for k, s in enumerate(self.sets):
            for j, (desc, _keep) in enumerate(s.descs[:2]):
                frames[k].lighting[j] = C.pointer(desc)
            frames[k].lit, frames[k].antialiased, frames[k].bloom, frames[k].out = s.lit_p, s.aa_p, s.mc, s.out_p
        self.handle = ctx.chain_create(cp, frames, tonemap_flags, (0 if exchange else _abi.CHAIN_NO_EXCHANGE) | (_abi.CHAIN_CAPTURE if capture else 0), self.work.cuda_stream,
                                       reduce_stream.cuda_stream if reduce_stream is not None else None, second_stream.cuda_stream if second_stream is not None else None)

    def register_direct_exchange(self, allgather):
        for s in self.sets:
            s.register_direct_exchange(allgather)

    def unregister_direct_exchange(self):
        for s in self.sets:
            s.unregister_direct_exchange()

    def submit(self, lighting_events=None):
        if lighting_events is not None:
            e0, e1 = lighting_events
            self.ctx.chain_submit(self.handle, e0.cuda_event, e1.cuda_event)
        else:
            self.ctx.chain_submit(self.handle)
        self.submitted += 1

    def flush(self):
        self.ctx.chain_flush(self.handle)

    def image(self, frame_index):
        return self.sets[frame_index % 2].out

    def graphs(self):
        return self.ctx.chain_graphs(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.chain_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
