"""One rank's compute of the sharded chain at N ranks, emulated on one GPU without exchanges: one stream against two streams
(A(i+1) = lighting + copy + mip-0 and mip-1 rows beside B(i) = mips 2-5 + tonemap rows), enqueued call by call from Python against the
library's own loop (sah_chain_submit, round 5).
usage: chain_two_streams.py [world] [rank] [--strict-tonemap] [--rebuild-copies] [--no-probe-updates] [--priorities]
  the tonemap runs in tolerance mode, as bench.py's chain workloads do, unless --strict-tonemap is given;
  the context tracks its fp32 copy of the irradiance atlas (sah_gi::probe_generation = SAH_GENERATION_TRACKED) and every frame re-widens
  the blocks of 1024 probes — the reference's r.GI.Cache.UpdatesPerFrame — through sah_probe_notify_updated, unless --no-probe-updates;
  --rebuild-copies: probe_generation 0, the whole atlas widened every frame (what rounds 3-4 measured)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from androidrenderer_amd import _abi, chain, frame, images, lib, synth

strict = "--strict-tonemap" in sys.argv
rebuild = "--rebuild-copies" in sys.argv
probe_updates = "--no-probe-updates" not in sys.argv and not rebuild
argv = [a for a in sys.argv if not a.startswith("--")]
world = int(argv[1]) if len(argv) > 1 else 8
rank = int(argv[2]) if len(argv) > 2 else 3
W, H = 3840, 2160
fr = frame.LightingInputs(W, H, seed=2, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_CACHE, flavour="atrium", shadowmap_res=4096, synth_device="cuda")
fr.probe_generation = 0 if rebuild else _abi.GENERATION_TRACKED
dev = fr.device_arrays("cuda")
ctx = lib.Context(0)
prio = "--priorities" in sys.argv  # the lighting stream above the two others (hipStreamCreateWithPriority through torch)
torch.cuda.set_stream(torch.cuda.Stream(priority=-1) if prio else torch.cuda.Stream())  # a work stream of its own: the null stream cannot be captured into a graph
s1 = torch.cuda.current_stream()
s2 = torch.cuda.Stream()
ctx.set_stream(s1.cuda_stream)
tm = 0 if strict else _abi.TONEMAP_TOLERANCE_1CODE
sets = [chain.ShardedChain(ctx, fr, dev, rank, world, tonemap_flags=tm) for _ in range(2)]
cells = synth.rng(33).permutation(32 * 32 * 32)[:1024]
probe_ids = torch.from_numpy(np.stack([cells % 32, (cells // 32) % 32, cells // 1024], axis=-1).astype(np.int32).reshape(-1)).cuda()
irr_vol = images.volume(dev["probe_irr"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
N = 300


def maintain():
    if probe_updates:
        ctx.probe_notify_updated(irr_vol, probe_ids.data_ptr(), 1024)


def one_stream():
    for i in range(N):
        s = sets[i % 2]
        maintain()
        s.lighting()
        s.reduce()
        s.composite()


def two_streams():
    a_done = [None, None]
    b_done = [None, None]
    for i in range(N + 1):
        if i < N:
            s = sets[i % 2]
            ctx.set_stream(s1.cuda_stream)
            if b_done[i % 2] is not None:
                s1.wait_event(b_done[i % 2])
            maintain()
            s.lighting()
            s.reduce()
            a_done[i % 2] = s1.record_event()
        if i >= 1:
            j = i - 1
            ctx.set_stream(s2.cuda_stream)
            s2.wait_event(a_done[j % 2])
            with torch.cuda.stream(s2):
                sets[j % 2].composite()
            b_done[j % 2] = s2.record_event()
    ctx.set_stream(s1.cuda_stream)


s3 = torch.cuda.Stream()


def three_streams():
    """lighting | copy + mip 0 + mip 1 | mips 2-5 + composite, each on a stream of its own: the lighting of frame i + 1 starts when that of frame i
    ends (not behind its copy and mip rows), ordered by events against the last readers of the buffer set it overwrites"""
    l_done, r_done, c_done = [None, None], [None, None], [None, None]
    for i in range(N + 2):
        if i < N:
            k = i % 2
            ctx.set_stream(s1.cuda_stream)
            if r_done[k] is not None:
                s1.wait_event(r_done[k])  # `lit` of this set was last read by the copy of frame i - 2
            maintain()
            sets[k].lighting()
            l_done[k] = s1.record_event()
            ctx.set_stream(s2.cuda_stream)
            s2.wait_event(l_done[k])
            if c_done[k] is not None:
                s2.wait_event(c_done[k])  # antialiased / mip 0 / mip 1 of this set were last read by the composite of frame i - 2
            with torch.cuda.stream(s2):
                sets[k].reduce()
            r_done[k] = s2.record_event()
        if 1 <= i <= N:
            j = (i - 1) % 2
            ctx.set_stream(s3.cuda_stream)
            s3.wait_event(r_done[j])
            with torch.cuda.stream(s3):
                sets[j].composite()
            c_done[j] = s3.record_event()
    ctx.set_stream(s1.cuda_stream)


def library_loop(second, capture=False, reduce=None):
    pc = chain.NativePipelinedChain(ctx, fr, dev, rank, world, None, second, tonemap_flags=tm, exchange=False, capture=capture, reduce_stream=reduce)

    def run():
        for _ in range(N):
            maintain()
            pc.submit()
        pc.flush()
    return run


lib_one, lib_two, graph_one, graph_two = library_loop(None), library_loop(s2), library_loop(None, True), library_loop(s2, True)
lib_three, graph_three = library_loop(s3, False, s2), library_loop(s3, True, s2)
for name, fn in (("python, one stream", one_stream), ("python, two streams", two_streams), ("python, three streams", three_streams), ("library loop, one stream", lib_one), ("library loop, two streams", lib_two),
                 ("library loop, three streams", lib_three), ("library loop + graphs, one", graph_one), ("library loop + graphs, two", graph_two),
                 ("library loop + graphs, three", graph_three)) * 2:
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    host = (time.perf_counter() - t) / N
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / N
    print(f"world {world} rank {rank}: {name:28s} {dt * 1e3:.4f} ms per frame (host enqueue {host * 1e3:.4f} ms)", flush=True)
