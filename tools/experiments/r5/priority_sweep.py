"""Stream priorities for the library's sharded frame loop (sah_chain_*): one rank's rows of a world-N plan on one GPU, no exchange, as
tools/experiments/chain_two_streams.py measures it, for every assignment of HIP stream priorities to the loop's streams.
  two streams:   work (lighting + copy + mip rows) | post (mips 2.. + composite)
  three streams: lighting | reduce (copy + mip rows) | post
The streams are made with hipStreamCreateWithPriority (torch's own constructor knows two levels) and handed to torch as external streams.
usage: priority_sweep.py [world] [rank] [--reps R]"""
import ctypes
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch

from androidrenderer_amd import _abi, chain, frame, images, lib, synth

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
world = int(argv[0]) if len(argv) > 0 else 8
rank = int(argv[1]) if len(argv) > 1 else 3
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
W, H = 3840, 2160
fr = frame.LightingInputs(W, H, seed=2, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_CACHE, flavour="atrium", shadowmap_res=4096, synth_device="cuda")
fr.probe_generation = _abi.GENERATION_TRACKED
dev = fr.device_arrays("cuda")
ctx = lib.Context(0)
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(0), ctypes.c_int(0)
assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
print(f"stream priority range: least {lo.value} .. greatest {hi.value}", flush=True)
levels = sorted({hi.value, 0, lo.value})  # numerically lower = higher priority
streams = {}


def stream(key):
    """one stream per (priority, slot): the slots of an assignment must be different streams"""
    if key not in streams:
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, key[0]) == 0  # hipStreamNonBlocking
        streams[key] = torch.cuda.ExternalStream(h.value)
    return streams[key]


tm = _abi.TONEMAP_TOLERANCE_1CODE
cells = synth.rng(33).permutation(32 * 32 * 32)[:1024]
probe_ids = torch.from_numpy(np.stack([cells % 32, (cells // 32) % 32, cells // 1024], axis=-1).astype(np.int32).reshape(-1)).cuda()
irr_vol = images.volume(dev["probe_irr"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
N = 300


def measure(prios):
    ss = [stream((p, k)) for k, p in enumerate(prios)]
    torch.cuda.set_stream(ss[0])
    ctx.set_stream(ss[0].cuda_stream)
    if len(ss) == 2:
        pc = chain.NativePipelinedChain(ctx, fr, dev, rank, world, None, ss[1], tonemap_flags=tm, exchange=False)
    else:
        pc = chain.NativePipelinedChain(ctx, fr, dev, rank, world, None, ss[2], tonemap_flags=tm, exchange=False, reduce_stream=ss[1])

    def run():
        for _ in range(N):
            ctx.probe_notify_updated(irr_vol, probe_ids.data_ptr(), 1024)
            pc.submit()
        pc.flush()
    best = 1e9
    for _ in range(reps + 1):  # the first pass warms up
        torch.cuda.synchronize()
        t = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / N
        best = min(best, dt) if _ else 1e9
    pc.close()
    return best


results = []
for n in (2, 3):
    for prios in itertools.product(levels, repeat=n):
        ms = measure(prios) * 1e3
        results.append((ms, n, prios))
        print(f"world {world} rank {rank}: {n} streams, priorities {prios}: {ms:.4f} ms per frame", flush=True)
print("--- sorted")
for ms, n, prios in sorted(results):
    print(f"{ms:.4f} ms  {n} streams  {prios}")
