"""Time of one pass over a row band against the band's height: what does not shrink with the rows (VERDICT r4 item 1 d — a rank of eight
shades 292 of 2160 rows in 62 us where 353 / 8 = 44 would be its share).  Lighting RT + cache (tiled kernel), lighting CSM + LPV (fast kernel)
and the tolerance composite, bands centred on the frame.    [SAH_HIP_LIBRARY=build_ab/<variant>.so] python tools/experiments/r5/band_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from androidrenderer_amd import _abi, chain, frame, lib

W, H = 3840, 2160
ROWS = [72, 144, 292, 584, 1080, 2160]
ctx = lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


print(f"library: {os.environ.get('SAH_HIP_LIBRARY', 'in-tree')}")
for name, sun, gi in (("lighting RT + cache (tiled)", _abi.SHADOW_MODE_RT, _abi.GI_CACHE), ("lighting CSM + LPV (fast)", _abi.SHADOW_MODE_CSM, _abi.GI_LPV)):
    fr = frame.LightingInputs(W, H, seed=2, sun_mode=sun, gi=gi, flavour="atrium", shadowmap_res=4096, synth_device="cuda")
    fr.probe_generation = fr.lpv_generation = 1
    dev = fr.device_arrays("cuda")
    lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    full = None
    for rows in ROWS:
        r0 = (H - rows) // 2 if rows < H else 0
        fr.row_begin, fr.row_end = (r0, r0 + rows) if rows < H else (0, 0)
        desc, keep = fr.describe(dev, lit)
        us = timed(lambda: ctx.lighting(desc))
        if rows == H:
            full = us
        print(f"{name:30s} rows {rows:5d}: {us:8.1f} us   ({us / rows * 1e3:7.1f} ns per row)", flush=True)
    print(f"{name:30s} fixed part of a 292-row band if the slope were the whole frame's: see rows 292 against {full:.1f} * 292 / 2160 = {full * 292 / 2160:.1f} us")
    del dev, lit
    torch.cuda.empty_cache()

# the tolerance composite over bands (chain buffers of the unsharded frame)
fr = frame.LightingInputs(W, H, seed=2, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_NONE, flavour="atrium", synth_device="cuda")
dev = fr.device_arrays("cuda")
sc = chain.ShardedChain(ctx, fr, dev, 0, 1, tonemap_flags=_abi.TONEMAP_TOLERANCE_1CODE)
sc.step(gather=False)
for rows in (72, 144, 270, 540, 1080, 2160):
    r0 = (H - rows) // 2 if rows < H else 0
    us = timed(lambda: ctx.tonemap(sc.aa_p, sc.mc, sc.out_p, r0, r0 + rows, flags=_abi.TONEMAP_TOLERANCE_1CODE))
    print(f"{'tolerance composite':30s} rows {rows:5d}: {us:8.1f} us   ({us / rows * 1e3:7.1f} ns per row)", flush=True)
