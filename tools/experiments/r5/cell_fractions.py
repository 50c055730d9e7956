"""VERDICT r4 item 2, "the wave-uniform probe-cell path": how many waves of the cache-GI lighting kernel hold pixels of ONE probe cell (so
that probe indices, validity bytes and atlas origins could live in scalar registers), and how many of a pixel's eight probes are evaluated.
One Lighting pass (RT sun + irradiance cache, the atrium frame of bench.py's 4k_probe_gi_chain) through the counter build
(tools/experiments/r5/variants.py tiled_cell_stats -> build_ab/tiled_cell_stats.so).

    python tools/experiments/r5/cell_fractions.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SAH_HIP_LIBRARY", os.path.join(ROOT, "build_ab", "tiled_cell_stats.so"))
import torch  # noqa: E402

from androidrenderer_amd import _abi, frame, lib  # noqa: E402

W, H = 3840, 2160
fr = frame.LightingInputs(W, H, seed=2, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_CACHE, flavour="atrium", shadowmap_res=4096, synth_device="cuda")
dev = fr.device_arrays("cuda")
ctx = lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
desc, keep = fr.describe(dev, lit)
L = lib.load()
L.sah_debug_cell_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
ctx.lighting(desc)
torch.cuda.synchronize()
assert L.sah_debug_cell_stats(None, 1) == 0
ctx.lighting(desc)
torch.cuda.synchronize()
s = (C.c_uint64 * 16)()
assert L.sah_debug_cell_stats(s, 0) == 0
waves, px, waves_same, px_same, probes, full, px_any = [int(v) for v in s[:7]]
pct = lambda a, b: f"{100.0 * a / max(b, 1):6.2f} %"
print(f"{W}x{H} atrium frame, RT sun + irradiance cache (waves are 32 x 2 pixels)")
print(f"pixels that gather                    {px:9d} = {pct(px, W * H)} of the frame, in {waves} waves ({pct(full, waves)} of them full)")
print(f"waves whose pixels share ONE cell     {waves_same:9d} = {pct(waves_same, waves)} of the waves")
print(f"pixels in their wave's first cell     {px_same:9d} = {pct(px_same, px)} of the gathering pixels")
print(f"probes evaluated per gathering pixel  {probes / max(px, 1):9.3f} of 8   (pixels with at least one: {pct(px_any, px)})")
