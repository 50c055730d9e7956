"""What would a Lighting launch cost without the sky path in its register budget?  The same frames with the sky bound (SKY = true bodies: 4 waves per
SIMD whatever the surface path needs) and without a sky (SKY = false bodies: 5-8 waves) — the difference is what a separate sky kernel may cost
before it loses.  1280 x 720 RT sun (configs[0]) and row bands of the 4K CSM + LPV frame.     python tools/experiments/r6/sky_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch

from androidrenderer_amd import _abi, frame, lib

ctx = lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=200):
    for _ in range(20):
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
for (W, H, sun, gi, bands, name) in ((1280, 720, _abi.SHADOW_MODE_RT, _abi.GI_NONE, [720], "720p RT sun"),
                                      (3840, 2160, _abi.SHADOW_MODE_CSM, _abi.GI_LPV, [72, 144, 292, 2160], "4K CSM + LPV"),
                                      (3840, 2160, _abi.SHADOW_MODE_CSM, _abi.GI_NONE, [2160], "4K CSM only")):
    for sky in (True, False):
        fr = frame.LightingInputs(W, H, seed=2, sun_mode=sun, gi=gi, flavour="atrium", shadowmap_res=4096, synth_device="cuda", sky=sky)
        if sun == _abi.SHADOW_MODE_RT:
            fr.arrays["shadow_mask"] = np.ones((H, W), dtype=np.float32)
        fr.lpv_generation = 1
        dev = fr.device_arrays("cuda")
        lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
        for rows in bands:
            r0 = (H - rows) // 2 if rows < H else 0
            fr.row_begin, fr.row_end = (r0, r0 + rows) if rows < H else (0, 0)
            desc, keep = fr.describe(dev, lit)
            print(f"{name:14s} rows {rows:5d} sky {'bound' if sky else 'none '}: {timed(lambda: ctx.lighting(desc)):7.2f} us", flush=True)
        del dev, lit
