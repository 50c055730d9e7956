"""How many (non-sky) pixels of a bench.py lighting workload leave the fast kernel for the fix-up kernel?   python tools/experiments/r5/deferred_count.py [workload ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from androidrenderer_amd import _abi, frame, lib  # noqa: E402

for name in sys.argv[1:] or ["4k_deferred_gi", "4k_deferred_gi_scene_shadow", "4k_deferred_only", "720p_deferred_only", "4k_deferred_gi_random"]:
    wl = bench.WORKLOADS[name]
    W, H = wl["res"]
    sun_mode = {"off": _abi.SHADOW_MODE_OFF, "csm": _abi.SHADOW_MODE_CSM, "rt": _abi.SHADOW_MODE_RT}[wl["sun"]]
    gi_kind = {"none": _abi.GI_NONE, "lpv": _abi.GI_LPV}[wl["gi"]]
    fr = frame.LightingInputs(W, H, seed=2, sun_mode=sun_mode, gi=gi_kind, flavour=wl["gbuffer"], shadowmap_res=4096, synth_device="cuda", shadow=wl.get("shadow", "noise"))
    dev = fr.device_arrays("cuda")
    ctx = lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    desc, keep = fr.describe(dev, lit)
    ctx.lighting(desc)
    ctx.lighting(desc)
    torch.cuda.synchronize()
    print(f"{name}: {ctx.deferred_pixels()} of {W * H} pixels deferred to the fix-up kernel", flush=True)
    ctx.close()
