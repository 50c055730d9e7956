"""Experiment: the headline frame lit by the library named in SAH_HIP_LIBRARY; saves the RGBA16F image, or compares two saved images.
    approx_bound.py save <path.npy>      |      approx_bound.py compare <strict.npy> <other.npy>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if sys.argv[1] == "save":
    import torch
    from androidrenderer_amd import _abi, frame, lib
    W, H = 3840, 2160
    fr = frame.LightingInputs(W, H, seed=2, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium", shadowmap_res=4096, synth_device="cuda")
    dev = fr.device_arrays("cuda")
    lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    ctx = lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d, keep = fr.describe(dev, lit)
    ctx.lighting(d)
    torch.cuda.synchronize()
    print("deferred pixels:", ctx.deferred_pixels())
    np.save(sys.argv[2], lit.cpu().numpy().view(np.uint16))
else:
    from tests import util
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = util.f16_ulp_diff(a, b)
    hist = [int((d == k).sum()) for k in range(4)] + [int((d >= 4).sum())]
    print("ulp histogram [0,1,2,3,>=4] over", d.size, "channel values:", hist, "max", int(d.max()))
    px = (d.max(axis=2) > 1)
    print("pixels with a channel > 1 ulp:", int(px.sum()), "of", px.size)
