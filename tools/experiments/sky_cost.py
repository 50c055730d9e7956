import sys, time
sys.path.insert(0, '.')
import torch
from androidrenderer_amd import _abi, frame, lib
ctx = lib.Context(device=0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for sky in (True, False, True, False):
    fr = frame.LightingInputs(3840, 2160, seed=2, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium", shadowmap_res=4096, sky=sky, synth_device="cuda")
    d = fr.device_arrays("cuda")
    lit = torch.zeros((2160, 3840, 4), dtype=torch.int16, device="cuda")
    desc, keep = fr.describe(d, lit)
    for _ in range(20): ctx.lighting(desc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ctx.lighting(desc)
    e1.record(); torch.cuda.synchronize()
    print("sky", sky, "%.4f ms" % (e0.elapsed_time(e1) / 200))
