"""Wall time per sah_lighting call for a 1/N row shard of the 4K frame (host + launch overhead shows up when the shard is small)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from androidrenderer_amd import _abi, frame, lib
W, H = 3840, 2160
fr = frame.LightingInputs(W, H, seed=2, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium", shadowmap_res=4096, synth_device="cuda")
dev = fr.device_arrays("cuda")
lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
ctx = lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for n in (1, 2, 4, 8):
    fr.row_begin, fr.row_end = 0, H // n
    d, keep = fr.describe(dev, lit)
    for _ in range(20):
        ctx.lighting(d)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        ctx.lighting(d)
    t_host = (time.perf_counter() - t) / 200
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t) / 200
    print(f"1/{n} of the frame: host enqueue {t_host*1e6:7.1f} us/call, end-to-end {t_all*1e6:7.1f} us/call")
