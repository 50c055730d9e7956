import sys, os
sys.path.insert(0, "/root/repo")
import torch
from androidrenderer_amd import _abi, chain, lib
from tests import util
torch.cuda.set_stream(torch.cuda.Stream())  # (the null stream cannot be captured)
ctx = lib.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
f = util.LightingFrame(160, 90, seed=33, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium", shadowmap_res=256)
dev = f.device_arrays()
pc = chain.NativePipelinedChain(ctx, f, dev, 0, 1, None, torch.cuda.Stream(), capture=True)
for i in range(9):
    pc.submit()
pc.flush(); torch.cuda.synchronize()
print(pc.graphs())
