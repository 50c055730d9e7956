"""sah_lpv_propagate's 32 dependent launches, eager against a captured HIP graph (torch.cuda.graph captures the library's launches on
the stream the context was given): does graph replay close the gaps between the launches?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from androidrenderer_amd import _abi, images, lib, synth
from tests import util

ctx = lib.Context(device=0)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ctx.set_stream(side.cuda_stream)
    vols = synth.lpv_volumes(4, 5)
    a_t = [util.to_torch(v.view(np.uint16)) for v in vols]
    b_t = [torch.zeros_like(t) for t in a_t]
    av = [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in a_t]
    bv = [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in b_t]

    def run():
        ctx.lpv_propagate(av, bv, 4, 32)

    def rate(fn, iters=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    print("eager   32 steps: %.4f ms" % rate(run), flush=True)
    run()
    torch.cuda.synchronize()
    want = [t.clone() for t in a_t + b_t]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        run()
    print("graph   32 steps: %.4f ms" % rate(g.replay), flush=True)
    torch.cuda.synchronize()
ctx.close()
