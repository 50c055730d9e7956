"""Fourth hypothesis check for the `rocprofv3 --pmc` SIGSEGV: bench.py's light_stats() is a loop over 1024 lights of about twenty launches each —
one-element kernels on 0-dim tensors between kernels over 8K frames — that the host queues far ahead of the GPU: does the fault need a FULL
queue (tens of thousands of profiled dispatches outstanding)?   usage: full_queue.py <report> <height> <width> <lights>"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(HERE, "segv_probe.so"))
probe.segv_probe_install.argtypes = [ctypes.c_char_p]
assert probe.segv_probe_install(sys.argv[1].encode()) == 0
H, W, n = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ws = torch.rand((H, W, 3), device="cuda") * 20.0
lo = torch.rand(((H + 15) // 16, (W + 15) // 16, 3), device="cuda") * 20.0
hi = lo + 1.0
L = torch.rand((n, 4), device="cuda") * 20.0
per_tile = torch.zeros(lo.shape[:2], device="cuda")
per_px = torch.zeros((H, W), device="cuda")
torch.cuda.synchronize()
print("inputs ready", flush=True)
for i in range(n):  # the loop of bench.py light_stats(), operator for operator
    c, r = L[i, :3], L[i, 3]
    d = torch.clamp(torch.maximum(lo - c, c - hi), min=0.0)
    per_tile += ((d * d).sum(-1) <= (r * 1.0001 + 1e-6) ** 2).float()
    per_px += (((ws - c) ** 2).sum(-1) <= (r * 1.001) ** 2).float()
    if i % 64 == 0:
        print("light", i, flush=True)
print("queued", flush=True)
torch.cuda.synchronize()
print("done", float(per_px.mean()), flush=True)
