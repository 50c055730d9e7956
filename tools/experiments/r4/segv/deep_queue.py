"""Second hypothesis check for the `rocprofv3 --pmc` SIGSEGV: is it the DEPTH of the queue — many profiled dispatches in flight because every
kernel is slow (8K planes, strided three-channel operands as in synth.atrium_gbuffer) and the host never waits?
usage: deep_queue.py <report> <height> <width> <launches>"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(HERE, "segv_probe.so"))
probe.segv_probe_install.argtypes = [ctypes.c_char_p]
assert probe.segv_probe_install(sys.argv[1].encode()) == 0
H, W, launches = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
noise = torch.rand((H, W, 7), device="cuda")
col = torch.rand((H, W, 3), device="cuda")
torch.cuda.synchronize()
print("inputs ready", flush=True)
for i in range(launches):
    m = col <= 0.0031308                      # the two ops the bench runs faulted in
    col = torch.where(m, col * 12.92, col + 0.02 * noise[..., 0:3])
    if i % 100 == 0:
        print("launch", i, flush=True)        # (no synchronisation: the queue is as deep as the host can make it)
torch.cuda.synchronize()
print("done", launches, flush=True)
