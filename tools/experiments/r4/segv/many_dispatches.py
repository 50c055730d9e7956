"""Hypothesis check for the `rocprofv3 --pmc` SIGSEGV (profiles/README.md "8K under --pmc"): does a process that merely issues MANY small
kernel launches on one stream fault at a fixed launch count, whatever the tensor size?   usage: many_dispatches.py <report> <elements> <launches>"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(HERE, "segv_probe.so"))
probe.segv_probe_install.argtypes = [ctypes.c_char_p]
assert probe.segv_probe_install(sys.argv[1].encode()) == 0
n, launches = int(sys.argv[2]), int(sys.argv[3])
a = torch.ones(n, device="cuda")
for i in range(launches):
    a.mul_(1.0)
    if i % 256 == 0:
        torch.cuda.synchronize()
        print("launch", i, flush=True)
torch.cuda.synchronize()
print("done", launches, flush=True)
