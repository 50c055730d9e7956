"""Third hypothesis check for the `rocprofv3 --pmc` SIGSEGV: a large pageable host-to-device copy (synth.atrium_gbuffer uploads an (H, W, 7) fp32
noise array: 929 MB at 8K, 232 MB at 4K) followed by ordinary elementwise launches on slices of it.
usage: h2d_then_launch.py <report> <height> <width> <channels>"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(HERE, "segv_probe.so"))
probe.segv_probe_install.argtypes = [ctypes.c_char_p]
assert probe.segv_probe_install(sys.argv[1].encode()) == 0
H, W, C = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
host = np.random.default_rng(1).uniform(-1.0, 1.0, (H, W, C)).astype(np.float32)
print("host array", host.nbytes >> 20, "MiB", flush=True)
for rep in range(4):
    noise = torch.from_numpy(host).to("cuda")
    print("uploaded", rep, flush=True)
    for i in range(40):
        col = (0.8 + 0.02 * noise[..., 0:3]).clamp(0.0, 1.0)
        m = col <= 0.0031308
        col = torch.where(m, col * 12.92, 1.055 * col.clamp_min(1e-12).pow(1.0 / 2.4) - 0.055)
    torch.cuda.synchronize()
    print("launched", rep, flush=True)
print("done", flush=True)
