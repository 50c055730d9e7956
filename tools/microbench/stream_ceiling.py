"""What a plain device-to-device stream reaches on this MI355X (the practical ceiling the 8 TB/s roofline fractions should be read
against): torch's copy kernel and hipMemcpyAsync on buffers the size of the benchmark's planes and larger."""
import torch

def rate(fn, nbytes, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2 * nbytes / (ms * 1e-3) / 1e9  # read + write

for mb in (66, 133, 265, 1061, 4244):
    n = mb * 1000 * 1000
    x = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
    y = torch.empty_like(x)
    ms, gbs = rate(lambda: y.copy_(x), n)
    x4, y4 = x.view(torch.float32), y.view(torch.float32)
    ms2, gbs2 = rate(lambda: torch.add(x4, 1.0, out=y4), n)
    print(f"{mb:5d} MB  copy_ {ms:7.4f} ms {gbs:7.1f} GB/s ({gbs / 80:.1f} % of 8 TB/s)   add(out=) {ms2:7.4f} ms {gbs2:7.1f} GB/s ({gbs2 / 80:.1f} %)", flush=True)
    del x, y
