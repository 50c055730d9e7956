"""Randomised HIP-vs-oracle sweep of the post chain (copy scene, bloom pyramid, tonemap composite): odd and tiny resolutions, aspect
ratios from 1:4 to 4:1, output resolutions other than the scene's, fewer than six mips, row bands, and scenes with negative, huge,
infinite, NaN, denormal and zero texels.  Complements tests/test_post_gpu.py; prints one line per case and a summary.

    python tools/stress_post.py [--cases 120]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from androidrenderer_amd import _abi, images, lib, synth  # noqa: E402
from tests import util  # noqa: E402

F16 = _abi.FORMAT_R16G16B16A16_SFLOAT


def make_scene(g, w, h, flavour):
    s = synth.hdr_scene(w, h, seed=int(g.integers(1 << 30)))
    f = s.reshape(h, w, 4)
    n = max(1, w * h // 50)
    ys, xs = g.integers(0, h, n), g.integers(0, w, n)
    if flavour == "signed":
        f[ys, xs, :3] *= np.float16(-1.0)
    elif flavour == "huge":
        f[ys, xs, :3] = np.float16(65504.0)
    elif flavour == "nonfinite":
        f[ys[: n // 2], xs[: n // 2], int(g.integers(3))] = np.float16(np.inf)
        f[ys[n // 2:], xs[n // 2:], int(g.integers(3))] = np.float16(np.nan)
    elif flavour == "tiny":
        f[..., :3] = (f[..., :3].astype(np.float32) * 1e-6).astype(np.float16)  # denormal halves
    elif flavour == "black":
        f[..., :3] = 0
        f[ys, xs, :3] = np.float16(3.0)
    return s.view(np.uint16).reshape(h, w, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120)
    args = ap.parse_args()
    import torch
    ctx = lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    o = util.oracle()
    sizes = [(320, 180), (257, 131), (64, 64), (511, 77), (96, 400), (33, 17), (8, 8), (1, 1), (3, 200), (700, 40), (1280, 720), (130, 129)]
    flavours = ["plain", "signed", "huge", "nonfinite", "tiny", "black"]
    bad = 0
    tol_hist = np.zeros(3, np.int64)
    for case in range(args.cases):
        g = synth.rng(7000 + case)
        w, h = sizes[case % len(sizes)]
        flavour = flavours[(case // len(sizes)) % len(flavours)]
        scene = make_scene(g, w, h, flavour)
        nm = int(g.choice([6, 6, 6, 4, 2, 1, 0]))
        ow, oh = (w, h) if g.random() < 0.6 else (max(1, int(w * g.choice([0.5, 1.5, 2.0]))), max(1, int(h * g.choice([0.5, 1.5, 2.0]))))
        sp = images.plane(scene, F16)
        fails = []
        # copy scene into an antialiased target of the output resolution
        ref_aa = np.zeros((oh, ow, 4), dtype=np.uint16)
        assert o.orc_copy_scene(C.byref(sp), C.byref(images.plane(ref_aa, F16))) == 0
        sc = util.to_torch(scene)
        aa = torch.zeros((oh, ow, 4), dtype=torch.int16, device="cuda")
        ctx.copy_scene(images.plane(sc, F16), images.plane(aa, F16))
        if util.f16_ulp_diff(util.from_torch(aa, np.uint16), ref_aa).max() != 0:  # (any NaN equals any NaN: x86 and gfx950 differ in the sign of a generated NaN)
            fails.append("copy")
        # bloom pyramid of the scene
        sizes_m = images.bloom_mip_sizes(w, h, 6)
        ref_m = [np.zeros((mh, mw, 4), dtype=np.uint16) for (mw, mh) in sizes_m]
        assert o.orc_bloom(C.byref(sp), C.byref(images.mipchain(ref_m))) == 0
        got_m = [torch.zeros(m.shape, dtype=torch.int16, device="cuda") for m in ref_m]
        ctx.bloom(images.plane(sc, F16), images.mipchain(got_m))
        for i, (a, b) in enumerate(zip(got_m, ref_m)):
            if util.f16_ulp_diff(util.from_torch(a, np.uint16), b).max() != 0:
                fails.append(f"mip{i}")
        # tonemap, whole image or three row bands
        ref = np.zeros((oh, ow, 4), dtype=np.uint8)
        assert o.orc_tonemap(C.byref(sp), C.byref(images.mipchain(ref_m[:nm])), C.byref(images.plane(ref, _abi.FORMAT_R8G8B8A8_SRGB)), 0, 0) == 0
        out = torch.zeros((oh, ow, 4), dtype=torch.uint8, device="cuda")
        op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
        chain = images.mipchain(got_m[:nm])
        if oh >= 3 and g.random() < 0.5:
            c1, c2 = sorted(int(v) for v in g.integers(0, oh + 1, 2))
            for r0, r1 in ((0, c1), (c1, c2), (c2, oh)):
                if r1 > r0:
                    ctx.tonemap(images.plane(sc, F16), chain, op, r0, r1)
        else:
            ctx.tonemap(images.plane(sc, F16), chain, op)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        # tolerance mode of the composite against the strict image: |code difference| histogram (finite scenes: must stay <= 1)
        out_t = torch.zeros((oh, ow, 4), dtype=torch.uint8, device="cuda")
        ctx.tonemap(images.plane(sc, F16), chain, images.plane(out_t, _abi.FORMAT_R8G8B8A8_SRGB), flags=_abi.TONEMAP_TOLERANCE_1CODE)
        torch.cuda.synchronize()
        dt = np.abs(out_t.cpu().numpy().astype(np.int32) - ref.astype(np.int32))
        hist = np.bincount(np.minimum(dt.reshape(-1), 2), minlength=3)
        tol_hist += hist
        if flavour != "nonfinite" and dt.max() > 1:
            fails.append(f"tolerance-mode tonemap (max {int(dt.max())})")
        if not np.array_equal(got, ref):
            fails.append(f"tonemap({int((got != ref).sum())} codes, max {int(np.abs(got.astype(int) - ref.astype(int)).max())})")
        bad += bool(fails)
        print(f"case {case:3d}: {w}x{h} -> {ow}x{oh}, {flavour}, {nm} mips: {'ok' if not fails else 'MISMATCH ' + ' '.join(fails)}; tolerance mode: "
              f"{int(hist[1])} codes at 1, {int(hist[2])} beyond", flush=True)
    print(f"{args.cases} cases, {bad} with mismatches")
    print(f"tolerance-mode composite (SAH_TONEMAP_TOLERANCE_1CODE) against the strict one, all cases: |code difference| histogram [0, 1, >= 2] = "
          f"{tol_hist.tolist()} (the >= 2 entries, if any, are pixels next to inf / NaN texels of the 'nonfinite' scenes)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
