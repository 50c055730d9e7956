"""Randomised parity sweep on the GPU, beyond what tests/ runs every time: for many seeds / sizes / modes the fast Lighting kernel
(1, 2 and 4 pixels per thread) must equal the general kernel bit for bit, the tiled kernel's culled light loop must equal the
brute-force one, and a sample of frames is also checked against the CPU oracle.  usage: tools/stress_parity.py [--seeds 12] [--big]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--big", action="store_true", help="also one 3840x2160 atrium frame (fast vs general)")
    args = ap.parse_args()
    import torch

    from androidrenderer_amd import _abi, lib, synth
    from tests import util

    ctx = lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bad = 0
    sizes = [(256, 144), (331, 77), (640, 360), (128, 512), (1001, 63)]
    n = 0
    for seed in range(args.seeds):
        W, H = sizes[seed % len(sizes)]
        for sun in (_abi.SHADOW_MODE_OFF, _abi.SHADOW_MODE_CSM, _abi.SHADOW_MODE_RT):
            for gi in (_abi.GI_NONE, _abi.GI_LPV):
                flavour = "atrium" if (seed + sun + gi) % 2 else "random"
                shadow = "scene" if seed % 3 == 0 else "noise"
                f = util.LightingFrame(W, H, seed=1000 + seed * 7 + sun * 3 + gi, sun_mode=sun, gi=gi, flavour=flavour, shadowmap_res=512, shadow=shadow,
                                       flags=_abi.LIGHTING_DEFAULT_FLAGS if seed % 4 else 0)
                dev = f.device_arrays()
                ctx.debug_set(force_general=True, force_ppt=0)
                ref = f.run_hip(ctx, dev)
                for ppt in (1, 2, 4):
                    ctx.debug_set(force_general=False, force_ppt=ppt)
                    got = f.run_hip(ctx, dev)
                    n += 1
                    if not np.array_equal(got, ref):
                        d = util.f16_ulp_diff(got, ref)
                        bad += 1
                        print(f"MISMATCH seed={seed} {W}x{H} sun={sun} gi={gi} {flavour} ppt={ppt}: {util.report_ulp('fast vs general', d)}", flush=True)
                if seed % 4 == 0:
                    orc = f.run_oracle()
                    if not np.array_equal(orc, ref):
                        bad += 1
                        print(f"MISMATCH vs oracle seed={seed} sun={sun} gi={gi}: {util.report_ulp('general vs oracle', util.f16_ulp_diff(ref, orc))}", flush=True)
        # tiled kernel: culled == brute force, hot forms on
        base = util.LightingFrame(W, H, seed=2000 + seed, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE, flavour="atrium" if seed % 2 else "random")
        lights = synth.point_lights(base.view, 40 + 37 * seed, 2.0 + seed % 5, seed=3000 + seed)
        f = util.LightingFrame(W, H, seed=2000 + seed, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV if seed % 2 else _abi.GI_NONE,
                               flavour="atrium" if seed % 2 else "random", lights=lights)
        dev = f.device_arrays()
        ctx.debug_set(force_general=False, force_ppt=0)
        culled = f.run_hip(ctx, dev)
        f.flags |= _abi.LIGHTING_BRUTE_FORCE_LIGHTS
        brute = f.run_hip(ctx, dev)
        n += 1
        if not np.array_equal(culled, brute):
            bad += 1
            print(f"MISMATCH lights culled vs brute seed={seed}", flush=True)
        for gi in (_abi.GI_CACHE, _abi.GI_RTGI):
            if seed % 3 == 0:
                fc = util.LightingFrame(min(W, 320), min(H, 180), seed=4000 + seed, sun_mode=_abi.SHADOW_MODE_RT, gi=gi, flavour="atrium", num_extra_rays=seed % 4)
                n += 1
                if not np.array_equal(fc.run_hip(ctx), fc.run_oracle()):
                    bad += 1
                    print(f"MISMATCH tiled gi={gi} vs oracle seed={seed}", flush=True)
        print(f"seed {seed}: {n} comparisons so far, {bad} mismatches", flush=True)
    if args.big:
        f = util.LightingFrame(3840, 2160, seed=2, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium", shadowmap_res=4096, synth_device="cuda")
        dev = f.device_arrays()
        ctx.debug_set(force_general=True, force_ppt=0)
        ref = f.run_hip(ctx, dev)
        ctx.debug_set(force_general=False, force_ppt=0)
        got = f.run_hip(ctx, dev)
        ok = np.array_equal(got, ref)
        bad += 0 if ok else 1
        print("4K atrium CSM+LPV fast == general:", ok, flush=True)
    print("TOTAL mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
