"""CPU baseline of all five BASELINE.json configs: the CPU oracle (oracle/, built -O3 -march=native on this machine, OpenMP
over rows) timed on a bounded band of rows of each config's frame — the lighting pass, and for the chain configs the post chain as well.
The oracle is a restatement of the reference shaders, not the reference's Vulkan-on-lavapipe path (which cannot be built here: SURVEY.md §8-c).

    python tools/cpu_baselines.py [--seconds 6] > profiles/r3_cpu_baselines.txt
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from androidrenderer_amd import _abi, frame, images, scene, synth  # noqa: E402

CONFIGS = [  # (BASELINE.json configs[i], resolution, flavour, sun, gi, lights, radius, chain)
    ("configs[0] 1280x720 single directional light, deferred shading only", (1280, 720), "atrium", "csm", "none", 0, 0.0, False),
    ("configs[1] 1920x1080 random G-buffer, 1 directional + 64 point lights", (1920, 1080), "random", "csm", "none", 64, 6.0, False),
    ("configs[2] 3840x2160 256 point lights", (3840, 2160), "atrium", "csm", "none", 256, 4.0, False),
    ("configs[3] 3840x2160 GI probe gather + AO + tonemap chain", (3840, 2160), "atrium", "rt", "cache", 0, 0.0, True),
    ("configs[4] 7680x4320 1024 lights + GI", (7680, 4320), "atrium", "csm", "lpv", 1024, 3.0, False),
    ("headline   3840x2160 deferred + LPV GI (bench.py default)", (3840, 2160), "atrium", "csm", "lpv", 0, 0.0, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    args = ap.parse_args()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    o = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_native.so"))
    o.orc_lighting.argtypes = [C.POINTER(_abi.LightingDesc)]
    o.orc_copy_scene.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.Plane)]
    o.orc_bloom.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain)]
    o.orc_tonemap.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain), C.POINTER(_abi.Plane), C.c_uint32, C.c_uint32]
    flags = open(os.path.join(ROOT, "oracle", "liboracle_native.flags")).read().strip()
    import bench
    cores = bench.usable_cpus()  # affinity capped by the cgroup quota: the threads that can actually run
    C.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    print(f"# CPU oracle ({flags}), {cores} OpenMP threads; per config: a band of rows sized for ~{args.seconds:.0f} s, median of 3 repetitions")
    print(f"# {'config':72s} {'pass':22s} {'rows':>6s} {'s':>8s} {'Mpx/s':>10s}")
    sun = {"csm": _abi.SHADOW_MODE_CSM, "rt": _abi.SHADOW_MODE_RT}
    gi = {"none": _abi.GI_NONE, "lpv": _abi.GI_LPV, "cache": _abi.GI_CACHE}
    for name, (W, H), flavour, s, g, nl, rad, chain in CONFIGS:
        import torch
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        lights = synth.point_lights(scene.SceneView.default(W, H), nl, rad, seed=8) if nl else None
        fr = frame.LightingInputs(W, H, seed=2, sun_mode=sun[s], gi=gi[g], flavour=flavour, shadowmap_res=4096, lights=lights, synth_device=dev)
        lit = np.zeros((H, W, 4), np.uint16)
        d, keep = fr.describe(fr.arrays, lit)

        def timed(r0, r1):
            d.row_begin, d.row_end = r0, r1
            t = time.perf_counter()
            assert o.orc_lighting(C.byref(d)) == 0
            return time.perf_counter() - t
        # band sizing: a short band is all thread start-up (256 threads), so grow the band until one repetition takes seconds / 3 or it is the frame
        mid = H // 2
        timed(mid, mid + 8)
        rows = 64
        while True:
            r0 = max(0, mid - rows // 2)
            r1 = min(H, r0 + rows)
            if timed(r0, r1) >= args.seconds / 3 or r1 - r0 >= H:
                break
            rows *= 2
        ts = sorted(timed(r0, r1) for _ in range(3))
        print(f"  {name:72s} {'lighting':22s} {r1 - r0:6d} {ts[1]:8.3f} {W * (r1 - r0) / ts[1] / 1e6:10.3f}", flush=True)
        if chain:
            sc = synth.hdr_scene(W, H, seed=11).view(np.uint16)
            aa = np.zeros_like(sc)
            mips = [np.zeros((mh, mw, 4), np.uint16) for (mw, mh) in images.bloom_mip_sizes(W, H, 6)]
            out = np.zeros((H, W, 4), np.uint8)
            sp, ap_ = images.plane(sc, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)
            mc, op = images.mipchain(mips), images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
            t = time.perf_counter()
            assert o.orc_copy_scene(C.byref(sp), C.byref(ap_)) == 0 and o.orc_bloom(C.byref(ap_), C.byref(mc)) == 0
            t_cb = time.perf_counter() - t
            band = max(16, H // 8)
            t = time.perf_counter()
            assert o.orc_tonemap(C.byref(ap_), C.byref(mc), C.byref(op), H // 2, H // 2 + band) == 0
            t_tm = (time.perf_counter() - t) * H / band
            print(f"  {'':72s} {'copy + bloom (whole)':22s} {H:6d} {t_cb:8.3f} {W * H / t_cb / 1e6:10.3f}")
            print(f"  {'':72s} {'tonemap (scaled)':22s} {band:6d} {t_tm:8.3f} {W * H / t_tm / 1e6:10.3f}")
            total = ts[1] * H / (r1 - r0) + t_cb + t_tm
            print(f"  {'':72s} {'whole chain':22s} {H:6d} {total:8.3f} {W * H / total / 1e6:10.3f}")


if __name__ == "__main__":
    main()
