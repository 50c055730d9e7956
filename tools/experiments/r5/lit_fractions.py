"""what fraction of the headline frame's pixels needs the PCF / the BRDF, and how often a thread's four pixels share their
LPV footprint.  Runs ONE Lighting pass of bench.py's headline workload (or --workload NAME) through the counter build of the fast kernel
(tools/experiments/r5/variants.py fast_stats -> build_ab/fast_stats.so) and prints the counters.

    SAH_HIP_LIBRARY=$PWD/build_ab/fast_stats.so python tools/experiments/r5/lit_fractions.py [--workload 4k_deferred_gi]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SAH_HIP_LIBRARY", os.path.join(ROOT, "build_ab", "fast_stats.so"))
import torch  # noqa: E402

import bench  # noqa: E402
from androidrenderer_amd import _abi, frame, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k_deferred_gi")
args = ap.parse_args()
wl = bench.WORKLOADS[args.workload]
W, H = wl["res"]
sun_mode = {"off": _abi.SHADOW_MODE_OFF, "csm": _abi.SHADOW_MODE_CSM, "rt": _abi.SHADOW_MODE_RT}[wl["sun"]]
gi_kind = {"none": _abi.GI_NONE, "lpv": _abi.GI_LPV}[wl["gi"]]
fr = frame.LightingInputs(W, H, seed=2, sun_mode=sun_mode, gi=gi_kind, flavour=wl["gbuffer"], shadowmap_res=4096, synth_device="cuda", shadow=wl.get("shadow", "noise"))
dev = fr.device_arrays("cuda")
ctx = lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
desc, keep = fr.describe(dev, lit)
L = lib.load()
L.sah_debug_fast_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
ctx.lighting(desc)  # (first call: tables, gather copy)
torch.cuda.synchronize()
assert L.sah_debug_fast_stats(None, 1) == 0
ctx.lighting(desc)
torch.cuda.synchronize()
s = (C.c_uint64 * 16)()
assert L.sah_debug_fast_stats(s, 0) == 0
px, surf, lit_n, unsh, waves, pcf_now, brdf_now, pcf_dense, brdf_dense, thr_surf, thr_casc, thr_cell, thr = [int(v) for v in s[:13]]
pct = lambda a, b: f"{100.0 * a / max(b, 1):6.2f} %"
print(f"workload {args.workload}: {W}x{H}, {wl['gbuffer']} G-buffer, shadow map '{wl.get('shadow', 'noise')}'")
print(f"pixels {px}  surface (depth != 0, not deferred) {surf} = {pct(surf, px)}")
print(f"  ndotl > 0                      {lit_n:10d} = {pct(lit_n, px)} of pixels, {pct(lit_n, surf)} of surface pixels")
print(f"  ndotl > 0 and shadow != 0      {unsh:10d} = {pct(unsh, px)} of pixels, {pct(unsh, surf)} of surface pixels")
print(f"waves {waves} (x 4 pixel slots = {4 * waves} wave-evaluations of each stage at most)")
print(f"  PCF  evaluations as voted today {pcf_now:8d} = {pct(pcf_now, 4 * waves)}   if each wave's lit pixels were dense: {pcf_dense:8d} = {pct(pcf_dense, 4 * waves)}")
print(f"  BRDF evaluations as voted today {brdf_now:8d} = {pct(brdf_now, 4 * waves)}   if each wave's unshadowed pixels were dense: {brdf_dense:8d} = {pct(brdf_dense, 4 * waves)}")
print(f"threads {thr}: all four pixels surface {thr_surf} = {pct(thr_surf, thr)}")
print(f"  ... sharing the LPV cascade                {thr_casc:9d} = {pct(thr_casc, thr)} of threads")
print(f"  ... sharing cascade AND footprint base cell {thr_cell:9d} = {pct(thr_cell, thr)} of threads")
