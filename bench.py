#!/usr/bin/env python3
"""Headline benchmark: lit Mpixels/s of the fused deferred-lighting + GI pass at 4K on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A step = one Lighting pass (sun CSM + LPV GI gather + AO + emissive + sky, SURVEY.md §8 a0-a3,a6) over one synthetic
3840x2160 G-buffer already resident in HBM.  N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is
sharded by contiguous row blocks, each rank shades its rows, and the lit rows are re-assembled on every rank with an RCCL
all-gather (the exchange step BASELINE.json's north_star names); total work is fixed => "scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` (dominant kernel vs HBM peak,
kernel time measured with HIP events on the launch stream) and, at N = 1, `cpu_baseline` (the CPU oracle, OpenMP, timed
on a bounded row band of the same frame).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PIXEL = 36   # colour 4 + normals 8 + data 4 + emission 4 + depth 4 + AO 4 read, lit 8 written

WORKLOADS = {
    # name: (width, height, flavour, sun_mode, gi)
    "4k_deferred_gi": (3840, 2160, "atrium", "csm", "lpv"),
    "4k_deferred_gi_random": (3840, 2160, "random", "csm", "lpv"),
    "4k_deferred_only": (3840, 2160, "atrium", "csm", "none"),
    "1080p_deferred_gi": (1920, 1080, "atrium", "csm", "lpv"),
    "8k_deferred_gi": (7680, 4320, "atrium", "csm", "lpv"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="4k_deferred_gi", choices=sorted(WORKLOADS))
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather (diagnostic only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU-oracle sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from androidrenderer_amd import _abi, images, lib, scene, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    W, H, flavour, sun_name, gi_name = WORKLOADS[args.workload]
    sun_mode = {"off": _abi.SHADOW_MODE_OFF, "csm": _abi.SHADOW_MODE_CSM, "rt": _abi.SHADOW_MODE_RT}[sun_name]
    gi_kind = {"none": _abi.GI_NONE, "lpv": _abi.GI_LPV}[gi_name]

    # ---- inputs (identical on every rank: generated from fixed seeds) ----------------------------------------------
    view = scene.SceneView.default(W, H)
    sun = scene.DirectionalLight(shadow_mode=sun_mode)
    sun.update_shadow_cascades(view, resolution=4096)
    if flavour == "atrium":
        g_np = synth.atrium_gbuffer(W, H, view, seed=2, device=str(dev))
    else:
        g_np = synth.random_gbuffer(W, H, seed=1)
    host = dict(g_np)
    host["ao"] = synth.ao_plane(W, H, 3)
    luts = synth.sky_luts(7)
    host["sky_t"], host["sky_v"] = luts["transmittance"], luts["sky_view"]
    host["shadowmap"] = synth.shadowmap(4096, 4, 6)
    lpv = scene.LpvCascades()
    lpv.update_cascade_transforms(view, sun)
    host["lpv_r"], host["lpv_g"], host["lpv_b"] = synth.lpv_volumes(4, 5)

    def up(a):
        if a.dtype == np.uint16:
            return torch.from_numpy(a.view(np.int16)).to(dev)
        return torch.from_numpy(a).to(dev)

    d_arr = {k: up(v) for k, v in host.items()}
    # row shard of this rank: ceil(H / world) rows per gather slot, clipped to the image (tests/test_dist_cpu.py)
    rows_per = -(-H // world)
    r0 = min(rank * rows_per, H)
    r1 = min(r0 + rows_per, H)
    lit_full = torch.zeros((rows_per * world, W, 4), dtype=torch.int16, device=dev)  # equal slots for the all-gather
    lit = lit_full[:H]

    gb = images.gbuffer(d_arr)
    lit_p = images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)
    ao_p = images.plane(d_arr["ao"], _abi.FORMAT_R32_SFLOAT)
    sm_v = images.volume(d_arr["shadowmap"], _abi.FORMAT_D16_UNORM)
    sky = _abi.SkyLuts(images.plane(d_arr["sky_t"], _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(d_arr["sky_v"], _abi.FORMAT_R16G16B16A16_SFLOAT))
    gi = _abi.GI()
    gi.kind = gi_kind
    if gi_kind == _abi.GI_LPV:
        gi.lpv_red = images.volume(d_arr["lpv_r"], _abi.FORMAT_R16G16B16A16_SFLOAT)
        gi.lpv_green = images.volume(d_arr["lpv_g"], _abi.FORMAT_R16G16B16A16_SFLOAT)
        gi.lpv_blue = images.volume(d_arr["lpv_b"], _abi.FORMAT_R16G16B16A16_SFLOAT)
        gi.lpv_cascades = C.cast(lpv.matrices, C.POINTER(_abi.LpvCascadeMatrices))
        gi.lpv_num_cascades = 4
        gi.lpv_exposure = float(np.float32(math.pi) * np.float32(10.0))

    desc = _abi.LightingDesc()
    desc.gbuffer = C.pointer(gb)
    desc.lit = C.pointer(lit_p)
    desc.ao = C.pointer(ao_p)
    desc.view = C.pointer(view.gpu_data)
    desc.sun = C.pointer(sun.constants)
    desc.shadowmap = C.pointer(sm_v)
    desc.sky = C.pointer(sky)
    desc.gi = C.pointer(gi)
    desc.flags = _abi.LIGHTING_DEFAULT_FLAGS
    desc.row_begin, desc.row_end = (r0, r1) if world > 1 else (0, 0)

    ctx = lib.Context(device=local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    gather = world > 1 and not args.no_gather
    lit_bytes = lit_full.view(torch.uint8).view(-1)  # RCCL has no int16: the rows travel as bytes
    shard_bytes = rows_per * W * 8
    my_slot = lit_bytes[rank * shard_bytes:(rank + 1) * shard_bytes]

    def step():
        if r1 > r0:
            ctx.lighting(desc)
        if gather:
            dist.all_gather_into_tensor(lit_bytes, my_slot)  # in place: the input is this rank's slot of the output

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        if r1 > r0:
            ctx.lighting(desc)
        ev[i][1].record()
        if gather:
            dist.all_gather_into_tensor(lit_bytes, my_slot)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kernel_ms_mean = sum(kernel_ms) / len(kernel_ms)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        px = W * H
        value = px * args.steps / elapsed / 1e6
        my_px = W * (r1 - r0) if world > 1 else px
        achieved = BYTES_PER_PIXEL * my_px / (kernel_ms_mean * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload)
            except Exception:
                traffic = None
        out = {
            "metric": "lit Mpixels/sec (deferred+GI pass) at 4K",
            "value": round(value, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {W}x{H} fused deferred lighting (sun {sun_name.upper()} 4x4096^2 D16 PCF + "
                            f"{gi_name.upper()} GI gather + AO + emissive + sky), {flavour} G-buffer",
                "resolution": [W, H],
                "gbuffer": flavour,
                "parallelism": "row-shard x%d + RCCL all-gather of lit rows" % world if world > 1 else "single GPU",
                "gather": bool(gather),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": "sah::k_lighting",
                "kernel_ms_mean": round(kernel_ms_mean, 5),
                "kernel_ms_min": round(kernel_ms[0], 5),
                "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * my_px,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, view, sun, lpv, W, H, sun_mode, gi_kind, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(host, view, sun, lpv, W, H, sun_mode, gi_kind, target_s):
    """Times the CPU oracle (a port, OpenMP over rows) on a bounded band of rows of the same frame."""
    from androidrenderer_amd import _abi, images
    try:
        from tests import util
        o = util.oracle()
    except Exception as e:  # oracle .so missing and no compiler: report, don't fail the bench
        return {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    lit = np.zeros((H, W, 4), dtype=np.uint16)
    gb = images.gbuffer(host)
    lit_p = images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)
    ao_p = images.plane(host["ao"], _abi.FORMAT_R32_SFLOAT)
    sm_v = images.volume(host["shadowmap"], _abi.FORMAT_D16_UNORM)
    sky = _abi.SkyLuts(images.plane(host["sky_t"], _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(host["sky_v"], _abi.FORMAT_R16G16B16A16_SFLOAT))
    gi = _abi.GI()
    gi.kind = gi_kind
    if gi_kind == _abi.GI_LPV:
        gi.lpv_red = images.volume(host["lpv_r"], _abi.FORMAT_R16G16B16A16_SFLOAT)
        gi.lpv_green = images.volume(host["lpv_g"], _abi.FORMAT_R16G16B16A16_SFLOAT)
        gi.lpv_blue = images.volume(host["lpv_b"], _abi.FORMAT_R16G16B16A16_SFLOAT)
        gi.lpv_cascades = C.cast(lpv.matrices, C.POINTER(_abi.LpvCascadeMatrices))
        gi.lpv_num_cascades = 4
        gi.lpv_exposure = float(np.float32(math.pi) * np.float32(10.0))
    d = _abi.LightingDesc()
    d.gbuffer, d.lit, d.ao = C.pointer(gb), C.pointer(lit_p), C.pointer(ao_p)
    d.view, d.sun = C.pointer(view.gpu_data), C.pointer(sun.constants)
    d.shadowmap, d.sky, d.gi = C.pointer(sm_v), C.pointer(sky), C.pointer(gi)
    d.flags = _abi.LIGHTING_DEFAULT_FLAGS
    cores = os.cpu_count() or 1
    mid = H // 2
    # calibrate on 16 rows, then size the band for ~target_s seconds
    d.row_begin, d.row_end = mid, min(H, mid + 16)
    t = time.perf_counter()
    o.orc_lighting(C.byref(d))
    dt = max(time.perf_counter() - t, 1e-4)
    rows = int(max(16, min(H, 16 * target_s / dt)))
    r0 = max(0, mid - rows // 2)
    d.row_begin, d.row_end = r0, min(H, r0 + rows)
    # a many-core host finishes the whole frame in well under target_s: repeat it so the sample is ~target_s of work
    reps = 1
    if rows >= H:
        reps = int(max(1, min(64, target_s / max(dt * H / 16.0, 1e-3))))
    t = time.perf_counter()
    for _ in range(reps):
        o.orc_lighting(C.byref(d))
    dt = time.perf_counter() - t
    npx = W * (d.row_end - d.row_begin) * reps
    return {"value": round(npx / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"rows [{d.row_begin},{d.row_end}) of the same {W}x{H} frame x {reps} repetitions ({npx} px) in {dt:.2f} s; CPU "
                      f"oracle (oracle/, g++ -O2 -fopenmp, {cores} threads) — a restatement, not the reference's Vulkan/lavapipe path"}


if __name__ == "__main__":
    main()
