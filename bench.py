#!/usr/bin/env python3
"""Headline benchmark: lit Mpixels/s of the fused deferred-lighting + GI pass at 4K on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A step = one Lighting pass (default workload: sun CSM + LPV GI gather + AO + emissive + sky, SURVEY.md §8 a0-a3,a6) over one
synthetic 3840x2160 G-buffer already resident in HBM.  N > 1 (launched by torch.distributed.run, one rank per GPU): the
frame is sharded by contiguous row blocks, each rank shades its rows, and the lit rows are re-assembled on every rank with
an RCCL all-gather (the exchange step BASELINE.json's north_star names); total work is fixed => "scaling": "strong".
The `*_chain` workloads add the post chain to the step: copy scene + bloom pyramid on the gathered frame (replicated),
tonemap composite on this rank's rows, all-gather of the final RGBA8 rows.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` (dominant kernel vs HBM peak,
kernel time measured with HIP events on the launch stream) and, at N = 1, `cpu_baseline` (the CPU oracle, OpenMP, timed
on a bounded row band of the same frame).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# BASELINE.json configs[1..4] plus diagnostics.  The headline (`metric`) is 4k_deferred_gi.
WORKLOADS = {
    "4k_deferred_gi": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv"),
    "4k_deferred_gi_scene_shadow": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv", shadow="scene"),  # CSM ray-cast from the atrium
    # inputs made on the GPU by the producer passes (f1, f2, f4): rasterised G-buffer and shadow cascades, LPV from RSM -> VPLs -> propagation
    "4k_deferred_gi_produced": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv", produced=True),
    "4k_deferred_gi_random": dict(res=(3840, 2160), gbuffer="random", sun="csm", gi="lpv"),
    "4k_deferred_only": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="none"),
    "1080p_deferred_gi": dict(res=(1920, 1080), gbuffer="atrium", sun="csm", gi="lpv"),
    "8k_deferred_gi": dict(res=(7680, 4320), gbuffer="atrium", sun="csm", gi="lpv"),
    # light radii per SURVEY.md §8-d: 64 lights r = 6 m, 256 lights r = 4 m, 1024 lights r = 3 m
    "1080p_64_lights": dict(res=(1920, 1080), gbuffer="random", sun="csm", gi="none", lights=64, radius=6.0),            # configs[1]
    "4k_256_lights": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="none", lights=256, radius=4.0),             # configs[2]
    "4k_probe_gi_chain": dict(res=(3840, 2160), gbuffer="atrium", sun="rt", gi="cache", chain=True),                     # configs[3]
    "4k_lpv_gi_chain": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv", chain=True),
    "8k_1024_lights_gi": dict(res=(7680, 4320), gbuffer="atrium", sun="csm", gi="lpv", lights=1024, radius=3.0),         # configs[4]
}


def light_stats(torch, fr, d_arr, lights_np, dev):
    """Mean lights per 16x16 tile after culling and mean lights shaded per surface pixel (SURVEY.md §8-d asks for both with every
    number).  Statistics only: positions are recomputed here in plain fp32 torch, not with the kernel's exact operator order."""
    v = fr.view.gpu_data
    W, H = fr.width, fr.height
    ip = torch.tensor(v.inverse_projection[:], dtype=torch.float32, device=dev).reshape(4, 4).T
    iv = torch.tensor(v.inverse_view[:], dtype=torch.float32, device=dev).reshape(4, 4).T
    depth = d_arr["depth"].view(torch.float32)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    ndc = torch.stack([(xs + 1.0) / v.render_resolution[0] * 2 - 1, (ys + 1.0) / v.render_resolution[1] * 2 - 1, depth, torch.ones_like(depth)], dim=-1)
    vs = ndc @ ip.T
    vs = torch.cat([vs[..., :3] / vs[..., 3:4], torch.ones_like(depth)[..., None]], dim=-1)
    ws = (vs @ iv.T)[..., :3]
    surf = (depth != 0) & torch.isfinite(ws).all(dim=-1)
    th, tw = (H + 15) // 16, (W + 15) // 16
    big = torch.full((th * 16, tw * 16, 3), float("inf"), device=dev)
    lo_src, hi_src = big.clone(), -big
    lo_src[:H, :W][surf] = ws[surf]
    hi_src[:H, :W][surf] = ws[surf]
    lo = lo_src.reshape(th, 16, tw, 16, 3).amin(dim=(1, 3))
    hi = hi_src.reshape(th, 16, tw, 16, 3).amax(dim=(1, 3))
    L = torch.from_numpy(lights_np).to(dev)
    per_tile = torch.zeros((th, tw), device=dev)
    per_px = torch.zeros((H, W), device=dev)
    for i in range(L.shape[0]):
        c, r = L[i, :3], L[i, 3]
        d = torch.clamp(torch.maximum(lo - c, c - hi), min=0.0)
        per_tile += ((d * d).sum(-1) <= (r * 1.0001 + 1e-6) ** 2).float()
        per_px += (((ws - c) ** 2).sum(-1) <= (r * 1.001) ** 2).float()
    return {"lights_per_tile_mean": round(float(per_tile.mean()), 2), "lights_per_tile_max": int(per_tile.max()),
            "lights_shaded_per_pixel_mean": round(float(per_px[surf].mean()) if bool(surf.any()) else 0.0, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="4k_deferred_gi", choices=sorted(WORKLOADS))
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather (diagnostic only)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: one lit target, the all-gather of frame i finishes before frame i+1 is shaded")
    ap.add_argument("--force-gather", action="store_true", help="N=1: run the (one-rank) all-gather path anyway (rehearsal of the N>1 loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU-oracle sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from androidrenderer_amd import _abi, frame, images, lib, scene, synth

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
    use_dist = world > 1 or args.force_gather
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the one JSON line by pointing
        # fd 1 at stderr until the result is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)
    elif args.force_gather:
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29511", rank=0, world_size=1, device_id=dev)

    wl = WORKLOADS[args.workload]
    W, H = wl["res"]
    sun_mode = {"off": _abi.SHADOW_MODE_OFF, "csm": _abi.SHADOW_MODE_CSM, "rt": _abi.SHADOW_MODE_RT}[wl["sun"]]
    gi_kind = {"none": _abi.GI_NONE, "lpv": _abi.GI_LPV, "cache": _abi.GI_CACHE, "rtgi": _abi.GI_RTGI}[wl["gi"]]
    n_lights = wl.get("lights", 0)
    chain = bool(wl.get("chain"))

    # ---- inputs (identical on every rank: generated from fixed seeds) ----------------------------------------------
    lights = None
    if n_lights:
        lights = synth.point_lights(scene.SceneView.default(W, H), n_lights, wl["radius"], seed=8)
    fr = frame.LightingInputs(W, H, seed=2, sun_mode=sun_mode, gi=gi_kind, flavour=wl["gbuffer"], shadowmap_res=4096, lights=lights,
                              synth_device=str(dev), shadow=wl.get("shadow", "noise"))
    d_arr = fr.device_arrays(dev)
    bytes_per_pixel = fr.bytes_per_pixel()

    # row shard of this rank: ceil(H / world) rows per gather slot, clipped to the image (tests/test_dist_cpu.py)
    rows_per = -(-H // world)
    r0 = min(rank * rows_per, H)
    r1 = min(r0 + rows_per, H)
    if world > 1:
        fr.row_begin, fr.row_end = r0, r1
    ctx = lib.Context(device=local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if wl.get("produced"):  # overwrite the synthetic planes with what the library's own producer passes make of the atrium mesh
        from androidrenderer_amd import mesh
        geo_arrays = mesh.to_device(mesh.atrium(8).arrays(), dev)
        geo = mesh.geometry(geo_arrays, [])
        ctx.gbuffer_render(geo, fr.view.gpu_data, images.gbuffer(d_arr))
        ctx.shadow_render(geo, fr.sun.constants, 4, images.volume(d_arr["shadowmap"], _abi.FORMAT_D16_UNORM))
        rsm_t = {"flux": torch.zeros((4, 128, 128, 4), dtype=torch.uint8, device=dev), "normals": torch.zeros((4, 128, 128, 4), dtype=torch.uint8, device=dev),
                 "depth": torch.zeros((4, 128, 128), dtype=torch.int16, device=dev)}
        rsm = _abi.RsmTargets(images.volume(rsm_t["flux"], _abi.FORMAT_R8G8B8A8_SRGB), images.volume(rsm_t["normals"], _abi.FORMAT_R8G8B8A8_UNORM),
                              images.volume(rsm_t["depth"], _abi.FORMAT_D16_UNORM))
        ctx.rsm_render(geo, fr.sun.constants, fr.lpv.matrices, 4, rsm)
        vols = [d_arr[k] for k in ("lpv_r", "lpv_g", "lpv_b")]
        for v in vols:
            v.zero_()
        vd = [images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in vols]
        vpls = torch.zeros((4096, 4), dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        for c in range(4):
            ctx.lpv_extract_vpls(rsm, fr.lpv.matrices, c, 0.25, vpls.data_ptr(), count.data_ptr())
            ctx.lpv_inject_vpls(vpls.data_ptr(), count.data_ptr(), 4096, fr.lpv.matrices, c, 4, vd)
        scratch = [torch.zeros_like(v) for v in vols]
        ctx.lpv_propagate(vd, [images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in scratch], 4, 32)
        torch.cuda.synchronize()
    gather = use_dist and not args.no_gather
    # N > 1: two lit targets, so that the all-gather of frame i (RCCL's own stream) overlaps the shading of frame i + 1; a target is
    # reused only after its gather has completed (stream-level wait).  Every gather finishes inside the timed region.
    nbuf = 2 if (gather and not chain and not args.no_overlap) else 1
    shard_bytes = rows_per * W * 8
    bufs = []
    for _ in range(nbuf):
        lf = torch.zeros((rows_per * world, W, 4), dtype=torch.int16, device=dev)  # equal slots for the all-gather
        desc_b, keep_b = fr.describe(d_arr, lf[:H])
        lb = lf.view(torch.uint8).view(-1)  # RCCL has no int16: the rows travel as bytes
        bufs.append({"lit": lf[:H], "desc": desc_b, "keep": keep_b, "bytes": lb, "slot": lb[rank * shard_bytes:(rank + 1) * shard_bytes]})
    pending = [None] * nbuf
    lit = bufs[0]["lit"]

    if chain:
        aa = torch.zeros((H, W, 4), dtype=torch.int16, device=dev)
        mips = [torch.zeros((mh, mw, 4), dtype=torch.int16, device=dev) for (mw, mh) in images.bloom_mip_sizes(W, H, 6)]
        final_full = torch.zeros((rows_per * world, W, 4), dtype=torch.uint8, device=dev)
        lit_p = images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)
        aa_p = images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)
        mc = images.mipchain(mips)
        final_p = images.plane(final_full[:H], _abi.FORMAT_R8G8B8A8_SRGB)
        final_bytes = final_full.view(-1)
        fshard = rows_per * W * 4
        final_slot = final_bytes[rank * fshard:(rank + 1) * fshard]

    def post():
        ctx.copy_scene(lit_p, aa_p)
        ctx.bloom(aa_p, mc)
        if r1 > r0:
            ctx.tonemap(aa_p, mc, final_p, *((r0, r1) if world > 1 else (0, 0)))
        if gather:
            dist.all_gather_into_tensor(final_bytes, final_slot)

    def step(i, e0=None, e1=None):
        b = bufs[i % nbuf]
        if pending[i % nbuf] is not None:
            pending[i % nbuf].wait()  # the compute stream waits for the gather that last used this target
            pending[i % nbuf] = None
        if e0 is not None:
            e0.record()
        if r1 > r0:
            ctx.lighting(b["desc"])
        if e1 is not None:
            e1.record()
        if gather:  # in place: the input is this rank's slot of the output
            if nbuf > 1:
                pending[i % nbuf] = dist.all_gather_into_tensor(b["bytes"], b["slot"], async_op=True)
            else:
                dist.all_gather_into_tensor(b["bytes"], b["slot"])
        if chain:
            post()

    def drain():
        for k in range(nbuf):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    ev =[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, ev[i][0], ev[i][1])
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kernel_ms_mean = sum(kernel_ms) / len(kernel_ms)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        px = W * H
        value = px * args.steps / elapsed / 1e6
        my_px = W * (r1 - r0) if world > 1 else px
        achieved = bytes_per_pixel * my_px / (kernel_ms_mean * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload)
            except Exception:
                traffic = None
        sun_txt = {"csm": "sun CSM 4x4096^2 D16 PCF", "rt": "sun RT (shadow-mask plane, half-precision BRDF)", "off": "sun off"}[wl["sun"]]
        gi_txt = {"none": "no GI", "lpv": "LPV GI gather + AO", "cache": "irradiance-cache probe gather", "rtgi": "RTGI reconstruction"}[wl["gi"]]
        parts = [sun_txt, gi_txt, "emissive", "sky"]
        if n_lights:
            parts.insert(1, f"{n_lights} point lights (r={wl['radius']} m) with LDS tile culling")
        what = "fused deferred lighting (" + " + ".join(parts) + ")"
        if chain:
            what += " + copy scene + bloom pyramid + tonemap composite"
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
                "workload": f"{args.workload}: {W}x{H} {what}, {wl['gbuffer']} G-buffer"
                            + (" (G-buffer, shadow cascades and LPV made on the GPU by the producer passes)" if wl.get("produced") else ""),
                "resolution": [W, H],
                "gbuffer": wl["gbuffer"],
                "parallelism": "row-shard x%d + RCCL all-gather of lit rows" % world if world > 1 else "single GPU",
                "gather": bool(gather),
                "gather_overlapped_with_next_frame": bool(gather and nbuf > 1),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": "sah::k_lighting_tiled" if (n_lights or gi_kind in (_abi.GI_CACHE, _abi.GI_RTGI)) else "sah::k_lighting_fast",
                "kernel_ms_mean": round(kernel_ms_mean, 5),
                "kernel_ms_min": round(kernel_ms[0], 5),
                "algorithmic_bytes_per_launch": bytes_per_pixel * my_px,
            },
        }
        if n_lights:
            out["config"].update(light_stats(torch, fr, d_arr, lights, dev))
        vpath = os.path.join(ROOT, "profiles", "valu.json")  # second roofline: VALU instruction stream of the dominant kernel
        if os.path.exists(vpath):
            try:
                vinfo = json.load(open(vpath)).get(args.workload)
                if vinfo:
                    out["roofline"]["valu"] = vinfo
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fr, args.cpu_seconds)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)  # teardown chatter goes to stderr as well
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(fr, target_s):
    """Times the CPU oracle (a port, OpenMP over rows) on a bounded band of rows of the same frame (lighting pass only)."""
    try:
        from tests import util
        o = util.oracle()
    except Exception as e:  # oracle .so missing and no compiler: report, don't fail the bench
        return {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    W, H = fr.width, fr.height
    lit = np.zeros((H, W, 4), dtype=np.uint16)
    fr.row_begin = fr.row_end = 0
    d, keep = fr.describe(fr.arrays, lit)
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
