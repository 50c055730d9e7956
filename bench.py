#!/usr/bin/env python3
"""Headline benchmark: lit Mpixels/s of the fused deferred-lighting + GI pass at 4K on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A step = one Lighting pass (default workload: sun CSM + LPV GI gather + AO + emissive + sky, SURVEY.md §8 a0-a3,a6) over one
synthetic 3840x2160 G-buffer already resident in HBM.  N > 1 (launched by torch.distributed.run, one rank per GPU): the
frame is sharded by contiguous row blocks, each rank shades its rows, and the lit rows are re-assembled on every rank by the
LIBRARY's exchange entry point (sah_allgather_rows: RCCL all-gather, on a side stream so that the gather of frame i runs beside
the shading of frame i + 1; `--torch-gather` uses torch.distributed instead, `--no-overlap` / `--no-gather` are diagnostics);
total work is fixed => "scaling": "strong".
The `*_chain` workloads are the whole frame: lighting, copy scene, bloom pyramid, tonemap composite.  At N > 1 they run the sharded
chain of androidrenderer_amd/chain.py: every rank shades its rows (+ halo), reduces them to its rows of bloom mips 0 and 1, all ranks
exchange the quarter-resolution mip 1, build the smaller mips redundantly, composite their rows of the final image and exchange the
RGBA8 rows — in reversed rank order, because the composite samples the scene upside down.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` (dominant kernel vs HBM peak,
kernel time measured with HIP events on the launch stream) and, at N = 1, `cpu_baseline` (the CPU oracle, OpenMP, -O3
-march=native, median of 5 repetitions of a bounded row band of the same frame).
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
    # configs[3] with nothing synthetic between the mesh and the image: the G-buffer rasterised from the atrium mesh, the AO plane and the sun's
    # shadow mask traced every step against the structure sah_rt_build made of the same mesh (reference defaults: 1 AO ray of 8 m, 8 shadow rays),
    # and 1024 probes of the irradiance cache traced (400 GI rays each) and folded into the atlases every step
    "4k_probe_gi_chain_traced": dict(res=(3840, 2160), gbuffer="atrium", sun="rt", gi="cache", chain=True, traced=True),
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
        if i % 32 == 31:
            # about twenty launches per light, queued far ahead of an 8K frame's kernels: with 16,384 dispatches outstanding — the queue's
            # packet count — rocprofv3 --pmc faults inside librocprofiler-sdk's queue interception (profiles/README.md "8K under --pmc")
            torch.cuda.synchronize()
    return {"lights_per_tile_mean": round(float(per_tile.mean()), 2), "lights_per_tile_max": int(per_tile.max()),
            "lights_shaded_per_pixel_mean": round(float(per_px[surf].mean()) if bool(surf.any()) else 0.0, 3)}


VALU_PEAK_TFLOPS = 157.3   # same guide: FP32 vector peak (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz)
SIMDS, MAX_CLOCK_HZ = 1024, 2.4e9


def roofline(workload, world, achieved_gbs, kernel_ms_mean, kernel_ms_min, scope, algorithmic_bytes, my_px, kernel):
    """The metric's roofline is HBM (BASELINE.json: "achieved HBM GB/s vs roofline"): achieved / peak / frac are that one, measured
    live.  `bound` names the roofline that actually binds the workload, and the other two are carried beside it:
      valu_issue  VALU issue cycles of one launch / (1024 SIMDs x 2.4 GHz x the LIVE kernel time).  The cycles are a MODEL: the measured
                  SQ_INSTS_VALU of a separate rocprofv3 --pmc pass of the same build and workload (profiles/roofline_static.json says
                  which file; PMC counters cannot be read from inside a run) x the cycles per instruction of the kernel's static VALU
                  mix priced with the measured issue costs (tools/pmc_to_static.py, profiles/r1_valu_issue_cost.txt).  Beside it:
                  `valu_busy_rocprof`, rocprof's VALUBusy (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles) — it charges every instruction a whole
                  quad-cycle, so it overstates kernels made of 2-cycle fp32 ops and can exceed 1 — and `lower_bound`, every
                  instruction at the cheapest cost (2.3 cycles);
      fp32        algorithmic FLOP per pixel (counted once from the per-pixel operator list) x pixels / LIVE kernel time
                  against the 157.3 TFLOP/s vector peak.
    `traffic` is the measured HBM traffic per launch of that same static pass (FETCH_SIZE doubled per the guide's gfx950 correction +
    WRITE_SIZE), scaled by this rank's share of the pixels; null when the workload has no static record."""
    st = {}
    try:
        st = json.load(open(os.path.join(ROOT, "profiles", "roofline_static.json"))).get(workload, {})
    except Exception:
        pass
    t = kernel_ms_mean * 1e-3
    share = my_px / st["pixels"] if st.get("pixels") else 1.0
    hbm_frac = achieved_gbs / HBM_PEAK_GBS
    out = {
        "bound": None,  # named below, and only when this workload has a static record to compare the HBM fraction with
        "achieved": round(achieved_gbs, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(hbm_frac, 4),
        "traffic": None,
        "kernel": kernel,
        "kernel_ms_mean": round(kernel_ms_mean, 5),
        "kernel_ms_min": None if kernel_ms_min is None else round(kernel_ms_min, 5),
        "kernel_ms_scope": scope,
        "algorithmic_bytes_per_launch": algorithmic_bytes,
    }
    fracs = {"hbm": hbm_frac}
    if st.get("hbm_traffic_bytes_per_launch"):
        out["traffic"] = int(st["hbm_traffic_bytes_per_launch"] * share)
        out["traffic_source"] = f"{st['source']} (static: separate --pmc passes of this build, not this run)"
    if st.get("valu_wave_insts_per_launch"):
        simd_cycles = SIMDS * MAX_CLOCK_HZ * t
        insts = st["valu_wave_insts_per_launch"] * share
        busy = st.get("valu_active_cycles_per_launch", 0.0) * share / simd_cycles
        lower = insts * 2.3 / simd_cycles
        f = st["valu_model_issue_cycles_per_launch"] * share / simd_cycles if st.get("valu_model_issue_cycles_per_launch") else busy
        fracs["valu"] = f
        # the static mix prices every instruction once, the kernel runs its loops' instructions many times: a model that comes out above
        # 1 says "at the issue roofline, and the executed mix is cheaper than the static one" — reported as 1 with the raw value beside it
        out["valu_issue"] = {"frac": round(min(f, 1.0), 4), **({"model_uncapped": round(f, 4)} if f > 1.0 else {}), "model_cycles_per_inst": st.get("valu_model_cycles_per_inst"), "valu_busy_rocprof": round(busy, 4),
                             "lower_bound": round(lower, 4), "insts_per_px": round(st["valu_wave_insts_per_launch"] * 64 / st["pixels"], 1),
                             "peak": "1024 SIMDs x 2.4 GHz", "source": f"{st['source']} (static) + static ISA mix x profiles/r1_valu_issue_cost.txt"}
    if st.get("flops_per_px"):
        tf = st["flops_per_px"] * my_px / t / 1e12
        out["fp32"] = {"achieved": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / VALU_PEAK_TFLOPS, 4),
                       "flops_per_px": st["flops_per_px"], "source": "DESIGN.md §7c operator count (static)"}
    # the binding roofline is the larger of the fractions — when there is more than the HBM one to compare: a workload without a static
    # record (profiles/roofline_static.json) says null rather than claim "hbm" by default
    out["bound"] = max(fracs, key=fracs.get) if len(fracs) > 1 else None
    if out["bound"] is None:
        out["bound_note"] = "no static counter record for this workload in profiles/roofline_static.json: only the HBM fraction is known"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: 4k_deferred_gi on one GPU (the headline pass); 4k_probe_gi_chain at N > 1 (BASELINE.json configs[3]: probe-GI "
                         "lighting + post chain, row-sharded, final RGBA8 image gathered)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather (diagnostic only)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: one lit target, the all-gather of frame i finishes before frame i+1 is shaded")
    ap.add_argument("--force-gather", action="store_true", help="N=1: run the exchange path anyway, through a one-rank RCCL communicator (rehearsal of the N>1 loop)")
    ap.add_argument("--shadow-samples", type=float, default=None, help="traced workload: sun shadow rays per pixel (default: the reference's 8)")
    ap.add_argument("--atrium-subdiv", type=int, default=8, help="traced workload: tessellation of the atrium the rays are traced against (8: 23.8 K triangles, the "
                    "default; 24: 214 K)")
    ap.add_argument("--rt-bounces", type=int, default=0, help="traced workload: sah_rt_set_bounces for the GI generators (the reference: 0)")
    ap.add_argument("--watchdog-s", type=float, default=300.0, help="N>1: end the rank when a phase makes no progress for this long (0: never)")
    ap.add_argument("--one-work-stream", action="store_true", help="N>1 chain: mips 1.. + tonemap of frame i on the lighting stream instead of beside the lighting of frame i+1")
    ap.add_argument("--exchange", choices=["rccl", "ipc"], default="rccl", help="N>1: how the library's gathers travel — ncclAllGather (default) or the direct "
                    "exchange over peer-mapped memory (sah_ipc_*: every rank copies its rows straight into every peer's buffer; handles go through torch.distributed)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N>1 launched on a box with ONE GPU: every rank uses cuda:0, torch.distributed runs over "
                    "gloo and the exchange is the direct one (RCCL refuses two ranks on a device) — the whole N-rank control flow of this file, "
                    "for rehearsal; the number it prints is not a scaling measurement")
    ap.add_argument("--torch-gather", action="store_true", help="N>1: gather with torch.distributed instead of the library's sah_allgather_rows")
    ap.add_argument("--ramp-ms", type=float, default=200.0, help="untimed: run the step back to back for this long before the W warm-up steps, so that the "
                    "timed region does not start in the GPU's idle power state (reported in config.clock_ramp_ms)")
    ap.add_argument("--repack-lpv", action="store_true", help="LPV / cache workloads: lpv_generation = probe_generation = 0, i.e. the library rebuilds its gather copy of the LPV on every "
                    "step (5 us + a launch), as it must when the volumes change every frame; default: the volumes of this benchmark never change, so "
                    "their change counter stays at 1 and the copy made by the first step is kept (config.lpv_gather_copy says which)")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2], help="one-GPU chain workloads: 2 = the post chain of a frame runs on a second "
                    "stream beside the lighting of the next one (as the N > 1 loop always does); every frame still completes inside the timed region")
    ap.add_argument("--strict-tonemap", action="store_true", help="chain workloads: the strict composite (codes bit-identical to the oracle) instead of "
                    "SAH_TONEMAP_TOLERANCE_1CODE (within one R8G8B8A8 code of it: north_star's tolerance for the final image)")
    ap.add_argument("--synth-device", choices=["cuda", "cpu"], default="cuda", help="where the synthetic inputs are generated (same values either way).  cpu: no torch "
                    "kernel is launched before the timed passes — for rocprofv3 --pmc runs of the 8K workloads, whose input synthesis on the GPU dies inside the "
                    "profiler's dispatch interception (profiles/README.md, round 4)")
    ap.add_argument("--no-light-stats", action="store_true", help="light workloads: skip the lights-per-tile / per-pixel statistics (torch kernels on full-frame tensors)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU-oracle sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from androidrenderer_amd import _abi, chain as chain_mod, frame, images, lib, scene, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    rehearsal = args.rehearse_on_one_gpu and world > 1
    if rehearsal:
        local_rank = 0
        args.exchange = "ipc"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if rehearsal else dev  # where the small all-reduces of this file live (gloo has no GPU tensors here)
    exchange = world > 1 or args.force_gather  # the exchange step is part of the loop
    torch_pg = world > 1 or (args.force_gather and args.torch_gather)  # torch.distributed: barrier + max over ranks (+ --torch-gather)
    saved_stdout = None
    if exchange:
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the one JSON line by pointing
        # fd 1 at stderr until the result is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N > 1: a collective that never completes would leave the launcher waiting for ever.  A watchdog thread ends the rank with a message
    # naming the phase when nothing has moved for --watchdog-s seconds (phases are marked with beat(); 0 disables it).
    heart = {"t": time.monotonic(), "phase": "process group"}

    def beat(phase):
        heart["t"], heart["phase"] = time.monotonic(), phase

    if world > 1 and args.watchdog_s > 0:
        import threading

        def watch():
            while True:
                time.sleep(5.0)
                if time.monotonic() - heart["t"] > args.watchdog_s:
                    print(f"[bench] rank {rank}: no progress for {args.watchdog_s:.0f} s in phase '{heart['phase']}' — giving up", file=sys.stderr, flush=True)
                    os._exit(124)
        threading.Thread(target=watch, daemon=True).start()
    if rehearsal:
        dist.init_process_group(backend="gloo")
    elif world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)
    elif torch_pg:
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29511", rank=0, world_size=1, device_id=dev)

    if args.workload is None:
        args.workload = "4k_probe_gi_chain" if world > 1 else "4k_deferred_gi"
    wl = WORKLOADS[args.workload]
    W, H = wl["res"]
    sun_mode = {"off": _abi.SHADOW_MODE_OFF, "csm": _abi.SHADOW_MODE_CSM, "rt": _abi.SHADOW_MODE_RT}[wl["sun"]]
    gi_kind = {"none": _abi.GI_NONE, "lpv": _abi.GI_LPV, "cache": _abi.GI_CACHE, "rtgi": _abi.GI_RTGI}[wl["gi"]]
    n_lights = wl.get("lights", 0)
    chain = bool(wl.get("chain"))

    beat("inputs")
    # ---- inputs (identical on every rank: generated from fixed seeds) ----------------------------------------------
    lights = None
    if n_lights:
        lights = synth.point_lights(scene.SceneView.default(W, H), n_lights, wl["radius"], seed=8)
    fr = frame.LightingInputs(W, H, seed=2, sun_mode=sun_mode, gi=gi_kind, flavour=wl["gbuffer"], shadowmap_res=4096, lights=lights,
                              synth_device=str(dev) if args.synth_device == "cuda" else "cpu", shadow=wl.get("shadow", "noise"))
    fr.lpv_generation = 0 if args.repack_lpv else 1
    fr.probe_generation = 0 if args.repack_lpv else 1  # (the traced workload folds probes every step: sah_probe_update drops the copy anyway)
    d_arr = fr.device_arrays(dev)
    bytes_per_pixel = fr.bytes_per_pixel()

    beat("library context + communicator")
    # ---- the library context: its own communicator unless --torch-gather ------------------------------------------
    gather = exchange and not args.no_gather
    lib_gather = gather and not args.torch_gather
    comm_id = None
    use_ipc = lib_gather and args.exchange == "ipc" and world > 1
    if lib_gather and not use_ipc:
        if world > 1:  # rank 0's ncclUniqueId to everybody
            box = [lib.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm_id = box[0]
        else:
            comm_id = lib.comm_unique_id()
    comm_note = None
    try:
        ctx = lib.Context(device=local_rank, rank=rank, world=world, comm_id=comm_id)
        ok = 1
    except Exception as err:  # the communicator could not be built on this rank: every rank falls back together, and the line says so
        ctx, ok, comm_note = None, 0, str(err)
    if world > 1 and lib_gather:
        flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        if not lib_gather:
            raise SystemExit(f"sah_create failed: {comm_note}")
        print(f"[bench] rank {rank}: library communicator unavailable ({comm_note}); all ranks use torch.distributed for the exchange", file=sys.stderr)
        lib_gather = False
        del ctx
        ctx = lib.Context(device=local_rank, rank=rank, world=world, comm_id=None)
        comm_note = comm_note or "another rank failed to build the library communicator"
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def allgather_handles(b):  # the channel the direct exchange's IPC handles travel over
        out_h = [None] * world
        dist.all_gather_object(out_h, b)
        return out_h
    if use_ipc:
        chain_mod.connect_direct_exchange(ctx, allgather_handles)
    comm_stream = None
    if lib_gather and not args.no_overlap:
        comm_stream = torch.cuda.Stream(device=dev)
        if not chain:
            ctx.comm_set_stream(comm_stream.cuda_stream)

    # row shard of this rank: ceil(H / world) rows per gather slot, clipped to the image (androidrenderer_amd/shard.py)
    rows_per = -(-H // world)
    r0, r1 = shard.lighting_rows(H, world, rank)
    if world > 1:
        fr.row_begin, fr.row_end = r0, r1
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

    traced = None
    if wl.get("traced"):
        from androidrenderer_amd import mesh
        geo_arrays = mesh.to_device(mesh.atrium(args.atrium_subdiv).arrays(), dev)
        geo = mesh.geometry(geo_arrays, [])
        ctx.gbuffer_render(geo, fr.view.gpu_data, images.gbuffer(d_arr))
        ctx.rt_set_bounces(args.rt_bounces)
        noise_t = torch.from_numpy(synth.rng(31).integers(0, 256, (128, 128, 4), dtype=np.uint8)).to(dev)
        planes_rt = (images.plane(d_arr["depth"], _abi.FORMAT_D32_SFLOAT), images.plane(d_arr["normals"], _abi.FORMAT_R16G16B16A16_SFLOAT),
                     images.plane(noise_t, _abi.FORMAT_R8G8B8A8_UNORM), images.plane(d_arr["ao"], _abi.FORMAT_R32_SFLOAT),
                     images.plane(d_arr["shadow_mask"], _abi.FORMAT_R32_SFLOAT))
        # the irradiance cache's own rays: r.GI.Cache.UpdatesPerFrame = 1024 probes x 400 GI rays, folded into the atlases the Lighting
        # pass samples (irradiance_cache.cpp:21-23, 585-724)
        import ctypes as C
        n_probes = 1024
        cells = synth.rng(33).permutation(32 * 32 * 32)[:n_probes]
        probe_ids = torch.from_numpy(np.stack([cells % 32, (cells // 32) % 32, cells // 1024], axis=-1).astype(np.int32).reshape(-1)).to(dev)
        trace_results = torch.zeros((n_probes, 20, 20, 4), dtype=torch.int16, device=dev)
        light_cache = torch.zeros((32, 416, 416), dtype=torch.int32, device=dev)
        average = torch.zeros((32, 32, 32), dtype=torch.int32, device=dev)
        sky_luts = _abi.SkyLuts(images.plane(d_arr["sky_t"], _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(d_arr["sky_v"], _abi.FORMAT_R16G16B16A16_SFLOAT))
        pt = _abi.ProbeTraceDesc()
        for c, (cmin, spacing) in enumerate(fr.probe_cascades()):
            pt.cascades[c].probe_spacing = spacing
            for i in range(3):
                pt.cascades[c].min[i] = cmin[i]
        pt.probes_to_update, pt.num_probes = probe_ids.data_ptr(), n_probes
        pt.sun, pt.sky, pt.noise = C.pointer(fr.sun.constants), C.pointer(sky_luts), C.pointer(planes_rt[2])
        pt.probe_irradiance = images.volume(d_arr["probe_irr"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
        pt.probe_depth = images.volume(d_arr["probe_depth"], _abi.FORMAT_R16G16_SFLOAT)
        pt.probe_validity = images.volume(d_arr["probe_val"], _abi.FORMAT_R8_UNORM)
        pt.probe_size[0], pt.probe_size[1] = 5, 6
        pt.trace_results = images.volume(trace_results, _abi.FORMAT_R16G16B16A16_SFLOAT)
        atlases = _abi.ProbeAtlases(pt.probe_irradiance, images.volume(light_cache, _abi.FORMAT_B10G11R11_UFLOAT_PACK32), pt.probe_depth,
                                    images.volume(average, _abi.FORMAT_B10G11R11_UFLOAT_PACK32), pt.probe_validity)
        rtgi_rb, rtgi_ri = (torch.zeros((H, W, 4), dtype=torch.int16, device=dev) for _ in range(2))
        if args.shadow_samples is not None:
            fr.sun.constants.num_shadow_samples = args.shadow_samples
        e = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        torch.cuda.synchronize()
        e[0].record()
        rt_stats = ctx.rt_build(geo)
        e[1].record()

        trace_rows = [(0, 0)]  # N > 1: the rows this rank's lighting reads (set below, once the row plan exists); the structure and the probes are replicated

        def trace_planes():
            ctx.probe_trace(pt)
            ctx.probe_update(atlases, pt.trace_results, probe_ids.data_ptr(), n_probes)
            for r0, r1 in trace_rows:
                ctx.rt_set_rows(r0, r1)
                ctx.rtao(fr.view.gpu_data, planes_rt[0], planes_rt[1], planes_rt[2], 1, 8.0, planes_rt[3])
                ctx.sun_shadow_mask(fr.view.gpu_data, fr.sun.constants, planes_rt[0], planes_rt[1], planes_rt[2], planes_rt[4])
            ctx.rt_set_rows(0, 0)
        trace_planes()  # warm
        torch.cuda.synchronize()
        e[2].record()
        ctx.rtao(fr.view.gpu_data, planes_rt[0], planes_rt[1], planes_rt[2], 1, 8.0, planes_rt[3])
        e[3].record()
        ctx.sun_shadow_mask(fr.view.gpu_data, fr.sun.constants, planes_rt[0], planes_rt[1], planes_rt[2], planes_rt[4])
        e[4].record()
        ctx.probe_trace(pt)
        e[5].record()
        ctx.probe_update(atlases, pt.trace_results, probe_ids.data_ptr(), n_probes)
        e[6].record()
        # (not part of this workload's frame — the RTGI mode's generator, one GI ray per pixel — timed once for the record)
        ctx.rtgi_trace(fr.view.gpu_data, fr.sun.constants, sky_luts, planes_rt[0], planes_rt[1], planes_rt[2],
                       images.plane(rtgi_rb, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(rtgi_ri, _abi.FORMAT_R16G16B16A16_SFLOAT))
        e[7].record()
        torch.cuda.synchronize()
        tr_dist = trace_results.view(torch.float16)[..., 3].float()
        traced = {"triangles": rt_stats[0], "levels": rt_stats[2], "gi_bounces": args.rt_bounces, "rt_build_ms": round(e[0].elapsed_time(e[1]), 4), "rtao_ms": round(e[2].elapsed_time(e[3]), 4),
                  "sun_shadow_mask_ms": round(e[3].elapsed_time(e[4]), 4), "probe_trace_ms": round(e[4].elapsed_time(e[5]), 4),
                  "probe_update_ms": round(e[5].elapsed_time(e[6]), 4), "probes_per_frame": n_probes,
                  "rtgi_trace_ms_not_in_frame": round(e[6].elapsed_time(e[7]), 4),
                  "shadow_samples": float(fr.sun.constants.num_shadow_samples), "ao_unoccluded_fraction": round(float((d_arr["ao"] == 1).float().mean()), 4),
                  "mask_lit_fraction": round(float(d_arr["shadow_mask"].mean()), 4),
                  "mask_pixels_0_between_1": [round(float((d_arr["shadow_mask"] == 0).float().mean()), 4),
                                              round(float(((d_arr["shadow_mask"] > 0) & (d_arr["shadow_mask"] < 1)).float().mean()), 4),
                                              round(float((d_arr["shadow_mask"] == 1).float().mean()), 4)],
                  "probe_rays_hit_front_back_miss": [round(float((tr_dist > 0).float().mean()), 4), round(float((tr_dist < 0).float().mean()), 4)],
                  "rtgi_rays_hit_fraction": round(float((rtgi_rb.view(torch.float16)[..., 3].float() != 0).float().mean()), 4)}

    tm_flags = 0 if args.strict_tonemap else _abi.TONEMAP_TOLERANCE_1CODE
    pipelined = chain and gather and lib_gather and comm_stream is not None
    # one GPU, no exchange: the same two-frames-in-flight loop when asked for (--frames-in-flight 2): the post chain of frame i on a second
    # stream beside the lighting of frame i + 1 (the gathers of PipelinedChain are no-ops without a communicator)
    if chain and not pipelined and world == 1 and not gather and traced is None and args.frames_in_flight == 2:
        pipelined = True
        comm_stream = torch.cuda.Stream(device=dev)
    pc = sc = None

    def build_pipelined():
        # the whole frame, sharded, two frames in flight: both exchanges run on the side stream beside compute (chain.py: PipelinedChain)
        pc = chain_mod.PipelinedChain(ctx, fr, d_arr, rank, world, comm_stream, None if args.one_work_stream else torch.cuda.Stream(device=dev),
                                      tonemap_flags=tm_flags)
        if use_ipc:
            pc.register_direct_exchange(allgather_handles)
        sc = pc.sets[0]
        if traced is not None and world > 1:
            trace_rows[:] = [r for r in (tuple(sc.plan.lit_rows), tuple(sc.plan.lit_wrap_rows)) if r[1] > r[0]]

        def step(i, e0=None, e1=None):
            if traced is not None:
                trace_planes()  # on the work stream, in front of this frame's lighting
            pc.submit((e0, e1) if e0 is not None else None)

        def drain():
            pc.flush()
            ctx.comm_wait()
        return pc, sc, step, drain

    def build_stepwise():
        # the whole frame, sharded (chain.py), one frame at a time: every exchange goes through the library (torch path only with --torch-gather)
        sc = chain_mod.ShardedChain(ctx, fr, d_arr, rank, world, tonemap_flags=tm_flags)
        if use_ipc:
            sc.register_direct_exchange(allgather_handles)
        q, per = sc.plan.mip1_rows_per_rank, sc.plan.rows_per_rank
        mip_bytes, out_bytes = sc.mip1_alloc.view(torch.uint8).view(-1), sc.out_alloc.view(-1)
        mip_slot_bytes, out_slot_bytes = q * sc.mip1_alloc.shape[1] * 8, per * W * 4
        if traced is not None and world > 1:
            trace_rows[:] = [r for r in (tuple(sc.plan.lit_rows), tuple(sc.plan.lit_wrap_rows)) if r[1] > r[0]]

        def step(i, e0=None, e1=None):
            if traced is not None:
                trace_planes()
            if e0 is not None:
                e0.record()
            sc.lighting()
            if e1 is not None:
                e1.record()
            sc.reduce()
            if gather and lib_gather:
                sc.exchange_mip()
            elif gather:
                dist.all_gather_into_tensor(mip_bytes, mip_bytes[rank * mip_slot_bytes:(rank + 1) * mip_slot_bytes])
            sc.composite()
            if gather and lib_gather:
                sc.exchange_final()
            elif gather:  # torch has no reversed-rank gather: gather in rank order into a scratch image (diagnostic path, not assembled)
                k = sc.plan.out_slot
                dist.all_gather_into_tensor(out_bytes, out_bytes[k * out_slot_bytes:(k + 1) * out_slot_bytes].clone())

        def drain():
            ctx.comm_wait()
        return None, sc, step, drain

    def chain_px(sc):
        return W * (sc.plan.lit_rows[1] - sc.plan.lit_rows[0] + sc.plan.lit_wrap_rows[1] - sc.plan.lit_wrap_rows[0]) if world > 1 else W * H

    if pipelined:
        pc, sc, step, drain = build_pipelined()
        my_px = chain_px(sc)
    elif chain:
        pc, sc, step, drain = build_stepwise()
        my_px = chain_px(sc)
    else:
        # N > 1: two lit targets, so that the all-gather of frame i (side stream) overlaps the shading of frame i + 1; a target is reused
        # only after its gather has completed (stream-level wait).  Every gather finishes inside the timed region.
        nbuf = 2 if (gather and not args.no_overlap) else 1
        shard_bytes = rows_per * W * 8
        bufs = []
        for _ in range(nbuf):
            lf = torch.zeros((rows_per * world, W, 4), dtype=torch.int16, device=dev)  # equal slots for the all-gather
            desc_b, keep_b = fr.describe(d_arr, lf[:H])
            lb = lf.view(torch.uint8).view(-1)  # RCCL has no int16: the rows travel as bytes
            if use_ipc:
                ctx.ipc_register(lf.data_ptr(), lf.numel() * 2, allgather_handles(ctx.ipc_export(lf.data_ptr(), lf.numel() * 2)))
            bufs.append({"lit": lf[:H], "desc": desc_b, "keep": keep_b, "bytes": lb, "slot": lb[rank * shard_bytes:(rank + 1) * shard_bytes],
                         "plane": images.plane(lf[:H], _abi.FORMAT_R16G16B16A16_SFLOAT)})
        pending = [None] * nbuf

        def step(i, e0=None, e1=None):
            b = bufs[i % nbuf]
            if pending[i % nbuf] is not None:  # the compute stream waits for the gather that last used this target (and only for that one)
                if lib_gather:
                    torch.cuda.current_stream().wait_event(pending[i % nbuf])
                else:
                    pending[i % nbuf].wait()
                pending[i % nbuf] = None
            if e0 is not None:
                e0.record()
            if r1 > r0:
                ctx.lighting(b["desc"])
            if e1 is not None:
                e1.record()
            if gather and lib_gather:  # in place: this rank's rows are its slot of the image
                ctx.allgather_rows(b["plane"], rows_per, rows_per * world)
                if comm_stream is not None:
                    pending[i % nbuf] = comm_stream.record_event()
            elif gather:
                if nbuf > 1:
                    pending[i % nbuf] = dist.all_gather_into_tensor(b["bytes"], b["slot"], async_op=True)
                else:
                    dist.all_gather_into_tensor(b["bytes"], b["slot"])

        def drain():
            ctx.comm_wait()
            for k in range(nbuf):
                if pending[k] is not None:
                    if not lib_gather:
                        pending[k].wait()
                    pending[k] = None
        my_px = W * (r1 - r0) if world > 1 else W * H

    # N > 1: the same workload unsharded on this rank's GPU, timed before the sharded loop — strong scaling of ONE workload can then be
    # read off this line alone (the driver's N = 1 run measures the headline lighting pass, not necessarily this workload)
    beat("unsharded reference")
    single_gpu = None
    ref_image = None  # chain workloads: the unsharded frame's final image, to hold the sharded loop's last frames against after the timed region
    if world > 1 or args.force_gather:
        if chain:
            ref = chain_mod.ShardedChain(ctx, fr, d_arr, 0, 1, tonemap_flags=tm_flags)
            ref_step = lambda: ref.step(gather=False)
        else:
            ref_lit = torch.zeros((H, W, 4), dtype=torch.int16, device=dev)
            saved_rows = (fr.row_begin, fr.row_end)
            fr.row_begin = fr.row_end = 0
            ref_desc, ref_keep = fr.describe(d_arr, ref_lit)
            fr.row_begin, fr.row_end = saved_rows
            ref_step = lambda: ctx.lighting(ref_desc)
        for _ in range(5):
            ref_step()
        torch.cuda.synchronize()
        r_e0, r_e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r_e0.record()
        for _ in range(30):
            ref_step()
        r_e1.record()
        torch.cuda.synchronize()
        ref_ms = r_e0.elapsed_time(r_e1) / 30
        single_gpu = {"ms_per_step": round(ref_ms, 5), "value": round(W * H / (ref_ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
                      "note": "the same workload unsharded on rank 0's GPU, 30 steps, GPU time between two events"}
        del ref_step
        if chain:
            if gather and lib_gather and traced is None:  # (traced: a rank traces only its own rows of the AO / shadow-mask planes)
                ref_image = ref.out.clone()
            del ref
        torch.cuda.empty_cache()

    # N > 1, two frames in flight: before anything is timed, three frames of the loop are held against the unsharded frame on every
    # rank.  The loop's streams and events have met more than one GPU only here: if a frame differs, every rank falls back together to
    # the one-frame-at-a-time loop (exchanges on the work stream's own order), checks that one too, and the line says what ran.
    preflight = None
    if pipelined and ref_image is not None and exchange:
        beat("pre-flight")
        for i in range(3):
            step(i)
        drain()
        torch.cuda.synchronize()
        ok = int(bool(torch.equal(pc.image(1), ref_image)) and bool(torch.equal(pc.image(2), ref_image)))
        if os.environ.get("SAH_BENCH_FAIL_PREFLIGHT") == "1":  # test hook: take the fall-back path
            ok = 0
        if torch_pg:
            t = torch.tensor([ok], dtype=torch.int32, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        preflight = {"two_frames_in_flight_ok": bool(ok), "fallback": None}
        if not ok:
            print(f"[bench] rank {rank}: the two-frames-in-flight loop's frames differ from the unsharded frame: falling back to one frame at a time", file=sys.stderr)
            if use_ipc:  # the registrations are found by address: give them back before the allocator can hand the addresses out again
                pc.unregister_direct_exchange()
            pc = None
            pipelined = False
            torch.cuda.empty_cache()
            pc, sc, step, drain = build_stepwise()
            my_px = chain_px(sc)
            for i in range(2):
                step(i)
            drain()
            torch.cuda.synchronize()
            ok2 = int(bool(torch.equal(sc.out, ref_image)))
            if torch_pg:
                t = torch.tensor([ok2], dtype=torch.int32, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok2 = int(t.item())
            preflight["fallback"] = "one frame at a time"
            preflight["fallback_ok"] = bool(ok2)

    if args.ramp_ms > 0:
        t_ramp = time.perf_counter()
        k = 0
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            for _ in range(8):
                step(k)
                k += 1
            drain()
            torch.cuda.synchronize()
    beat("warm-up")
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    beat("timed region")
    if torch_pg:
        dist.barrier()
        torch.cuda.synchronize()
    # GPU time of the timed region: ONE pair of HIP events on the launch stream around all K steps (round 2 recorded a pair per step:
    # each record is a barrier packet of ~10 us on the queue, 20 us per 190 us step that were measurement, not work —
    # profiles/r3_trace_bench20_gaps.txt).  Created and recorded once before t0 so that their lazy creation is outside the region.
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    g1.record()
    torch.cuda.synchronize()
    # chain workloads: the roofline is the Lighting kernel's, one pass of several in a step, so those keep a pair per step around the
    # sah_lighting call (2 x ~10 us on a ~0.9 ms step); created and recorded before t0 as well
    ev = []
    if chain:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a_, b_ in ev:
            a_.record()
            b_.record()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    g0.record()
    for i in range(args.steps):
        if chain:
            step(i, ev[i][0], ev[i][1])
        else:
            step(i)
    drain()
    g1.record()
    torch.cuda.synchronize()
    if torch_pg:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loop_ms = g0.elapsed_time(g1) / args.steps
    if chain:
        per_step = sorted(a_.elapsed_time(b_) for a_, b_ in ev)
        kernel_ms_mean, kernel_ms_min = sum(per_step) / len(per_step), per_step[0]
        kernel_scope = "HIP events around each sah_lighting call of the timed region, on its stream (main kernel + fix-up)" + (
            "; two frames in flight: the previous frame's post chain runs beside it on a second stream, so this is not the kernel's time alone"
            if pipelined and not args.one_work_stream else "")
    else:
        kernel_ms_mean, kernel_ms_min = loop_ms, None
        kernel_scope = ("one pair of HIP events on the launch stream around all K steps of the timed region, / K: everything a sah_lighting call "
                        "enqueues (main kernel with its sky workgroups + fix-up) and the gaps between launches")
    if torch_pg:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # after the timed region: every rank holds the last frames its loop assembled (both buffer sets of the pipelined loop) against the
    # unsharded frame it rendered itself before the loop — the gathers moved the right rows to the right places on every rank, or not
    beat("verification")
    sharded_equals_unsharded = None
    if ref_image is not None:
        outs = [pc.image(args.steps - 1), pc.image(args.steps - 2)] if pipelined and args.steps > 1 else [pc.image(args.steps - 1) if pipelined else sc.out]
        same = int(all(bool(torch.equal(o, ref_image)) for o in outs))
        if os.environ.get("SAH_BENCH_FAIL_VERIFY") == "1":  # test hook: the line must then carry no throughput and the exit code say so
            same = 0
        if torch_pg:
            t = torch.tensor([same], dtype=torch.int32, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            same = int(t.item())
        sharded_equals_unsharded = bool(same)
        if not same:
            print(f"[bench] rank {rank}: the sharded loop's final image differs from the unsharded frame", file=sys.stderr)
    # A loop that moved wrong rows has no throughput: the line then carries value null and an error, and every rank exits non-zero (the
    # verdicts above are all-reduced, so the ranks agree).  Likewise when the pre-flight fell back and the fall-back was wrong as well.
    failure = None
    if sharded_equals_unsharded is False:
        failure = "the sharded loop's last frames differ from the unsharded frame on at least one rank"
    elif preflight is not None and preflight.get("fallback") and not preflight.get("fallback_ok"):
        failure = "pre-flight: the two-frames-in-flight loop AND the one-frame-at-a-time fall-back differ from the unsharded frame"

    if rank == 0:
        px = W * H
        value = px * args.steps / elapsed / 1e6
        achieved = bytes_per_pixel * my_px / (kernel_ms_mean * 1e-3) / 1e9
        sun_txt = {"csm": "sun CSM 4x4096^2 D16 PCF", "rt": "sun RT (shadow-mask plane, half-precision BRDF)", "off": "sun off"}[wl["sun"]]
        gi_txt = {"none": "no GI", "lpv": "LPV GI gather + AO", "cache": "irradiance-cache probe gather", "rtgi": "RTGI reconstruction"}[wl["gi"]]
        parts = [sun_txt, gi_txt, "emissive", "sky"]
        if n_lights:
            parts.insert(1, f"{n_lights} point lights (r={wl['radius']} m) with LDS tile culling")
        what = "fused deferred lighting (" + " + ".join(parts) + ")"
        if chain:
            what += " + copy scene + bloom pyramid + tonemap composite"
        if world == 1:
            par = "single GPU" + (" + one-rank RCCL communicator (rehearsal of the exchange)" if exchange else "")
        elif chain:
            par = f"row-shard x{world}: lighting rows + halo, all-gather of bloom mip 1, all-gather of the RGBA8 rows (reversed rank order)"
            if rehearsal:
                par += " — REHEARSAL: all ranks on one GPU, not a scaling measurement"
        else:
            par = f"row-shard x{world} + RCCL all-gather of the lit rows"
        out = {
            "metric": "lit Mpixels/sec (deferred+GI pass) at 4K",
            "value": None if failure else round(value, 1),
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
                "clock_ramp_ms": args.ramp_ms,
                "tonemap": None if not chain else ("strict" if args.strict_tonemap else "SAH_TONEMAP_TOLERANCE_1CODE (within one code of the strict composite)"),
                "lpv_gather_copy": None if gi_kind != _abi.GI_LPV else ("rebuilt every step (lpv_generation 0)" if args.repack_lpv else
                                                                          "kept across steps (lpv_generation 1: the LPV volumes of this benchmark never change)"),
                "probe_gather_copy": None if gi_kind != _abi.GI_CACHE else (
                    "rebuilt every step (sah_probe_update writes the atlas every step)" if wl.get("traced") else
                    "rebuilt every step (probe_generation 0)" if args.repack_lpv else "kept across steps (probe_generation 1: the atlases of this benchmark never change)"),
                "gbuffer": wl["gbuffer"],
                "parallelism": par,
                "gather": bool(gather),
                "gather_through": (("sah_allgather_rows (library, direct exchange over peer-mapped memory)" if use_ipc else "sah_allgather_rows (library, RCCL)") if lib_gather else "torch.distributed" + (f" (fallback: {comm_note})" if comm_note else "")) if gather else None,
                "gather_overlapped_with_next_frame": bool(gather and not args.no_overlap and (pipelined or not chain)),
                "post_chain_beside_next_frames_lighting": bool(pipelined and not args.one_work_stream),
                "frames_in_flight": 2 if pipelined else 1,
                "same_workload_on_one_gpu": single_gpu,
                # the N = 1 run of this file measures the headline lighting pass, a different workload from the sharded chain: the strong
                # scaling of THIS workload is its throughput here over its throughput unsharded on one of these GPUs
                # (null in a rehearsal: there every rank — and the one-GPU reference, timed while the others wait — shares ONE GPU)
                "speedup_vs_same_workload_on_one_gpu": None if (not single_gpu or rehearsal or failure) else round(value / single_gpu["value"], 3),
                "sharded_equals_unsharded": sharded_equals_unsharded,
                "preflight": preflight,
                "traced": traced,
            },
            "roofline": roofline(args.workload, world, achieved, kernel_ms_mean, kernel_ms_min, kernel_scope, bytes_per_pixel * my_px, my_px,
                                 "sah::k_lighting_tiled" if (n_lights or gi_kind in (_abi.GI_CACHE, _abi.GI_RTGI)) else "sah::k_lighting_fast"),
        }
        if failure:
            out["error"] = failure
            out["measured_but_invalid_Mpixels_per_s"] = round(value, 1)
        if n_lights and not args.no_light_stats:
            out["config"].update(light_stats(torch, fr, d_arr, lights, dev))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fr, args.cpu_seconds)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)  # teardown chatter goes to stderr as well
    ctx.close()
    if torch_pg:
        dist.barrier()
        dist.destroy_process_group()
    if failure:
        sys.exit(3)


def usable_cpus():
    """the CPUs this process may actually run on: its affinity mask, capped by the cgroup's CPU quota (a one-GPU box shares a 256-thread
    host: os.cpu_count() says 256 where 16 are granted, and 256 OpenMP threads on 16 CPUs time the scheduler, not the oracle)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(fr, target_s):
    """Times the CPU oracle (a port of the reference shaders, OpenMP over rows) on a bounded band of rows of the same frame
    (lighting pass only): built here with -O3 -march=native (oracle/Makefile target `native`), one warm-up, then the
    median of 5 repetitions."""
    import subprocess
    flags = "g++ -O2 (oracle/liboracle.so: the -O3 -march=native build failed)"
    so = os.path.join(ROOT, "oracle", "liboracle_native.so")
    try:
        # -march=native code must not travel: the build is tied to this machine's CPU (model + feature flags) and redone anywhere else
        import hashlib
        cpu = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
        key = hashlib.sha1("".join(cpu).encode()).hexdigest()
        key_file = os.path.join(ROOT, "oracle", "liboracle_native.cpu")
        same_cpu = os.path.exists(so) and os.path.exists(key_file) and open(key_file).read().strip() == key
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")] + ([] if same_cpu else ["-B"]) + ["native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        open(key_file, "w").write(key)
        flags = open(os.path.join(ROOT, "oracle", "liboracle_native.flags")).read().strip()
    except Exception:
        so = os.path.join(ROOT, "oracle", "liboracle.so")
    try:
        if not os.path.exists(so):
            from tests import util
            util.build_oracle()
        from androidrenderer_amd import _abi
        o = C.CDLL(so)
        o.orc_lighting.argtypes = [C.POINTER(_abi.LightingDesc)]
    except Exception as e:  # no oracle and no compiler: report, don't fail the bench
        return {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    W, H = fr.width, fr.height
    lit = np.zeros((H, W, 4), dtype=np.uint16)
    fr.row_begin = fr.row_end = 0
    d, keep = fr.describe(fr.arrays, lit)
    cores = usable_cpus()
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(cores)  # the oracle's OpenMP runtime: as many threads as there are CPUs to run them
    except Exception:
        cores = os.cpu_count() or 1  # (the runtime's own default then applies)
    mid = H // 2

    def timed(r0, r1):
        d.row_begin, d.row_end = r0, r1
        t = time.perf_counter()
        o.orc_lighting(C.byref(d))
        return time.perf_counter() - t
    timed(mid, min(H, mid + 16))                      # warm-up (thread pool, page faults)
    dt = max(timed(mid, min(H, mid + 32)), 1e-4)      # calibration
    reps = 5
    rows = int(max(32, min(H, 32 * (target_s / reps) / dt)))
    r0 = max(0, mid - rows // 2)
    r1 = min(H, r0 + rows)
    times = sorted(timed(r0, r1) for _ in range(reps))
    med = times[reps // 2]
    npx = W * (r1 - r0)
    return {"value": round(npx / med / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"rows [{r0},{r1}) of the same {W}x{H} frame ({npx} px): median of {reps} repetitions = {med:.3f} s "
                      f"(min {times[0]:.3f}, max {times[-1]:.3f}); CPU oracle (oracle/, {flags}, {cores} OpenMP threads) — a restatement of the "
                      f"reference shaders, not its Vulkan/lavapipe path"}


if __name__ == "__main__":
    main()
