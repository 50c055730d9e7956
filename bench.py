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

# BASELINE.json configs[0..4] plus diagnostics.  The headline (`metric`) is 4k_deferred_gi.
WORKLOADS = {
    "4k_deferred_gi": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv"),
    "4k_deferred_gi_scene_shadow": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv", shadow="scene"),  # CSM ray-cast from the atrium
    # inputs made on the GPU by the producer passes (f1, f2, f4): rasterised G-buffer and shadow cascades, LPV from RSM -> VPLs -> propagation
    "4k_deferred_gi_produced": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv", produced=True),
    "4k_deferred_gi_random": dict(res=(3840, 2160), gbuffer="random", sun="csm", gi="lpv"),
    "4k_deferred_only": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="none"),
    # configs[0]: 1280x720, a single directional light, deferred shading only — the reference's default sun (RT mode, directional_light.rt.slang) with
    # every shadow ray unoccluded (mask = 1), no GI overlay
    "720p_deferred_only": dict(res=(1280, 720), gbuffer="atrium", sun="rt", gi="none", mask="ones"),                      # configs[0]
    "1080p_deferred_gi": dict(res=(1920, 1080), gbuffer="atrium", sun="csm", gi="lpv"),
    "8k_deferred_gi": dict(res=(7680, 4320), gbuffer="atrium", sun="csm", gi="lpv"),
    # light radii per SURVEY.md §8-d: 64 lights r = 6 m, 256 lights r = 4 m, 1024 lights r = 3 m
    "1080p_64_lights": dict(res=(1920, 1080), gbuffer="random", sun="csm", gi="none", lights=64, radius=6.0),            # configs[1]
    "4k_256_lights": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="none", lights=256, radius=4.0),             # configs[2]
    "4k_probe_gi_chain": dict(res=(3840, 2160), gbuffer="atrium", sun="rt", gi="cache", chain=True),                     # configs[3]
    "4k_lpv_gi_chain": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv", chain=True),
    # the LPV mode's honest frame: the volumes' upkeep inside EVERY timed step, as the reference's frame has it (scene_renderer.cpp:374-401 ->
    # light_propagation_volume.cpp:548-760 inject_indirect_sun_light, 970-1063 propagate_lighting): clear, RSM of the atrium mesh for the four
    # cascades, VPL extraction and injection per cascade, 32 propagation steps (the last one stores the Lighting pass's gather copy) — then
    # lighting, copy scene, bloom, composite
    "4k_lpv_gi_frame": dict(res=(3840, 2160), gbuffer="atrium", sun="csm", gi="lpv", chain=True, lpv_frame=True),
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
                  mix priced with the measured issue costs (tools/pmc_to_static.py, profiles/r1_valu_issue_cost.txt).  Beside it
                  `lower_bound`, every instruction at the cheapest cost (2.3 cycles).  (rocprof's VALUBusy — SQ_ACTIVE_INST_VALU x 4 / SIMD
                  cycles — charged every instruction a whole quad-cycle and read 1.16-1.36 for the tiled kernels: a mis-scaled counter
                  is not a fraction, and the line no longer carries it.)
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
        # what the memory system really moved against what the algorithm needs (> 1: lines fetched more than once, or only partly used), and
        # that traffic as a fraction of the HBM peak over the live kernel time — the memory roofline the kernel actually sits under
        out["traffic_over_algorithmic"] = round(out["traffic"] / algorithmic_bytes, 3) if algorithmic_bytes else None
        out["traffic_frac_of_peak"] = round(out["traffic"] / t / 1e9 / HBM_PEAK_GBS, 4)
        fracs["hbm"] = max(hbm_frac, out["traffic"] / t / 1e9 / HBM_PEAK_GBS)
    if st.get("valu_wave_insts_per_launch"):
        simd_cycles = SIMDS * MAX_CLOCK_HZ * t
        insts = st["valu_wave_insts_per_launch"] * share
        lower = insts * 2.3 / simd_cycles
        f = st["valu_model_issue_cycles_per_launch"] * share / simd_cycles if st.get("valu_model_issue_cycles_per_launch") else lower
        fracs["valu"] = f
        # the static mix prices every instruction once, the kernel runs its loops' instructions many times: a model that comes out above
        # 1 says "at the issue roofline, and the executed mix is cheaper than the static one" — reported as 1 with the raw value beside it
        out["valu_issue"] = {"frac": round(min(f, 1.0), 4), **({"model_uncapped": round(f, 4)} if f > 1.0 else {}), "model_cycles_per_inst": st.get("valu_model_cycles_per_inst"),
                             "lower_bound": round(lower, 4), "insts_per_px": round(st["valu_wave_insts_per_launch"] * 64 / st["pixels"], 1),
                             "peak": "1024 SIMDs x 2.4 GHz", "source": f"{st['source']} (static) + static ISA mix x profiles/r1_valu_issue_cost.txt"}
    if st.get("flops_per_px"):
        tf = st["flops_per_px"] * my_px / t / 1e12
        out["fp32"] = {"achieved": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / VALU_PEAK_TFLOPS, 4),
                       "flops_per_px": st["flops_per_px"], "source": "DESIGN.md §7c operator count (static)"}
    # the binding roofline is the larger of the fractions — when there is more than the HBM one to compare: a workload without a static
    # record (profiles/roofline_static.json) says null rather than claim "hbm" by default
    out["bound"] = max(fracs, key=fracs.get) if len(fracs) > 1 else None
    if out["bound"] is not None and max(fracs.values()) < 0.6:
        # neither the issue model nor the measured memory traffic comes near its roofline: what the launch waits for is latency — too few waves
        # per SIMD, or a launch too short to fill the chip for long (small frames, row bands, gathers that miss)
        out["bound_by_fraction"] = out["bound"]
        out["bound"] = "latency"
        out["bound_note"] = (f"largest fraction {max(fracs.values()):.2f} ({out['bound_by_fraction']}): neither VALU issue nor HBM traffic reaches 0.6 of its peak — "
                             "latency / occupancy bound")
    if out["bound"] is None:
        out["bound_note"] = "no static counter record for this workload in profiles/roofline_static.json: only the HBM fraction is known"
    return out


class Run:
    """The state the stages of one benchmark run share (plain attributes; each stage below says what it adds)."""


def parse_args(argv=None):
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
    ap.add_argument("--two-work-streams", action="store_true", help="N>1 chain: lighting and copy + mip rows share the work stream, mips 2.. + tonemap run on a second one "
                    "(rounds 3-4's shape; default since round 5: three streams — the lighting of frame i+1 does not wait for the copy and mip rows of frame i)")
    ap.add_argument("--python-loop", action="store_true", help="two frames in flight: enqueue every pass from Python (chain.PipelinedChain) instead of through the library's own "
                    "frame loop (sah_chain_submit, chain.NativePipelinedChain: the default)")
    ap.add_argument("--exchange", choices=["rccl", "ipc"], default="rccl", help="N>1: how the library's gathers travel — ncclAllGather (default) or the direct "
                    "exchange over peer-mapped memory (sah_ipc_*: every rank copies its rows straight into every peer's buffer; handles go through torch.distributed)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N>1 launched on a box with ONE GPU: every rank uses cuda:0, torch.distributed runs over "
                    "gloo and the exchange is the direct one (RCCL refuses two ranks on a device) — the whole N-rank control flow of this file, "
                    "for rehearsal; the number it prints is not a scaling measurement")
    ap.add_argument("--torch-gather", action="store_true", help="N>1: gather with torch.distributed instead of the library's sah_allgather_rows")
    ap.add_argument("--ramp-ms", type=float, default=200.0, help="untimed: run the step back to back for this long before the W warm-up steps, so that the "
                    "timed region does not start in the GPU's idle power state (reported in config.clock_ramp_ms)")
    ap.add_argument("--lpv-copy", choices=["propagate", "rebuild", "kept"], default="propagate",
                    help="LPV workloads: where the library's interleaved gather copy of the three volumes comes from.  propagate (default): the frame's LPV "
                         "maintenance is the library's own — sah_lpv_propagate runs once before the loop on the synthetic volumes (one step; the benchmark's "
                         "volumes do not change afterwards, the reference's frame would run it every frame: light_propagation_volume.cpp:970-1063), its last "
                         "step stores the gather copy beside the volumes (SAH_GENERATION_TRACKED), and no Lighting pass rebuilds it; the same pass with the "
                         "copy rebuilt inside every step is timed beside it (config.lpv_gather_copy_rebuilt_every_step).  rebuild: lpv_generation 0, k_lpv_pack "
                         "runs inside every timed step — a frame whose volumes somebody else's pass rewrites.  kept: lpv_generation 1 on the synthetic volumes "
                         "(rounds 1-4's default)")
    ap.add_argument("--repack-lpv", action="store_true", help="(rounds 3-4; now the default) same as --lpv-copy rebuild --probe-copy rebuild")
    ap.add_argument("--probe-copy", choices=["patched", "rebuild", "kept"], default="patched",
                    help="irradiance-cache workloads: the library's fp32 copy of the irradiance atlas.  patched (default): the context tracks the atlas "
                         "(SAH_GENERATION_TRACKED) and every step re-widens the blocks of 1024 probes — the reference's r.GI.Cache.UpdatesPerFrame — through "
                         "sah_probe_notify_updated (the traced workload's own sah_probe_update does it by itself); rebuild: probe_generation 0, the whole atlas "
                         "every step; kept: probe_generation 1")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2], help="one-GPU chain workloads: 2 = the post chain of a frame runs on a second "
                    "stream beside the lighting of the next one (as the N > 1 loop always does); every frame still completes inside the timed region")
    ap.add_argument("--strict-tonemap", action="store_true", help="chain workloads: the strict composite (codes bit-identical to the oracle) instead of "
                    "SAH_TONEMAP_TOLERANCE_1CODE (within one R8G8B8A8 code of it: north_star's tolerance for the final image)")
    ap.add_argument("--synth-device", choices=["cuda", "cpu"], default="cuda", help="where the synthetic inputs are generated (same values either way).  cpu: no torch "
                    "kernel is launched before the timed passes (diagnostic: profiles/README.md \"8K under --pmc\" — the fault it was added for turned out to be "
                    "the light statistics' 16,384 outstanding dispatches, not the input synthesis)")
    ap.add_argument("--no-light-stats", action="store_true", help="light workloads: skip the lights-per-tile / per-pixel statistics (torch kernels on full-frame tensors)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU-oracle sample")
    args = ap.parse_args(argv)
    if args.repack_lpv:
        args.lpv_copy, args.probe_copy = "rebuild", "rebuild"
    return args


def self_launch(args, argv=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks as a CHILD process —
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` —
    relay its one JSON line to stdout (everything else it wrote there goes to stderr) and return its exit code.  Never exec: this process
    has not touched the GPU (nothing before this point imports torch; tests/test_bench_host.py holds it to that), and it stays the parent
    until the ranks have gone."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")  # (torch.distributed.run would set it, with a warning on stdout's neighbour)
    print("[bench] no launcher around --gpus %d: starting the ranks as a child process: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    line = None
    for text in child.stdout:
        if text.startswith('{"metric"'):
            line = text.rstrip("\n")
        else:
            sys.stderr.write(text)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] the ranks exited with code 0 but printed no result line", file=sys.stderr, flush=True)
        rc = 4
    return rc


def setup_distributed(args):
    """torch, the process group, this rank's device and the stdout discipline.  Adds: torch, dist, world, rank, local_rank, dev, red_dev, rehearsal,
    exchange, torch_pg, saved_stdout, beat."""
    import torch
    import torch.distributed as dist
    R = Run()
    R.args, R.torch, R.dist = args, torch, dist
    R.world = int(os.environ.get("WORLD_SIZE", "1"))
    R.rank = int(os.environ.get("RANK", "0"))
    R.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if R.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {R.world}: launch as `python bench.py --gpus N ...` (the file starts its own ranks) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    R.rehearsal = args.rehearse_on_one_gpu and R.world > 1
    if R.rehearsal:
        R.local_rank = 0
        args.exchange = "ipc"
    elif R.local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {R.rank} (local rank {R.local_rank}) has no GPU — this box shows {torch.cuda.device_count()}; --gpus N needs N GPUs "
                         "(to rehearse the N-rank control flow on one GPU: --rehearse-on-one-gpu)")
    torch.cuda.set_device(R.local_rank)
    R.dev = torch.device("cuda", R.local_rank)
    R.red_dev = torch.device("cpu") if R.rehearsal else R.dev  # where the small all-reduces of this file live (gloo has no GPU tensors here)
    R.exchange = R.world > 1 or args.force_gather  # the exchange step is part of the loop
    R.torch_pg = R.world > 1 or (args.force_gather and args.torch_gather)  # torch.distributed: barrier + max over ranks (+ --torch-gather)
    R.saved_stdout = None
    if R.exchange:
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the one JSON line by pointing
        # fd 1 at stderr until the result is printed
        sys.stdout.flush()
        R.saved_stdout = os.dup(1)
        os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N > 1: a collective that never completes would leave the launcher waiting for ever.  A watchdog thread ends the rank with a message
    # naming the phase when nothing has moved for --watchdog-s seconds (phases are marked with beat(); 0 disables it).
    heart = {"t": time.monotonic(), "phase": "process group"}

    def beat(phase):
        heart["t"], heart["phase"] = time.monotonic(), phase
    R.beat = beat
    if R.world > 1 and args.watchdog_s > 0:
        import threading

        def watch():
            while True:
                time.sleep(5.0)
                if time.monotonic() - heart["t"] > args.watchdog_s:
                    print(f"[bench] rank {R.rank}: no progress for {args.watchdog_s:.0f} s in phase '{heart['phase']}' — giving up", file=sys.stderr, flush=True)
                    os._exit(124)
        threading.Thread(target=watch, daemon=True).start()
    if R.rehearsal:
        dist.init_process_group(backend="gloo")
    elif R.world > 1:
        dist.init_process_group(backend="nccl", device_id=R.dev)
    elif R.torch_pg:
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29511", rank=0, world_size=1, device_id=R.dev)
    return R


def all_ranks_min(R, value):
    """an int every rank agrees on: the minimum over the ranks (1 = everybody fine)"""
    if not R.torch_pg:
        return int(value)
    t = R.torch.tensor([int(value)], dtype=R.torch.int32, device=R.red_dev)
    R.dist.all_reduce(t, op=R.dist.ReduceOp.MIN)
    return int(t.item())


def make_inputs(R):
    """The workload's synthetic frame, identical on every rank (fixed seeds).  Adds: wl, W, H, sun_mode, gi_kind, n_lights, chain, lights, fr, d_arr,
    bytes_per_pixel, probe_ids."""
    from androidrenderer_amd import _abi, frame, scene, synth
    args = R.args
    if args.workload is None:
        args.workload = "4k_probe_gi_chain" if R.world > 1 else "4k_deferred_gi"
    R.wl = wl = WORKLOADS[args.workload]
    R.W, R.H = wl["res"]
    R.sun_mode = {"off": _abi.SHADOW_MODE_OFF, "csm": _abi.SHADOW_MODE_CSM, "rt": _abi.SHADOW_MODE_RT}[wl["sun"]]
    R.gi_kind = {"none": _abi.GI_NONE, "lpv": _abi.GI_LPV, "cache": _abi.GI_CACHE, "rtgi": _abi.GI_RTGI}[wl["gi"]]
    R.n_lights = wl.get("lights", 0)
    R.chain = bool(wl.get("chain"))
    R.beat("inputs")
    R.lights = synth.point_lights(scene.SceneView.default(R.W, R.H), R.n_lights, wl["radius"], seed=8) if R.n_lights else None
    fr = frame.LightingInputs(R.W, R.H, seed=2, sun_mode=R.sun_mode, gi=R.gi_kind, flavour=wl["gbuffer"], shadowmap_res=4096, lights=R.lights,
                              synth_device=str(R.dev) if args.synth_device == "cuda" else "cpu", shadow=wl.get("shadow", "noise"))
    if wl.get("mask") == "ones":  # "deferred shading only": every shadow ray of the RT-mode sun unoccluded
        fr.arrays["shadow_mask"] = np.ones((R.H, R.W), dtype=np.float32)
    fr.lpv_generation = {"propagate": _abi.GENERATION_TRACKED, "rebuild": 0, "kept": 1}[args.lpv_copy]  # (propagate: see propagate_lpv below)
    # (the traced workload folds probes every step: sah_probe_update keeps a tracked copy current by itself)
    fr.probe_generation = {"patched": _abi.GENERATION_TRACKED, "rebuild": 0, "kept": 1}[args.probe_copy]
    R.fr = fr
    R.d_arr = fr.device_arrays(R.dev)
    R.bytes_per_pixel = fr.bytes_per_pixel()
    R.probe_ids = None
    if R.gi_kind == _abi.GI_CACHE and args.probe_copy == "patched" and not wl.get("traced"):
        cells = synth.rng(33).permutation(32 * 32 * 32)[:1024]
        R.probe_ids = R.torch.from_numpy(np.stack([cells % 32, (cells // 32) % 32, cells // 1024], axis=-1).astype(np.int32).reshape(-1)).to(R.dev)


def make_context(R):
    """The library context with its own communicator (unless --torch-gather), the direct exchange's connection and the side stream.
    Adds: ctx, gather, lib_gather, use_ipc, comm_note, comm_stream, allgather_handles, rows_per, r0, r1."""
    from androidrenderer_amd import chain as chain_mod, lib, shard
    args, torch, dist = R.args, R.torch, R.dist
    R.beat("library context + communicator")
    R.gather = R.exchange and not args.no_gather
    R.lib_gather = R.gather and not args.torch_gather
    comm_id = None
    R.use_ipc = R.lib_gather and args.exchange == "ipc" and R.world > 1
    if R.lib_gather and not R.use_ipc:
        if R.world > 1:  # rank 0's ncclUniqueId to everybody
            box = [lib.comm_unique_id() if R.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm_id = box[0]
        else:
            comm_id = lib.comm_unique_id()
    R.comm_note = None
    try:
        R.ctx = lib.Context(device=R.local_rank, rank=R.rank, world=R.world, comm_id=comm_id)
        ok = 1
    except Exception as err:  # the communicator could not be built on this rank: every rank falls back together, and the line says so
        R.ctx, ok, R.comm_note = None, 0, str(err)
    if R.world > 1 and R.lib_gather:
        ok = all_ranks_min(R, ok)
    if not ok:
        if not R.lib_gather:
            raise SystemExit(f"sah_create failed: {R.comm_note}")
        print(f"[bench] rank {R.rank}: library communicator unavailable ({R.comm_note}); all ranks use torch.distributed for the exchange", file=sys.stderr)
        R.lib_gather = False
        R.ctx = lib.Context(device=R.local_rank, rank=R.rank, world=R.world, comm_id=None)
        R.comm_note = R.comm_note or "another rank failed to build the library communicator"
    R.ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def allgather_handles(b):  # the channel the direct exchange's IPC handles travel over
        out_h = [None] * R.world
        dist.all_gather_object(out_h, b)
        return out_h
    R.allgather_handles = allgather_handles
    if R.use_ipc:
        chain_mod.connect_direct_exchange(R.ctx, allgather_handles)
    R.comm_stream = None
    if R.lib_gather and not args.no_overlap:
        R.comm_stream = torch.cuda.Stream(device=R.dev)
        if not R.chain:
            R.ctx.comm_set_stream(R.comm_stream.cuda_stream)
    # row shard of this rank: ceil(H / world) rows per gather slot, clipped to the image (androidrenderer_amd/shard.py)
    R.rows_per = -(-R.H // R.world)
    R.r0, R.r1 = shard.lighting_rows(R.H, R.world, R.rank)
    if R.world > 1:
        R.fr.row_begin, R.fr.row_end = R.r0, R.r1


def propagate_lpv(R):
    """--lpv-copy propagate: the workload's LPV volumes become what ONE step of the library's sah_lpv_propagate makes of the synthetic ones — the
    step that, in a frame, ends the LPV maintenance and stores the Lighting pass's gather copy beside the volumes.  Host copies follow, so that the
    CPU baseline shades the same inputs."""
    from androidrenderer_amd import _abi, frame, images
    torch, ctx, fr, d_arr = R.torch, R.ctx, R.fr, R.d_arr
    keys = ("lpv_r", "lpv_g", "lpv_b")
    b_t = [torch.zeros_like(d_arr[k]) for k in keys]
    ctx.lpv_propagate([images.volume(d_arr[k], _abi.FORMAT_R16G16B16A16_SFLOAT) for k in keys], [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in b_t], 4, 1)
    torch.cuda.synchronize()
    for k, t in zip(keys, b_t):
        d_arr[k] = t
        fr.arrays[k] = frame.from_torch(t, np.uint16).view(np.float16).reshape(fr.arrays[k].shape)


def produce_inputs(R):
    """`produced` workloads: overwrite the synthetic planes with what the library's own producer passes make of the atrium mesh (rasterised
    G-buffer and shadow cascades, LPV from RSM -> VPLs -> propagation)."""
    from androidrenderer_amd import _abi, images, mesh
    torch, ctx, fr, d_arr, dev = R.torch, R.ctx, R.fr, R.d_arr, R.dev
    geo = mesh.geometry(mesh.to_device(mesh.atrium(8).arrays(), dev), [])
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
    R.keep_produced = (geo, rsm_t, scratch)


def setup_lpv_frame(R):
    """`lpv_frame` workload: the LPV's per-frame upkeep as part of the step.  The G-buffer and the shadow cascades are rasterised once from the
    atrium mesh (they are the path's inputs); every step then clears the volumes, renders the four cascades' RSM of the same mesh, extracts and
    injects each cascade's VPLs and runs the reference's 32 propagation steps, whose last one stores the Lighting pass's gather copy
    (SAH_GENERATION_TRACKED).  Adds: lpv_upkeep(), lpv_frame (the report's dict: each part timed once, on its own)."""
    from androidrenderer_amd import _abi, images, mesh
    torch, ctx, fr, d_arr, dev = R.torch, R.ctx, R.fr, R.d_arr, R.dev
    geo = mesh.geometry(mesh.to_device(mesh.atrium(R.args.atrium_subdiv).arrays(), dev), [])
    ctx.gbuffer_render(geo, fr.view.gpu_data, images.gbuffer(d_arr))
    ctx.shadow_render(geo, fr.sun.constants, 4, images.volume(d_arr["shadowmap"], _abi.FORMAT_D16_UNORM))
    rsm_t = {"flux": torch.zeros((4, 128, 128, 4), dtype=torch.uint8, device=dev), "normals": torch.zeros((4, 128, 128, 4), dtype=torch.uint8, device=dev),
             "depth": torch.zeros((4, 128, 128), dtype=torch.int16, device=dev)}
    rsm = _abi.RsmTargets(images.volume(rsm_t["flux"], _abi.FORMAT_R8G8B8A8_SRGB), images.volume(rsm_t["normals"], _abi.FORMAT_R8G8B8A8_UNORM),
                          images.volume(rsm_t["depth"], _abi.FORMAT_D16_UNORM))
    vols = [d_arr[k] for k in ("lpv_r", "lpv_g", "lpv_b")]
    vd = [images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in vols]
    scratch = [torch.zeros_like(v) for v in vols]
    sd = [images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in scratch]
    vpls = torch.zeros((4096, 4), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    steps = 32  # light_propagation_volume.cpp: r.LPV.NumPropagationSteps

    def clear():
        ctx.lpv_clear(vd[0], vd[1], vd[2], None, 4)

    def render_rsm():
        ctx.rsm_render(geo, fr.sun.constants, fr.lpv.matrices, 4, rsm)

    def inject():
        for c in range(4):
            ctx.lpv_extract_vpls(rsm, fr.lpv.matrices, c, 0.25, vpls.data_ptr(), count.data_ptr())
            ctx.lpv_inject_vpls(vpls.data_ptr(), count.data_ptr(), 4096, fr.lpv.matrices, c, 4, vd)

    def propagate():
        ctx.lpv_propagate(vd, sd, 4, steps)  # an even number of steps: the result (and the gather copy's source) is `vols` again

    def upkeep():
        clear()
        render_rsm()
        inject()
        propagate()
    fr.lpv_generation = _abi.GENERATION_TRACKED
    parts = {}
    for name, fn in (("clear", clear), ("rsm_render", render_rsm), ("extract_inject_x4", inject), (f"propagate_x{steps}", propagate), ("upkeep", upkeep)):
        for _ in range(3):
            upkeep()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        parts[name + "_ms"] = round(e0.elapsed_time(e1) / n, 4)
    upkeep()
    torch.cuda.synchronize()
    lit_cells = [int((v.view(torch.int16).reshape(-1, 4) != 0).any(dim=1).sum()) for v in vols]
    R.lpv_upkeep = upkeep
    R.keep_lpv_frame = (geo, rsm_t, scratch, vpls, count, vd, sd, rsm)
    R.lpv_frame = dict(parts, propagation_steps=steps, triangles=int(mesh.atrium(R.args.atrium_subdiv).arrays()["indices"].size // 3) if False else None,
                       rsm="4 cascades x 128 x 128", non_zero_cells_rgb=lit_cells,
                       note="each part timed on its own (20 calls back to back) before the timed region; `upkeep` = one frame's clear + RSM + inject + propagate")
    R.lpv_frame.pop("triangles")


def setup_traced(R):
    """`traced` workload: the G-buffer rasterised from the atrium mesh, the AO plane and the sun's shadow mask traced every step against the
    structure sah_rt_build made of the same mesh, 1024 probes of the irradiance cache traced and folded into the atlases every step
    (irradiance_cache.cpp:21-23, 585-724).  Adds: traced (the report's dict), trace_planes(), trace_rows."""
    from androidrenderer_amd import _abi, images, mesh, synth
    args, torch, ctx, fr, d_arr, dev, W, H = R.args, R.torch, R.ctx, R.fr, R.d_arr, R.dev, R.W, R.H
    geo = mesh.geometry(mesh.to_device(mesh.atrium(args.atrium_subdiv).arrays(), dev), [])
    ctx.gbuffer_render(geo, fr.view.gpu_data, images.gbuffer(d_arr))
    ctx.rt_set_bounces(args.rt_bounces)
    noise_t = torch.from_numpy(synth.rng(31).integers(0, 256, (128, 128, 4), dtype=np.uint8)).to(dev)
    planes_rt = (images.plane(d_arr["depth"], _abi.FORMAT_D32_SFLOAT), images.plane(d_arr["normals"], _abi.FORMAT_R16G16B16A16_SFLOAT),
                 images.plane(noise_t, _abi.FORMAT_R8G8B8A8_UNORM), images.plane(d_arr["ao"], _abi.FORMAT_R32_SFLOAT),
                 images.plane(d_arr["shadow_mask"], _abi.FORMAT_R32_SFLOAT))
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
    R.trace_rows = [(0, 0)]  # N > 1: the rows this rank's lighting reads (set once the row plan exists); the structure and the probes are replicated

    def trace_planes():
        ctx.probe_trace(pt)
        ctx.probe_update(atlases, pt.trace_results, probe_ids.data_ptr(), n_probes)
        for r0, r1 in R.trace_rows:
            ctx.rt_set_rows(r0, r1)
            ctx.rtao(fr.view.gpu_data, planes_rt[0], planes_rt[1], planes_rt[2], 1, 8.0, planes_rt[3])
            ctx.sun_shadow_mask(fr.view.gpu_data, fr.sun.constants, planes_rt[0], planes_rt[1], planes_rt[2], planes_rt[4])
        ctx.rt_set_rows(0, 0)
    R.trace_planes = trace_planes
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
    R.keep_traced = (geo, noise_t, trace_results, light_cache, average, sky_luts, pt, atlases, planes_rt, probe_ids)
    R.traced = {"triangles": rt_stats[0], "levels": rt_stats[2], "gi_bounces": args.rt_bounces, "rt_build_ms": round(e[0].elapsed_time(e[1]), 4), "rtao_ms": round(e[2].elapsed_time(e[3]), 4),
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


def frame_maintenance(R):
    """What a frame does to the GI side tables before its Lighting pass, on the work stream: the traced workload traces and folds its probes;
    the other irradiance-cache workloads tell the context which 1024 probes "the frame's update shaders" rewrote (--probe-copy patched)."""
    if R.traced is not None:
        R.trace_planes()
    elif getattr(R, "lpv_upkeep", None) is not None:
        R.lpv_upkeep()
    elif R.probe_ids is not None:
        R.ctx.probe_notify_updated(R.irr_volume, R.probe_ids.data_ptr(), 1024)


def build_loop(R, force_stepwise=False):
    """The step of the timed loop.  Three shapes: the whole frame sharded with two frames in flight (chain workloads with a library
    exchange, or one GPU with --frames-in-flight 2), the whole frame one at a time, the Lighting pass alone (with its all-gather of
    the lit rows at N > 1).  Adds: step(i, e0, e1), drain(), my_px, pc, sc, pipelined."""
    from androidrenderer_amd import _abi, chain as chain_mod, images
    args, torch, dist, ctx, fr, d_arr, dev, W, H, world, rank = R.args, R.torch, R.dist, R.ctx, R.fr, R.d_arr, R.dev, R.W, R.H, R.world, R.rank
    R.tm_flags = tm_flags = 0 if args.strict_tonemap else _abi.TONEMAP_TOLERANCE_1CODE
    if R.gi_kind == _abi.GI_CACHE:
        R.irr_volume = images.volume(d_arr["probe_irr"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
    pipelined = R.chain and R.gather and R.lib_gather and R.comm_stream is not None and not force_stepwise
    # one GPU, no exchange: the same two-frames-in-flight loop when asked for (--frames-in-flight 2): the post chain of frame i on a second
    # stream beside the lighting of frame i + 1 (the gathers are no-ops without a communicator)
    if R.chain and not pipelined and world == 1 and not R.gather and R.traced is None and args.frames_in_flight == 2 and not force_stepwise:
        pipelined = True
        if R.comm_stream is None:
            R.comm_stream = torch.cuda.Stream(device=dev)
    R.pipelined = pipelined
    R.pc = R.sc = None
    R.work_streams = 1

    def chain_px(sc):
        return W * (sc.plan.lit_rows[1] - sc.plan.lit_rows[0] + sc.plan.lit_wrap_rows[1] - sc.plan.lit_wrap_rows[0]) if world > 1 else W * H

    def set_trace_rows(sc):
        if R.traced is not None and world > 1:
            R.trace_rows[:] = [r for r in (tuple(sc.plan.lit_rows), tuple(sc.plan.lit_wrap_rows)) if r[1] > r[0]]

    if pipelined:
        # the whole frame, sharded, two frames in flight: both exchanges run on the side stream beside compute.  The loop itself lives in
        # the library (sah_chain_submit: chain.NativePipelinedChain); --python-loop enqueues the same order pass by pass (chain.PipelinedChain)
        post_stream = None if args.one_work_stream else torch.cuda.Stream(device=dev)
        if args.python_loop:
            pc = chain_mod.PipelinedChain(ctx, fr, d_arr, rank, world, R.comm_stream, post_stream, tonemap_flags=tm_flags)
        else:  # lighting | copy + mip rows | mips 2.. + composite, each on a stream of its own (sah_hip.h "the row-sharded frame as a loop of this library")
            reduce_stream = None if (args.one_work_stream or args.two_work_streams) else torch.cuda.Stream(device=dev)
            pc = chain_mod.NativePipelinedChain(ctx, fr, d_arr, rank, world, R.comm_stream, post_stream, tonemap_flags=tm_flags, reduce_stream=reduce_stream)
        R.work_streams = 1 if args.one_work_stream else (2 if (args.python_loop or args.two_work_streams) else 3)
        if R.use_ipc:
            pc.register_direct_exchange(R.allgather_handles)
        R.pc, R.sc = pc, pc.sets[0]
        set_trace_rows(R.sc)

        def step(i, e0=None, e1=None):
            frame_maintenance(R)  # on the work stream, in front of this frame's lighting
            pc.submit((e0, e1) if e0 is not None else None)

        def drain():
            pc.flush()
            ctx.comm_wait()
        R.my_px = chain_px(R.sc)
    elif R.chain:
        # the whole frame, sharded (chain.py), one frame at a time: every exchange goes through the library (torch path only with --torch-gather)
        sc = chain_mod.ShardedChain(ctx, fr, d_arr, rank, world, tonemap_flags=tm_flags)
        if R.use_ipc:
            sc.register_direct_exchange(R.allgather_handles)
        q, per = sc.plan.mip1_rows_per_rank, sc.plan.rows_per_rank
        mip_bytes, out_bytes = sc.mip1_alloc.view(torch.uint8).view(-1), sc.out_alloc.view(-1)
        mip_slot_bytes, out_slot_bytes = q * sc.mip1_alloc.shape[1] * 8, per * W * 4
        R.sc = sc
        set_trace_rows(sc)
        gather, lib_gather = R.gather, R.lib_gather

        def step(i, e0=None, e1=None):
            frame_maintenance(R)
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
        R.my_px = chain_px(sc)
    else:
        # N > 1: two lit targets, so that the all-gather of frame i (side stream) overlaps the shading of frame i + 1; a target is reused
        # only after its gather has completed (stream-level wait).  Every gather finishes inside the timed region.
        gather, lib_gather, comm_stream, rows_per, r0, r1 = R.gather, R.lib_gather, R.comm_stream, R.rows_per, R.r0, R.r1
        nbuf = 2 if (gather and not args.no_overlap) else 1
        shard_bytes = rows_per * W * 8
        bufs = []
        for _ in range(nbuf):
            lf = torch.zeros((rows_per * world, W, 4), dtype=torch.int16, device=dev)  # equal slots for the all-gather
            desc_b, keep_b = fr.describe(d_arr, lf[:H])
            lb = lf.view(torch.uint8).view(-1)  # RCCL has no int16: the rows travel as bytes
            if R.use_ipc:
                ctx.ipc_register(lf.data_ptr(), lf.numel() * 2, R.allgather_handles(ctx.ipc_export(lf.data_ptr(), lf.numel() * 2)))
            bufs.append({"lit": lf[:H], "desc": desc_b, "keep": keep_b, "bytes": lb, "slot": lb[rank * shard_bytes:(rank + 1) * shard_bytes],
                         "plane": images.plane(lf[:H], _abi.FORMAT_R16G16B16A16_SFLOAT)})
        pending = [None] * nbuf
        R.bufs = bufs

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
        R.my_px = W * (r1 - r0) if world > 1 else W * H
    R.step, R.drain = step, drain


def time_unsharded(R):
    """N > 1 (or --force-gather): the same workload unsharded on this rank's GPU, timed before the sharded loop — the strong scaling of ONE
    workload can then be read off this line alone (the driver's N = 1 run measures the headline lighting pass, not necessarily this
    workload) — and, for chain workloads, the unsharded frame's final image to hold the sharded loop's frames against.  Adds: single_gpu, ref_image."""
    from androidrenderer_amd import chain as chain_mod
    torch, ctx, fr, d_arr, W, H = R.torch, R.ctx, R.fr, R.d_arr, R.W, R.H
    R.beat("unsharded reference")
    R.single_gpu = R.ref_image = None
    if not (R.world > 1 or R.args.force_gather):
        return
    if R.chain:
        ref = chain_mod.ShardedChain(ctx, fr, d_arr, 0, 1, tonemap_flags=R.tm_flags)
        ref_step = lambda: ref.step(gather=False)
    else:
        ref_lit = torch.zeros((H, W, 4), dtype=torch.int16, device=R.dev)
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
    R.single_gpu = {"ms_per_step": round(ref_ms, 5), "value": round(W * H / (ref_ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
                    "note": "the same workload unsharded on rank 0's GPU, 30 steps, GPU time between two events"}
    if R.chain and R.gather and R.lib_gather and R.traced is None:  # (traced: a rank traces only its own rows of the AO / shadow-mask planes)
        R.ref_image = ref.out.clone()
    del ref_step
    torch.cuda.empty_cache()


def preflight(R):
    """N > 1, two frames in flight: before anything is timed, three frames of the loop are held against the unsharded frame on every rank.
    The loop's streams and events have met more than one GPU only here: if a frame differs, every rank falls back together to the
    one-frame-at-a-time loop (exchanges in the work stream's own order), checks that one too, and the line says what ran.  Adds: preflight."""
    torch = R.torch
    R.preflight = None
    if not (R.pipelined and R.ref_image is not None and R.exchange):
        return
    R.beat("pre-flight")
    for i in range(3):
        R.step(i)
    R.drain()
    torch.cuda.synchronize()
    ok = int(bool(torch.equal(R.pc.image(1), R.ref_image)) and bool(torch.equal(R.pc.image(2), R.ref_image)))
    if os.environ.get("SAH_BENCH_FAIL_PREFLIGHT") == "1":  # test hook: take the fall-back path
        ok = 0
    ok = all_ranks_min(R, ok)
    R.preflight = {"two_frames_in_flight_ok": bool(ok), "fallback": None}
    if ok:
        return
    print(f"[bench] rank {R.rank}: the two-frames-in-flight loop's frames differ from the unsharded frame: falling back to one frame at a time", file=sys.stderr)
    if R.use_ipc:  # the registrations are found by address: give them back before the allocator can hand the addresses out again
        R.pc.unregister_direct_exchange()
    if hasattr(R.pc, "close"):
        R.pc.close()
    R.pc = None
    torch.cuda.empty_cache()
    build_loop(R, force_stepwise=True)
    for i in range(2):
        R.step(i)
    R.drain()
    torch.cuda.synchronize()
    ok2 = all_ranks_min(R, int(bool(torch.equal(R.sc.out, R.ref_image))))
    R.preflight["fallback"] = "one frame at a time"
    R.preflight["fallback_ok"] = bool(ok2)


def clock_ramp(R, step, drain, synchronize, clock=time.perf_counter):
    """args.ramp_ms of untimed steps before the warm-up (clocks up, caches and tables made), eight at a time.  WHEN TO STOP IS ONE DECISION
    FOR THE WHOLE JOB: every step holds the exchanges of the sharded loop, so every rank must make the same number of them — a rank that
    read its own clock alone would stop one round before or after a peer whose ramp began a millisecond apart, and the extra gathers
    would wait for partners that never come (RCCL: for ever; the direct exchange: its two seconds, found by a four-rank rehearsal in
    round 5).  The ranks stop together as soon as any of them has had its time.  Returns the number of steps made."""
    if R.args.ramp_ms <= 0:
        return 0
    t_ramp = clock()
    k = 0
    while True:
        for _ in range(8):
            step(k)
            k += 1
        drain()
        synchronize()
        if not all_ranks_min(R, int((clock() - t_ramp) * 1e3 < R.args.ramp_ms)):
            return k


def run_timed(R):
    """Clock ramp, W warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides.  Adds: elapsed (max over ranks),
    kernel_ms_mean, kernel_ms_min, kernel_scope."""
    args, torch, dist, step, drain = R.args, R.torch, R.dist, R.step, R.drain
    clock_ramp(R, step, drain, torch.cuda.synchronize)
    R.beat("warm-up")
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    R.beat("timed region")
    if R.torch_pg:
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
    # sah_lighting call (2 x ~10 us on a ~0.6 ms step); created and recorded before t0 as well
    ev = []
    if R.chain:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a_, b_ in ev:
            a_.record()
            b_.record()
        torch.cuda.synchronize()
    rebuilds0 = R.ctx.copy_rebuilds()
    t0 = time.perf_counter()
    g0.record()
    for i in range(args.steps):
        if R.chain:
            step(i, ev[i][0], ev[i][1])
        else:
            step(i)
    drain()
    g1.record()
    torch.cuda.synchronize()
    if R.torch_pg:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    R.rebuilds_in_timed_region = {"lpv": R.ctx.copy_rebuilds()[0] - rebuilds0[0], "irradiance_atlas": R.ctx.copy_rebuilds()[1] - rebuilds0[1]}
    # how many pixels of the last pass the fast kernel handed to its fix-up launch (0 for the frames a renderer produces: the launch then
    # returns at once; the tiled kernel re-evaluates inline and has no such list)
    R.fixup_pixels_last_step = R.ctx.deferred_pixels() if not R.chain and not R.n_lights and R.gi_kind in (0, 1) else None
    loop_ms = g0.elapsed_time(g1) / args.steps
    if R.chain:
        per_step = sorted(a_.elapsed_time(b_) for a_, b_ in ev)
        R.kernel_ms_mean, R.kernel_ms_min = sum(per_step) / len(per_step), per_step[0]
        R.kernel_scope = "HIP events around each sah_lighting call of the timed region, on its stream (main kernel + fix-up)" + (
            "; two frames in flight: the previous frame's post chain runs beside it on a second stream, so this is not the kernel's time alone"
            if R.pipelined and not args.one_work_stream else "")
    else:
        R.kernel_ms_mean, R.kernel_ms_min = loop_ms, None
        R.kernel_scope = ("one pair of HIP events on the launch stream around all K steps of the timed region, / K: everything a sah_lighting call "
                          "enqueues (gather-copy rebuild if any, main kernel with its sky workgroups, fix-up) and the gaps between launches")
    if R.torch_pg:
        t = torch.tensor([elapsed], dtype=torch.float64, device=R.red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    R.elapsed = elapsed


def verify(R):
    """After the timed region: every rank holds the last frames its loop assembled (both buffer sets of the pipelined loop) against the
    unsharded frame it rendered itself before the loop — the gathers moved the right rows to the right places on every rank, or not.
    A loop that moved wrong rows has no throughput: the line then carries value null and an error, and every rank exits non-zero (the
    verdicts are all-reduced, so the ranks agree).  Adds: sharded_equals_unsharded, failure."""
    torch, args = R.torch, R.args
    R.beat("verification")
    R.sharded_equals_unsharded = None
    if R.ref_image is not None:
        outs = [R.pc.image(args.steps - 1), R.pc.image(args.steps - 2)] if R.pipelined and args.steps > 1 else [R.pc.image(args.steps - 1) if R.pipelined else R.sc.out]
        same = int(all(bool(torch.equal(o, R.ref_image)) for o in outs))
        if os.environ.get("SAH_BENCH_FAIL_VERIFY") == "1":  # test hook: the line must then carry no throughput and the exit code say so
            same = 0
        same = all_ranks_min(R, same)
        R.sharded_equals_unsharded = bool(same)
        if not same:
            print(f"[bench] rank {R.rank}: the sharded loop's final image differs from the unsharded frame", file=sys.stderr)
    R.failure = None
    if R.sharded_equals_unsharded is False:
        R.failure = "the sharded loop's last frames differ from the unsharded frame on at least one rank"
    elif R.preflight is not None and R.preflight.get("fallback") and not R.preflight.get("fallback_ok"):
        R.failure = "pre-flight: the two-frames-in-flight loop AND the one-frame-at-a-time fall-back differ from the unsharded frame"


def time_longer_run(R):
    """The driver times 20 steps of a 0.17 ms pass: 3.4 ms, over before a power sample sees it.  On one GPU the line therefore carries, beside the
    driver's K steps, the same loop over at least 200 steps (GPU time between two events, after the timed region): what the K-step number is good
    for can be read off the two."""
    torch = R.torch
    if not (R.world == 1 and not R.exchange and R.args.steps < 200):
        return None
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R.drain()
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        R.step(i)
    R.drain()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return {"steps": n, "ms_per_step": round(ms, 5), "value": round(R.W * R.H / (ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
            "note": "the same loop over 200 steps, GPU time between two events, after the timed region"}


def time_exchange_only(R):
    """N > 1, chain workloads through the library's exchange: the frame's two gathers (bloom mip 1, the final RGBA8 rows) by themselves, with no
    compute beside them — what the frame rate cannot exceed whatever the kernels do (the sharded loop overlaps them with compute: frame time =
    max(compute, exchange)).  EVERY rank runs it (the gathers are collective), the same fixed number of times; outside the timed region.
    Returns the report's dict or None."""
    if not (R.world > 1 and R.pipelined and R.gather and R.lib_gather and R.pc is not None and not R.failure):
        return None
    torch, ctx = R.torch, R.ctx
    sc = R.pc.sets[0]
    n = 20
    R.beat("exchange-only timing")
    try:
        R.drain()
        torch.cuda.synchronize()
        R.dist.barrier()
        for _ in range(3):
            sc.exchange_mip()
            sc.exchange_final()
        ctx.comm_wait()
        torch.cuda.synchronize()
        R.dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            sc.exchange_mip()
            sc.exchange_final()
        ctx.comm_wait()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        R.dist.barrier()
    except Exception as e:  # (a diagnostic must not cost the run its line)
        return {"error": str(e)[:200]}
    p = sc.plan
    mip_bytes = p.mip1_rows_per_rank * sc.mip1_alloc.shape[1] * 8 * (R.world - 1)
    out_bytes = p.rows_per_rank * R.W * 4 * (R.world - 1)
    return {"ms_per_frame": round(ms, 5), "bytes_arriving_per_rank": int(mip_bytes + out_bytes),
            "arriving_GB_per_s_per_rank": round((mip_bytes + out_bytes) / (ms * 1e-3) / 1e9, 1),
            "note": "the frame's two gathers alone, 20 frames back to back behind 3 untimed ones, host clock of rank 0 around a synchronise; the sharded loop "
                    "runs them beside compute, so ms_per_step >= this"}


def time_with_rebuild(R):
    """One GPU, LPV lighting workloads under --lpv-copy propagate: the same pass with lpv_generation 0 — k_lpv_pack inside every step, what the
    frame pays when its volumes are rewritten by a pass that is not the library's.  Outside the timed region; returns the report's dict or None."""
    from androidrenderer_amd import _abi
    torch, ctx, fr, d_arr = R.torch, R.ctx, R.fr, R.d_arr
    if not (R.world == 1 and R.gi_kind == _abi.GI_LPV and not R.chain and not R.exchange and R.args.lpv_copy == "propagate"):
        return None
    saved = fr.lpv_generation
    fr.lpv_generation = 0
    lit = torch.zeros((R.H, R.W, 4), dtype=torch.int16, device=R.dev)
    desc, keep = fr.describe(d_arr, lit)
    fr.lpv_generation = saved
    # (200 calls at least, behind 20 untimed ones: round 5 timed 20 calls behind 10 on the freshly allocated target and read 0.203 ms where a
    # 200-step run of the same pass gave 0.172 — a short loop measures its own start, not the pass)
    n = max(200, R.args.steps)
    for _ in range(20):
        ctx.lighting(desc)
    torch.cuda.synchronize()
    before = ctx.copy_rebuilds()[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ctx.lighting(desc)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return {"ms_per_step": round(ms, 5), "value": round(R.W * R.H / (ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "steps": n,
            "gather_copy_rebuilds_during_it": ctx.copy_rebuilds()[0] - before,
            "note": "the same pass with lpv_generation 0 (k_lpv_pack + the pass in every step), GPU time between two events, after the timed region"}


def metric_name(workload, chain, W, H):
    """`metric` names what the line's `value` counts.  The headline workload carries BASELINE.json's metric verbatim; every other workload says
    which pass or frame its pixels went through, so that two lines of different workloads are not read as one curve (at N > 1 the default
    workload is the configs[3] frame, not the N = 1 headline pass)."""
    if workload == "4k_deferred_gi":
        return "lit Mpixels/sec (deferred+GI pass) at 4K"
    res = {(3840, 2160): "4K", (7680, 4320): "8K", (1920, 1080): "1080p", (1280, 720): "720p"}.get((W, H), f"{W}x{H}")
    if chain:
        return f"final-image Mpixels/sec ({workload}: lighting + copy scene + bloom + tonemap frame) at {res}"
    return f"lit Mpixels/sec ({workload} lighting pass) at {res}"


def report(R):
    """Rank 0 prints the ONE JSON line."""
    from androidrenderer_amd import _abi
    args, wl, W, H, world = R.args, R.wl, R.W, R.H, R.world
    px = W * H
    value = px * args.steps / R.elapsed / 1e6
    achieved = R.bytes_per_pixel * R.my_px / (R.kernel_ms_mean * 1e-3) / 1e9
    sun_txt = {"csm": "sun CSM 4x4096^2 D16 PCF", "rt": "sun RT (shadow-mask plane, half-precision BRDF)", "off": "sun off"}[wl["sun"]]
    if wl.get("mask") == "ones":
        sun_txt = "sun RT, every shadow ray unoccluded (mask = 1): a single directional light, half-precision BRDF"
    gi_txt = {"none": "no GI", "lpv": "LPV GI gather + AO", "cache": "irradiance-cache probe gather", "rtgi": "RTGI reconstruction"}[wl["gi"]]
    parts = [sun_txt, gi_txt, "emissive", "sky"]
    if R.n_lights:
        parts.insert(1, f"{R.n_lights} point lights (r={wl['radius']} m) with LDS tile culling")
    what = "fused deferred lighting (" + " + ".join(parts) + ")"
    if R.chain:
        what += " + copy scene + bloom pyramid + tonemap composite"
    if world == 1:
        par = "single GPU" + (" + one-rank RCCL communicator (rehearsal of the exchange)" if R.exchange else "")
    elif R.chain:
        par = f"row-shard x{world}: lighting rows + halo, all-gather of bloom mip 1, all-gather of the RGBA8 rows (reversed rank order)"
        if R.rehearsal:
            par += " — REHEARSAL: all ranks on one GPU, not a scaling measurement"
    else:
        par = f"row-shard x{world} + RCCL all-gather of the lit rows"
    gather, lib_gather = R.gather, R.lib_gather
    out = {
        "metric": metric_name(args.workload, R.chain, W, H),
        "value": None if R.failure else round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(R.elapsed / args.steps * 1e3, 5),
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
            "tonemap": None if not R.chain else ("strict" if args.strict_tonemap else "SAH_TONEMAP_TOLERANCE_1CODE (within one code of the strict composite)"),
            "lpv_gather_copy": None if R.gi_kind != _abi.GI_LPV else (
                "written by the last step of sah_lpv_propagate beside the volumes (SAH_GENERATION_TRACKED; the propagation ran once before the loop — the "
                "benchmark's volumes do not change — where a frame runs it every frame): no Lighting pass rebuilds it" if args.lpv_copy == "propagate" else
                "rebuilt inside every step (lpv_generation 0: k_lpv_pack + the pass — a frame whose volumes somebody else's propagation rewrites)"
                if args.lpv_copy == "rebuild" else "kept across steps (lpv_generation 1: the LPV volumes of this benchmark never change)"),
            "gather_copy_rebuilds_in_timed_region": R.rebuilds_in_timed_region,
            "fixup_pixels_last_step": R.fixup_pixels_last_step,
            "probe_gather_copy": None if R.gi_kind != _abi.GI_CACHE else (
                "tracked by the context (SAH_GENERATION_TRACKED): sah_probe_update re-widens the blocks of the 1024 probes it folds every step" if (wl.get("traced") and args.probe_copy == "patched") else
                "tracked by the context (SAH_GENERATION_TRACKED): every step re-widens the blocks of 1024 probes (sah_probe_notify_updated: r.GI.Cache.UpdatesPerFrame)" if args.probe_copy == "patched" else
                "rebuilt every step (probe_generation 0)" if args.probe_copy == "rebuild" else "kept across steps (probe_generation 1: the atlases of this benchmark never change)"),
            "gbuffer": wl["gbuffer"],
            "parallelism": par,
            "gather": bool(gather),
            "gather_through": (("sah_allgather_rows (library, direct exchange over peer-mapped memory)" if R.use_ipc else "sah_allgather_rows (library, RCCL)") if lib_gather else "torch.distributed" + (f" (fallback: {R.comm_note})" if R.comm_note else "")) if gather else None,
            "gather_overlapped_with_next_frame": bool(gather and not args.no_overlap and (R.pipelined or not R.chain)),
            "post_chain_beside_next_frames_lighting": bool(R.pipelined and not args.one_work_stream),
            "frames_in_flight": 2 if R.pipelined else 1,
            "frame_loop": None if not R.pipelined else ("Python, pass by pass (chain.PipelinedChain)" if args.python_loop else "the library's (sah_chain_submit)"),
            "work_streams": None if not R.pipelined else R.work_streams,
            "same_workload_on_one_gpu": R.single_gpu,
            # the N = 1 run of this file measures the headline lighting pass, a different workload from the sharded chain: the strong
            # scaling of THIS workload is its throughput here over its throughput unsharded on one of these GPUs
            # (null in a rehearsal: there every rank — and the one-GPU reference, timed while the others wait — shares ONE GPU)
            "speedup_vs_same_workload_on_one_gpu": None if (not R.single_gpu or R.rehearsal or R.failure) else round(value / R.single_gpu["value"], 3),
            "sharded_equals_unsharded": R.sharded_equals_unsharded,
            "preflight": R.preflight,
            "traced": R.traced,
            "lpv_frame": R.lpv_frame,
        },
        "roofline": roofline(args.workload, world, achieved, R.kernel_ms_mean, R.kernel_ms_min, R.kernel_scope, R.bytes_per_pixel * R.my_px, R.my_px,
                             "sah::k_lighting_tiled" if (R.n_lights or R.gi_kind in (_abi.GI_CACHE, _abi.GI_RTGI)) else "sah::k_lighting_fast"),
    }
    if getattr(R, "exchange_only", None) is not None:
        out["config"]["exchange_only"] = R.exchange_only
    if world > 1:
        # what a scaling curve of THIS line has to be read against: the same workload unsharded on one of these GPUs, timed by this run (the
        # N = 1 run of this file measures the headline lighting pass — another workload; value(N) / value(1) across the two is not a speed-up)
        out["same_workload_on_one_gpu"] = R.single_gpu
        out["speedup_vs_same_workload_on_one_gpu"] = out["config"]["speedup_vs_same_workload_on_one_gpu"]
    if R.with_rebuild is not None:
        out["config"]["lpv_gather_copy_rebuilt_every_step"] = R.with_rebuild
    if R.longer_run is not None:
        out["config"]["same_loop_over_200_steps"] = R.longer_run
    if R.failure:
        out["error"] = R.failure
        out["measured_but_invalid_Mpixels_per_s"] = round(value, 1)
    if R.n_lights and not args.no_light_stats:
        out["config"].update(light_stats(R.torch, R.fr, R.d_arr, R.lights, R.dev))
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(R.fr, args.cpu_seconds)
    if R.saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(R.saved_stdout, 1)
    print(json.dumps(out), flush=True)
    if R.saved_stdout is not None:
        os.dup2(2, 1)  # teardown chatter goes to stderr as well


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    R = setup_distributed(args)
    make_inputs(R)
    make_context(R)
    R.lpv_upkeep = R.lpv_frame = None
    if R.wl.get("produced"):
        produce_inputs(R)  # (its propagation's last step has stored the gather copy as well)
    elif R.wl.get("lpv_frame"):
        setup_lpv_frame(R)
    elif R.gi_kind == 1 and args.lpv_copy == "propagate":  # _abi.GI_LPV
        propagate_lpv(R)
    R.traced = None
    if R.wl.get("traced"):
        setup_traced(R)
    build_loop(R)
    time_unsharded(R)
    preflight(R)
    run_timed(R)
    verify(R)
    R.exchange_only = time_exchange_only(R)
    R.longer_run = time_longer_run(R) if R.rank == 0 and not R.failure else None
    R.with_rebuild = time_with_rebuild(R) if R.rank == 0 and not R.failure else None
    if R.rank == 0:
        report(R)
    if R.pc is not None and hasattr(R.pc, "close"):
        R.pc.close()
    R.ctx.close()
    if R.torch_pg:
        R.dist.barrier()
        R.dist.destroy_process_group()
    if R.failure:
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
