"""Copies what tools/refresh_profiles.sh left under gpurun_out/refresh/ into profiles/ (the tracked, judged copies), named per round.
usage: python tools/copy_profiles.py [round-prefix, default r2]"""
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r6"
SRC, DST = "gpurun_out/refresh", "profiles"
shutil.copy(f"{SRC}/bench.json", f"{DST}/{R}_bench_4k_deferred_gi.json")
shutil.copy(f"{SRC}/ktrace/kt_kernel_stats.csv", f"{DST}/{R}_kernel_stats_4k_deferred_gi.csv")
shutil.copy(f"{SRC}/pmc.txt", f"{DST}/{R}_pmc_4k_deferred_gi.txt")
if os.path.exists(f"{SRC}/ktrace_chain/kt_kernel_stats.csv"):
    shutil.copy(f"{SRC}/ktrace_chain/kt_kernel_stats.csv", f"{DST}/{R}_kernel_stats_4k_probe_gi_chain.csv")
shutil.copy(f"{SRC}/passes.txt", f"{DST}/{R}_passes_4k.txt")
if os.path.exists(f"{SRC}/ktrace_lpv_frame/kt_kernel_stats.csv"):
    shutil.copy(f"{SRC}/ktrace_lpv_frame/kt_kernel_stats.csv", f"{DST}/{R}_kernel_stats_4k_lpv_gi_frame.csv")
if os.path.exists(f"{SRC}/roofline_static.json"):
    shutil.copy(f"{SRC}/roofline_static.json", f"{DST}/roofline_static.json")
for src, dst in (("pmc_rt_cache_tiled.txt", "pmc_rt_cache_tiled.txt"), ("pmc_tonemap.txt", "pmc_tonemap.txt"), ("cpu_baselines.txt", "cpu_baselines.txt"),
                 ("rehearse_n2.json", "rehearse_n2_one_gpu.json"), ("strict_chain.json", "bench_4k_probe_gi_chain_strict_tonemap.json"),
                 ("repack.json", "bench_4k_deferred_gi_lpv_copy_rebuilt_every_step.json"), ("kept.json", "bench_4k_deferred_gi_lpv_copy_kept.json"), ("bench200.json", "bench_4k_deferred_gi_200_steps.json"), ("chain_fif2.json", "bench_4k_probe_gi_chain_two_frames_in_flight.json"),
                 ("lpv_chain_fif2.json", "bench_4k_lpv_gi_chain_two_frames_in_flight.json")):
    if os.path.exists(f"{SRC}/{src}"):
        shutil.copy(f"{SRC}/{src}", f"{DST}/{R}_{dst}")
# other workloads: replace the lines of the workloads that were re-run, keep the rest (the 8K ones come from their own run)
path = f"{DST}/{R}_bench_other_workloads.jsonl"
old = {}
if os.path.exists(path):
    for line in open(path):
        if line.strip().startswith("{"):
            d = json.loads(line)
            old[d["config"]["workload"].split(":")[0]] = line.strip()
for f in sorted(os.listdir(SRC)):
    if f.startswith("bench_") and f.endswith(".json"):
        line = open(f"{SRC}/{f}").read().strip()
        if line.startswith("{"):
            old[json.loads(line)["config"]["workload"].split(":")[0]] = line
with open(path, "w") as out:
    for k in old:
        out.write(old[k] + "\n")
print("copied; workloads:", list(old))
# second half (tools/refresh_extras.sh)
X = "gpurun_out/refresh_extras"
for src, dst in (("pmc_rt.txt", "pmc_rt_kernels.txt"), ("ktrace_traced/kt_kernel_stats.csv", "kernel_stats_4k_probe_gi_chain_traced.csv"),
                 ("stress_rt.txt", "stress_rt.txt"), ("stress_parity.txt", "stress_parity.txt"), ("stress_post.txt", "stress_post.txt"),
                 ("stress_raster.txt", "stress_raster.txt"), ("segv_fixed.txt", "segv_fixed.txt")):
    if os.path.exists(f"{X}/{src}"):
        shutil.copy(f"{X}/{src}", f"{DST}/{R}_{dst}")
