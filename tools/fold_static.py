"""Folds gpurun_out/<R>_static/<R>_pmc_static_<workload>.txt (tools/static_all.sh; R from the environment, default r5) into profiles/: the summaries of the
library's own kernels (torch's input-synthesis kernels dropped) as profiles/<R>_pmc_static_<workload>.txt, and one record per workload in
profiles/roofline_static.json through tools/pmc_to_static.py.   usage: fold_static.py [<workload> ...]   (default: every summary present)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.environ.get("R", "r5")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src = os.path.join(ROOT, "gpurun_out", f"{R}_static")
names = sys.argv[1:] or sorted(re.sub(rf"^{R}_pmc_static_|\.txt$", "", os.path.basename(p)) for p in glob.glob(src + f"/{R}_pmc_static_*.txt"))
for wl in names:
    spec = bench.WORKLOADS[wl]
    keep, on = [], True
    for line in open(os.path.join(src, f"{R}_pmc_static_{wl}.txt")):
        if line.startswith("pmc group"):
            keep.append(line)
        elif not line.startswith(" "):
            on = "sah::" in line
            if on:
                keep.append(line)
        elif on:
            keep.append(line)
    out = os.path.join(ROOT, "profiles", f"{R}_pmc_static_{wl}.txt")
    head = (f"rocprofv3 --pmc <group> -- python3 bench.py --workload {wl} --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0"
            + (" --synth-device cpu --no-light-stats" if wl.startswith("8k_") else "") + "   (one pass per group; per-launch means; library kernels only)\n")
    open(out, "w").write(head + "".join(keep))
    tiled = bool(spec.get("lights")) or spec.get("gi") == "cache"
    w, h = spec["res"]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_to_static.py"), wl, out, "k_lighting_tiled" if tiled else "k_lighting_fast", str(w), str(h)],
                   check=True, stdout=subprocess.DEVNULL)
    print("folded", wl)
