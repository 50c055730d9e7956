"""Re-runs tools/isa_stages.py for the two lighting kernels and replaces the static tables inside profiles/r4_fast_ablation.txt and
profiles/r4_cache_ablation.txt (the block from the '# k_lighting_…: static instructions per stage' line to its 'total' line)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
JOBS = (("profiles/r4_fast_ablation.txt", "androidrenderer_amd/csrc/lighting.hip", "k_lighting_fast<1, 1, 4, false>", "tools/experiments/r4/stages_fast_csm_lpv.json", ["--per", "4"]),
        ("profiles/r4_cache_ablation.txt", "androidrenderer_amd/csrc/lighting_tiled.hip", "k_lighting_tiled<2, 2, false>", "tools/experiments/r4/stages_tiled_rt_cache.json", []))
for path, src, kernel, stages, extra in JOBS:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stages.py"), os.path.join(ROOT, src), kernel, os.path.join(ROOT, stages)] + extra,
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    table = "".join(l + "\n" for l in out.splitlines() if not l.startswith("# unmatched") and not l.startswith("#   "))
    text = open(os.path.join(ROOT, path)).read()
    m = re.search(r"^# k_lighting_[^\n]*static instructions per stage.*?^total[^\n]*\n", text, re.S | re.M)
    assert m, path
    open(os.path.join(ROOT, path), "w").write(text[:m.start()] + table + text[m.end():])
    print(path, "table refreshed:", table.splitlines()[-1])
