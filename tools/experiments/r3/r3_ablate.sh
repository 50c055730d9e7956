#!/bin/bash
# experiment: kernel time of the tiled cache kernel with parts of the probe loop compiled out (results are garbage; timing only)
set -o pipefail
mkdir -p gpurun_out
for v in base 1 2 3 4; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/androidrenderer_amd/_exp/libsah_skip$v.so; fi
  timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 10 2>gpurun_out/r3_ablate.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant $v', r['kernel_ms_mean'], r['kernel_ms_min'])" || { tail -20 gpurun_out/r3_ablate.err; exit 1; }
done
