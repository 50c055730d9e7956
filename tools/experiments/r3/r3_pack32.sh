#!/bin/bash
# experiment: fp32-expanded LPV gather copy against the fp16 one
set -o pipefail
mkdir -p gpurun_out
for v in base pack32 base pack32; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/androidrenderer_amd/_exp/libsah_pack32.so; fi
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>gpurun_out/r3_pack32.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant $v', d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min'])" || { tail -20 gpurun_out/r3_pack32.err; exit 1; }
done
export SAH_HIP_LIBRARY=$PWD/androidrenderer_amd/_exp/libsah_pack32.so
timeout -k 10 600 python -m pytest tests/test_lighting_gpu.py -x -q -m gpu 2>&1 | tail -3
