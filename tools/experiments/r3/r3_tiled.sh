#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lighting_gpu.py tests/test_lighting_ext_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/r3_tiled_tests.log 2>&1 || { tail -30 gpurun_out/r3_tiled_tests.log; exit 1; }
tail -2 gpurun_out/r3_tiled_tests.log
for w in 4k_probe_gi_chain 4k_256_lights 1080p_64_lights 8k_1024_lights_gi; do
timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline 2> gpurun_out/r3_tiled_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min'])" || { tail -20 gpurun_out/r3_tiled_bench.err; exit 1; }
done
