#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lighting_gpu.py tests/test_golden.py tests/test_comm_gpu.py tests/test_lighting_ext_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r3_cache_tests.log 2>&1 || { tail -30 gpurun_out/r3_cache_tests.log; exit 1; }
tail -3 gpurun_out/r3_cache_tests.log
for extra in "" "--repack-lpv"; do
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 10 $extra 2>gpurun_out/r3_cache_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$extra', d['ms_per_step'], r['kernel'], r['kernel_ms_mean'], r['kernel_ms_min'], d['config'].get('probe_gather_copy'))" || { tail -20 gpurun_out/r3_cache_bench.err; exit 1; }
done
timeout -k 10 600 python tools/stress_parity.py --seeds 12 > gpurun_out/r3_stress_parity.log 2>&1 || { tail -30 gpurun_out/r3_stress_parity.log; exit 1; }
tail -3 gpurun_out/r3_stress_parity.log
