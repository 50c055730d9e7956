#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lighting_gpu.py tests/test_lighting_ext_gpu.py tests/test_golden.py tests/test_fullsize_gpu.py tests/test_shard_chain.py -x -q -m gpu > gpurun_out/r4_round8_tests.log 2>&1 || { tail -40 gpurun_out/r4_round8_tests.log; exit 1; }
tail -3 gpurun_out/r4_round8_tests.log
for w in 4k_deferred_gi 4k_deferred_gi_random 4k_deferred_only 8k_deferred_gi 4k_lpv_gi_chain; do
timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s ms/step %.4f  lighting %.4f' % ('$w', d['ms_per_step'], d['roofline']['kernel_ms_mean']))"
done
timeout -k 10 300 python tools/bench_passes.py --only "lighting" --iters 50 2>/dev/null | grep -i "lighting" | tee gpurun_out/r4_round8_passes.txt
