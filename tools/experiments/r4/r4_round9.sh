#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lighting_gpu.py tests/test_lighting_ext_gpu.py tests/test_golden.py tests/test_fullsize_gpu.py tests/test_shard_chain.py -x -q -m gpu > gpurun_out/r4_round9_tests.log 2>&1 || { tail -40 gpurun_out/r4_round9_tests.log; exit 1; }
tail -3 gpurun_out/r4_round9_tests.log
for w in 4k_deferred_gi 4k_deferred_only 4k_probe_gi_chain 4k_256_lights 1080p_64_lights; do
timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s ms/step %.4f  lighting %.4f' % ('$w', d['ms_per_step'], d['roofline']['kernel_ms_mean']))"
done
