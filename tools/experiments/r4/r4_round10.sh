#!/bin/bash
# cache-GI trims (cascade pick behind a wave vote, debug colours by selects): parity, then timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lighting_gpu.py tests/test_lighting_ext_gpu.py tests/test_golden.py tests/test_fullsize_gpu.py tests/test_shard_chain.py -x -q -m gpu > gpurun_out/r4_round10_tests.log 2>&1 || { tail -30 gpurun_out/r4_round10_tests.log; exit 1; }
tail -2 gpurun_out/r4_round10_tests.log
for wl in 4k_probe_gi_chain 4k_deferred_gi; do
  timeout -k 10 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>> gpurun_out/r4_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s ms/step %.4f  lighting %.4f' % ('$wl', d['ms_per_step'], d['roofline']['kernel_ms_mean']))"
done
