#!/bin/bash
# ray-tracing parity tests + the traced workload's per-kernel times (one gpurun call)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_rt.py tests/test_golden.py -x -q -m gpu > gpurun_out/r3_rt_tests.log 2>&1 || { tail -30 gpurun_out/r3_rt_tests.log; exit 1; }
tail -2 gpurun_out/r3_rt_tests.log
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain_traced --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r3_rt_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['config']['traced']))" || { tail -20 gpurun_out/r3_rt_bench.err; exit 1; }
