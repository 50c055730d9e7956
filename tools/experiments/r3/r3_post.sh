#!/bin/bash
# post chain: parity tests, then the passes' times
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_post_gpu.py tests/test_golden.py tests/test_shard_chain.py -x -q -m gpu > gpurun_out/r3_post_tests.log 2>&1 || { tail -30 gpurun_out/r3_post_tests.log; exit 1; }
tail -2 gpurun_out/r3_post_tests.log
timeout -k 10 300 python tools/bench_passes.py --only "copy,bloom,tonemap,off/none" --iters 50 2>/dev/null | grep -i "copy\|bloom\|tonemap\|off/none"
