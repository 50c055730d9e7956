#!/bin/bash
# round 4, post chain: parity tests of everything the advisor fixes and the new post kernels touch, then the passes' times
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_post_gpu.py tests/test_golden.py tests/test_shard_chain.py tests/test_rt.py tests/test_comm_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r4_post_tests.log 2>&1 || { tail -40 gpurun_out/r4_post_tests.log; exit 1; }
tail -3 gpurun_out/r4_post_tests.log
timeout -k 10 300 python tools/bench_passes.py --only "copy,bloom,tonemap,off/none" --iters 50 2>/dev/null | grep -i "copy\|bloom\|tonemap\|off/none" | tee gpurun_out/r4_post_passes.txt
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_chain.err | tee gpurun_out/r4_chain.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4k_probe_gi_chain ms/step', d['ms_per_step'], 'lighting', d['roofline']['kernel_ms_mean'])"
