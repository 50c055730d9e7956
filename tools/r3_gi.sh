#!/bin/bash
# round 3: full GPU suite, the ray-tracing stress sweep with the GI rays, and the traced chain workload
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3_gpu_tests.log
timeout -k 10 600 python tools/stress_rt.py --cases 40 > gpurun_out/r3_stress_rt.log 2>&1 || { tail -30 gpurun_out/r3_stress_rt.log; exit 1; }
tail -2 gpurun_out/r3_stress_rt.log
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain_traced --steps 20 --warmup 5 > gpurun_out/r3_bench_traced.json 2> gpurun_out/r3_bench_traced.err || { tail -30 gpurun_out/r3_bench_traced.err; exit 1; }
cat gpurun_out/r3_bench_traced.json
