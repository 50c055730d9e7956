#!/bin/bash
# smoke(), the whole GPU suite and the driver's bench command (what the round-end tiers run)
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1 || { tail -20 gpurun_out/r3_smoke.log; exit 1; }
tail -3 gpurun_out/r3_smoke.log
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r3_gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r3_bench.err | cut -c1-300
