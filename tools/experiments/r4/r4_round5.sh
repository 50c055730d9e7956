#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_post_gpu.py tests/test_golden.py tests/test_shard_chain.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r4_round5_tests.log 2>&1 || { tail -40 gpurun_out/r4_round5_tests.log; exit 1; }
tail -3 gpurun_out/r4_round5_tests.log
timeout -k 10 300 python tools/bench_passes.py --only "tonemap composite, tol" --iters 100 2>/dev/null | grep -i "tonemap"
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_chain.err | tee gpurun_out/r4_chain.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4k_probe_gi_chain ms/step', d['ms_per_step'], 'lighting', d['roofline']['kernel_ms_mean'])"
for v in base lights_pow5_f32 lights_inv_r lights_both; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  timeout -k 10 200 python bench.py --workload 4k_256_lights --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s 4k_256_lights ms/step %.4f' % ('$v', d['ms_per_step']))"
done
unset SAH_HIP_LIBRARY
# the round-3 abort (gpurun_out/pmc_lights/8k_1024_lights_gi.log): which ingredient makes it? one run each, no loop
run() { echo "== $*"; timeout -k 10 150 "$@" > gpurun_out/r4_segv_$N.log 2>&1; echo "rc=$? $(grep -c SIGSEGV gpurun_out/r4_segv_$N.log) SIGSEGV lines"; N=$((N+1)); }
N=0
run rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU -d gpurun_out/r4_segv0 -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 --workload 8k_1024_lights_gi
run rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/r4_segv1 -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 --workload 8k_1024_lights_gi
run rocprofv3 --pmc SQ_THREAD_CYCLES_VALU -d gpurun_out/r4_segv2 -o pmc --output-format csv -- python3 -c "import torch; a=torch.ones(7680*4320, device='cuda'); b=a*2.0; torch.cuda.synchronize(); print('ok', float(b[5]))"
run rocprofv3 --pmc SQ_THREAD_CYCLES_VALU -d gpurun_out/r4_segv3 -o pmc --output-format csv -- python3 -c "import torch; a=torch.ones(3840*2160, device='cuda'); b=a*2.0; torch.cuda.synchronize(); print('ok', float(b[5]))"
