#!/bin/bash
# emissive blends behind a wave vote (both lighting kernels), power-of-two probe spacing as a multiply, irradiance / weight through one refined
# reciprocal: parity, then timings, then the dynamic instruction count of the cache kernel
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lighting_gpu.py tests/test_lighting_ext_gpu.py tests/test_golden.py tests/test_fullsize_gpu.py tests/test_shard_chain.py tests/test_host_facade_gpu.py -x -q -m gpu > gpurun_out/r4_round11_tests.log 2>&1 || { tail -30 gpurun_out/r4_round11_tests.log; exit 1; }
tail -2 gpurun_out/r4_round11_tests.log
for wl in 4k_probe_gi_chain 4k_deferred_gi 4k_deferred_only 4k_256_lights 4k_deferred_gi_random; do
  timeout -k 10 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>> gpurun_out/r4_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s ms/step %.4f  lighting %.4f' % ('$wl', d['ms_per_step'], d['roofline']['kernel_ms_mean']))"
done
timeout -k 10 300 python tools/bench_passes.py --only "lighting" --iters 200 2>/dev/null | grep lighting
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d gpurun_out/r4_r11_pmc -o pmc --output-format csv -- python3 bench.py --workload 4k_probe_gi_chain --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 > gpurun_out/r4_r11_pmc.log 2>&1 && python3 tools/pmc_summary.py gpurun_out/r4_r11_pmc k_lighting_tiled
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d gpurun_out/r4_r11_pmc2 -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 > gpurun_out/r4_r11_pmc2.log 2>&1 && python3 tools/pmc_summary.py gpurun_out/r4_r11_pmc2 k_lighting_fast
rm -rf gpurun_out/r4_r11_pmc gpurun_out/r4_r11_pmc2
