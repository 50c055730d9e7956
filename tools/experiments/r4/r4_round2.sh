#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/microbench/half_math_check | tee gpurun_out/r4_half_math_check.txt
timeout -k 10 1100 python -m pytest tests/test_lighting_gpu.py tests/test_lighting_ext_gpu.py tests/test_golden.py tests/test_post_gpu.py tests/test_shard_chain.py tests/test_fullsize_gpu.py tests/test_rt.py -x -q -m gpu > gpurun_out/r4_round2_tests.log 2>&1 || { tail -40 gpurun_out/r4_round2_tests.log; exit 1; }
tail -3 gpurun_out/r4_round2_tests.log
timeout -k 10 300 python tools/bench_passes.py --only "copy,bloom,tonemap,RT+cache,RT only" --iters 50 2>/dev/null | grep -i "copy\|bloom\|tonemap\|RT" | tee gpurun_out/r4_round2_passes.txt
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_chain.err | tee gpurun_out/r4_chain.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4k_probe_gi_chain ms/step', d['ms_per_step'], 'lighting', d['roofline']['kernel_ms_mean'])"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_kt_bloom -o kt --output-format csv -- python3 tools/bench_passes.py --only "bloom" --iters 20 > gpurun_out/r4_kt_bloom.log 2>&1; python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4_kt_bloom/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bloom' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
