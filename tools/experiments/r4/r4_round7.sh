#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_post_gpu.py tests/test_golden.py tests/test_shard_chain.py tests/test_fullsize_gpu.py tests/test_comm_gpu.py -x -q -m gpu > gpurun_out/r4_round7_tests.log 2>&1 || { tail -40 gpurun_out/r4_round7_tests.log; exit 1; }
tail -3 gpurun_out/r4_round7_tests.log
timeout -k 10 300 python tools/bench_passes.py --only "bloom,tonemap composite, tol" --iters 100 2>/dev/null | grep -i "bloom\|tonemap"
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_chain.err | tee gpurun_out/r4_chain.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4k_probe_gi_chain ms/step', d['ms_per_step'], 'lighting', d['roofline']['kernel_ms_mean'])"
timeout -k 10 300 python tools/experiments/chain_two_streams.py 8 3 2>/dev/null | tail -2
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_kt_n8 -o kt --output-format csv -- python3 tools/experiments/chain_two_streams.py 8 3 > gpurun_out/r4_kt_n8.log 2>&1; python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4_kt_n8/**/*kernel_stats.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'sah::' in r['Name']]
    for r in rows: print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f'%float(r['AverageNs'])).rjust(10))
PY
