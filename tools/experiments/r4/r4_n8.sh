#!/bin/bash
# one rank of eight, emulated on one GPU (no exchanges): frame time on one / two work streams, and the kernels of its frame
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/experiments/chain_two_streams.py 8 3 2>/dev/null | tee gpurun_out/r4_n8.txt
timeout -k 10 300 python tools/experiments/chain_two_streams.py 4 1 2>/dev/null | tee -a gpurun_out/r4_n8.txt
timeout -k 10 300 python tools/experiments/chain_two_streams.py 2 0 2>/dev/null | tee -a gpurun_out/r4_n8.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_kt_n8 -o kt --output-format csv -- python3 tools/experiments/chain_two_streams.py 8 3 > gpurun_out/r4_kt_n8.log 2>&1; python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4_kt_n8/**/*kernel_stats.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'sah::' in r['Name']]
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows: print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f'%float(r['AverageNs'])).rjust(10), ('%.1f%%'%(100*float(r['TotalDurationNs'])/tot)).rjust(7))
PY
