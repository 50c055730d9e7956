#!/bin/bash
# one rank of eight / four / two, emulated on one GPU (no exchanges): frame time on one, two and three work streams, enqueued from Python
# and by the library's loop (sah_chain_submit), with and without HIP graphs; then the kernels of the world-8 frame (rocprofv3 --kernel-trace)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r6_n8.txt
for w in "8 3" "4 1" "2 0"; do timeout -k 10 300 python tools/experiments/chain_two_streams.py $w 2>/dev/null | tee -a gpurun_out/r6_n8.txt; done
echo "--- the same, whole irradiance atlas widened every frame (probe_generation 0: rounds 3-4's setting)" | tee -a gpurun_out/r6_n8.txt
timeout -k 10 300 python tools/experiments/chain_two_streams.py 8 3 --rebuild-copies 2>/dev/null | head -9 | tee -a gpurun_out/r6_n8.txt
rm -rf gpurun_out/r6_kt_n8
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r6_kt_n8 -o kt --output-format csv -- python3 tools/experiments/chain_two_streams.py 8 3 > gpurun_out/r6_kt_n8.log 2>&1; python3 - <<'PY' | tee -a gpurun_out/r6_n8.txt
import csv,glob
print("--- kernel trace of the world-8 frame (rocprofv3 --kernel-trace --stats; all phases of the script: launches of the multi-stream phases overlap and run longer each): calls, average ns, share")
for f in glob.glob('gpurun_out/r6_kt_n8/**/*kernel_stats.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'sah::' in r['Name']]
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows: print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f'%float(r['AverageNs'])).rjust(10), ('%.1f%%'%(100*float(r['TotalDurationNs'])/tot)).rjust(7))
PY
