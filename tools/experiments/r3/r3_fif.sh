#!/bin/bash
set -o pipefail
for w in 4k_probe_gi_chain 4k_lpv_gi_chain; do for f in 1 2 1 2; do
timeout -k 10 300 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --frames-in-flight $f 2>gpurun_out/r3_fif.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w frames_in_flight $f', d['ms_per_step'], d['value'], r['kernel_ms_mean'], d['config'].get('frames_in_flight'))" || { tail -20 gpurun_out/r3_fif.err; exit 1; }
done; done
