#!/bin/bash
# the ray-tracing passes against the atrium at two tessellations (23.8 K and 214 K triangles), 3840 x 2160, and the GI
# generators with the hit stage's bounce branch on (sah_rt_set_bounces 1, 2).  Parity tests of the bounce branch first.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_rt.py -x -q -m gpu > gpurun_out/r4_rt_tests.log 2>&1 || { tail -30 gpurun_out/r4_rt_tests.log; exit 1; }
tail -2 gpurun_out/r4_rt_tests.log
out=gpurun_out/r4_rt_scaling.txt
: > $out
for cfg in "8 0" "24 0" "8 1" "8 2" "24 1"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --workload 4k_probe_gi_chain_traced --atrium-subdiv $1 --rt-bounces $2 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r4_rt_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['traced']
print('atrium(%s) bounces %s:' % ('$1', '$2'), {k:t[k] for k in ('triangles','levels','rt_build_ms','rtao_ms','sun_shadow_mask_ms','probe_trace_ms','probe_update_ms','rtgi_trace_ms_not_in_frame')}, 'frame ms/step', d['ms_per_step'])" >> $out || { tail -20 gpurun_out/r4_rt_bench.err; exit 1; }
done
cat $out
