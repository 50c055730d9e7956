#!/bin/bash
# compiler scheduling options on the two lighting translation units (tools/experiments/r4/variants.py sched_*): same source, same bits; timing only
set -o pipefail
mkdir -p gpurun_out
{
for v in ${VARIANTS_:-base sched_max_ilp sched_max_clause sched_iter_minreg sched_bias0 sched_bias100 sched_no_post sched_relaxed_occ}; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  [ $v = base ] || [ -f build_ab/$v.so ] || continue
  for wl in ${WLS_:-4k_deferred_gi 4k_probe_gi_chain 4k_256_lights}; do
    timeout -k 10 200 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-light-stats 2>gpurun_out/r4_sched.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-20s %-20s ms/step %.4f  lighting %.4f' % ('$v', '$wl', d['ms_per_step'], d['roofline']['kernel_ms_mean']))" || { tail -5 gpurun_out/r4_sched.err; exit 1; }
  done
done
} | tee gpurun_out/r4_sched.txt
