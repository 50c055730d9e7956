#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for v in base prio_tm prio_np_tm base prio_tm prio_np_tm prio_tm_strict; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"; timeout -k 10 300 python tools/bench_passes.py --only "tonemap composite" --iters 200 2>/dev/null | grep tonemap
done
for v in base prio_rt base prio_rt; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain_traced --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['traced']; print('$v', {k:t[k] for k in ('rtao_ms','sun_shadow_mask_ms','probe_trace_ms','rtgi_trace_ms_not_in_frame')})"
done
} | tee gpurun_out/r4_sched3.txt
