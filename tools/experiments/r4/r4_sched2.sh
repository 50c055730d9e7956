#!/bin/bash
# -enable-post-misched=0 on the other translation units, one at a time: the passes each of them owns (timing only; same instructions)
set -o pipefail
mkdir -p gpurun_out
{
run() { # variant, bench_passes filter
  if [ $1 = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$1.so; fi
  echo "== $1"
  timeout -k 10 300 python tools/bench_passes.py --only "$2" --iters 100 2>/dev/null | grep -v "^$\|amdgpu.ids"
}
run base "copy scene,bloom chain,tonemap,lpv propagate,lpv clear,probe,raster,lpv inject"
run np_post "copy scene,bloom chain"
run np_tonemap_tol "tonemap composite, tol"
run np_tonemap "tonemap composite"
run np_lpv "lpv propagate,lpv clear"
run np_probes "probe"
run np_raster "raster,lpv inject"
for v in base np_rt; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain_traced --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['traced']; print('$v', {k:t[k] for k in ('rt_build_ms','rtao_ms','sun_shadow_mask_ms','probe_trace_ms','rtgi_trace_ms_not_in_frame')})"
done
} | tee gpurun_out/r4_sched2.txt
