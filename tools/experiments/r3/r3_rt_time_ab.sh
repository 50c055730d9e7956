#!/bin/bash
# timing-only A/B of ray-tracing variants (results of SKIP builds are wrong on purpose: no tests)
for v in base ${VARIANTS-}; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/androidrenderer_amd/_exp/libsah_$v.so; fi
  timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain_traced --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r3_rt_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['traced']; print('$v', {k:t[k] for k in ('rtao_ms','sun_shadow_mask_ms','probe_trace_ms','rtgi_trace_ms_not_in_frame')})" || { tail -20 gpurun_out/r3_rt_bench.err; exit 1; }
done
