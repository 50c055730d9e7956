#!/bin/bash
# A/B builds of the ray tracer: libraries androidrenderer_amd/_exp/libsah_<variant>.so built with SAH_EXTRA_HIPCC_FLAGS (see README.md)
set -o pipefail
mkdir -p gpurun_out
for v in base ${VARIANTS-}; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/androidrenderer_amd/_exp/libsah_$v.so; fi
  timeout -k 10 300 python -m pytest tests/test_rt.py -x -q -m gpu 2>&1 | tail -1
  timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain_traced --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r3_rt_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['traced']; print('$v', d['ms_per_step'], {k:t[k] for k in ('rt_build_ms','rtao_ms','sun_shadow_mask_ms','probe_trace_ms','rtgi_trace_ms_not_in_frame')})" || { tail -20 gpurun_out/r3_rt_bench.err; exit 1; }
done
