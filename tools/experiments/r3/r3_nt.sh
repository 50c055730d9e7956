#!/bin/bash
set -o pipefail
for v in base nt base nt; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/androidrenderer_amd/_exp/libsah_$v.so; fi
  echo "== $v"; python tools/bench_passes.py --only "lighting" --iters 100 2>&1 | grep -E "off/none|CSM\+LPV atrium|RT only"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'])"
done
