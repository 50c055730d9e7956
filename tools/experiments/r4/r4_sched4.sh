#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for v in base tm_prio_ilp tm_prio_clause tm_prio_bias100 tm_prio_revlocal base; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"; timeout -k 10 300 python tools/bench_passes.py --only "tonemap composite, tol" --iters 200 2>/dev/null | grep tonemap
done
for v in base post_ilp post_clause base; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"; timeout -k 10 300 python tools/bench_passes.py --only "copy scene,bloom chain" --iters 200 2>/dev/null | grep "copy\|bloom"
done
} | tee gpurun_out/r4_sched4.txt
