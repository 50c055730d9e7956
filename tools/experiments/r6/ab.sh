#!/bin/bash
# A/B of bench.py workloads: build_ab/<partner>.so against the in-tree library, alternating, in one run on one box.
#   tools/experiments/r6/ab.sh <partner> <reps> <workload> [<workload> ...]        (extra bench.py arguments: AB_ARGS="...")
P=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do
  for w in "$@"; do
    for lib in $P product; do
      if [ $lib = product ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$lib.so; fi
      python bench.py --workload $w --no-cpu-baseline --steps ${AB_STEPS:-200} --warmup 20 $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['config']['workload'].split(':')[0], d['ms_per_step'], d['roofline']['kernel_ms_mean'], flush=True)"
    done
  done
done
