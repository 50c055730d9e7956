#!/bin/bash
# A/B of tools/bench_passes.py passes: build_ab/<partner>.so ... against the in-tree library, alternating.
#   tools/experiments/r6/ab_passes.sh "<--only filter>" <reps> <partner> [<partner> ...]
F=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do
  for lib in "$@" product; do
    if [ $lib = product ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$lib.so; fi
    python tools/bench_passes.py --only "$F" 2>/dev/null | grep " ms " | sed "s/^/$lib: /"
  done
done
