#!/bin/bash
# A/B of the light workloads: the product library against build_ab/head.so (the previous commit's sources), alternating, in one run.
# usage: ab_lights.sh [workloads...]   (on the GPU box; output on stdout)
W=${@:-"4k_256_lights 1080p_64_lights"}
for rep in 1 2; do
  for w in $W; do
    for lib in head product; do
      if [ $lib = head ]; then export SAH_HIP_LIBRARY=$PWD/build_ab/head.so; else unset SAH_HIP_LIBRARY; fi
      python bench.py --workload $w --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['config']['workload'].split(':')[0], d['ms_per_step'], d['roofline']['kernel_ms_mean'])"
    done
  done
done
