#!/bin/bash
# A/B timing of alternative builds of the library (build_ab/*.so, see SAH_HIP_LIBRARY in androidrenderer_amd/lib.py):
#   tools/ab.sh "<bench args>" name1 name2 ...     ("base" = the in-tree libsah_hip.so)
ARGS=$1; shift
for n in "$@"; do
  if [ "$n" = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$n.so; fi
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline $ARGS | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['kernel_ms_min'])"
done
