#!/bin/bash
# A/B of the fix-up kernel's grid cap (lighting.hip: SAH_FIXUP_MAX_WGS, product value 2048) on a frame that lists nothing and on one that lists
# 29 K pixels.  Build the variants first, each into its own library:
#   for n in 512 1024 4096; do SAH_EXTRA_HIPCC_FLAGS="-DSAH_FIXUP_MAX_WGS=$n" SAH_HIP_LIBRARY=$PWD/build_ab/fix$n.so python -m androidrenderer_amd.build; done
# then on the GPU box: bash tools/experiments/r5/ab_fixup_grid.sh
for rep in 1 2; do for w in 4k_deferred_gi 4k_deferred_gi_random; do for lib in product fix512 fix1024 fix4096; do
  if [ $lib = product ]; then unset SAH_HIP_LIBRARY; else [ -f build_ab/$lib.so ] || continue; export SAH_HIP_LIBRARY=$PWD/build_ab/$lib.so; fi
  python bench.py --workload $w --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['config']['workload'].split(':')[0], d['ms_per_step'])"
done; done; done
