#!/bin/bash
# timing-only ablation of k_copy_bloom_mip0 (tools/experiments/r4/variants.py cbm_*): what the copy's four taps per cell, the antialiased
# stores and the mip 0 filter cost, next to the two separate passes.  Images of the variants are wrong on purpose.
set -o pipefail
mkdir -p gpurun_out
{
for v in base cbm_one_tap cbm_no_aa_store cbm_no_filter cbm_one_tap_no_filter; do
  if [ $v = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/bench_passes.py --only "copy scene,bloom chain" --iters 200 2>gpurun_out/r4_cbm.err | grep -v "^$" || { tail -5 gpurun_out/r4_cbm.err; exit 1; }
done
} | tee gpurun_out/r4_cbm_ablate.txt
