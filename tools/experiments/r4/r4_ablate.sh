#!/bin/bash
# timing-only ablation builds (tools/experiments/r4/variants.py): kernel time of the headline kernel and of the tiled cache kernel with one
# stage replaced by a stand-in.  Images of the variants are wrong on purpose.
set -o pipefail
mkdir -p gpurun_out
run() {  # name workload
  if [ "$1" = base ]; then unset SAH_HIP_LIBRARY; else export SAH_HIP_LIBRARY=$PWD/build_ab/$1.so; fi
  timeout -k 10 200 python bench.py --workload $2 --steps 60 --warmup 10 --no-cpu-baseline $3 2>gpurun_out/r4_ablate.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s %-20s ms/step %.4f  kernel mean %.4f  min %s' % ('$1', '$2', d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min']))" || { tail -20 gpurun_out/r4_ablate.err; exit 1; }
}
{
for v in base fast_no_pcf fast_no_brdf fast_no_lpv_taps fast_no_lpv_loads fast_no_lpv_select fast_no_geometry fast_no_deferral fast_no_decode; do run $v 4k_deferred_gi || exit 1; done
run base 4k_deferred_only || exit 1
for v in base tiled_no_rt_sun tiled_no_cache_brdf tiled_no_cheb tiled_no_depth_dir tiled_no_depth_lookup tiled_no_irr_taps tiled_no_lookups tiled_no_pixel_setup tiled_skeleton; do run $v 4k_probe_gi_chain || exit 1; done
} | tee gpurun_out/r4_ablate.txt
