#!/bin/bash
# shadow-mask time against the number of sun samples (where do the 8 rays per pixel go?)
for n in 1 2 4 8 16; do
  timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain_traced --steps 5 --warmup 2 --no-cpu-baseline --shadow-samples $n 2> gpurun_out/r3_mask.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['config']['traced']; print('samples $n', t['sun_shadow_mask_ms'], t['mask_pixels_0_between_1'])" || { tail -20 gpurun_out/r3_mask.err; exit 1; }
done
