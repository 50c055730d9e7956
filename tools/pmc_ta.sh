#!/bin/bash
# texture-addresser load of the ray-tracing kernels (one or two derived counters per pass: more abort rocprofv3)
export TMPDIR=/tmp
O=gpurun_out/pmc_ta; rm -rf $O; mkdir -p $O
i=0
for g in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  timeout -k 10 150 rocprofv3 --pmc $g -d $O/g$i -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 --workload 4k_probe_gi_chain_traced > $O/g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
for k in k_rtao k_sun_shadow_mask k_probe_trace k_lighting_tiled; do python3 tools/pmc_summary.py $O $k; done
