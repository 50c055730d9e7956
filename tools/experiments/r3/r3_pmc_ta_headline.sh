#!/bin/bash
# texture-addresser load of the headline kernel (one or two derived counters per pass)
export TMPDIR=/tmp
O=gpurun_out/pmc_ta_headline; rm -rf $O; mkdir -p $O
i=0
for g in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  timeout -k 10 150 rocprofv3 --pmc $g -d $O/g$i -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 > $O/g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 tools/pmc_summary.py $O k_lighting_fast
