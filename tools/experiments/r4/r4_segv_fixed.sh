#!/bin/bash
# after the fix (light_stats waits every 32 lights): the command that used to fault, four counters, GPU synthesis, statistics on
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/r4_segv_fixed -o pmc --output-format csv -- python3 bench.py --workload 8k_1024_lights_gi --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 > gpurun_out/r4_segv_fixed.log 2>&1
echo "rc=$? $(grep -a -c '"metric"' gpurun_out/r4_segv_fixed.log) bench line(s); $(grep -a -o 'lights_per_tile_mean[^,]*' gpurun_out/r4_segv_fixed.log | head -1)"
rm -rf gpurun_out/r4_segv_fixed
