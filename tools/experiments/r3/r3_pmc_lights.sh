#!/bin/bash
# VALU instructions per pixel of the light-loop workloads
export TMPDIR=/tmp
O=gpurun_out/pmc_lights; rm -rf $O; mkdir -p $O
for w in 4k_256_lights 8k_1024_lights_gi; do
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU -d $O/$w -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 --workload $w > $O/$w.log 2>&1 || echo "$w failed"
echo "== $w"; python3 tools/pmc_summary.py $O/$w k_lighting_tiled
done
