#!/bin/bash
# Second half of the evidence refresh (tools/refresh_profiles.sh is the first): PMC of the ray-tracing kernels and the randomised
# HIP-vs-oracle sweeps.  Output under gpurun_out/refresh_extras/, copied into profiles/ by tools/copy_profiles.py.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/refresh_extras
rm -rf $O; mkdir -p $O
PMC_ARGS="--no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 --workload 4k_probe_gi_chain_traced" PMC_KERNEL=k_rtao bash tools/pmc_collect.sh $O/pmc_rt > $O/pmc_rt.txt 2>&1 && echo "pmc rt ok"
for k in k_sun_shadow_mask k_probe_trace k_rtgi_trace; do python3 tools/pmc_summary.py $O/pmc_rt $k >> $O/pmc_rt.txt; done
# texture-addresser load of the same kernels (derived counters: one or two per pass)
echo "# TA counters (tools/pmc_ta.sh): TA_BUSY_avr is cycles per TA, GRBM_GUI_ACTIVE the sum over the 8 XCDs" >> $O/pmc_rt.txt
bash tools/pmc_ta.sh >> $O/pmc_rt.txt 2>&1 && echo "pmc ta ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ktrace_traced -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --workload 4k_probe_gi_chain_traced --steps 10 --warmup 3 > $O/ktrace_traced.log 2>&1 && echo "ktrace traced ok"
timeout -k 10 400 python3 tools/stress_rt.py --cases 40 > $O/stress_rt.txt 2>&1 && echo "stress rt ok"
timeout -k 10 400 python3 tools/stress_parity.py --seeds 12 > $O/stress_parity.txt 2>&1 && echo "stress parity ok"
timeout -k 10 400 python3 tools/stress_post.py > $O/stress_post.txt 2>&1 && echo "stress post ok"
timeout -k 10 400 python3 tools/stress_raster.py > $O/stress_raster.txt 2>&1 && echo "stress raster ok"
# the 8K command that used to die under --pmc (light_stats' 16,384 outstanding dispatches), once, after the fix: its one result line
bash tools/experiments/r4/r4_segv_fixed.sh > $O/segv_fixed.txt 2>&1 && echo "segv fixed: $(cat $O/segv_fixed.txt)"
# what travels back from the GPU box is limited (64 MiB): the raw counter and trace CSVs have been summarised above
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -path "*/g[0-9]*" -name "*.csv" -delete
du -sh $O | tail -1
