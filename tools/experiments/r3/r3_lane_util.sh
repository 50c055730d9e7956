export TMPDIR=/tmp
O=gpurun_out/lane_util; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES -d $O/g0 -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 --workload 4k_probe_gi_chain_traced > $O/g0.log 2>&1 || { echo failed; tail -5 $O/g0.log; }
for k in k_rtao k_sun_shadow_mask k_probe_trace k_rtgi_trace k_lighting_tiled; do python3 tools/pmc_summary.py $O $k; done
