#!/bin/bash
# Regenerates the evidence under gpurun_out/ that gets copied into profiles/ (run on the GPU box from the repo root):
#   bench line, rocprofv3 kernel-trace stats of the same command, PMC passes (traffic), per-pass table.
set -u
export TMPDIR=/tmp
O=gpurun_out/refresh
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err && echo "bench ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > $O/ktrace.log 2>&1 && echo "ktrace ok"
bash tools/pmc_collect.sh $O/pmc > $O/pmc.txt 2>&1 && echo "pmc ok"
python3 tools/bench_passes.py > $O/passes.txt 2>&1 && echo "passes ok"
for w in 1080p_64_lights 4k_256_lights 4k_probe_gi_chain 4k_lpv_gi_chain 4k_deferred_gi_random 4k_deferred_gi_scene_shadow 4k_deferred_gi_produced 8k_deferred_gi 8k_1024_lights_gi; do python3 bench.py --workload $w --steps 50 --warmup 5 --cpu-seconds 3 > $O/bench_$w.json 2>> $O/bench.err && echo "$w ok"; done
# the counter group pmc_collect.sh leaves out (derived TA / TCP counters): one run each, evidence kept for the cause
for c in TA_BUSY_avr TCP_TCC_READ_REQ_sum "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --pmc $c -d $O/ta_$n -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/ta_$n.log 2>&1; echo "TA/TCP probe '$c': rc=$?" >> $O/ta_probe.txt
  tail -3 $O/ta_$n.log >> $O/ta_probe.txt
done
rocprofv3 --list-avail > $O/list_avail.txt 2>&1; grep -c "TA_\|TCP_" $O/list_avail.txt >> $O/ta_probe.txt
