#!/bin/bash
# Regenerates the evidence under gpurun_out/refresh/ that tools/copy_profiles.py copies into profiles/ (run on the GPU box from the repo root):
#   bench line with the driver's arguments, rocprofv3 kernel-trace stats of the same command, PMC passes, per-pass table, other workloads,
#   CPU baselines of the five configs, the N = 2 rehearsal on one GPU.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${R:-r6}
O=gpurun_out/refresh
rm -rf $O; mkdir -p $O
bash tools/pmc_collect.sh $O/pmc > $O/pmc.txt 2>&1 && echo "pmc ok"
PMC_ARGS="--no-cpu-baseline --steps 5 --warmup 1 --workload 4k_probe_gi_chain" PMC_KERNEL=k_lighting_tiled bash tools/pmc_collect.sh $O/pmc_tiled > $O/pmc_rt_cache_tiled.txt 2>&1 && echo "pmc tiled ok"
# the static half of bench.py's roofline object comes from THESE passes (the headline and the cache-GI workload, all nine counter groups; the
# other workloads' records come from tools/static_all.sh + tools/fold_static.py): folded into profiles/roofline_static.json
# before any bench line is printed
cp $O/pmc.txt profiles/${R}_pmc_4k_deferred_gi.txt; cp $O/pmc_rt_cache_tiled.txt profiles/${R}_pmc_rt_cache_tiled.txt
python3 tools/pmc_to_static.py 4k_deferred_gi profiles/${R}_pmc_4k_deferred_gi.txt k_lighting_fast 3840 2160 450 > $O/static.log 2>&1 && python3 tools/pmc_to_static.py 4k_probe_gi_chain profiles/${R}_pmc_rt_cache_tiled.txt k_lighting_tiled 3840 2160 >> $O/static.log 2>&1 && python3 tools/pmc_to_static.py 4k_probe_gi_chain_traced profiles/${R}_pmc_rt_cache_tiled.txt k_lighting_tiled 3840 2160 >> $O/static.log 2>&1 && cp profiles/roofline_static.json $O/roofline_static.json && echo "static ok"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err && echo "bench ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/ktrace.log 2>&1 && echo "ktrace ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ktrace_chain -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --workload 4k_probe_gi_chain --steps 50 --warmup 5 > $O/ktrace_chain.log 2>&1 && echo "ktrace chain ok"
PMC_SCRIPT=tools/bench_passes.py PMC_ARGS="--only tonemap --iters 5" PMC_KERNEL=k_tonemap bash tools/pmc_collect.sh $O/pmc_tm > $O/pmc_tonemap.txt 2>&1 && echo "pmc tonemap ok"
python3 tools/bench_passes.py --iters 200 > $O/passes.txt 2>&1 && echo "passes ok"
for w in 720p_deferred_only 1080p_64_lights 4k_256_lights 4k_probe_gi_chain 4k_lpv_gi_chain 4k_lpv_gi_frame 4k_probe_gi_chain_traced 4k_deferred_gi_random 4k_deferred_gi_produced 4k_deferred_gi_scene_shadow 4k_deferred_only 1080p_deferred_gi 8k_deferred_gi 8k_1024_lights_gi; do python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2>> $O/bench.err && echo "$w ok"; done
python3 bench.py --workload 4k_probe_gi_chain --strict-tonemap --steps 50 --warmup 5 --no-cpu-baseline > $O/strict_chain.json 2>> $O/bench.err && echo "strict chain ok"
python3 bench.py --workload 4k_probe_gi_chain --frames-in-flight 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/chain_fif2.json 2>> $O/bench.err && echo "two frames in flight ok"
python3 bench.py --workload 4k_lpv_gi_chain --frames-in-flight 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/lpv_chain_fif2.json 2>> $O/bench.err && echo "two frames in flight (lpv) ok"
python3 bench.py --lpv-copy rebuild --steps 200 --warmup 20 --no-cpu-baseline > $O/repack.json 2>> $O/bench.err && echo "rebuild-every-step ok"
python3 bench.py --lpv-copy kept --steps 200 --warmup 20 --no-cpu-baseline > $O/kept.json 2>> $O/bench.err && echo "kept ok"
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench200.json 2>> $O/bench.err && echo "200 steps ok"
python3 tools/cpu_baselines.py --seconds 5 > $O/cpu_baselines.txt 2>> $O/bench.err && echo "cpu baselines ok"
# (round 6: no launcher around it — bench.py starts its own ranks)
timeout -k 10 200 python3 bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu > $O/rehearse_n2.json 2> $O/rehearse_n2.err && echo "rehearsal ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ktrace_lpv_frame -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --workload 4k_lpv_gi_frame --steps 30 --warmup 5 > $O/ktrace_lpv_frame.log 2>&1 && echo "ktrace lpv frame ok"
# what travels back from the GPU box is limited (64 MiB): the raw counter and trace CSVs have been summarised above
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -path "*/g[0-9]*" -name "*.csv" -delete
du -sh $O | tail -1
