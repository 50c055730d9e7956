#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_post_gpu.py tests/test_golden.py tests/test_shard_chain.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r4_round4_tests.log 2>&1 || { tail -40 gpurun_out/r4_round4_tests.log; exit 1; }
tail -3 gpurun_out/r4_round4_tests.log
timeout -k 10 300 python tools/bench_passes.py --only "tonemap composite, tol" --iters 100 2>/dev/null | grep -i "tonemap"
timeout -k 10 300 python bench.py --workload 4k_probe_gi_chain --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/r4_chain.err | tee gpurun_out/r4_chain.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4k_probe_gi_chain ms/step', d['ms_per_step'], 'lighting', d['roofline']['kernel_ms_mean'])"
i=0
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  timeout -k 10 150 rocprofv3 --pmc $g -d gpurun_out/r4_tm_pmc2/g$i -o pmc --output-format csv -- python3 tools/bench_passes.py --only "tonemap composite, tol" --iters 5 > gpurun_out/r4_tm_pmc2_g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/r4_tm_pmc2 k_tonemap_tol | tee gpurun_out/r4_tm_pmc2.txt
