#!/bin/bash
# tolerance tonemap: counters of the kernel (three --pmc passes) + the half-precision divide / root check
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r4_tm_pmc
i=0
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  timeout -k 10 150 rocprofv3 --pmc $g -d gpurun_out/r4_tm_pmc/g$i -o pmc --output-format csv -- python3 tools/bench_passes.py --only "tonemap composite, tol" --iters 5 > gpurun_out/r4_tm_pmc/g$i.log 2>&1 || echo "group $i failed: $(tail -2 gpurun_out/r4_tm_pmc/g$i.log | tr '\n' ' ')"
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/r4_tm_pmc k_tonemap | tee gpurun_out/r4_tm_pmc.txt
tools/microbench/half_math_check | tee gpurun_out/r4_half_math_check.txt
