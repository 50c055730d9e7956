#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for cfg in "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|1024|20000" "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|33177600|20000" "SQ_INSTS_VALU SQ_WAVES|1024|40000" "SQ_INSTS_VALU|1024|40000"; do
  IFS='|' read -r ctr n launches <<< "$cfg"
  timeout -k 10 250 rocprofv3 --pmc $ctr -d gpurun_out/r4_segv_count$i -o pmc --output-format csv -- python3 tools/experiments/r4/segv/many_dispatches.py gpurun_out/r4_segv_count_report$i.txt $n $launches > gpurun_out/r4_segv_count$i.log 2>&1
  echo "counters [$ctr] elements $n: rc=$? last: $(grep -a '^launch\|^done' gpurun_out/r4_segv_count$i.log | tail -1); $(head -3 gpurun_out/r4_segv_count_report$i.txt 2>/dev/null | tr '\n' ' ' | cut -c1-200)"
  rm -rf gpurun_out/r4_segv_count$i
  i=$((i+1))
done
