#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for cfg in "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|4320|7680|600" "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|2160|3840|600" "SQ_INSTS_VALU SQ_WAVES|4320|7680|600"; do
  IFS='|' read -r ctr h w launches <<< "$cfg"
  timeout -k 10 250 rocprofv3 --pmc $ctr -d gpurun_out/r4_segv_deep$i -o pmc --output-format csv -- python3 tools/experiments/r4/segv/deep_queue.py gpurun_out/r4_segv_deep_report$i.txt $h $w $launches > gpurun_out/r4_segv_deep$i.log 2>&1
  echo "counters [$ctr] ${w}x$h: rc=$? last: $(grep -a '^launch\|^done\|^inputs' gpurun_out/r4_segv_deep$i.log | tail -1); $(head -3 gpurun_out/r4_segv_deep_report$i.txt 2>/dev/null | tr '\n' ' ' | cut -c1-200)"
  grep -a -m3 "librocprofiler\|libhsa" gpurun_out/r4_segv_deep_report$i.txt 2>/dev/null | cut -c1-120
  rm -rf gpurun_out/r4_segv_deep$i
  i=$((i+1))
done
