#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for cfg in "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|4320|7680|1024" "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|2160|3840|1024" "SQ_INSTS_VALU SQ_WAVES|4320|7680|1024" "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|270|480|1024"; do
  IFS='|' read -r ctr h w n <<< "$cfg"
  timeout -k 10 280 rocprofv3 --pmc $ctr -d gpurun_out/r4_segv_full$i -o pmc --output-format csv -- python3 tools/experiments/r4/segv/full_queue.py gpurun_out/r4_segv_full_report$i.txt $h $w $n > gpurun_out/r4_segv_full$i.log 2>&1
  echo "counters [$ctr] ${w}x$h, $n lights: rc=$? last: $(grep -a '^light\|^done\|^queued\|^inputs' gpurun_out/r4_segv_full$i.log | tail -1); $(head -2 gpurun_out/r4_segv_full_report$i.txt 2>/dev/null | tr '\n' ' ' | cut -c1-160)"
  grep -a -m4 "librocprofiler\|libhsa" gpurun_out/r4_segv_full_report$i.txt 2>/dev/null | cut -c1-120
  rm -rf gpurun_out/r4_segv_full$i
  i=$((i+1))
done
