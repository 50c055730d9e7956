#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for cfg in "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|4320|7680|7" "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|2160|3840|7" "SQ_INSTS_VALU SQ_WAVES|4320|7680|7"; do
  IFS='|' read -r ctr h w c <<< "$cfg"
  timeout -k 10 250 rocprofv3 --pmc $ctr -d gpurun_out/r4_segv_h2d$i -o pmc --output-format csv -- python3 tools/experiments/r4/segv/h2d_then_launch.py gpurun_out/r4_segv_h2d_report$i.txt $h $w $c > gpurun_out/r4_segv_h2d$i.log 2>&1
  echo "counters [$ctr] ${w}x${h}x$c: rc=$? last: $(grep -a '^uploaded\|^done\|^launched\|^host' gpurun_out/r4_segv_h2d$i.log | tail -1); $(head -3 gpurun_out/r4_segv_h2d_report$i.txt 2>/dev/null | tr '\n' ' ' | cut -c1-200)"
  grep -a -m3 "librocprofiler\|libhsa" gpurun_out/r4_segv_h2d_report$i.txt 2>/dev/null | cut -c1-120
  rm -rf gpurun_out/r4_segv_h2d$i
  i=$((i+1))
done
# (round 5: the loop that re-ran the faulting command three times 'for the record of how often it faults' is gone — the cause is known:
#  profiles/README.md "8K under --pmc"; tools/experiments/r4/r4_segv_fixed.sh runs the fixed command once)
