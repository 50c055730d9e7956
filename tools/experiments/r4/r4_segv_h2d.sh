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
# and the real thing once more, for the record of how often it faults: the bench's own synthesis, 4 counters, three times
for k in 1 2 3; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/r4_segv_again -o pmc --output-format csv -- python3 tools/experiments/r4/segv/segv_probe.py gpurun_out/r4_segv_again_report$k.txt --workload 8k_deferred_gi --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 > gpurun_out/r4_segv_again$k.log 2>&1
  echo "bench 8k_deferred_gi, GPU synthesis, 4 counters, run $k: rc=$? $(grep -a -m1 'fault address' gpurun_out/r4_segv_again_report$k.txt 2>/dev/null | cut -c1-120) $(grep -a -m1 -o 'gpu_kernel_impl[^(]*<[^>]*>' gpurun_out/r4_segv_again_report$k.txt 2>/dev/null | head -1 | cut -c1-120)"
  rm -rf gpurun_out/r4_segv_again
done
