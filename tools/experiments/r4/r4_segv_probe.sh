#!/bin/bash
# Where does `rocprofv3 --pmc ... -- python3 bench.py --workload 8k_1024_lights_gi` fault?  (tools/experiments/r4/segv/)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/r4_segv_probe -o pmc --output-format csv -- python3 tools/experiments/r4/segv/segv_probe.py gpurun_out/r4_segv_report.txt --workload 8k_1024_lights_gi --no-cpu-baseline --steps 2 --warmup 1 --ramp-ms 0 > gpurun_out/r4_segv_probe.log 2>&1
echo "rc=$?"
if [ -s gpurun_out/r4_segv_report.txt ]; then
  head -40 gpurun_out/r4_segv_report.txt
  # nearest dynamic symbols of the frames in the profiler / runtime libraries
  grep -o '^/[^ (]*(+0x[0-9a-f]*)' gpurun_out/r4_segv_report.txt | sort -u | while read f; do
    lib=${f%%(*}; off=${f##*(+}; off=${off%)}
    case $lib in *python3*|*libtorch*|*_ctypes*|*libffi*) continue;; esac
    echo "$lib $off -> $(addr2line -f -C -e $lib $off 2>/dev/null | head -1)"
  done > gpurun_out/r4_segv_symbols.txt
  cat gpurun_out/r4_segv_symbols.txt
  a=$(grep -o 'fault address 0x[0-9a-f]*' gpurun_out/r4_segv_report.txt | cut -d' ' -f3)
  echo "fault address $a; the mappings around it:"
  python3 - "$a" <<'PY'
import sys
a = int(sys.argv[1], 16)
rows = []
for line in open("gpurun_out/r4_segv_report.txt"):
    p = line.split()
    if p and "-" in p[0] and len(p) >= 5:
        try:
            lo, hi = (int(x, 16) for x in p[0].split("-"))
        except ValueError:
            continue
        rows.append((lo, hi, line.rstrip()))
rows.sort()
for i, (lo, hi, line) in enumerate(rows):
    if hi >= a - (1 << 22) and lo <= a + (1 << 22):
        print(("  >>" if lo <= a < hi else "    "), line, f"[{(hi - lo) >> 10} KiB]")
PY
else
  tail -5 gpurun_out/r4_segv_probe.log
fi
