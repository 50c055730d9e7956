#!/bin/bash
# A static counter record for EVERY bench.py workload (profiles/roofline_static.json, folded in afterwards with
# tools/pmc_to_static.py on the summaries this writes), four rocprofv3 --pmc passes per workload, never combined with tracing.
# The 8K workloads synthesise their inputs on the host (--synth-device cpu --no-light-stats): see profiles/README.md "8K under --pmc".
# usage: [R=r5] tools/static_all.sh <workload> [<workload> ...]
set -u
R=${R:-r5}
export TMPDIR=/tmp
mkdir -p gpurun_out/${R}_static
GROUPS_=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
  "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
for wl in "$@"; do
  extra=""
  case $wl in 8k_*) extra="--synth-device cpu --no-light-stats";; esac
  d=gpurun_out/${R}_static/$wl
  rm -rf $d; mkdir -p $d
  i=0
  for g in "${GROUPS_[@]}"; do
    timeout -k 10 240 rocprofv3 --pmc $g -d $d/g$i -o pmc --output-format csv -- python3 bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 $extra > $d/g$i.log 2>&1 \
      || { echo "$wl group $i ($g) failed: $(tail -3 $d/g$i.log | tr '\n' ' ' | cut -c1-300)"; exit 1; }
    i=$((i+1))
  done
  { for g in "${GROUPS_[@]}"; do echo "pmc group: $g"; done; python3 tools/pmc_summary.py $d k_; } > gpurun_out/${R}_static/${R}_pmc_static_$wl.txt
  find $d -name "*.csv" -delete
  echo "$wl done: $(grep -c mean= gpurun_out/${R}_static/${R}_pmc_static_$wl.txt) counter means"
done
