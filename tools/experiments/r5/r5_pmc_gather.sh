#!/bin/bash
# is the 5.5x / 4.9x "traffic over algorithmic bytes" of the gather-heavy workloads real, or the guide's x2 FETCH_SIZE
# correction (stated for WIDE coalesced reads) applied to narrow gathers?  (1) which request-size counters gfx950 offers; (2) the
# calibration kernels of tools/microbench/fetch_calib.hip (known bytes / known distinct lines); (3) the two workloads, with the memory-side
# request counters beside the L2 miss count.  One rocprofv3 --pmc pass per group, never combined with tracing.
export TMPDIR=/tmp
O=gpurun_out/r5_pmc_gather; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_MISS[A-Z_]*\|TCC_HIT[A-Z_]*\|TCC_REQ[A-Z_]*\|TCC_READ[A-Z_]*\|TCC_BUBBLE[A-Z_]*\|FETCH_SIZE\|TCP_TCC_READ_REQ[A-Z_]*\|TCP_TOTAL_CACHE_ACCESSES[A-Z_]*" | sort -u > $O/avail.txt
echo "== counters offered:"; tr '\n' ' ' < $O/avail.txt; echo
GROUPS_=("FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_HIT_sum TCC_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum")
run() {  # name, command...
  local name=$1; shift
  local i=0
  for g in "${GROUPS_[@]}"; do
    timeout -k 10 200 rocprofv3 --pmc $g -d $O/$name/g$i -o pmc --output-format csv -- "$@" > $O/$name.g$i.log 2>&1 || echo "$name group $i ($g) failed: $(tail -1 $O/$name.g$i.log)"
    i=$((i+1))
  done
}
run calib ./tools/microbench/fetch_calib
echo "== calibration (tools/microbench/fetch_calib: 512 MiB / 128 MiB / 64 MiB streamed at 16 / 4 / 2 B per lane; 2^20 2-byte loads from distinct lines; 2^20 PCF footprints)"
grep "known bytes" $O/calib.g0.log
python3 tools/pmc_summary.py $O/calib ""
for w in 1080p_64_lights 4k_deferred_gi_random 4k_deferred_gi; do
  run $w python3 bench.py --workload $w --no-cpu-baseline --no-light-stats --steps 3 --warmup 1 --ramp-ms 0
  echo "== $w"
  python3 tools/pmc_summary.py $O/$w k_lighting
done
