#!/bin/bash
# (The derived TA_* / TCP_* counters are not in these groups: eight of them in ONE pass made rocprofv3 abort in round 1 — too many
#  hardware counter slots for one pass; one or two per pass collect fine, see tools/refresh_profiles.sh and profiles/r2_pmc_ta_tcp_probe.txt.)
# Collect hardware counters for one bench.py workload, one rocprofv3 --pmc pass per counter group (never combined with tracing),
# and print the per-kernel means.   usage: tools/pmc_collect.sh <out-dir> [bench.py args...]
set -u
OUT=${1:?out dir}; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
GROUPS_=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for g in "${GROUPS_[@]}"; do
  d="$OUT/g$i"
  echo "pmc group $i: $g"
  timeout -k 10 150 rocprofv3 --pmc $g -d "$d" -o pmc --output-format csv -- python3 ${PMC_SCRIPT:-bench.py} ${PMC_ARGS:---no-cpu-baseline --steps 5 --warmup 1} "$@" > "$OUT/g$i.log" 2>&1 \
    || echo "group $i ($g) failed: $(tail -2 "$OUT/g$i.log" | tr '\n' ' ')"
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" ${PMC_KERNEL:-k_lighting}
