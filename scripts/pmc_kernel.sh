#!/bin/bash
# scripts/pmc_kernel.sh <tag> <kernel substring> -- <command…>
#   One rocprofv3 --kernel-trace --stats run of <command> plus one --pmc run per counter group below (counters are collected in
#   their own runs: a PMC pass serialises kernels), every run reduced to the dispatches whose kernel name contains <kernel
#   substring>: gpurun_out/<tag>/kernel_stats.csv and gpurun_out/<tag>/pmc_summary.txt (mean per dispatch of every counter).
#   PMC_GROUPS="a b c|d e" overrides the groups ('|' between runs).
set -u
TAG=$1; KERNEL=$2; shift 2
[ "$1" = "--" ] && shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
DEFAULT_GROUPS="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD|SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
IFS='|' read -r -a GROUPS_LIST <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$@" > "$OUT/stats_stdout.log" 2> "$OUT/stats.log"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$OUT/kernel_stats.csv"; done
rm -rf "$OUT/stats"
: > "$OUT/pmc_summary.txt"
PASS=0
for counters in "${GROUPS_LIST[@]}"; do
  PASS=$((PASS + 1))
  rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmc_$PASS" -- "$@" > /dev/null 2> "$OUT/pmc_$PASS.log" || echo "pass $PASS ($counters) failed" >> "$OUT/pmc_summary.txt"
  find "$OUT/pmc_$PASS" -name "*counter_collection.csv" | head -1 | while read f; do
    python - "$f" "$KERNEL" >> "$OUT/pmc_summary.txt" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r.get("Kernel_Name", "")]
sums, counts = collections.defaultdict(float), collections.defaultdict(int)
for r in rows:
    sums[r["Counter_Name"]] += float(r["Counter_Value"]); counts[r["Counter_Name"]] += 1
for name in sums:
    print(f"{name:36s} {sums[name] / counts[name]:18.1f}  (mean of {counts[name]} dispatches)")
PY
  done
  rm -rf "$OUT/pmc_$PASS"
done
grep -i "$KERNEL" "$OUT/kernel_stats.csv" | head -3
cat "$OUT/pmc_summary.txt"
