#!/bin/bash
# scripts/profile_round.sh <tag> [bench args…] — one GPU-box session that produces the round's evidence:
#   1. bench.py (builds the index on the GPU, sweeps ef, CPU baseline)               → gpurun_out/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same workload at the chosen ef          → gpurun_out/<tag>/kernel_stats.csv
#   3. rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, cache + SQ mix), own run each   → gpurun_out/<tag>/pmc_*.csv
#   4. scripts/pmc_traffic.py                                                          → gpurun_out/<tag>/traffic.json
#   5. bench.py again with --traffic-json, so the committed line carries roofline.traffic → gpurun_out/<tag>/bench.json
# Copy gpurun_out/<tag>/ into profiles/ afterwards.
set -u
TAG=$1; shift 1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python "$REPO/bench.py" "$@" > "$OUT/bench.json" 2> "$OUT/bench.log"
cat "$OUT/bench.json"; tail -12 "$OUT/bench.log"
EF=$(python -c "import json,sys; print(json.load(open('$OUT/bench.json'))['config']['expansion_search'])" 2>/dev/null || echo 256)
QUICK="$* --expansion $EF --recall-queries 0 --no-cpu-baseline --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$REPO/bench.py" $QUICK > "$OUT/stats_bench.json" 2> "$OUT/stats.log"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$OUT/kernel_stats.csv"; done
# PROFILE_TRAFFIC_ONLY=1 keeps the two passes roofline.traffic needs (about half the GPU time of the full set)
GROUPS_LIMIT=${PROFILE_TRAFFIC_ONLY:+2}
PASS=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  PASS=$((PASS + 1))
  if [ -n "${GROUPS_LIMIT:-}" ] && [ "$PASS" -gt "$GROUPS_LIMIT" ]; then break; fi
  name=$(echo $counters | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmc_$name" -- python "$REPO/bench.py" $QUICK > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.log" || echo "pmc $counters failed"
  # keep per-dispatch counter rows of the search kernel only
  find "$OUT/pmc_$name" -name "*counter_collection.csv" | head -1 | while read f; do head -1 "$f" > "$OUT/pmc_$name.csv"; grep search_kernel "$f" >> "$OUT/pmc_$name.csv"; done
  rm -rf "$OUT/pmc_$name" "$OUT/pmc_$name.json"
done
rm -rf "$OUT/stats"
python "$REPO/scripts/pmc_traffic.py" "$OUT/pmc_FETCH_SIZE.csv" "$OUT/pmc_WRITE_SIZE.csv" > "$OUT/traffic.json" && cat "$OUT/traffic.json"
python "$REPO/bench.py" "$@" --expansion $EF --traffic-json "$OUT/traffic.json" > "$OUT/bench.json" 2>> "$OUT/bench.log"
cat "$OUT/bench.json"
du -sh "$OUT"; head -8 "$OUT/kernel_stats.csv"
