#!/bin/bash
# scripts/profile_round.sh <tag> [bench args…] — one GPU-box session that produces a workload's evidence FROM ONE BINARY:
#   1. bench.py (builds the index on the GPU, sweeps ef; no CPU baseline yet)            → picks the expansion
#   2. rocprofv3 --kernel-trace --stats of the same workload at that expansion             → gpurun_out/<tag>/kernel_stats.csv
#   3. rocprofv3 --pmc passes, one counter group per run (FETCH_SIZE, WRITE_SIZE [, more])  → gpurun_out/<tag>/pmc_*.csv
#   4. scripts/pmc_traffic.py (with the workload string and the source hash of this tree)   → gpurun_out/<tag>/traffic.json
#   5. bench.py again, full, with --traffic-json: the line carries roofline.traffic         → gpurun_out/<tag>/bench.json
# Copy gpurun_out/<tag>/ into profiles/ afterwards. PROFILE_TRAFFIC_ONLY=1 keeps the two PMC passes roofline.traffic needs.
set -u
TAG=$1; shift 1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ -z "${PROFILE_EF:-}" ]; then
  python "$REPO/bench.py" "$@" --no-cpu-baseline --no-stress-rows --no-placement-check --no-secondary --steps 5 --warmup 2 > "$OUT/pick.json" 2> "$OUT/pick.log"
  tail -4 "$OUT/pick.log"
  EF=$(python -c "import json; print(json.load(open('$OUT/pick.json'))['config']['expansion_search'])" 2>/dev/null || echo 256)
else
  EF=$PROFILE_EF   # the expansion is known (an earlier session picked it): no sweep
fi
# (PROFILE_WARMUP: enough warm-up launches for the engine's placement trials — up to 8 + 3 + 3, one per chip-filling launch — to be over
#  before the timed steps: a trial inside them would put its judge launches among "the last launches" the trace is trimmed to)
WARMUP=${PROFILE_WARMUP:-1}
QUICK="$* --expansion $EF --recall-queries 0 --no-cpu-baseline --no-stress-rows --no-host-api --no-placement-check --no-secondary --steps ${PROFILE_STEPS:-5} --warmup $WARMUP"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$REPO/bench.py" $QUICK > "$OUT/stats_bench.json" 2> "$OUT/stats.log"
QUICK="$QUICK --no-tune"   # the counter passes: which level the placement lands on does not enter the bytes

find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$OUT/kernel_stats.csv"; done
# the timed launches alone (the whole-process average above also covers the placement draws' candidates)
find "$OUT/stats" -name "*kernel_trace.csv" | head -1 | while read f; do
  python "$REPO/scripts/trace_timed.py" "$f" "$OUT/stats_bench.json" > "$OUT/kernel_trace_timed.json" || echo "trace_timed failed"
done
GROUPS_LIMIT=${PROFILE_TRAFFIC_ONLY:+4}
PASS=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  PASS=$((PASS + 1))
  if [ -n "${GROUPS_LIMIT:-}" ] && [ "$PASS" -gt "$GROUPS_LIMIT" ]; then break; fi
  name=$(echo $counters | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmc_$name" -- python "$REPO/bench.py" $QUICK > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.log" || echo "pmc $counters failed"
  # keep per-dispatch counter rows of the search kernel only
  find "$OUT/pmc_$name" -name "*counter_collection.csv" | head -1 | while read f; do head -1 "$f" > "$OUT/pmc_$name.csv"; grep search_kernel "$f" >> "$OUT/pmc_$name.csv"; done
  rm -rf "$OUT/pmc_$name" "$OUT/pmc_$name.json"
done
rm -rf "$OUT/stats"
python "$REPO/scripts/pmc_traffic.py" "$OUT/pmc_FETCH_SIZE.csv" "$OUT/pmc_WRITE_SIZE.csv" search_kernel "$OUT/stats_bench.json" \
    "$OUT/pmc_TCC_EA0_RDREQ_sum_TCC_EA0_RDREQ_32B_sum_.csv" "$OUT/pmc_TCC_EA0_WRREQ_sum_TCC_EA0_WRREQ_64B_sum_.csv" > "$OUT/traffic.json" && cat "$OUT/traffic.json"
python "$REPO/bench.py" "$@" --expansion $EF --traffic-json "$OUT/traffic.json" --wave-clock --no-secondary --warmup $WARMUP > "$OUT/bench.json" 2> "$OUT/bench.log"
cat "$OUT/bench.json"
du -sh "$OUT"; head -8 "$OUT/kernel_stats.csv"
