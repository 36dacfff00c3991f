#!/bin/bash
# scripts/profile_round.sh <tag> <n> [bench args…] — one GPU-box session that produces the round's evidence:
#   1. bench.py at the given size (builds the index once into /dev/shm)            → gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (cached index)           → gpurun_out/<tag>_stats/
#   3. rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ mix), each in its own run  → gpurun_out/<tag>_pmc_*/
# Copy the summaries you want judged from gpurun_out/ into profiles/ afterwards.
set -u
TAG=$1; N=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
COMMON="--n $N --cache-dir /dev/shm $*"
python "$REPO/bench.py" $COMMON > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.log"
cat "$OUT/${TAG}_bench.json"; tail -8 "$OUT/${TAG}_bench.log"
EF=$(python -c "import json,sys; print(json.load(open('$OUT/${TAG}_bench.json'))['config']['expansion_search'])" 2>/dev/null || echo 256)
QUICK="$COMMON --expansion $EF --recall-queries 0 --no-cpu-baseline --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- python "$REPO/bench.py" $QUICK > "$OUT/${TAG}_stats.json" 2> "$OUT/${TAG}_stats.log"
for counters in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  name=$(echo $counters | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $counters --output-format csv -d "$OUT/${TAG}_pmc_$name" -- python "$REPO/bench.py" $QUICK > "$OUT/${TAG}_pmc_$name.json" 2> "$OUT/${TAG}_pmc_$name.log" || echo "pmc $counters failed"
done
# keep the merged-back payload small: summaries and per-dispatch counter rows of OUR kernel only
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*counter_collection.csv" | while read f; do head -1 "$f" > "$f.search"; grep search_kernel "$f" >> "$f.search"; rm "$f"; done
du -sh "$OUT"; find "$OUT" -name "*kernel_stats.csv" | head -3 | while read f; do echo "== $f"; head -8 "$f"; done
find "$OUT" -name "*.search" | while read f; do echo "== $f"; head -4 "$f" | cut -c1-400; done
