#!/bin/bash
# scripts/r03_remap.sh <tag> [processes]: one physical scratch block under several mappings (USEARCH_AMD_SCRATCH_REMAP), in a few processes
set -u
TAG=$1; N=${2:-3}
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$REPO"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
for i in $(seq 1 $N); do
  USEARCH_AMD_SCRATCH_REMAP=12 USEARCH_AMD_PLACEMENT_LOG=1 timeout -s KILL 400 python bench.py --expansion 608 --recall-queries 0 --no-cpu-baseline --no-stress-rows --no-placement-check --no-host-api --steps 3 > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.log"
  echo "--- process $i rc=$?"; grep "usearch_amd\]" "$OUT/bench_$i.log" | cut -c1-400
  python -c "
import json; d=json.load(open('$OUT/bench_$i.json')); print('kernel_ms', d['roofline']['kernel_ms'], d['config']['placement']['matrix'])"
done
