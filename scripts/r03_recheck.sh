#!/bin/bash
# scripts/r03_recheck.sh <tag>: C4 and C5 at the final sources, bench lines only (no counter passes)
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$REPO"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
timeout -s KILL 170 python bench.py --config c4 --expansion 80 --recall-queries 0 --no-cpu-baseline --no-stress-rows --no-placement-check --no-host-api --steps 5 > "$OUT/c4.json" 2> "$OUT/c4.log"; echo "c4 rc=$?"
python -c "
import json; d=json.load(open('$OUT/c4.json')); print('c4', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['kernel_instantiation'])"
timeout -s KILL 170 python bench.py --config c5 --expansion 64 --recall-queries 0 --no-cpu-baseline --no-stress-rows --no-placement-check --no-host-api --steps 5 > "$OUT/c5.json" 2> "$OUT/c5.log"; echo "c5 rc=$?"
python -c "
import json; d=json.load(open('$OUT/c5.json')); print('c5', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['kernel_instantiation'])"
