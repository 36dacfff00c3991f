#!/bin/bash
# scripts/r03_twocells.sh <tag> — the 2-cells-per-lane `top` build (64 < expansion <= 128 on short rows) against the 4-cell one,
# after the whole GPU suite.
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
echo "=== suite $(date +%T)"
timeout -s KILL 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "rc=$?"; tail -3 "$OUT/pytest.log"
echo "=== i8 $(date +%T)"
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 96 --dtype i8 --ef 80 128 --queries 100000 --modes 2 --steps 3 \
    --env "" USEARCH_AMD_NO_TWO_CELLS=1 > "$OUT/i8.log" 2>&1
grep "^ef=\|^---\|GPU-built\|rror" "$OUT/i8.log"
echo "=== b1 $(date +%T)"
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 128 --dtype b1 --ef 96 --queries 100000 --modes 2 --steps 3 \
    --env "" USEARCH_AMD_NO_TWO_CELLS=1 > "$OUT/b1.log" 2>&1
grep "^ef=\|^---\|GPU-built\|rror" "$OUT/b1.log"
echo "=== done $(date +%T)"
