#!/bin/bash
# scripts/r03_pair.sh <tag> [quick] — the two-queries-per-wave walk: parity first (small, under short timeouts), then A/B against
# the one-query kernel on 20M-vector slices of the two short-row configurations, then the phase clock (`make PHASES=1` build).
set -u
TAG=$1
QUICK=${2:-}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
if [ -z "$QUICK" ]; then
  echo "=== parity with USEARCH_AMD_PAIR=1 $(date +%T)"
  USEARCH_AMD_PAIR=1 timeout -s KILL 400 python -m pytest tests/test_gpu_golden.py tests/test_gpu_search_parity.py tests/test_gpu_fullsize.py \
      tests/test_gpu_dropin.py -m gpu -q -x > "$OUT/pytest_pair.log" 2>&1
  echo "rc=$?"; tail -4 "$OUT/pytest_pair.log"
fi
if [ "$QUICK" != "phases" ]; then
echo "=== b1 sweep $(date +%T)"
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 128 --dtype b1 --ef 64 --queries 100000 --modes 2 4 --steps 3 > "$OUT/b1.log" 2>&1
grep "^ef=\|GPU-built\|rror" "$OUT/b1.log"
echo "=== i8 sweep $(date +%T)"
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 96 --dtype i8 --ef 64 80 --queries 100000 --modes 2 4 --steps 3 > "$OUT/i8.log" 2>&1
grep "^ef=\|GPU-built\|rror" "$OUT/i8.log"
fi
echo "=== phases $(date +%T)"
export USEARCH_AMD_LIBRARY=$REPO/usearch_amd/lib_phases/libusearch_amd.so
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 128 --dtype b1 --ef 64 --queries 100000 --modes 4 --steps 1 \
    --env USEARCH_AMD_PHASES=1 > "$OUT/b1_phases.log" 2>&1
grep "phases ef=64" "$OUT/b1_phases.log" | grep -v "grid=1:" | tail -1
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 96 --dtype i8 --ef 80 --queries 100000 --modes 4 --steps 1 \
    --env USEARCH_AMD_PHASES=1 > "$OUT/i8_phases.log" 2>&1
grep "phases ef=80" "$OUT/i8_phases.log" | grep -v "grid=1:" | tail -1
echo "=== done $(date +%T)"
