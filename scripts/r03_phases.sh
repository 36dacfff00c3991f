#!/bin/bash
# scripts/r03_phases.sh <tag> — per-phase shader-clock share of the short-row walks (one-query kernel with its visited set in a
# global slab / in LDS; the two-queries-per-wave kernel both ways), `make PHASES=1` build.
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
export USEARCH_AMD_LIBRARY=$REPO/usearch_amd/lib_phases/libusearch_amd.so
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 128 --dtype b1 --ef 64 --queries 100000 --modes 2 1 4 5 --steps 1 \
    --env USEARCH_AMD_PHASES=1,USEARCH_AMD_HASH_CAP=4096 USEARCH_AMD_PHASES=1 > "$OUT/b1.log" 2>&1
grep "^ef=\|^---\|phases ef=64" "$OUT/b1.log" | grep -v "grid=1:"
timeout -s KILL 400 python scripts/sweep.py --n 20000000 --dim 96 --dtype i8 --ef 80 --queries 100000 --modes 2 1 4 5 --steps 1 \
    --env USEARCH_AMD_PHASES=1,USEARCH_AMD_HASH_CAP=4096 > "$OUT/i8.log" 2>&1
grep "^ef=\|^---\|phases ef=80" "$OUT/i8.log" | grep -v "grid=1:"
echo "=== done $(date +%T)"
