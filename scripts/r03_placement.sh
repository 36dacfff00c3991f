#!/bin/bash
# scripts/r03_placement.sh <tag> — placement: hipMalloc draws against physical chunks mapped through the virtual-memory API; the
# gather probe and the translation probe against the real batch
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout -s KILL 600 python scripts/placement_study.py --copies 5 --parts 1 --chunks-mb 0 2 64 1024 16384 --no-engine-draws > "$OUT/study.log" 2>&1
grep -v "amdgpu.ids" "$OUT/study.log" | tail -50
echo "=== done $(date +%T)"
