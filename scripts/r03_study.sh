#!/bin/bash
# scripts/r03_study.sh <tag> [study args…] — scripts/placement_study.py alone
set -u
TAG=$1; shift 1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout -s KILL 600 python scripts/placement_study.py "$@" > "$OUT/study.log" 2>&1; grep -v "amdgpu.ids" "$OUT/study.log" | tail -30
echo "=== done $(date +%T)"
