#!/bin/bash
# scripts/r03_suite.sh <tag> [study args…] — the whole GPU suite, then (optionally) the placement study with the given arguments
set -u
TAG=$1; shift 1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
echo "=== suite $(date +%T)"
timeout -s KILL 1200 python -m pytest tests -m gpu -q --maxfail=10 > "$OUT/pytest.log" 2>&1; echo "rc=$?"; grep -v "^$" "$OUT/pytest.log" | tail -25
if [ $# -gt 0 ]; then
  echo "=== placement study $(date +%T)"
  timeout -s KILL 600 python scripts/placement_study.py "$@" > "$OUT/study.log" 2>&1; grep -v "amdgpu.ids" "$OUT/study.log" | tail -30
fi
echo "=== done $(date +%T)"
