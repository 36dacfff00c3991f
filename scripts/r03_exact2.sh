#!/bin/bash
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$REPO"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
timeout -s KILL 600 python -m pytest tests/test_gpu_exact.py -q > "$OUT/pytest.log" 2>&1; echo "tests rc=$?"; tail -3 "$OUT/pytest.log"
