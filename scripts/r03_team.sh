#!/bin/bash
# scripts/r03_team.sh <tag>: the team build (four waves per query) — the GPU suite, then latencies with and without it on the headline index
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$REPO"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
timeout -s KILL 900 python -m pytest tests -m gpu -q --maxfail=8 > "$OUT/pytest.log" 2>&1; echo "tests rc=$?"; tail -12 "$OUT/pytest.log" | cut -c1-300
timeout -s KILL 400 python scripts/latency_check.py > "$OUT/latency.log" 2>&1; echo "latency rc=$?"; grep -v amdgpu.ids "$OUT/latency.log" | tail -12
