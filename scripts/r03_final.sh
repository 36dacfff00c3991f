#!/bin/bash
# scripts/r03_final.sh <tag> <what…> — the round's evidence sessions: headline | c2 | c4 | c5 | tests
set -u
TAG=$1; shift 1
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$REPO"
mkdir -p "$REPO/gpurun_out/$TAG"
for what in "$@"; do
  echo "=== $what $(date +%T)"
  case $what in
    headline) timeout -s KILL 1500 bash scripts/profile_round.sh $TAG/headline ;;
    c2) PROFILE_TRAFFIC_ONLY=1 timeout -s KILL 900 bash scripts/profile_round.sh $TAG/c2 --config c2 ;;
    c4) PROFILE_EF=80 PROFILE_TRAFFIC_ONLY=1 timeout -s KILL 1800 bash scripts/profile_round.sh $TAG/c4 --config c4 ;;
    c5) PROFILE_EF=64 PROFILE_TRAFFIC_ONLY=1 timeout -s KILL 1800 bash scripts/profile_round.sh $TAG/c5 --config c5 ;;
    headline_quick) PROFILE_EF=608 PROFILE_TRAFFIC_ONLY=1 timeout -s KILL 1500 bash scripts/profile_round.sh $TAG/headline ;;
    exact) # bench.py --exact under the kernel trace, then on its own (the line that is kept)
      mkdir -p "$REPO/gpurun_out/$TAG/exact"; cd /tmp
      timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/$TAG/exact/stats" -- python "$REPO/bench.py" --exact --no-cpu-baseline --steps 5 > "$REPO/gpurun_out/$TAG/exact/stats_bench.json" 2> "$REPO/gpurun_out/$TAG/exact/stats.log"
      find "$REPO/gpurun_out/$TAG/exact/stats" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$REPO/gpurun_out/$TAG/exact/kernel_stats.csv"; done
      rm -rf "$REPO/gpurun_out/$TAG/exact/stats"
      timeout -s KILL 600 python "$REPO/bench.py" --exact > "$REPO/gpurun_out/$TAG/exact/bench.json" 2> "$REPO/gpurun_out/$TAG/exact/bench.log"
      cat "$REPO/gpurun_out/$TAG/exact/bench.json"; head -5 "$REPO/gpurun_out/$TAG/exact/kernel_stats.csv"; cd "$REPO" ;;
    tests) timeout -s KILL 1200 python -m pytest tests -m gpu -q --maxfail=10 > "$REPO/gpurun_out/$TAG/pytest.log" 2>&1; echo "rc=$?"; tail -4 "$REPO/gpurun_out/$TAG/pytest.log" ;;
  esac
done
echo "=== done $(date +%T)"
