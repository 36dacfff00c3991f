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
    tests) timeout -s KILL 1200 python -m pytest tests -m gpu -q --maxfail=10 > "$REPO/gpurun_out/$TAG/pytest.log" 2>&1; echo "rc=$?"; tail -4 "$REPO/gpurun_out/$TAG/pytest.log" ;;
  esac
done
echo "=== done $(date +%T)"
