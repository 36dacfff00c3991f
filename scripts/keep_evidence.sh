#!/bin/bash
# scripts/keep_evidence.sh <gpurun_out/…/dir> <profiles/…/dir>: copies one profile_round.sh output directory into profiles/,
# with the counter CSVs trimmed to the timed launches (scripts/trim_pmc.py)
set -eu
FROM=$1; TO=$2
mkdir -p "$TO"
for f in bench.json bench.log kernel_stats.csv kernel_trace_timed.json traffic.json stats_bench.json pick.log; do
  [ -f "$FROM/$f" ] && cp "$FROM/$f" "$TO/$f"
done
for f in "$FROM"/pmc_*.csv; do
  [ -f "$f" ] && python3 "$(dirname "$0")/trim_pmc.py" "$f" "$TO/$(basename "$f")"
done
du -sh "$TO"
