#!/bin/bash
# round 5, session 21: the headline bench line with roofline.traffic from session 20's PMC passes (traffic.json recomputed over the timed
# steps alone: one of the nineteen dispatches session 20's own post-processing averaged was a placement trial's judge launch)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final/headline
timeout 600 python bench.py --expansion 608 --traffic-json profiles/r05_final/headline/traffic.json --wave-clock --warmup 14 > gpurun_out/r05_final/headline/bench.json 2> gpurun_out/r05_final/headline/bench.log
tail -c 1500 gpurun_out/r05_final/headline/bench.json
