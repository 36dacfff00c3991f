#!/bin/bash
# round 5, session 20 (final sources): the headline's evidence with enough warm-up launches for the placement trials to be over
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final
PROFILE_WARMUP=14 PROFILE_EF=608 PROFILE_TRAFFIC_ONLY=1 timeout 1500 bash scripts/profile_round.sh r05_final/headline > gpurun_out/r05_s20_profile.log 2>&1
tail -3 gpurun_out/r05_s20_profile.log | cut -c1-300
cat gpurun_out/r05_final/headline/kernel_trace_timed.json
