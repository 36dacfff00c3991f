#!/bin/bash
# round 5, session 8: which fold takes the tiles (phase-clock build of exact_tiled.hip), the corner test again
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s8; mkdir -p $OUT
export TMPDIR=/tmp
USEARCH_AMD_LIBRARY=$PWD/usearch_amd/lib_phases/libusearch_amd.so timeout 400 python scripts/exact_knockout.py --combos 64,66,3 --repeats 2 > $OUT/phases.log 2>&1; grep -E "knockout|exact phases" $OUT/phases.log | tail -12
