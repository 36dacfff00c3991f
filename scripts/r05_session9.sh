#!/bin/bash
# round 5, session 9: the list inserts behind the compiler's back (no wait for the fills in flight): tests, then 0 / 64 / 32 / 0
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s9; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exact.py -q -x > $OUT/pytest_exact.log 2>&1; tail -5 $OUT/pytest_exact.log | cut -c1-300
timeout 400 python scripts/exact_knockout.py --combos 0,64,1,3,0 --repeats 3 > $OUT/knockout.log 2>&1; grep -E "knockout" $OUT/knockout.log
