#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s10; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exact.py -q -x > $OUT/pytest_exact.log 2>&1; tail -3 $OUT/pytest_exact.log | cut -c1-300
timeout 400 python scripts/exact_knockout.py --combos 0,64,1,3,0 --repeats 3 > $OUT/knockout.log 2>&1; grep -E "knockout" $OUT/knockout.log
