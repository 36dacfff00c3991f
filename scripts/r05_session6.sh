#!/bin/bash
# round 5, session 6: the wide exact tile with its fills in the buffer form (one 32-bit offset per lane, pass and chunk in the scalar
# offset), exact tests first; then the knockouts again (what the fold is made of: bits 8 and 16); the NaN-safe register `top`
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exact.py -q -x > $OUT/pytest_exact.log 2>&1; tail -3 $OUT/pytest_exact.log
timeout 400 python scripts/exact_knockout.py --combos 0,1,8,16,24,3,7,0 > $OUT/knockout.log 2>&1; grep knockout $OUT/knockout.log
timeout 600 python -m pytest tests/test_gpu_search_parity.py -q -x -k "team or benchmarked or tie" > $OUT/pytest_b.log 2>&1; tail -3 $OUT/pytest_b.log | cut -c1-300
