#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s10; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "cos i8 128 9000 16" "l2sq i8 96 20011 16" "cos i8 200 9000 16"; do
echo "== $cfg"; timeout 300 python scripts/probes_exact_diff.py $cfg 2>&1 | grep attempt
done
echo "== global lists forced, k=10"; USEARCH_AMD_EXACT_GLOBAL_LISTS=1 timeout 300 python scripts/probes_exact_diff.py cos i8 200 9000 10 2>&1 | grep attempt
timeout 600 python -m pytest tests/test_gpu_exact.py -q -x > $OUT/pytest_exact.log 2>&1; tail -3 $OUT/pytest_exact.log | cut -c1-300
timeout 400 python scripts/exact_knockout.py --combos 0,64,1,3,0 --repeats 3 > $OUT/knockout.log 2>&1; grep -E "knockout" $OUT/knockout.log
