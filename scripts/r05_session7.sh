#!/bin/bash
# round 5, session 7: the wide exact tile's fold with the thresholds inside the sums (f16 cos / ip): tests, then the same launch
# with the fused fold (0), with the general fold forced (32) and without any fold (1)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exact.py -q -x > $OUT/pytest_exact.log 2>&1; tail -15 $OUT/pytest_exact.log | cut -c1-300
timeout 400 python scripts/exact_knockout.py --combos 0,32,1,0 > $OUT/knockout.log 2>&1; grep knockout $OUT/knockout.log
timeout 300 python bench.py --exact --no-cpu-baseline --steps 10 --warmup 3 > $OUT/exact_bench.json 2> $OUT/exact_bench.log; tail -c 700 $OUT/exact_bench.json
