#!/bin/bash
# round 5, session 5: the fixes of this round's second sitting under the GPU tests they touch, then where the wide exact tile's time
# goes (scripts/exact_knockout.py) and its bench line with the wave number held in a scalar register (no spills in the cos kernels)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_dropin.py tests/test_gpu_sharded.py tests/test_gpu_filtered.py -q -x > $OUT/pytest_a.log 2>&1; tail -4 $OUT/pytest_a.log
timeout 600 python -m pytest tests/test_gpu_search_parity.py -q -x -k "team or benchmarked" > $OUT/pytest_b.log 2>&1; tail -4 $OUT/pytest_b.log
timeout 400 python scripts/exact_knockout.py > $OUT/knockout.log 2>&1; grep knockout $OUT/knockout.log
timeout 400 python bench.py --exact --no-cpu-baseline --steps 10 --warmup 3 > $OUT/exact_bench.json 2> $OUT/exact_bench.log; tail -c 600 $OUT/exact_bench.json
