#!/bin/bash
# scripts/r03_exact.sh <tag>: the wide exact-search tile against the bit-exact kernel and the 64-query tile, plus the tests it touches
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$REPO"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
timeout -s KILL 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_build.py -q -x > "$OUT/pytest.log" 2>&1; echo "tests rc=$?"; tail -3 "$OUT/pytest.log"
timeout -s KILL 300 python -m pytest tests/test_gpu_search_parity.py -q -k "large_expansion" >> "$OUT/pytest.log" 2>&1; echo "parity rc=$?"; tail -2 "$OUT/pytest.log"
timeout -s KILL 600 python scripts/exact_check.py --vectors 10000000 --dim 768 --dtype f16 --queries 10000 --tiles 256 64 > "$OUT/exact_f16.log" 2>&1; echo "f16 rc=$?"; cat "$OUT/exact_f16.log" | grep -v amdgpu.ids
timeout -s KILL 300 python scripts/exact_check.py --vectors 20000000 --dim 96 --dtype i8 --queries 4096 --tiles 256 64 > "$OUT/exact_i8.log" 2>&1; echo "i8 rc=$?"; cat "$OUT/exact_i8.log" | grep -v amdgpu.ids
