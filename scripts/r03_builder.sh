#!/bin/bash
# scripts/r03_builder.sh <tag>: the builder after the insertion-beam change — its tests, the drop-in's, and the headline build time
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$REPO"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
timeout -s KILL 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_dropin.py tests/test_cpp_class.py -q -m gpu > "$OUT/pytest.log" 2>&1; echo "tests rc=$?"; tail -3 "$OUT/pytest.log"
timeout -s KILL 600 python bench.py --expansion 608 --recall-queries 0 --no-cpu-baseline --no-stress-rows --no-placement-check --no-host-api --steps 5 > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
grep "GPU build" "$OUT/bench.log"; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['roofline']['kernel_ms'], d['config']['index_build'])"
timeout -s KILL 600 python bench.py --config c5 --vectors 20000000 --expansion 64 --recall-queries 0 --no-cpu-baseline --no-stress-rows --no-placement-check --no-host-api --steps 5 > "$OUT/bench_b1.json" 2> "$OUT/bench_b1.log"; echo "b1 rc=$?"
grep "GPU build" "$OUT/bench_b1.log"
