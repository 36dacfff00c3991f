#!/bin/bash
# round 5, session 22: the b1 x 128 walk at 7 waves per SIMD (72 VGPRs, 23 spilled) against the product's 6 (80 VGPRs): 20M slice of C5
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s22; mkdir -p $OUT
export TMPDIR=/tmp
for lib in lib lib_w7 lib lib_w7; do
USEARCH_AMD_LIBRARY=$PWD/usearch_amd/$lib/libusearch_amd.so timeout 300 python bench.py --config c5 --n 20000000 --expansion 64 --steps 8 --warmup 2 --no-cpu-baseline --no-stress-rows --no-load-timing --no-host-api --no-placement-check --recall-queries 2000 > $OUT/c5_$lib.json 2> $OUT/c5_$lib.log
python - $OUT/c5_$lib.json $lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("c5", sys.argv[2], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "recall", d["config"].get("recall_at_k"), "waves", d["config"].get("persistent_waves"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
