#!/bin/bash
# round 5, session 11: the visited-set slabs of the short-row walks in three kinds of device memory (USEARCH_AMD_SCRATCH_MEMORY =
# 0 plain, 1 uncached, 2 fine-grained): 20M-vector slices of C5 (b1 x 128, ef 64) and C4 (i8 x 96, ef 80), batch 100 000
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s11; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "c5 64" "c4 80"; do set -- $cfg
for kind in 0 1 2; do
USEARCH_AMD_SCRATCH_MEMORY=$kind timeout 300 python bench.py --config $1 --n 20000000 --expansion $2 --steps 8 --warmup 2 --no-cpu-baseline --no-stress-rows --no-load-timing --no-host-api --no-placement-check --recall-queries 2000 > $OUT/$1_mem$kind.json 2> $OUT/$1_mem$kind.log
python - $OUT/$1_mem$kind.json $1 $kind <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "scratch memory kind", sys.argv[3], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "recall", d["config"].get("recall_at_k"), d["roofline"]["kernel_instantiation"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
done; done
