#!/bin/bash
# round 5, session 18: what a clear-free visited set could save at most in time — the slab's clear still issued, nobody waits for it
# (an experimental build of the two short-row units) — against the product library; 20M-vector slices of C5 and C4
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s18; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "c5 64" "c4 80"; do set -- $cfg
for lib in lib lib_noclear lib lib_noclear; do
USEARCH_AMD_LIBRARY=$PWD/usearch_amd/$lib/libusearch_amd.so timeout 300 python bench.py --config $1 --n 20000000 --expansion $2 --steps 8 --warmup 2 --no-cpu-baseline --no-stress-rows --no-load-timing --no-host-api --no-placement-check --recall-queries 2000 > $OUT/$1_$lib.json 2> $OUT/$1_$lib.log
python - $OUT/$1_$lib.json $1 $lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "recall", d["config"].get("recall_at_k"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
done; done
