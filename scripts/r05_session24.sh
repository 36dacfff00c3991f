#!/bin/bash
# round 5, session 24: a smaller frontier heap at small expansions makes room for more `seen` cells at the same residency (C5 slice)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s24; mkdir -p $OUT
export TMPDIR=/tmp
for cap in 0 384 320 256 0 384; do
if [ $cap = 0 ]; then unset USEARCH_AMD_NEXT_CAP; else export USEARCH_AMD_NEXT_CAP=$cap; fi
timeout 300 python bench.py --config c5 --n 20000000 --expansion 64 --steps 8 --warmup 2 --no-cpu-baseline --no-stress-rows --no-load-timing --no-host-api --no-placement-check --recall-queries 2000 > $OUT/c5_$cap.json 2> $OUT/c5_$cap.log
python - $OUT/c5_$cap.json $cap <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
    print("c5 next_cap", sys.argv[2], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "waves", c.get("persistent_waves"), "lds", c.get("lds_bytes_per_wave"), "passes", c.get("kernel_passes"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
