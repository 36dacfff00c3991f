#!/bin/bash
# round 5, session 27 (the last GPU seconds of the round): load-first probing of the short rows' visited set, parity and time
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_s27
timeout 100 python scripts/load_first_check.py > gpurun_out/r05_s27/check.log 2>&1; grep -E "identical|Error|error" gpurun_out/r05_s27/check.log | cut -c1-300
for sw in 0 1; do
USEARCH_AMD_PROBE_LOAD_FIRST=$sw timeout 60 python bench.py --config c5 --n 20000000 --expansion 64 --steps 8 --warmup 2 --no-cpu-baseline --no-stress-rows --no-load-timing --no-host-api --no-placement-check --recall-queries 2000 > gpurun_out/r05_s27/c5_$sw.json 2> gpurun_out/r05_s27/c5_$sw.log
python - gpurun_out/r05_s27/c5_$sw.json $sw <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("c5 slice, load first", sys.argv[2], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "recall", d["config"].get("recall_at_k"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
