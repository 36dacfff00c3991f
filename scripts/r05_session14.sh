#!/bin/bash
# round 5, session 14: the other BASELINE configurations on the final walk sources (bench lines)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_final/configs; mkdir -p $OUT
export TMPDIR=/tmp
for c in c1 c2 c4 c5; do
timeout 900 python bench.py --config $c --steps 10 --warmup 2 > $OUT/$c.json 2> $OUT/$c.log
python - $OUT/$c.json $c <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[2], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "frac", round(r["frac"], 4), "lines", round(r.get("lines_touched_frac") or 0, 4), "ef", d["config"]["expansion_search"], "recall", d["config"].get("recall_at_k"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
