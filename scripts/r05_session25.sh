#!/bin/bash
# round 5, session 25 (final sources): the whole GPU suite, smoke(), then the short-row configurations at full size with the `seen` cells
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final/configs
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r05_final/pytest_gpu.log 2>&1; tail -2 gpurun_out/r05_final/pytest_gpu.log | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for c in c5 c4; do
timeout 600 python bench.py --config $c --steps 10 --warmup 2 --no-stress-rows --no-load-timing > gpurun_out/r05_final/configs/${c}_seen.json 2> gpurun_out/r05_final/configs/${c}_seen.log
python - gpurun_out/r05_final/configs/${c}_seen.json $c <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[2], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "frac", round(r["frac"], 4), "lines", round(r.get("lines_touched_frac") or 0, 4), "ef", d["config"]["expansion_search"], "recall", d["config"].get("recall_at_k"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), "sources", d["config"]["sources"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
