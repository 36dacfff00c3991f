#!/bin/bash
# round 5, session 4: the default bench line (as the driver runs it) twice on the online placement trials, then the GPU suite
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s4; mkdir -p $OUT
export TMPDIR=/tmp
for run in 1 2; do
  USEARCH_AMD_PLACEMENT_LOG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_$run.json 2> $OUT/bench_$run.log
  grep -E "placement|trial|with the placement" $OUT/bench_$run.log | grep -v "MB @" | tail -30
  python - $OUT/bench_$run.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "first", r["frac_first_placement"], "after", r.get("kernel_ms_after_first_placement_check"), d["config"]["placement"]["matrix"])
PY
done
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
