#!/bin/bash
# round 5, session 15: (1) the wide exact tile with the two waves of a SIMD filling at different places of a chunk; (2) the placement
# trials reopened by a wider regime: C2 and the headline, driver style; (3) one query at a time with the team's helpers at priority 2
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s15; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_exact.py -q -x > $OUT/pytest_exact.log 2>&1; tail -2 $OUT/pytest_exact.log | cut -c1-200
timeout 300 python scripts/exact_knockout.py --combos 0,1,3,0 --repeats 3 > $OUT/knockout.log 2>&1; grep knockout $OUT/knockout.log
for run in c2 headline; do
  extra=""; [ $run = c2 ] && extra="--config c2"
  USEARCH_AMD_PLACEMENT_LOG=1 timeout 600 python bench.py $extra --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stress-rows > $OUT/$run.json 2> $OUT/$run.log
  grep -E "placement trial" $OUT/$run.log | cut -c1-220
  python - $OUT/$run.json $run <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "first placement", r.get("frac_first_placement"), d["config"]["placement"]["matrix"])
PY
done
timeout 400 python scripts/latency_check.py --ef 608 --batches 1 16 > $OUT/latency_default.log 2>&1; grep "ef=" $OUT/latency_default.log
USEARCH_AMD_LIBRARY=$PWD/usearch_amd/lib_prio/libusearch_amd.so timeout 400 python scripts/latency_check.py --ef 608 --batches 1 16 > $OUT/latency_prio.log 2>&1; grep "ef=" $OUT/latency_prio.log
