#!/bin/bash
# round 5, session 23: `seen` cells in LDS in front of the short-row walks' global visited set (fewer memory-side atomics): 20M slices of
# C5 and C4 with 0 / auto / 512 / 1024 / 2048 cells, one process each; then the parity tests the short rows touch
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s23; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "c5 64" "c4 80"; do set -- $cfg
for cells in 0 auto 512 1024 2048 0 auto; do
if [ $cells = auto ]; then unset USEARCH_AMD_SEEN_CELLS; else export USEARCH_AMD_SEEN_CELLS=$cells; fi
timeout 300 python bench.py --config $1 --n 20000000 --expansion $2 --steps 8 --warmup 2 --no-cpu-baseline --no-stress-rows --no-load-timing --no-host-api --no-placement-check --recall-queries 2000 > $OUT/$1_$cells.json 2> $OUT/$1_$cells.log
python - $OUT/$1_$cells.json $1 $cells <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "seen cells", sys.argv[3], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "recall", d["config"].get("recall_at_k"), "waves", d["config"].get("persistent_waves"), "lds", d["config"].get("lds_bytes_per_wave"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
done; done
unset USEARCH_AMD_SEEN_CELLS
timeout 900 python -m pytest tests/test_gpu_search_parity.py tests/test_gpu_fullsize.py tests/test_gpu_build.py tests/test_gpu_filtered.py -q -x > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log | cut -c1-200
