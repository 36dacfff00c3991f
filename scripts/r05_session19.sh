#!/bin/bash
# round 5, session 19 (final sources): the headline's evidence once more — kernel trace, the PMC passes roofline.traffic needs, the bench
# line (scripts/profile_round.sh) — a driver-style default run, the whole GPU suite, smoke()
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final
PROFILE_EF=608 PROFILE_TRAFFIC_ONLY=1 timeout 1500 bash scripts/profile_round.sh r05_final/headline > gpurun_out/r05_s19_profile.log 2>&1
tail -3 gpurun_out/r05_s19_profile.log | cut -c1-300
USEARCH_AMD_PLACEMENT_LOG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_final/driver_style.json 2> gpurun_out/r05_final/driver_style.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final/driver_style.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("driver-style: value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "first placement", r.get("frac_first_placement"), d["config"]["placement"]["matrix"], d["config"]["sources"])
PY
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05_final/pytest_gpu.log 2>&1; tail -2 gpurun_out/r05_final/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
