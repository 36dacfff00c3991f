#!/bin/bash
# round 5, session 12: the headline's evidence from the final walk sources — kernel trace, the PMC passes roofline.traffic needs, the
# bench line (scripts/profile_round.sh) — and a driver-style default run
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PROFILE_EF=608 PROFILE_TRAFFIC_ONLY=1 timeout 1500 bash scripts/profile_round.sh r05_final/headline > gpurun_out/r05_s12_profile.log 2>&1
tail -5 gpurun_out/r05_s12_profile.log | cut -c1-600
USEARCH_AMD_PLACEMENT_LOG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_final/driver_style.json 2> gpurun_out/r05_final/driver_style.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final/driver_style.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("driver-style: value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "first placement", r.get("frac_first_placement"), d["config"]["placement"]["matrix"])
PY
