#!/bin/bash
# round 5, session 26: the default bench line as the driver runs it, on the final sources (after the `seen` cells went in: the headline's
# instantiation — 8 lanes per row — does not contain them)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final
USEARCH_AMD_PLACEMENT_LOG=1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_final/driver_style_final_sources.json 2> gpurun_out/r05_final/driver_style_final_sources.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final/driver_style_final_sources.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("driver-style: value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "first placement", r.get("frac_first_placement"), d["config"]["placement"]["matrix"]["draws"], d["config"]["placement"]["matrix"]["kept"], d["config"]["sources"], r["kernel_instantiation"])
PY
