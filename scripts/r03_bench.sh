#!/bin/bash
# scripts/r03_bench.sh <tag> <name> [bench args…] — one bench.py line into gpurun_out/<tag>/<name>.json (+ .log)
set -u
TAG=$1; NAME=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
USEARCH_AMD_PLACEMENT_LOG=1 timeout -s KILL 1500 python bench.py "$@" > "$OUT/$NAME.json" 2> "$OUT/$NAME.log"; echo "rc=$?"
grep -v "amdgpu.ids\|ef=.*recall" "$OUT/$NAME.log" | tail -25; python - "$OUT/$NAME.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"value {d['value']:.0f} {d['unit']}  ms/step {d['ms_per_step']:.3f}  kernel {r['kernel_ms']:.3f} ms  frac {r['frac']:.4f}  first-placement frac {r.get('frac_first_placement')}  "
      f"lines-touched frac {r.get('lines_touched_frac')}  ef {d['config']['expansion_search']} recall {d['config']['recall_at_k']}")
print("placement", d["config"]["placement"]["matrix"]); print("cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("frontier_check"))
PY
