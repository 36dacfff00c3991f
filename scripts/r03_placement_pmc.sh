#!/bin/bash
# scripts/r03_placement_pmc.sh <tag> — slow and fast placements of the same index in ONE process under translation counters:
# does the address-translation path see them differently?
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PASS=0
for counters in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
                "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_CLIENT_UTCL1_INFLIGHT_sum" \
                "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_HIT_sum TCC_MISS_sum"; do
  PASS=$((PASS + 1))
  timeout -s KILL 600 rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmc$PASS" -- python "$REPO/scripts/placement_study.py" --copies 6 --parts 1 \
      --no-engine-draws > "$OUT/pmc$PASS.log" 2>&1 || echo "pass $PASS failed"
  grep "^ *[0-9]  " "$OUT/pmc$PASS.log"
  find "$OUT/pmc$PASS" -name "*counter_collection.csv" | head -1 | while read f; do head -1 "$f" > "$OUT/pmc$PASS.csv"; grep "search_kernel<99" "$f" >> "$OUT/pmc$PASS.csv"; done
  rm -rf "$OUT/pmc$PASS"
  python - "$OUT/pmc$PASS.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
big = [(d, c) for d, c in by.items()]
print(f"{len(big)} dispatches of the headline kernel")
for d, c in big[-24:]:
    print(d, " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
PY
done
echo "=== done $(date +%T)"
