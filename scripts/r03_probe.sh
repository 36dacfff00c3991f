#!/bin/bash
# scripts/r03_probe.sh <tag> — round 3, first GPU session: facts about the short-row walk before it is rebuilt.
#   1. WRITE_SIZE / FETCH_SIZE calibration on known patterns (scripts/probes/write_calib.hip)
#   2. 20M x 128 b1 (ef 64) and 20M x 96 i8 (ef 80): timing of the global-slab and LDS visited sets, then SQ counter passes
#      (instructions issued per hop: is the walk issue-bound?)
set -u
TAG=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

echo "=== calibration $(date +%T)"
"$REPO/scripts/probes/_bin/write_calib" > "$OUT/calib_plain.log" 2>&1; tail -4 "$OUT/calib_plain.log"
for counter in WRITE_SIZE FETCH_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $counter | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $counter --output-format csv -d "$OUT/calib_$name" -- "$REPO/scripts/probes/_bin/write_calib" > /dev/null 2> "$OUT/calib_$name.log" || echo "calib $counter failed"
  find "$OUT/calib_$name" -name "*counter_collection.csv" | head -1 | while read f; do cp "$f" "$OUT/calib_$name.csv"; done
  rm -rf "$OUT/calib_$name"
done
python - "$OUT" <<'EOF'
import csv, glob, sys, collections
for path in sorted(glob.glob(sys.argv[1] + "/calib_*.csv")):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        key = (r.get("Kernel_Name", "?")[:40], r.get("Counter_Name", "?"))
        agg.setdefault(key, []).append(float(r.get("Counter_Value", 0)))
    print(path.split("/")[-1])
    for key, values in agg.items():
        print("   ", key, [round(v) for v in values])
EOF

shape() { # <name> <dim> <dtype> <ef>
  echo "=== $1 timing $(date +%T)"
  timeout 600 python "$REPO/scripts/sweep.py" --n 20000000 --dim $2 --dtype $3 --ef $4 --queries 100000 --modes 2 1 --steps 2 \
      --env "" USEARCH_AMD_HASH_CAP=4096 > "$OUT/$1_timing.log" 2>&1
  grep "^ef=\|^---\|GPU-built" "$OUT/$1_timing.log"
  PASS=0
  for counters in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
                  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
                  "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
    PASS=$((PASS + 1))
    echo "--- $1 pmc pass $PASS"
    timeout 600 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$1_pmc$PASS" -- python "$REPO/scripts/sweep.py" --n 20000000 --dim $2 \
        --dtype $3 --ef $4 --queries 100000 --modes 2 --steps 1 > "$OUT/$1_pmc$PASS.log" 2>&1 || echo "pmc pass $PASS failed"
    find "$OUT/$1_pmc$PASS" -name "*counter_collection.csv" | head -1 | while read f; do head -1 "$f" > "$OUT/$1_pmc$PASS.csv"; grep search_kernel "$f" >> "$OUT/$1_pmc$PASS.csv"; done
    rm -rf "$OUT/$1_pmc$PASS"
    grep "^ef=" "$OUT/$1_pmc$PASS.log" | tail -1
  done
  python - "$OUT" "$1" <<'EOF'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
for path in sorted(glob.glob(f"{out}/{name}_pmc*.csv")):
    rows = list(csv.DictReader(open(path)))
    # the timed search launches are the LAST dispatches of the search kernel with the big grid; print the last dispatch's counters
    by_dispatch = collections.OrderedDict()
    for r in rows:
        by_dispatch.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        by_dispatch[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "?")
        by_dispatch[r["Dispatch_Id"]]["_kernel"] = r.get("Kernel_Name", "?")[:60]
    last = list(by_dispatch.items())[-2:]
    for dispatch, counters in last:
        print(path.split("/")[-1], dispatch, counters)
EOF
}
shape b1 128 b1 64
shape i8 96 i8 80
echo "=== done $(date +%T)"
