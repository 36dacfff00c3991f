#!/usr/bin/env python3
"""scripts/pmc_per_dispatch.py <counter_collection.csv> [kernel substring]: one line per dispatch of the widest grid of that kernel, in
dispatch order, with every counter of the pass — for A/B comparisons between launches of one process (scripts/fragment_study.py)."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if (sys.argv[2] if len(sys.argv) > 2 else "search_kernel") in r.get("Kernel_Name", "")]
if not rows:
    sys.exit("no such kernel in the file")
widest = max(int(r["Grid_Size"]) for r in rows)
table = collections.OrderedDict()
for r in rows:
    if int(r["Grid_Size"]) == widest:
        table.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({n for values in table.values() for n in values})
print("dispatch " + " ".join(f"{n:>28s}" for n in names))
for dispatch in sorted(table):
    print(f"{dispatch:8d} " + " ".join(f"{table[dispatch].get(n, float('nan')):28.0f}" for n in names))
