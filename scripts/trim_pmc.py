#!/usr/bin/env python3
"""scripts/trim_pmc.py <pmc csv> <out csv> [launches] [instantiation]: keeps, of a rocprofv3 counter_collection CSV, the rows of the last
`launches` (default 12) dispatches of the widest grid of `instantiation` (a substring of the kernel name, e.g.
"search_kernel<99, 12, 8, 3, 1, 16, 1>"; default: any search_kernel) — the timed launches of bench.py, which is what
scripts/pmc_traffic.py averages — so the evidence committed under profiles/ stays small. Everything else in the file is
the build's insertion searches, the placement self-searches and warm-up."""
import csv
import sys


def main():
    source, target = sys.argv[1], sys.argv[2]
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    wanted = sys.argv[4] if len(sys.argv) > 4 else "search_kernel"
    with open(source, newline="") as f:
        rows = list(csv.reader(f))
    header, rows = rows[0], rows[1:]
    grid, dispatch, name = header.index("Grid_Size"), header.index("Dispatch_Id"), header.index("Kernel_Name")
    rows = [r for r in rows if wanted in r[name]]
    widest = max(int(r[grid]) for r in rows)
    kept_ids = sorted({int(r[dispatch]) for r in rows if int(r[grid]) == widest})[-launches:]
    kept = [r for r in rows if int(r[dispatch]) in set(kept_ids)]
    with open(target, "w", newline="") as f:
        writer = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        writer.writerow(header)
        writer.writerows(kept)
    print(f"{source}: {len(rows)} rows -> {len(kept)} rows ({len(kept_ids)} dispatches of grid {widest})")


if __name__ == "__main__":
    main()
