#!/usr/bin/env python3
"""scripts/pmc_traffic.py <FETCH_SIZE csv> <WRITE_SIZE csv> [kernel substring] [bench line json] [RDREQ-by-size csv] [WRREQ / ATOMIC csv] → JSON with the HBM bytes per launch of the
search kernel, from rocprofv3 PMC passes collected SEPARATELY (TCC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass),
corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so wide (16 B/lane) reads are doubled; WRITE_SIZE is taken as is (uncalibrated). With a bench
line (the JSON bench.py printed in one of the profiled runs) the result also names the workload and the sources it was measured
on: bench.py attaches a `roofline.traffic` only to lines of that very workload built from those very sources.

Round 3 calibrated the two counters on the kernels' own patterns (profiles/r03_short_rows/README.md §1): WRITE_SIZE is exact for stores
but ALSO counts 64 bytes for every global atomic (atomics are executed at the memory side: one write request each), and FETCH_SIZE counts
64 bytes for a scattered 16-byte read. With the request-size passes (TCC_EA0_RDREQ_{32B,64B,128B}, TCC_EA0_WRREQ_{64B}, TCC_EA0_ATOMIC)
the bytes are counted instead of corrected: read = 32·n32 + 64·n64 + 128·n128 (+ 64 for requests of none of the three tallies), written =
64·w64 + 32·(w − w64) for the requests that are not atomics, and the atomics are reported on their own (`atomic_requests_per_launch`;
each moves 4 bytes of payload in a 64-byte request). `hbm_bytes_per_launch` then = read + written + 64 × atomics."""
import csv
import json
import sys


def mean_counter(path, name, kernel, last=0):
    """Mean per launch of `name` over the TIMED launches of `kernel`. With `last` (= warm-up + timed steps of the profiled
    bench run, which does nothing else after the index build: `--recall-queries 0 --no-host-api`) those are the last `last`
    dispatches of that instantiation — the index build of the same process may run the very same instantiation at the very same
    grid size (100M x 96 i8: expansion 80 and the builder's 128 share one build), so neither the name nor the grid tells them
    apart; dispatch order does. Without `last`: the (instantiation, grid size) group with the largest mean."""
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name and kernel in row["Kernel_Name"]:
                rows.append(row)
    if not rows:
        return None, 0, None
    if last:
        rows.sort(key=lambda row: int(row["Dispatch_Id"]))
        chosen = rows[-last:]
        values = [float(row["Counter_Value"]) for row in chosen]
        milliseconds = [(int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6 for row in chosen]
        label = (chosen[-1]["Kernel_Name"].split("(")[0] + f", last {len(chosen)} dispatches, grid {chosen[-1].get('Grid_Size', '')}, "
                 f"{sum(milliseconds) / len(milliseconds):.3f} ms each under the counter pass")
        return sum(values) / len(values), len(values), label
    groups = {}
    for row in rows:
        groups.setdefault((row["Kernel_Name"], row.get("Grid_Size", "")), []).append(float(row["Counter_Value"]))
    best = max(groups, key=lambda k: sum(groups[k]) / len(groups[k]))
    values = groups[best]
    return sum(values) / len(values), len(values), best[0].split("(")[0] + f" grid {best[1]}"


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "search_kernel"
    last = 0
    if len(sys.argv) > 4:  # the bench line names the timed instantiation exactly, and how many launches of it were timed
        try:
            line = json.load(open(sys.argv[4]))
            kernel = line["roofline"].get("kernel_instantiation") or kernel
            # the timed steps alone: the engine's placement trials (judge launches of this very instantiation at this very grid) run
            # among the warm-up launches, never among the timed ones when the warm-up is long enough (PROFILE_WARMUP)
            last = int(line["steps"])
        except (OSError, ValueError, KeyError):
            pass
    fetch_kib, fetch_n, kernel_name = mean_counter(fetch_csv, "FETCH_SIZE", kernel, last)
    write_kib, write_n, _ = mean_counter(write_csv, "WRITE_SIZE", kernel, last)
    fetch_bytes = fetch_kib * 1024 * 2 if fetch_kib is not None else None
    write_bytes = write_kib * 1024 if write_kib is not None else None
    total = (fetch_bytes or 0) + (write_bytes or 0) if fetch_bytes is not None else None
    exact = None
    if len(sys.argv) > 6:
        try:
            reads = {name: mean_counter(sys.argv[5], name, kernel, last)[0] for name in
                     ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}
            writes = {name: mean_counter(sys.argv[6], name, kernel, last)[0] for name in
                      ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_ATOMIC_sum")}
            if all(v is not None for v in reads.values()) and all(v is not None for v in writes.values()):
                other = max(0.0, reads["TCC_EA0_RDREQ_sum"] - reads["TCC_EA0_RDREQ_32B_sum"] - reads["TCC_EA0_RDREQ_64B_sum"]
                            - reads["TCC_EA0_RDREQ_128B_sum"])
                read_bytes = (32 * reads["TCC_EA0_RDREQ_32B_sum"] + 64 * reads["TCC_EA0_RDREQ_64B_sum"]
                              + 128 * reads["TCC_EA0_RDREQ_128B_sum"] + 64 * other)
                atomics = writes["TCC_EA0_ATOMIC_sum"]
                plain = max(0.0, writes["TCC_EA0_WRREQ_sum"] - atomics)
                plain64 = max(0.0, min(plain, writes["TCC_EA0_WRREQ_64B_sum"] - atomics))
                written = 64 * plain64 + 32 * (plain - plain64)
                exact = {"read_bytes_per_launch": read_bytes, "written_bytes_per_launch": written,
                         "atomic_requests_per_launch": atomics, "read_requests": reads, "write_requests": writes}
        except (OSError, KeyError, ValueError):
            exact = None
    if exact:
        total = exact["read_bytes_per_launch"] + exact["written_bytes_per_launch"] + 64 * exact["atomic_requests_per_launch"]
    identity = {}
    if len(sys.argv) > 4:
        try:
            line = json.load(open(sys.argv[4]))
            identity = {"workload": line["config"]["workload"], "sources": line["config"]["sources"]}
        except (OSError, ValueError, KeyError):
            pass
    print(json.dumps({**identity, "hbm_bytes_per_launch": total, "fetch_bytes_per_launch": fetch_bytes,
                      "write_bytes_per_launch": write_bytes, "by_request_size": exact,
                      "launches_averaged": [fetch_n, write_n], "kernel": kernel_name,
                      "correction": ("requests counted by size: 32·n32 + 64·n64 + 128·n128 read, non-atomic writes by size, + 64 B per "
                                     "memory-side atomic request" if exact else
                                     "FETCH_SIZE[KiB] x 1024 x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE[KiB] x 1024 "
                                     "(exact for stores; counts 64 B per global atomic)")}))


if __name__ == "__main__":
    main()
