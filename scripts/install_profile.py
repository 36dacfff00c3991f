#!/usr/bin/env python3
"""Copies one profile session (gpurun_out/<tag>/<workload>/, written by scripts/profile_round.sh) into profiles/<name>/<workload>/:
the bench line, the logs, the kernel-trace summary and traffic.json as they are, the PMC CSVs cut down to the timed dispatches
(the last steps + warmup launches of `roofline.kernel_instantiation` of the bench line) — the rest of those files is the index build.

    python scripts/install_profile.py gpurun_out/r02_final4 profiles/r02_final
"""
import csv
import json
import os
import shutil
import sys


def main(source: str, target: str) -> None:
    for workload in sorted(os.listdir(source)):
        directory = os.path.join(source, workload)
        bench_path = os.path.join(directory, "bench.json")
        if not os.path.isfile(bench_path):
            continue
        line = json.loads(open(bench_path).read().strip().splitlines()[-1])
        instantiation = line["roofline"]["kernel_instantiation"]
        out = os.path.join(target, workload)
        os.makedirs(out, exist_ok=True)
        for name in ("bench.json", "bench.log", "pick.log", "kernel_stats.csv", "kernel_trace_timed.json", "traffic.json"):
            if os.path.isfile(os.path.join(directory, name)):
                shutil.copyfile(os.path.join(directory, name), os.path.join(out, name))
        # the bench line of the very process the kernel trace was taken from: its HIP-event time is the one to hold against
        # kernel_stats.csv (bench.json is a later process on the same box — another placement of the arrays)
        if os.path.isfile(os.path.join(directory, "stats_bench.json")):
            shutil.copyfile(os.path.join(directory, "stats_bench.json"), os.path.join(out, "kernel_trace_bench.json"))
        for name in sorted(os.listdir(directory)):
            if not (name.startswith("pmc_") and name.endswith(".csv")):
                continue
            with open(os.path.join(directory, name), newline="") as handle:
                rows = list(csv.reader(handle))
            matching = [row for row in rows[1:] if instantiation + "(" in ",".join(row)]
            # the builder's insertion searches may be the same instantiation (short rows): the timed launches are the last
            # steps + warmup dispatches, one row per counter each — what scripts/pmc_traffic.py averages
            counters = len({row[rows[0].index("Counter_Name")] for row in matching}) or 1
            kept = [rows[0]] + matching[-(line["steps"] + line["warmup"]) * counters:]
            with open(os.path.join(out, name), "w", newline="") as handle:
                csv.writer(handle, quoting=csv.QUOTE_MINIMAL).writerows(kept)
            print(f"{workload}/{name}: {len(kept) - 1} of {len(rows) - 1} rows ({instantiation})")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
