#!/usr/bin/env python3
"""Average duration of the TIMED launches in a `rocprofv3 --kernel-trace` CSV.

`--stats` averages every launch of a kernel over the whole process; since the loader draws placements (bench.py
--placement-draws) the timed instantiation also runs on the candidates that lose, which walk at other speeds. The timed region is
the last steps + warmup launches of `roofline.kernel_instantiation` (profiled runs skip everything bench.py does after it), so
this prints their durations and average next to the whole-process figure.

    python scripts/trace_timed.py <kernel_trace.csv> <bench line .json>  >  kernel_trace_timed.json
"""
import csv
import json
import sys


def main(trace_path: str, line_path: str) -> None:
    line = json.loads(open(line_path).read().strip().splitlines()[-1])
    instantiation = line["roofline"]["kernel_instantiation"]
    wanted = line["steps"]  # the timed steps alone: placement trials run among the warm-up launches (see pmc_traffic.py)
    with open(trace_path, newline="") as handle:
        rows = list(csv.DictReader(handle))
    name_column = next(c for c in rows[0] if c.lower() == "kernel_name")
    begin = next(c for c in rows[0] if c.lower() == "start_timestamp")
    end = next(c for c in rows[0] if c.lower() == "end_timestamp")
    order = next((c for c in rows[0] if c.lower() == "dispatch_id"), begin)
    matching = sorted((row for row in rows if instantiation + "(" in row[name_column]), key=lambda row: int(row[order]))
    durations = [(int(row[end]) - int(row[begin])) / 1e6 for row in matching]
    timed = durations[-wanted:]
    print(json.dumps({"kernel_instantiation": instantiation, "launches_in_process": len(durations),
                      "average_ms_all_launches": sum(durations) / max(1, len(durations)),
                      "timed_launches": len(timed), "timed_durations_ms": [round(d, 3) for d in timed],
                      "average_ms_timed": sum(timed) / max(1, len(timed)),
                      "hip_event_kernel_ms_of_the_same_process": line["roofline"]["kernel_ms"]}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
