#!/usr/bin/env python3
"""scripts/latency_check.py [--vectors N --dim D --dtype T --ef E …]: one query at a time through the host API (what a `usearch_search`
loop sees) and small batches, with the team build (five waves per query) and without it (USEARCH_AMD_NO_TEAM=1) — same index, same
process; results compared key for key, distance bits and counters included."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--vectors", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--dtype", default="f16")
    p.add_argument("--ef", type=int, nargs="+", default=[608, 64])
    p.add_argument("--batches", type=int, nargs="+", default=[1, 16, 256])
    args = p.parse_args()
    import torch

    import usearch_amd
    metric = "l2sq" if args.dtype == "i8" else "cos"
    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(args.vectors, args.dim, args.dtype, 42, device)
    built = usearch_amd.build(None, metric, args.dtype, device_pointer=data.data_ptr(), count=args.vectors, stride=data.stride(0),
                              ndim=args.dim)
    del data
    torch.cuda.empty_cache()
    index = built.index
    queries = bench.synthetic_vectors_device(512, args.dim, args.dtype, 43, device).cpu().numpy().view(bench.NUMPY_STORAGE[args.dtype])
    for ef in args.ef:
        index.expansion_search = ef
        for batch in args.batches:
            row = {}
            for label, off in (("team", "0"), ("one wave", "1")):
                os.environ["USEARCH_AMD_NO_TEAM"] = off
                rounds = max(4, 64 // batch)
                for i in range(2):
                    index.search(queries[:batch], 10, dtype=args.dtype)
                t0 = time.perf_counter()
                for i in range(rounds):
                    got = index.search(queries[i * batch % 256:i * batch % 256 + batch], 10, dtype=args.dtype)
                seconds = (time.perf_counter() - t0) / rounds
                check = index.search(queries[:batch], 10, dtype=args.dtype)
                row[label] = (seconds, check)
            a, b = row["team"][1], row["one wave"][1]
            same = (np.array_equal(a.keys, b.keys) and np.array_equal(a.distances.view(np.uint32), b.distances.view(np.uint32))
                    and np.array_equal(a.visited_per_query, b.visited_per_query)
                    and np.array_equal(a.computed_per_query, b.computed_per_query))
            print(f"ef={ef} batch={batch}: team {row['team'][0] * 1e3:.3f} ms per call (kernel build {a.stats.variant} / {b.stats.variant}, scratch mode {a.stats.mode} / {b.stats.mode}), "
                  f"one wave per query {row['one wave'][0] * 1e3:.3f} ms; identical results: {same}", flush=True)


if __name__ == "__main__":
    main()
