#!/usr/bin/env python3
"""scripts/probe_mode_check.py — the three ways a short-row walk probes its wave-private visited-set slab (common.hpp `probe_mode_t`,
USEARCH_AMD_PROBE_MODE = 0 compare-and-swap | 1 a load first, the swap to claim | 2 no atomic: loads, plain stores, claims settled
by bits in LDS) on the same index and the same batch: keys / distance bits / counts / both traversal counters compared query by
query against mode 0, then the kernel time of each on a 100 000-query batch.

    python scripts/probe_mode_check.py [--vectors 2000000] [--queries 20000] [--timed-queries 100000]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--vectors", type=int, default=2_000_000)
    parser.add_argument("--queries", type=int, default=20_000)
    parser.add_argument("--timed-queries", type=int, default=100_000)
    parser.add_argument("--modes", type=int, nargs="+", default=[0, 1, 2])
    parser.add_argument("--early", type=int, nargs="+", default=[0], help="USEARCH_AMD_EARLY_ROWS values to cross the modes with (G = 2 rows)")
    args = parser.parse_args()
    import torch

    import bench
    import usearch_amd
    from usearch_amd import Tuning
    device = torch.device("cuda", 0)
    n = args.vectors
    for dtype, metric, dim, expansion in (("b1", "hamming", 128, 64), ("i8", "l2sq", 96, 80), ("i8", "l2sq", 96, 64)):
        data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
        built = usearch_amd.build(None, metric, dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
        index = built.index
        count = max(args.queries, args.timed_queries)
        queries_dev = bench.synthetic_vectors_device(count, dim, dtype, 43, device)
        queries = queries_dev.cpu().numpy().view(bench.NUMPY_STORAGE[dtype])
        outs = [torch.zeros((count, 10), dtype=torch.int64, device=device), torch.zeros((count, 10), dtype=torch.float32, device=device)] + \
               [torch.zeros(count, dtype=torch.int64, device=device) for _ in range(3)]
        answers = {}
        combos = [(mode, early) for mode in args.modes for early in (args.early if dtype == "i8" else args.early[:1])]
        for mode, early in combos:
            os.environ["USEARCH_AMD_PROBE_MODE"] = str(mode)
            os.environ["USEARCH_AMD_EARLY_ROWS"] = str(early)
            got = index.search(queries[:args.queries], 10, expansion=expansion, dtype=dtype, tuning=Tuning(mode=2))
            times = []
            for _ in range(5):
                stats = index.search_device(queries_dev.data_ptr(), args.timed_queries, queries_dev.stride(0), 10, expansion, outs[0].data_ptr(),
                                            outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(), timed=True,
                                            tuning=Tuning(mode=2))
                times.append(stats.kernel_ms)
            answers[(mode, early)] = (got, float(np.min(times[1:])), stats)
        base = answers[combos[0]][0]
        for mode, early in combos:
            got, ms, stats = answers[(mode, early)]
            same = (np.array_equal(base.keys, got.keys) and np.array_equal(base.distances.view(np.uint32), got.distances.view(np.uint32))
                    and np.array_equal(base.counts, got.counts) and np.array_equal(base.visited_per_query, got.visited_per_query)
                    and np.array_equal(base.computed_per_query, got.computed_per_query))
            print(f"{n}x{dim} {dtype} {metric} ef {expansion}: probe mode {mode} early rows {early} (ran {stats.probe_mode} / {stats.early_rows}, scratch mode {stats.mode}, {stats.grid} waves, "
                  f"{stats.lds_bytes} B LDS/wave, seen {stats.seen_cells}, claim bits {stats.claim_bits}): {ms:.3f} ms for {args.timed_queries} queries = "
                  f"{args.timed_queries / ms / 1e3:.2f} M QPS; identical to mode {args.modes[0]} on {args.queries} queries (keys, bits, counts, both counters): {same}",
                  flush=True)
        del index, built, data
        torch.cuda.empty_cache()
    os.environ.pop("USEARCH_AMD_PROBE_MODE", None)
    os.environ.pop("USEARCH_AMD_EARLY_ROWS", None)


if __name__ == "__main__":
    main()
