#!/usr/bin/env python3
"""scripts/plain_check.py — the short-row walk's kernel build cut for plain batches (kernels.hpp `plain_ak`) against the general build
(USEARCH_AMD_NO_PLAIN=1) on the same index and the same batch: keys / distance bits / counts / both traversal counters compared query
by query, then the kernel time of each on a 100 000-query batch (HIP events), alternating the two builds.

    python scripts/plain_check.py [--vectors 20000000] [--queries 20000] [--timed-queries 100000] [--shapes b1 i8]
    --once <0|1>: one build only, a few timed launches and nothing else (the command a rocprofv3 --pmc pass wraps)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SHAPES = {"b1": ("b1", "hamming", 128, 64), "i8": ("i8", "l2sq", 96, 80), "i8ef64": ("i8", "l2sq", 96, 64)}


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--vectors", type=int, default=20_000_000)
    parser.add_argument("--queries", type=int, default=20_000)
    parser.add_argument("--timed-queries", type=int, default=100_000)
    parser.add_argument("--shapes", nargs="+", default=["b1", "i8", "i8ef64"])
    parser.add_argument("--rounds", type=int, default=3)
    parser.add_argument("--once", type=int, default=None)
    args = parser.parse_args()
    import torch

    import bench
    import usearch_amd
    from usearch_amd import Tuning
    device = torch.device("cuda", 0)
    n = args.vectors
    for shape in args.shapes:
        dtype, metric, dim, expansion = SHAPES[shape]
        data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
        built = usearch_amd.build(None, metric, dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
        index = built.index
        count = max(args.queries, args.timed_queries)
        queries_dev = bench.synthetic_vectors_device(count, dim, dtype, 43, device)
        queries = queries_dev.cpu().numpy().view(bench.NUMPY_STORAGE[dtype])
        outs = [torch.zeros((count, 10), dtype=torch.int64, device=device), torch.zeros((count, 10), dtype=torch.float32, device=device)] + \
               [torch.zeros(count, dtype=torch.int64, device=device) for _ in range(3)]

        def timed(repeats):
            times, stats = [], None
            for _ in range(repeats):
                stats = index.search_device(queries_dev.data_ptr(), args.timed_queries, queries_dev.stride(0), 10, expansion, outs[0].data_ptr(),
                                            outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(), timed=True,
                                            tuning=Tuning(mode=2))
                times.append(stats.kernel_ms)
            return times, stats

        if args.once is not None:
            os.environ["USEARCH_AMD_NO_PLAIN"] = "0" if args.once else "1"
            times, stats = timed(6)
            print(f"{n}x{dim} {dtype} ef {expansion}: plain {stats.plain}: {min(times[1:]):.3f} ms", flush=True)
            continue
        answers, best, ran = {}, {0: [], 1: []}, {}
        for plain in (0, 1):
            os.environ["USEARCH_AMD_NO_PLAIN"] = "0" if plain else "1"
            answers[plain] = index.search(queries[:args.queries], 10, expansion=expansion, dtype=dtype, tuning=Tuning(mode=2))
        for _ in range(args.rounds):
            for plain in (0, 1):
                os.environ["USEARCH_AMD_NO_PLAIN"] = "0" if plain else "1"
                times, stats = timed(5)
                best[plain].append(float(np.min(times[1:])))
                ran[plain] = stats
        base, got = answers[0], answers[1]
        same = (np.array_equal(base.keys, got.keys) and np.array_equal(base.distances.view(np.uint32), got.distances.view(np.uint32))
                and np.array_equal(base.counts, got.counts) and np.array_equal(base.visited_per_query, got.visited_per_query)
                and np.array_equal(base.computed_per_query, got.computed_per_query))
        hops = float(got.visited_per_query.mean())
        peaks = index.last_peaks(args.timed_queries)  # of the last timed launch: the frontier's peak size and the visited set's final size, per query
        print(f"{n}x{dim} {dtype} ef {expansion}: frontier peak per query: median {int(np.median(peaks[:, 0]))}, 99.9 % {int(np.percentile(peaks[:, 0], 99.9))}, "
              f"max {int(peaks[:, 0].max())}; visited set: median {int(np.median(peaks[:, 1]))}, max {int(peaks[:, 1].max())}; passes of the last launch {ran[1].passes}",
              flush=True)
        for plain in (0, 1):
            stats = ran[plain]
            print(f"{n}x{dim} {dtype} {metric} ef {expansion}: plain build {'on' if plain else 'off'} (ran plain={stats.plain} / answers plain={answers[plain].stats.plain}, "
                  f"scratch mode {stats.mode}, {stats.grid} waves, {stats.lds_bytes} B LDS/wave, seen {stats.seen_cells}, aside {stats.aside_cells}, early rows {stats.early_rows}): "
                  f"{' / '.join(f'{ms:.3f}' for ms in best[plain])} ms for {args.timed_queries} queries, best {min(best[plain]):.3f} ms = "
                  f"{args.timed_queries / min(best[plain]) / 1e3:.2f} M QPS", flush=True)
        print(f"{n}x{dim} {dtype} ef {expansion}: plain against general {min(best[0]) / min(best[1]):.4f} x; identical on {args.queries} queries "
              f"(keys, bits, counts, both counters): {same}; {hops:.1f} hops per query", flush=True)
        del index, built, data
        torch.cuda.empty_cache()
    os.environ.pop("USEARCH_AMD_NO_PLAIN", None)


if __name__ == "__main__":
    main()
