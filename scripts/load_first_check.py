#!/usr/bin/env python3
"""scripts/load_first_check.py — the short rows' visited set probed by a load first (USEARCH_AMD_PROBE_LOAD_FIRST=1, kernels.hpp
`load_first`) against probing by compare-and-swap alone: same index, same batch, global-hash mode forced, keys / distance bits /
counts / both traversal counters compared query by query; then the time of each on a larger batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main() -> None:
    import torch

    import bench
    import usearch_amd
    from usearch_amd import Tuning
    device = torch.device("cuda", 0)
    for dtype, metric, dim, n, expansion in (("b1", "hamming", 128, 2_000_000, 64), ("i8", "l2sq", 96, 2_000_000, 80)):
        data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
        built = usearch_amd.build(None, metric, dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
        index = built.index
        queries = bench.synthetic_vectors_device(20_000, dim, dtype, 43, device).cpu().numpy().view(bench.NUMPY_STORAGE[dtype])
        answers = {}
        for switch in ("0", "1"):
            os.environ["USEARCH_AMD_PROBE_LOAD_FIRST"] = switch
            index.search(queries, 10, expansion=expansion, dtype=dtype, tuning=Tuning(mode=2))
            t0 = time.perf_counter()
            got = index.search(queries, 10, expansion=expansion, dtype=dtype, tuning=Tuning(mode=2))
            answers[switch] = (got, got.stats.kernel_ms, time.perf_counter() - t0)
        a, b = answers["0"][0], answers["1"][0]
        same = (np.array_equal(a.keys, b.keys) and np.array_equal(a.distances.view(np.uint32), b.distances.view(np.uint32))
                and np.array_equal(a.counts, b.counts) and np.array_equal(a.visited_per_query, b.visited_per_query)
                and np.array_equal(a.computed_per_query, b.computed_per_query))
        print(f"{n}x{dim} {dtype} {metric}, 20000 queries, ef {expansion}, mode {a.stats.mode}/{b.stats.mode}: compare-and-swap alone "
              f"{answers['0'][1]:.3f} ms, load first {answers['1'][1]:.3f} ms; identical keys, bits, counts and both counters: {same}", flush=True)
        del index, built, data
        torch.cuda.empty_cache()
    os.environ["USEARCH_AMD_PROBE_LOAD_FIRST"] = "0"


if __name__ == "__main__":
    main()
