#!/usr/bin/env python3
"""scripts/heap_bench.py — ticks per pop / push of the frontier heap in isolation: the reference-shaped pop (one level per LDS
round trip) against the five-levels-per-round-trip pop the traversal uses (diagnostic)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import usearch_amd  # noqa: E402

import usearch_amd.index
L = usearch_amd.index.test_hooks()  # the containers' micro-benchmarks live in their own library (csrc/test_hooks.hip)
L.usearch_amd_bench_heap.argtypes = [C.c_uint32] * 4 + [C.c_void_p] * 3 + [C.POINTER(C.c_char_p)]
for fill in (200, 1000, 2000):
    for waves in (1, 2048):
        sums = {}
        for serial in (1, 0):
            pops, pushes, sums[serial] = (np.zeros(waves, dtype=np.uint64) for _ in range(3))
            err = C.c_char_p()
            count = 2000
            L.usearch_amd_bench_heap(serial, fill, count, waves, pops.ctypes.data, pushes.ctypes.data,
                                     sums[serial].ctypes.data, C.byref(err))
            assert not err.value, err.value
            print(f"heap of {fill:5d}, {waves:5d} waves, {'serial  ' if serial else 'subtree '} pop: {pops.mean() / count:8.1f} ticks per pop, "
                  f"{pushes.mean() / count:8.1f} per push", flush=True)
        assert np.array_equal(sums[0], sums[1]), "the two pops disagree"
