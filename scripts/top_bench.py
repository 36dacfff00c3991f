#!/usr/bin/env python3
"""scripts/top_bench.py — cycles per insert of the register-resident result buffer in isolation (diagnostic)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import usearch_amd  # noqa: E402

import usearch_amd.index
L = usearch_amd.index.test_hooks()  # the containers' micro-benchmarks live in their own library (csrc/test_hooks.hip)
L.usearch_amd_bench_top.argtypes = [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p)]
for epl, limit in ((4, 256), (8, 512), (16, 640), (16, 1024)):
    for waves in (1, 256, 2048, 4096):
        ticks = np.zeros(waves, dtype=np.uint64)
        accepted = np.zeros(waves, dtype=np.uint64)
        err = C.c_char_p()
        count = 20000
        L.usearch_amd_bench_top(epl, count, limit, waves, ticks.ctypes.data, accepted.ctypes.data, C.byref(err))
        assert not err.value, err.value
        print(f"rows={epl:2d} limit={limit:4d} waves={waves:5d}: {ticks.mean() / accepted.mean():8.1f} ticks per accepted insert "
              f"({accepted.mean():.0f} of {count} accepted, {ticks.mean() / count:.1f} ticks per candidate)", flush=True)
