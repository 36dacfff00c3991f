#!/usr/bin/env python3
"""scripts/traversal_shape.py — what the reference's own traversal looks like at the headline's expansion, measured with the
CPU oracle on an index the compiled reference builds here (design study for the device kernel; CPU only):
how often the next hop is a node found during the hop before, how many candidates a hop admits, how much of the frontier heap
is dead weight (entries farther than the radius are never popped, the reference keeps them).

    python scripts/traversal_shape.py [--n 200000 --dim 768 --dtype f16 --ef 592 --queries 200]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oraclebind, refbind  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=200_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--dtype", default="f16")
    p.add_argument("--ef", type=int, nargs="+", default=[64, 256, 592])
    p.add_argument("--queries", type=int, default=200)
    args = p.parse_args()
    metric = "hamming" if args.dtype == "b1" else "l2sq" if args.dtype == "i8" else "cos"
    vectors = bench.synthetic_vectors(args.n, args.dim, args.dtype, seed=42)
    queries = bench.synthetic_vectors(args.queries, args.dim, args.dtype, seed=43)
    t0 = time.time()
    reference = refbind.RefIndex(args.dim, metric, args.dtype, 16, 128, 64)
    reference.add(np.arange(args.n, dtype=np.uint64), vectors, threads=bench.host_cores())
    print(f"reference built {args.n} x {args.dim} {args.dtype} in {time.time() - t0:.0f}s", flush=True)
    index = oraclebind.OracleIndex(reference.save_buffer())
    library = oraclebind.lib()
    library.uo_last_traversal_shape.argtypes = [C.c_void_p]
    keys, distances = np.zeros(10, dtype=np.uint64), np.zeros(10, dtype=np.float32)
    for ef in args.ef:
        total = np.zeros(8)
        peak = 0
        for q in queries:
            library.uo_search(C.byref(index.ix), q.ctypes.data, oraclebind.SCALAR[args.dtype], 10, ef, 0, 0, None, None,
                              keys.ctypes.data, distances.ctypes.data, None, None)
            shape = np.zeros(8)
            library.uo_last_traversal_shape(shape.ctypes.data)
            total += shape
            peak = max(peak, library.uo_last_peak_next())
        hops = total[0]
        print(f"ef={ef:4d}: {hops / len(queries):6.0f} hops/query, {total[3] / hops:5.1f} fresh neighbours and {total[2] / hops:4.1f} "
              f"admitted per hop; next hop is a newcomer of the hop before in {100 * total[1] / hops:4.1f} % of the hops; frontier: "
              f"{total[6] / hops:6.0f} entries on average (peak {peak}), {100 * total[7] / (total[6] / 16):4.1f} % of them still poppable; "
              f"at the end {total[4] / len(queries):6.0f} entries, {100 * total[5] / max(total[4], 1):4.1f} % poppable", flush=True)


if __name__ == "__main__":
    main()
