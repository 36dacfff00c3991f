#!/usr/bin/env python3
"""scripts/exact_check.py --vectors N --dim D --dtype T [--queries Q] — the tiled (matrix-unit) exact search against the bit-exact
wave-per-query one on the same index: equality of keys / distance bits (i8) or tolerance (f16, bf16), and both kernel times."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--vectors", type=int, nargs="+", default=[2_000_000])
    p.add_argument("--dim", type=int, default=96)
    p.add_argument("--dtype", default="i8")
    p.add_argument("--queries", type=int, default=256)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--tiles", type=int, nargs="+", default=[0], help="USEARCH_AMD_EXACT_TILE settings to time (0 = the default choice)")
    args = p.parse_args()
    import torch

    import usearch_amd
    metric = "l2sq" if args.dtype == "i8" else "cos"
    device = torch.device("cuda", 0)
    for n in args.vectors:
        data = bench.synthetic_vectors_device(n, args.dim, args.dtype, 42, device)
        built = usearch_amd.build(None, metric, args.dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0),
                                  ndim=args.dim, expansion_add=16, connectivity=4)
        del data
        torch.cuda.empty_cache()
        queries = bench.synthetic_vectors_device(args.queries, args.dim, args.dtype, 43, device).cpu().numpy().view(
            bench.NUMPY_STORAGE[args.dtype])
        plain = built.index.search(queries, args.k, dtype=args.dtype, exact=True)
        flops = 2.0 * len(queries) * n * args.dim
        for tile in args.tiles:
            os.environ["USEARCH_AMD_EXACT_TILE"] = str(tile)
            for _ in range(2):
                tiled = built.index.search(queries, args.k, dtype=args.dtype, exact="tiled")
            same_keys = float((plain.keys == tiled.keys).mean())
            same_bits = float((plain.distances.view(np.uint32) == tiled.distances.view(np.uint32)).mean())
            worst = float(np.nanmax(np.abs(plain.distances - tiled.distances)))
            print(f"n={n} {args.dtype}x{args.dim}, {len(queries)} queries, tile setting {tile}: {tiled.stats.kernel_ms:.1f} ms = "
                  f"{flops / tiled.stats.kernel_ms / 1e9:.0f} T(FL)OP/s; against the wave-per-query kernel ({plain.stats.kernel_ms:.1f} ms): "
                  f"keys equal {same_keys:.4f}, distance bits equal {same_bits:.4f}, max |diff| {worst:.3g}, "
                  f"counts equal {bool(np.array_equal(plain.counts, tiled.counts))}", flush=True)
        del built


if __name__ == "__main__":
    main()
