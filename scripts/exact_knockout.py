#!/usr/bin/env python3
"""scripts/exact_knockout.py — where the wide exact tile's time goes, by taking parts of it out (timing only: with any part out the
results are wrong). Needs a library built with the knock-outs compiled in — `make -C usearch_amd/csrc EXTRA=-DUSEARCH_AMD_EXPERIMENT_EXACT_KNOCKOUT
OUT=$PWD/usearch_amd/lib_knockout OBJ=$PWD/usearch_amd/lib_knockout/obj`, loaded with USEARCH_AMD_LIBRARY=usearch_amd/lib_knockout/libusearch_amd.so; the
product library compiles them away. `USEARCH_AMD_EXACT_KNOCKOUT` is read per launch (csrc/exact_tiled.hip: 1 = no fold, 2 = no fills after the
prologue's, 4 = no wait for the fills and no barrier, 8 = thresholds refreshed once, 16 = no per-block tests). Prints kernel ms and T(FL)OP/s per combination, same process, same data.

    python scripts/exact_knockout.py [--n 10000000] [--dim 768] [--queries 10000] [--dtype f16] [--combos 0,1,4,5,3,7]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--n", type=int, default=10_000_000)
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--queries", type=int, default=10_000)
    parser.add_argument("--dtype", default="f16")
    parser.add_argument("--k", type=int, default=10)
    parser.add_argument("--combos", default="0,1,4,5,3,7,0")
    parser.add_argument("--repeats", type=int, default=4)
    args = parser.parse_args()
    import torch

    import bench
    import usearch_amd
    device = torch.device("cuda", 0)
    metric = "l2sq" if args.dtype == "i8" else "cos"
    data = bench.synthetic_vectors_device(args.n, args.dim, args.dtype, 42, device)
    built = usearch_amd.build(None, metric, args.dtype, connectivity=4, expansion_add=16, device=0, device_pointer=data.data_ptr(),
                              count=args.n, stride=data.stride(0), ndim=args.dim)
    del data
    torch.cuda.empty_cache()
    index = built.index
    queries = bench.synthetic_vectors_device(args.queries, args.dim, args.dtype, 43, device)
    keys = torch.zeros((args.queries, args.k), dtype=torch.int64, device=device)
    distances = torch.zeros((args.queries, args.k), dtype=torch.float32, device=device)
    counts = torch.zeros(args.queries, dtype=torch.int64, device=device)
    stream = torch.cuda.Stream(device)

    def step() -> float:
        return index.exact_search_device(queries.data_ptr(), args.queries, queries.stride(0), args.k, keys.data_ptr(),
                                         distances.data_ptr(), counts.data_ptr(), stream=stream.cuda_stream, tiled=True)

    operations = 2.0 * args.queries * args.n * args.dim
    for _ in range(3):
        step()
    names = {1: "no fold", 2: "no fills", 4: "no barrier", 8: "thresholds once", 16: "no block tests", 32: "general fold forced", 64: "fused fold without its rare path", 128: "fold at priority 3"}
    for combo in [int(c) for c in args.combos.split(",")]:
        os.environ["USEARCH_AMD_EXACT_KNOCKOUT"] = str(combo)
        step()
        ms = [step() for _ in range(args.repeats)]
        what = " + ".join(names[b] for b in (1, 2, 4, 8, 16, 32, 64, 128) if combo & b) or "the kernel as it is"
        print(f"knockout {combo} ({what}): kernel {np.mean(ms):.1f} ms (min {np.min(ms):.1f}) = "
              f"{operations / (np.mean(ms) / 1e3) / 1e12:.0f} T(FL)OP/s", flush=True)
    os.environ["USEARCH_AMD_EXACT_KNOCKOUT"] = "0"


if __name__ == "__main__":
    main()
