#!/usr/bin/env python3
"""scripts/sweep.py — tuning sweep of the search kernel on one GPU: builds (or reuses) one reference-built index and times
the batched search under different scratch placements / persistent-wave counts / unroll depths.

    python scripts/sweep.py --n 500000 --dim 768 --dtype f16 --cache-dir /dev/shm --ef 64 256 \
        --modes 1 2 --waves 4 8 16 --variants 1 2 3 4

Development tool (not part of the product or of the test-suite); prints one line per variant to stdout.
"""
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
    p.add_argument("--n", type=int, default=500_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--dtype", default="f16")
    p.add_argument("--queries", type=int, nargs="+", default=[10_000], help="batch sizes, one sweep each on the same index")
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--ef", type=int, nargs="+", default=[64, 256])
    p.add_argument("--modes", type=int, nargs="+", default=[1, 2])
    p.add_argument("--waves", type=int, nargs="+", default=[0])
    p.add_argument("--variants", type=int, nargs="+", default=[0])
    p.add_argument("--frontiers", type=int, nargs="+", default=[0], help="0 = auto, 1 = reference heap, 2 = open cells of top")
    p.add_argument("--n-queries-check", type=int, default=0, help="also compare distances / counters across settings")
    p.add_argument("--cache-dir", default="/dev/shm")
    p.add_argument("--gather", action="store_true")
    p.add_argument("--build-threads", type=int, default=0)
    p.add_argument("--env", nargs="*", default=[""],
                   help="environment settings to A/B on the same index, one run each: 'NAME=V,NAME2=V2' ('' = none); the "
                        "engine reads its USEARCH_AMD_* switches at every search call")
    args = p.parse_args()
    metric = "hamming" if args.dtype == "b1" else "l2sq" if args.dtype == "i8" else "cos"

    import torch
    import usearch_amd

    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(args.n, args.dim, args.dtype, 42, device)
    t0 = time.time()
    built = usearch_amd.build(None, metric, args.dtype, device_pointer=data.data_ptr(), count=args.n,
                              stride=data.stride(0), ndim=args.dim)
    print(f"GPU-built {args.n} in {time.time() - t0:.1f}s", flush=True)
    del data
    torch.cuda.empty_cache()
    index = built.index
    for batch in args.queries:
        print(f"=== batch of {batch} queries", flush=True)
        device = torch.device("cuda", 0)
        queries = bench.synthetic_vectors_device(batch, args.dim, args.dtype, 43, device)
        queries_host = queries.cpu().numpy().view(bench.NUMPY_STORAGE[args.dtype])
        keys = torch.zeros((batch, args.k), dtype=torch.int64, device=device)
        dists = torch.zeros((batch, args.k), dtype=torch.float32, device=device)
        counts = torch.zeros(batch, dtype=torch.int64, device=device)
        visited = torch.zeros(batch, dtype=torch.int64, device=device)
        computed = torch.zeros(batch, dtype=torch.int64, device=device)
        bpv, m0 = index.bytes_per_vector, 2 * index.connectivity
        if args.gather:
            # ceiling: the same row-gather + distance loop with no graph dependencies (one wave per list of random slots)
            rng = np.random.default_rng(7)
            for per in (256, 2048):
                nq = min(batch, 8192)
                slots = rng.integers(0, len(index), size=(nq, per), dtype=np.uint32)
                for _ in range(2):
                    index.distances(queries_host[:nq], slots)
                ms = index.last_distances_ms
                print(f"gather ceiling: {nq} waves x {per} random rows of {index.row_stride} B: {ms:.3f} ms = "
                      f"{nq * per * index.row_stride / ms / 1e6:.1f} GB/s", flush=True)
        reference_keys = {}
        for ef in args.ef:
          for setting in args.env:
            for name in [n for n in os.environ if n.startswith("USEARCH_AMD_")]:
                del os.environ[name]
            for pair in filter(None, setting.split(",")):
                os.environ[pair.split("=")[0]] = pair.split("=")[1]
            if setting:
                print(f"--- {setting}", flush=True)
            for mode in args.modes:
                for waves in args.waves:
                    for variant in args.variants:
                      for frontier in args.frontiers:
                        tuning = usearch_amd.Tuning(mode=mode, waves_per_cu=waves, variant=variant, frontier=frontier, wave_clock=1)
                        ms = []
                        try:
                            for step in range(args.steps + 1):
                                stats = index.search_device(queries.data_ptr(), batch, queries.stride(0), args.k, ef,
                                                            keys.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                                                            visited.data_ptr(), computed.data_ptr(), timed=True, tuning=tuning)
                                if step:
                                    ms.append(stats.kernel_ms)
                        except RuntimeError as error:
                            print(f"ef={ef:4d} mode={mode} waves/cu={waves} variant={variant} frontier={frontier}: {error}", flush=True)
                            continue
                        c = computed.cpu().numpy().astype(np.float64)
                        v = visited.cpu().numpy().astype(np.float64)
                        step_bytes = float(np.sum(c * bpv + v * 4 * m0 + args.k * 8 + bpv))
                        best = min(ms)
                        k_host = keys.cpu().numpy()
                        d_host = dists.cpu().numpy().view(np.uint32)
                        same = reference_keys.setdefault(ef, (k_host, d_host, c, v))
                        differing = [int((same[0] != k_host).any(axis=1).sum()), int((same[1] != d_host).any(axis=1).sum()),
                                     int((same[2] != c).sum()), int((same[3] != v).sum())]
                        identical = "yes" if not any(differing) else f"queries differing in keys/distances/computed/hops: {differing}"
                        peaks = index.last_peaks(batch)
                        print(f"ef={ef:4d} mode={stats.mode} waves/cu={waves:2d} grid={stats.grid:5d} lds={stats.lds_bytes:6d} "
                              f"variant={stats.variant} frontier={stats.frontier} passes={stats.passes} ms={best:8.3f} "
                              f"(mean {np.mean(ms):8.3f}) qps={batch / best * 1e3:10.0f} GB/s={step_bytes / best / 1e6:8.1f} "
                              f"dist/q={c.mean():.0f} hops/q={v.mean():.0f} peak_next={peaks[:, 0].max()} "
                              f"visits_max={peaks[:, 1].max()} tail_idle={stats.tail_idle:.4f} span_ms={stats.span_ms:.3f} "
                              f"same as the first setting: {identical}", flush=True)


if __name__ == "__main__":
    main()
