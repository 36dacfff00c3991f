#!/usr/bin/env python3
"""scripts/build_probe.py — GPU index construction at scale: time split, work counters and graph quality next to a
reference-built index of the same data (recall@10 of the SAME GPU search over both graphs, exact ground truth on GPU).

    python scripts/build_probe.py --n 200000 --dim 768 --dtype f16 [--reference] [--max-batch 65536]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (first: the engine then shares torch's HIP runtime)

import usearch_amd  # noqa: E402
from bench import host_cores, synthetic_vectors, synthetic_vectors_device  # noqa: E402


def recall_at(found, truth):
    k = truth.shape[1]
    return float(np.mean([len(np.intersect1d(found[i], truth[i])) / k for i in range(len(truth))]))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--n", type=int, default=200_000)
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--dtype", default="f16")
    parser.add_argument("--metric", default=None)
    parser.add_argument("--queries", type=int, default=1000)
    parser.add_argument("--max-batch", type=int, default=0)
    parser.add_argument("--batch-divisor", type=int, default=0)
    parser.add_argument("--reference", action="store_true", help="also build with the reference for comparison")
    parser.add_argument("--save-check", action="store_true", help="serialize and let the reference load the image")
    parser.add_argument("--device-data", action="store_true", help="generate the vectors on the GPU")
    args = parser.parse_args()
    metric = args.metric or ("hamming" if args.dtype == "b1" else "l2sq" if args.dtype == "i8" else "cos")
    queries = synthetic_vectors(args.queries, args.dim, args.dtype, seed=43)
    t0 = time.time()
    if args.device_data:
        data = synthetic_vectors_device(args.n, args.dim, args.dtype, 42, torch.device("cuda", 0))
        torch.cuda.synchronize()
        generate_seconds = time.time() - t0
        t0 = time.time()
        built = usearch_amd.build(None, metric, args.dtype, device_pointer=data.data_ptr(), count=args.n,
                                  stride=data.stride(0), ndim=args.dim, max_batch=args.max_batch,
                                  batch_divisor=args.batch_divisor)
        seconds = time.time() - t0
        if args.reference:
            vectors = data.cpu().numpy()
        del data
        torch.cuda.empty_cache()
    else:
        vectors = synthetic_vectors(args.n, args.dim, args.dtype, seed=42)
        generate_seconds = time.time() - t0
        t0 = time.time()
        built = usearch_amd.build(vectors, metric, args.dtype, max_batch=args.max_batch,
                                  batch_divisor=args.batch_divisor)
        seconds = time.time() - t0
    stats = built.stats.as_dict()
    out = {"generate_seconds": generate_seconds, "n": args.n, "dim": args.dim, "dtype": args.dtype, "metric": metric, "gpu_build_seconds": seconds,
           "vectors_per_second": args.n / seconds, "stats": stats,
           "per_node": {k: stats[k] / args.n for k in ("search_distances", "search_hops", "select_distances",
                                                          "reverse_distances", "repruned_lists")}}
    t0 = time.time()
    exact = built.index.search(queries, 10, dtype=args.dtype, exact=True)
    truth = exact.keys
    out["exact_seconds"] = time.time() - t0
    out["exact_kernel_ms"] = exact.stats.kernel_ms
    out["gpu_graph_recall"] = {}
    for ef in (64, 128, 256, 512):
        got = built.index.search(queries, 10, dtype=args.dtype, expansion=ef)
        out["gpu_graph_recall"][ef] = {"recall": recall_at(got.keys, truth),
                                       "distances_per_query": got.computed_distances / len(queries)}
    if args.save_check or args.reference:
        from oracle import refbind
    if args.save_check:
        t0 = time.time()
        image = built.save_buffer()
        out["save_seconds"] = time.time() - t0
        t0 = time.time()
        theirs = refbind.RefIndex.from_buffer(image, view=True, dtype=args.dtype)
        out["reference_view_seconds"] = time.time() - t0
        theirs.expansion_search = 128
        found, *_ = theirs.search(queries, 10, dtype=args.dtype, threads=host_cores())
        out["reference_search_of_gpu_graph_recall_ef128"] = recall_at(found, truth)
    if args.reference:
        cores = host_cores()
        own = refbind.RefIndex(args.dim, metric, args.dtype, 16, 128, 64)
        t0 = time.time()
        own.add(np.arange(args.n, dtype=np.uint64), vectors, threads=2 * cores)
        out["reference_build_seconds"] = time.time() - t0
        out["reference_build_threads"] = 2 * cores
        theirs = usearch_amd.Index.restore(own.save_buffer())
        out["reference_graph_recall"] = {}
        for ef in (64, 128, 256, 512):
            got = theirs.search(queries, 10, dtype=args.dtype, expansion=ef)
            out["reference_graph_recall"][ef] = {"recall": recall_at(got.keys, truth),
                                                 "distances_per_query": got.computed_distances / len(queries)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
