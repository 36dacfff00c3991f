#!/usr/bin/env python3
"""scripts/graph_quality.py — is the graph the device builder links as good as the reference's? Same vectors, same M and
ef_construction; recall@10 against exact search and distances per query at a ladder of expansions, for (a) the reference's own
`add` loop on the host cores and (b) the device builder. What matters for the headline: the smallest expansion that reaches 0.95.

    python scripts/graph_quality.py --vectors 1000000 --threads 16
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import usearch_amd  # noqa: E402
import torch  # noqa: E402
from bench import synthetic_vectors, synthetic_vectors_device, recall_per_query, host_cores  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--vectors", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--queries", type=int, default=10_000)
    p.add_argument("--threads", type=int, default=0)
    p.add_argument("--expansions", type=int, nargs="+", default=[64, 96, 128, 160, 192, 224, 256, 320, 384, 512])
    p.add_argument("--variants", nargs="+", default=["gpu", "gpu:max_batch=16384", "gpu:max_batch=4096", "reference"])
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "graph_quality.json"))
    args = p.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    k = 10
    vectors = synthetic_vectors(args.vectors, args.dim, "f16", seed=42)
    queries = synthetic_vectors_device(args.queries, args.dim, "f16", 43, device).cpu().numpy().view(np.float16)
    data_dev = torch.from_numpy(vectors.view(np.uint8).reshape(args.vectors, -1)).to(device)
    rows = []
    truth = None
    for variant in args.variants:
        name, _, options = variant.partition(":")
        knobs = dict(item.split("=") for item in options.split(",") if item)
        t0 = time.time()
        if name == "reference":
            from oracle import refbind
            threads = args.threads or 2 * host_cores()
            ref = refbind.RefIndex(args.dim, "cos", "f16", 16, 128, 64)
            ref.add(np.arange(args.vectors, dtype=np.uint64), vectors, threads=threads)
            index = usearch_amd.Index.restore(ref.save_buffer(), device=0)
            del ref
            owner = None
        else:
            owner = usearch_amd.build(None, "cos", "f16", connectivity=16, expansion_add=128, device=0, device_pointer=data_dev.data_ptr(),
                                      count=args.vectors, stride=data_dev.stride(0), ndim=args.dim,
                                      max_batch=int(knobs.get("max_batch", 0)), batch_divisor=int(knobs.get("batch_divisor", 0)))
            index = owner.index
        seconds = time.time() - t0
        if truth is None:
            truth = index.search(queries, k, dtype="f16", exact="tiled").keys
        curve = []
        for ef in args.expansions:
            got = index.search(queries, k, expansion=ef, dtype="f16")
            curve.append({"ef": ef, "recall": float(np.mean(recall_per_query(got.keys, truth, k))),
                          "distances_per_query": float(got.computed_per_query.mean()), "hops_per_query": float(got.visited_per_query.mean())})
        rows.append({"builder": variant, "build_seconds": seconds, "curve": curve})
        print(f"[quality] {variant:24s} built in {seconds:7.1f}s | " +
              " ".join(f"ef{c['ef']}:{c['recall']:.4f}/{c['distances_per_query']:.0f}" for c in curve), flush=True)
        index.close() if owner is None else owner.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
