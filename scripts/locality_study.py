#!/usr/bin/env python3
"""scripts/locality_study.py — does the walk get faster when (a) the queries of a batch are handed out in an order that keeps
neighbours in vector space next to each other, and (b) the stored rows are laid out in that same order?

Everything is done with the PUBLIC entry points by permuting inputs: the rows before the build (keys = the original row numbers,
so results stay comparable), the queries before the search. Four cells per index: {rows as generated, rows in cluster order} x
{queries as generated, queries in cluster order}. The cluster order is generic (nothing of the generator is used): k-means
centroids on a sample, a greedy nearest-neighbour tour over the centroids, rows / queries sorted by the tour position of their
nearest centroid.

    python scripts/locality_study.py --vectors 10000000 --expansion 608
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import usearch_amd  # noqa: E402  (before torch: one HIP runtime)
import torch  # noqa: E402
from bench import synthetic_vectors_device, recall_per_query  # noqa: E402


def log(*a):
    print(*a, flush=True)


def kmeans_tour(data_f16: "torch.Tensor", clusters: int, iterations: int, sample: int, seed: int):
    """→ (centroids [C, d] f32 unit rows, tour_position [C]) ; cosine k-means on a sample, greedy tour over the centroids."""
    g = torch.Generator(device=data_f16.device)
    g.manual_seed(seed)
    n = data_f16.shape[0]
    pick = torch.randint(0, n, (min(sample, n),), generator=g, device=data_f16.device)
    x = torch.nn.functional.normalize(data_f16[pick].float(), dim=1)
    centroids = x[torch.randperm(len(x), generator=g, device=x.device)[:clusters]].clone()
    for _ in range(iterations):
        assign = torch.empty(len(x), dtype=torch.long, device=x.device)
        for b in range(0, len(x), 131072):
            assign[b:b + 131072] = (x[b:b + 131072] @ centroids.T).argmax(dim=1)
        sums = torch.zeros_like(centroids).index_add_(0, assign, x)
        counts = torch.bincount(assign, minlength=clusters).clamp(min=1).unsqueeze(1)
        fresh = torch.nn.functional.normalize(sums / counts, dim=1)
        empty = (sums.abs().sum(dim=1) == 0)
        fresh[empty] = centroids[empty]
        centroids = fresh
    sim = (centroids @ centroids.T).cpu().numpy()
    np.fill_diagonal(sim, -2.0)
    seen = np.zeros(clusters, dtype=bool)
    tour = [0]
    seen[0] = True
    for _ in range(clusters - 1):
        row = sim[tour[-1]].copy()
        row[seen] = -3.0
        nxt = int(row.argmax())
        tour.append(nxt)
        seen[nxt] = True
    position = np.empty(clusters, dtype=np.int64)
    position[np.array(tour)] = np.arange(clusters)
    return centroids, torch.from_numpy(position).to(data_f16.device)


def tour_keys(rows_f16: "torch.Tensor", centroids, position):
    out = torch.empty(rows_f16.shape[0], dtype=torch.long, device=rows_f16.device)
    for b in range(0, rows_f16.shape[0], 262144):
        x = rows_f16[b:b + 262144].float()
        out[b:b + 262144] = position[(x @ centroids.T).argmax(dim=1)]
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--vectors", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--queries", type=int, default=10_000)
    p.add_argument("--expansion", type=int, default=608)
    p.add_argument("--clusters", type=int, default=4096)
    p.add_argument("--steps", type=int, default=6)
    p.add_argument("--copies", type=int, default=2, help="fresh copies of every index (placement draws) the cells are timed on")
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "locality_study.json"))
    args = p.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    k = args.k

    data = synthetic_vectors_device(args.vectors, args.dim, "f16", 42, device)
    queries = synthetic_vectors_device(args.queries, args.dim, "f16", 43, device)
    data_f16 = data.view(torch.float16)
    t0 = time.time()
    centroids, position = kmeans_tour(data_f16, args.clusters, 6, 1_000_000, 7)
    row_key = tour_keys(data_f16, centroids, position)
    row_order = torch.argsort(row_key, stable=True)
    query_key = tour_keys(queries.view(torch.float16), centroids, position)
    query_order = torch.argsort(query_key, stable=True)
    torch.cuda.synchronize()
    log(f"[locality] cluster order of {args.vectors} rows + {args.queries} queries in {time.time() - t0:.1f}s "
        f"({args.clusters} centroids; biggest cluster {int(torch.bincount(row_key).max())} rows)")

    queries_sorted = queries[query_order].contiguous()
    q = args.queries
    keys_dev = torch.zeros((q, k), dtype=torch.int64, device=device)
    dist_dev = torch.zeros((q, k), dtype=torch.float32, device=device)
    counts_dev = torch.zeros(q, dtype=torch.int64, device=device)
    visited_dev = torch.zeros(q, dtype=torch.int64, device=device)
    computed_dev = torch.zeros(q, dtype=torch.int64, device=device)
    stream = torch.cuda.Stream(device)
    truth = None
    results = []

    def time_cells(index, label):
        nonlocal truth
        bpv = index.bytes_per_vector
        for qname, qs in (("queries as generated", queries), ("queries in cluster order", queries_sorted)):
            times = []
            for step in range(args.steps + 2):
                stats = index.search_device(qs.data_ptr(), q, qs.stride(0), k, args.expansion, keys_dev.data_ptr(),
                                            dist_dev.data_ptr(), counts_dev.data_ptr(), visited_dev.data_ptr(),
                                            computed_dev.data_ptr(), stream=stream.cuda_stream, timed=True,
                                            tuning=usearch_amd.Tuning(wave_clock=1))
                if step >= 2:
                    times.append(stats.kernel_ms)
            computed = computed_dev.cpu().numpy().astype(np.float64)
            visited = visited_dev.cpu().numpy().astype(np.float64)
            step_bytes = float(np.sum(computed * bpv + visited * 4 * 2 * index.connectivity + k * 8 + bpv))
            ms = float(np.mean(times))
            found = keys_dev.cpu().numpy().astype(np.uint64)
            if qs is queries_sorted:
                back = np.empty_like(found)
                back[query_order.cpu().numpy()] = found
                found = back
            recall = float(np.mean(recall_per_query(found, truth, k))) if truth is not None else None
            row = {"index": label, "queries": qname, "kernel_ms": ms, "min_ms": float(np.min(times)), "TBps": step_bytes / ms / 1e9,
                   "frac": step_bytes / ms / 1e9 / 8.0, "distances_per_query": float(computed.mean()),
                   "hops_per_query": float(visited.mean()), "recall": recall, "tail_idle": float(stats.tail_idle),
                   "placement": index.placement}
            results.append(row)
            log(f"[locality] {label:34s} | {qname:26s} | {ms:7.2f} ms (min {np.min(times):.2f}) | {row['TBps']:.2f} TB/s = "
                f"{row['frac'] * 100:.1f} % | {computed.mean():.0f} dist/q | recall {recall} | tail {stats.tail_idle:.3f}")

    for layout in ("rows as generated", "rows in cluster order"):
        if layout == "rows in cluster order":
            # permute in blocks: a second whole copy of the matrix next to torch's own would not fit comfortably
            permuted = torch.empty_like(data)
            for b in range(0, args.vectors, 1_000_000):
                permuted[b:b + 1_000_000] = data[row_order[b:b + 1_000_000]]
            del data
            data = permuted
            keys = row_order.cpu().numpy().astype(np.uint64)
        else:
            keys = None
        torch.cuda.synchronize()
        t0 = time.time()
        built = usearch_amd.build(None, "cos", "f16", keys=keys, connectivity=16, expansion_add=128, device=0,
                                  device_pointer=data.data_ptr(), count=args.vectors, stride=data.stride(0), ndim=args.dim)
        log(f"[locality] {layout}: built in {time.time() - t0:.1f}s")
        if truth is None:
            exact = built.index.search(queries.cpu().numpy().view(np.float16), k, dtype="f16", exact="tiled")
            truth = exact.keys
        time_cells(built.index, layout + " (builder's copy)")
        if args.copies > 0:
            image = built.save_buffer()
            for c in range(args.copies):
                copy = usearch_amd.Index.restore(image, device=0)
                time_cells(copy, layout + f" (copy {c + 1})")
                copy.close()
                del copy
            del image
        built.close()
        del built
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
