#!/usr/bin/env python3
"""scripts/tail_overlap.py — does handing the LAST T queries of a chip-filling batch to five-wave teams shorten the batch?

The one-wave kernel's launch drains: once the ticket counter runs dry the waves leave one after the other over a query's duration
and 5 % of the launch's wave-time is idle (`stats.tail_idle`). Prototype of the cheapest remedy: the batch is split,
the first Q − T queries go out as one persistent launch, the last T as a second, OVERLAPPING batch of `team_search_kernel` workgroups
(a batch of ≤ 2 × CUs queries gets them by itself) on another stream — its workgroups become resident as the first launch's waves
leave. Same results by construction (`test_team_and_one_wave_agree`). Reports wall time per step of the split against the whole.

    python scripts/tail_overlap.py [--vectors 10000000] [--expansion 608] [--tails 0 128 256 384 512]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--vectors", type=int, default=10_000_000)
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--dtype", default="f16")
    parser.add_argument("--queries", type=int, default=10_000)
    parser.add_argument("--expansion", type=int, default=608)
    parser.add_argument("--tails", type=int, nargs="+", default=[0, 128, 256, 384, 512, 0])
    parser.add_argument("--steps", type=int, default=12)
    parser.add_argument("--delay-us", type=float, default=300.0, help="the tail's launch follows the head's by this much")
    args = parser.parse_args()
    import torch

    import bench
    import usearch_amd
    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(args.vectors, args.dim, args.dtype, 42, device)
    built = usearch_amd.build(None, "cos", args.dtype, device_pointer=data.data_ptr(), count=args.vectors, stride=data.stride(0), ndim=args.dim)
    image = built.save_buffer()
    built.close()
    del data, built
    torch.cuda.empty_cache()
    usearch_amd.note_device_free()
    index = usearch_amd.Index.restore(image)
    del image
    q, k = args.queries, 10
    queries = bench.synthetic_vectors_device(q, args.dim, args.dtype, 43, device)
    outs = [torch.zeros((q, k), dtype=torch.int64, device=device), torch.zeros((q, k), dtype=torch.float32, device=device)] + \
           [torch.zeros(q, dtype=torch.int64, device=device) for _ in range(3)]
    streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
    row = queries.stride(0) * queries.element_size()

    def part(first: int, count: int, stream, stats_out: list):
        if not count:
            return
        pointers = [queries.data_ptr() + first * row, count, queries.stride(0), k, args.expansion,
                    outs[0].data_ptr() + first * k * 8, outs[1].data_ptr() + first * k * 4, outs[2].data_ptr() + first * 8,
                    outs[3].data_ptr() + first * 8, outs[4].data_ptr() + first * 8]
        stats_out.append(index.search_device(*pointers, stream=stream.cuda_stream, timed=False))

    def step(tail: int) -> float:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        head_stats, tail_stats = [], []
        workers = [threading.Thread(target=part, args=(0, q - tail, streams[0], head_stats))]
        if tail:
            def later():
                time.sleep(args.delay_us / 1e6)
                part(q - tail, tail, streams[1], tail_stats)
            workers.append(threading.Thread(target=later))
        for worker in workers:
            worker.start()
        for worker in workers:
            worker.join()
        torch.cuda.synchronize()
        step.variants = (head_stats[0].variant, tail_stats[0].variant if tail_stats else None)
        return (time.perf_counter() - t0) * 1e3

    for _ in range(6):
        step(0)
    reference = None
    for tail in args.tails:
        for _ in range(3):
            step(tail)
        times = [step(tail) for _ in range(args.steps)]
        keys = outs[0].cpu().numpy().copy()
        if reference is None:
            reference = keys
        print(f"tail {tail:4d} queries to teams: {np.mean(times):7.3f} ms per step (min {np.min(times):.3f}, kernel builds {step.variants}); "
              f"keys identical to the whole batch: {np.array_equal(keys, reference)}", flush=True)


if __name__ == "__main__":
    main()
