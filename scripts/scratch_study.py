#!/usr/bin/env python3
"""scripts/scratch_study.py — the block of per-wave visited-set slabs decides which speed the headline walk runs at (a fresh block
flips the batch by 13 %, profiles/r03_placement/). Is that a matter of the block's FOOTPRINT against the 256-MiB Infinity Cache?
One index, USEARCH_AMD_SCRATCH_REDRAW=1 (a fresh block for every launch), and per setting of {cells per slab, waves per CU} the
distribution of the batch time over fresh blocks.

    python scripts/scratch_study.py --launches 10
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import usearch_amd  # noqa: E402
import torch  # noqa: E402
from bench import synthetic_vectors_device  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--vectors", type=int, default=10_000_000)
    p.add_argument("--queries", type=int, default=10_000)
    p.add_argument("--expansion", type=int, default=608)
    p.add_argument("--launches", type=int, default=10)
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scratch_study.json"))
    args = p.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ["USEARCH_AMD_SCRATCH_DRAWS"] = os.environ["USEARCH_AMD_PLACEMENT_DRAWS"] = "1"
    k, q = 10, args.queries
    data = synthetic_vectors_device(args.vectors, 768, "f16", 42, device)
    queries = synthetic_vectors_device(q, 768, "f16", 43, device)
    built = usearch_amd.build(None, "cos", "f16", connectivity=16, expansion_add=128, device=0,
                              device_pointer=data.data_ptr(), count=args.vectors, stride=data.stride(0), ndim=768)
    del data
    torch.cuda.empty_cache()
    index = built.index
    out = [torch.zeros((q, k), dtype=torch.int64, device=device), torch.zeros((q, k), dtype=torch.float32, device=device)] + \
          [torch.zeros(q, dtype=torch.int64, device=device) for _ in range(3)]
    stream = torch.cuda.Stream(device)

    def launch(tuning):
        stats = index.search_device(queries.data_ptr(), q, queries.stride(0), k, args.expansion, *[t.data_ptr() for t in out],
                                    stream=stream.cuda_stream, timed=True, tuning=tuning)
        return stats

    rows = []
    os.environ["USEARCH_AMD_SCRATCH_REDRAW"] = "1"
    for cells, waves in ((0, 0), (65536, 0), (0, 4), (0, 6), (65536, 4), (0, 12), (0, 16)):
        tuning = usearch_amd.Tuning(hash_cap=cells, waves_per_cu=waves, wave_clock=1)
        if waves > 8:
            tuning.variant = 1  # the 4-loads build is the one cut for 16 waves per CU
        times, passes = [], 0
        for _ in range(args.launches):
            stats = launch(tuning)
            times.append(float(stats.kernel_ms))
            passes = max(passes, int(stats.passes))
        footprint = stats.grid * (cells or 32768) * 4 / 2**20
        rows.append({"cells": cells or 32768, "waves_per_cu": waves or 8, "grid": int(stats.grid), "footprint_MiB": footprint,
                     "kernel_ms": times, "passes": passes, "tail_idle": float(stats.tail_idle)})
        print(f"[scratch] cells {cells or 32768:6d} waves/CU {waves or 8:2d} grid {stats.grid:5d} footprint {footprint:6.0f} MiB passes {passes}: "
              + " ".join(f"{t:.2f}" for t in times) + f"  tail {stats.tail_idle:.3f}", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
