#!/usr/bin/env python3
"""scripts/variance_probe.py — where does the run-to-run spread of the headline kernel come from? (development tool)

One process: builds the 10M×768 f16 index on the GPU, then times the headline batch (10 000 queries, ef = 608) launch by launch
 (a) 40 launches back to back, a pause, 20 more        — drift over time (clocks, temperature),
 (b) on fresh snapshots restored from the saved image — the same bytes at other addresses (placement),
and prints the device's clocks / power before and after. Run it twice to see the spread across processes.
"""
import argparse
import subprocess
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=30).stdout
        keep = [line.strip() for line in out.splitlines() if any(word in line for word in ("sclk", "mclk", "fclk", "Power", "junction", "Junction"))]
        print("   smi: " + " | ".join(keep[:8]), flush=True)
    except Exception as error:  # noqa: BLE001
        print(f"   smi unavailable: {error}", flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--ef", type=int, default=608)
    p.add_argument("--batch", type=int, default=10_000)
    p.add_argument("--restores", type=int, default=3)
    p.add_argument("--pause", type=float, default=15.0)
    args = p.parse_args()
    import torch
    import usearch_amd
    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(args.n, args.dim, "f16", 42, device)
    t0 = time.time()
    built = usearch_amd.build(None, "cos", "f16", device_pointer=data.data_ptr(), count=args.n, stride=data.stride(0), ndim=args.dim)
    print(f"GPU-built {args.n} in {time.time() - t0:.1f}s", flush=True)
    del data
    torch.cuda.empty_cache()
    queries = bench.synthetic_vectors_device(args.batch, args.dim, "f16", 43, device)
    k = 10
    keys = torch.zeros((args.batch, k), dtype=torch.int64, device=device)
    dists = torch.zeros((args.batch, k), dtype=torch.float32, device=device)
    counts, visited, computed = (torch.zeros(args.batch, dtype=torch.int64, device=device) for _ in range(3))

    def series(index, launches, label):
        ms = []
        for _ in range(launches + 1):
            stats = index.search_device(queries.data_ptr(), args.batch, queries.stride(0), k, args.ef, keys.data_ptr(), dists.data_ptr(),
                                        counts.data_ptr(), visited.data_ptr(), computed.data_ptr(), timed=True)
            ms.append(stats.kernel_ms)
        ms = np.array(ms[1:])
        print(f"{label}: min {ms.min():.2f} median {np.median(ms):.2f} max {ms.max():.2f} ms; first 5 {np.round(ms[:5], 2).tolist()} "
              f"last 5 {np.round(ms[-5:], 2).tolist()}", flush=True)

    smi()
    series(built.index, 40, "built index, 40 launches")
    smi()
    if args.pause:
        time.sleep(args.pause)
        series(built.index, 20, f"after a {args.pause:.0f} s pause, 20 launches")
    image = built.save_buffer()
    built.close()
    del built
    torch.cuda.empty_cache()
    for round_ in range(args.restores):
        t0 = time.time()
        restored = usearch_amd.Index.restore(image)
        print(f"restored in {time.time() - t0:.2f} s", flush=True)
        series(restored, 10, f"fresh snapshot {round_ + 1} restored from the image, 10 launches")
        if round_ % 2 == 0:  # an odd-sized hole before the next one: the next snapshot lands elsewhere
            hole = torch.empty((1 << 30) + 12345 * 4096 * (round_ + 1), dtype=torch.uint8, device=device)
        restored.close()
        del restored
    smi()


if __name__ == "__main__":
    main()
