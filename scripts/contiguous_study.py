#!/usr/bin/env python3
"""scripts/contiguous_study.py — does PHYSICALLY CONTIGUOUS device memory (USEARCH_AMD_CONTIGUOUS=1: `hipDeviceMallocContiguous`
for the matrix, the lists and the block of visited-set slabs) end the placement lottery of the headline walk?

One index image; fresh copies of it are restored alternately with the switch off and on, every copy with the engine's placement
draws OFF (first placement of everything), and the headline batch is timed on each.

    python scripts/contiguous_study.py --vectors 10000000 --expansion 608 --copies 4
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
from bench import synthetic_vectors_device  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--vectors", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--dtype", default="f16")
    p.add_argument("--metric", default="cos")
    p.add_argument("--queries", type=int, default=10_000)
    p.add_argument("--expansion", type=int, default=608)
    p.add_argument("--copies", type=int, default=4)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "contiguous_study.json"))
    args = p.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ["USEARCH_AMD_SCRATCH_DRAWS"] = os.environ["USEARCH_AMD_PLACEMENT_DRAWS"] = "1"
    os.environ["USEARCH_AMD_PLACEMENT_LOG"] = "1"
    k, q = 10, args.queries
    data = synthetic_vectors_device(args.vectors, args.dim, args.dtype, 42, device)
    queries = synthetic_vectors_device(q, args.dim, args.dtype, 43, device)
    built = usearch_amd.build(None, args.metric, args.dtype, connectivity=16, expansion_add=128, device=0,
                              device_pointer=data.data_ptr(), count=args.vectors, stride=data.stride(0), ndim=args.dim)
    del data
    torch.cuda.empty_cache()
    image = built.save_buffer()
    keys_dev = torch.zeros((q, k), dtype=torch.int64, device=device)
    dist_dev = torch.zeros((q, k), dtype=torch.float32, device=device)
    counts_dev = torch.zeros(q, dtype=torch.int64, device=device)
    visited_dev = torch.zeros(q, dtype=torch.int64, device=device)
    computed_dev = torch.zeros(q, dtype=torch.int64, device=device)
    stream = torch.cuda.Stream(device)

    def timed(index):
        times = []
        for step in range(args.steps + 2):
            stats = index.search_device(queries.data_ptr(), q, queries.stride(0), k, args.expansion, keys_dev.data_ptr(),
                                        dist_dev.data_ptr(), counts_dev.data_ptr(), visited_dev.data_ptr(),
                                        computed_dev.data_ptr(), stream=stream.cuda_stream, timed=True)
            if step >= 2:
                times.append(stats.kernel_ms)
        return float(np.mean(times))

    rows = [{"copy": "builder's own", "contiguous": 0, "kernel_ms": timed(built.index)}]
    print(f"[contiguous] builder's own arrays: {rows[0]['kernel_ms']:.2f} ms", flush=True)
    built.close()
    del built
    held = []
    for c in range(args.copies):
        for mode in (0, 1):
            os.environ["USEARCH_AMD_CONTIGUOUS"] = str(mode)
            t0 = time.time()
            copy = usearch_amd.Index.restore(image, device=0)
            ms = timed(copy)
            rows.append({"copy": c, "contiguous": mode, "kernel_ms": ms, "restore_s": time.time() - t0})
            print(f"[contiguous] copy {c} contiguous={mode}: {ms:.2f} ms", flush=True)
            held.append(copy)
            if len(held) > 2:  # the two before stay resident, so that a fresh copy lands elsewhere
                held.pop(0).close()
    for mode in (0, 1):
        got = [r["kernel_ms"] for r in rows if r["contiguous"] == mode and r["copy"] != "builder's own"]
        print(f"[contiguous] contiguous={mode}: " + " ".join(f"{x:.2f}" for x in got) + f" ms (mean {np.mean(got):.2f})", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
