#!/usr/bin/env python3
"""scripts/placement_study.py — does the engine's millisecond gather probe predict the speed of the real walk, and where inside a
slow placement does the time go? One GPU, the headline index (10M x 768 f16) built once, its image restored several times with
the earlier copies still held (so that every copy lands elsewhere).

    for each copy:  probe GB/s of the whole matrix · of each eighth of it · the headline batch (10 000 queries, ef 608) in ms

Development tool (profiles/r03_placement/ keeps its output)."""
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
    p.add_argument("--n", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--dtype", default="f16")
    p.add_argument("--queries", type=int, default=10_000)
    p.add_argument("--ef", type=int, default=608)
    p.add_argument("--copies", type=int, default=6)
    p.add_argument("--parts", type=int, default=8)
    p.add_argument("--no-engine-draws", action="store_true")
    p.add_argument("--ballast-steps", type=int, default=0)
    p.add_argument("--ballast-gb", type=int, default=8)
    p.add_argument("--scratch-redraws", type=int, default=0,
                   help="on ONE resident copy: this many launches, each with a freshly allocated visited-set scratch block")
    p.add_argument("--chunks-mb", type=int, nargs="+", default=[0],
                   help="also place the matrix through the virtual-memory API in physical chunks of this many MB (0 = hipMalloc)")
    args = p.parse_args()
    metric = "hamming" if args.dtype == "b1" else "l2sq" if args.dtype == "i8" else "cos"

    import torch
    import usearch_amd

    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(args.n, args.dim, args.dtype, 42, device)
    t0 = time.time()
    built = usearch_amd.build(None, metric, args.dtype, device_pointer=data.data_ptr(), count=args.n, stride=data.stride(0),
                              ndim=args.dim)
    print(f"GPU-built {args.n} in {time.time() - t0:.1f}s", flush=True)
    del data
    image = built.save_buffer()
    del built
    torch.cuda.empty_cache()

    queries = bench.synthetic_vectors_device(args.queries, args.dim, args.dtype, 43, device)
    keys = torch.zeros((args.queries, 10), dtype=torch.int64, device=device)
    dists = torch.zeros((args.queries, 10), dtype=torch.float32, device=device)
    counts = torch.zeros(args.queries, dtype=torch.int64, device=device)
    visited = torch.zeros(args.queries, dtype=torch.int64, device=device)
    computed = torch.zeros(args.queries, dtype=torch.int64, device=device)

    def batch_ms(index) -> float:
        times = []
        for step in range(4):
            stats = index.search_device(queries.data_ptr(), args.queries, queries.stride(0), 10, args.ef, keys.data_ptr(),
                                        dists.data_ptr(), counts.data_ptr(), visited.data_ptr(), computed.data_ptr(), timed=True)
            if step:
                times.append(stats.kernel_ms)
        return min(times)

    if args.scratch_redraws:
        index = usearch_amd.Index.restore(image)

        def launches(count):
            out = []
            for _ in range(count):
                stats = index.search_device(queries.data_ptr(), args.queries, queries.stride(0), 10, args.ef, keys.data_ptr(),
                                            dists.data_ptr(), counts.data_ptr(), visited.data_ptr(), computed.data_ptr(), timed=True)
                out.append(round(stats.kernel_ms, 2))
            return out

        launches(3)
        print("same scratch block, launch after launch :", launches(args.scratch_redraws), flush=True)
        os.environ["USEARCH_AMD_SCRATCH_REDRAW"] = "1"
        print("a fresh scratch block for every launch  :", launches(args.scratch_redraws), flush=True)
        os.environ.pop("USEARCH_AMD_SCRATCH_REDRAW")
        print("same scratch block again                :", launches(args.scratch_redraws), flush=True)
        index.close()
        torch.cuda.empty_cache()
    if args.ballast_steps:
        # map the device's memory: park `ballast_gb` more gigabytes before every fresh scratch block, so that the blocks walk through
        # the whole physical space; the index stays where it is
        os.environ["USEARCH_AMD_PLACEMENT_DRAWS"] = "1"
        index = usearch_amd.Index.restore(image)
        os.environ["USEARCH_AMD_SCRATCH_REDRAW"] = "1"
        parked = []
        print(f"ballast GB  batch ms with a fresh scratch block (x3)   free GB", flush=True)
        for step in range(args.ballast_steps):
            times = []
            for _ in range(3):
                stats = index.search_device(queries.data_ptr(), args.queries, queries.stride(0), 10, args.ef, keys.data_ptr(),
                                            dists.data_ptr(), counts.data_ptr(), visited.data_ptr(), computed.data_ptr(), timed=True)
                times.append(round(stats.kernel_ms, 2))
            free_gb = torch.cuda.mem_get_info(device)[0] / 1e9
            print(f"{step * args.ballast_gb:8d}    {times}   {free_gb:6.1f}", flush=True)
            if free_gb < args.ballast_gb + 6:
                break
            parked.append(torch.empty(args.ballast_gb << 30, dtype=torch.uint8, device=device))
        os.environ.pop("USEARCH_AMD_SCRATCH_REDRAW")
        del parked
        index.close()
        torch.cuda.empty_cache()
    rows_per_part = args.n // args.parts
    for chunk_mb in args.chunks_mb:
        if chunk_mb < 0:  # the engine's default: ONE physical allocation of the whole array, mapped
            os.environ.pop("USEARCH_AMD_VMM_CHUNK_MB", None)
        else:
            os.environ["USEARCH_AMD_VMM_CHUNK_MB"] = str(chunk_mb)
        held = []
        how = "the engine's default (one mapped physical allocation)" if chunk_mb < 0 else "hipMalloc" if not chunk_mb \
            else f"hipMemCreate chunks of {chunk_mb} MB"
        print(f"--- matrix placed by {how}", flush=True)
        print("copy  load s  gather GB/s (x2)  pages M/s  latency ns rows / lists   batch ms   gather GB/s of each part", flush=True)
        for copy in range(args.copies):
            t0 = time.time()
            try:
                index = usearch_amd.Index.restore(image)
            except RuntimeError as error:
                print(f"{copy:4d}  restore failed: {error}", flush=True)
                break
            load_s = time.time() - t0
            held.append(index)
            whole = index.gather_probe()
            pages = index.translation_probe()
            latency = (index.latency_probe(), index.latency_probe(lists=True))
            parts = [index.gather_probe(i * rows_per_part, rows_per_part) for i in range(args.parts)] if args.parts > 1 else []
            ms = batch_ms(index)
            whole_again = index.gather_probe()
            latency_again = (index.latency_probe(), index.latency_probe(lists=True))
            print(f"{copy:4d}  {load_s:6.2f}  {whole:8.0f} {whole_again:8.0f}  {pages:9.0f}  {latency[0]:6.0f} {latency_again[0]:6.0f} / "
                  f"{latency[1]:6.0f} {latency_again[1]:6.0f}   {ms:8.3f}   " +
                  " ".join(f"{r:6.0f}" for r in parts), flush=True)
        for index in held:
            index.close()
        held.clear()
        torch.cuda.empty_cache()
    os.environ.pop("USEARCH_AMD_VMM_CHUNK_MB", None)
    # the engine's own choice
    for draws in (() if args.no_engine_draws else (6, 6, 1, 1)):
        os.environ["USEARCH_AMD_PLACEMENT_DRAWS"] = str(draws)
        t0 = time.time()
        index = usearch_amd.Index.restore(image)
        load_s = time.time() - t0
        print(f"draws={draws}: placement {index.placement}  load {load_s:.2f}s  batch {batch_ms(index):.3f} ms", flush=True)
        index.close()


if __name__ == "__main__":
    main()
