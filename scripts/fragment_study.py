#!/usr/bin/env python3
"""scripts/fragment_study.py — round 5: is the "placement lottery" of the headline walk (45 … 52 ms on the same bytes) the size of the
page-table FRAGMENTS an allocation ends up with? Every UTCL1 miss of the walk (≈ one per gathered row) is answered by the per-XCD
UTCL2; an entry of that cache covers one fragment, and the fragment a range gets is bounded by the alignment of its virtual address
relative to its physical frames. `hipMalloc` hands out 2-MB-aligned ranges (libhsakmt's default, HSA_MAX_VA_ALIGN = 9 → 4 KB << 9).

One process = one allocation policy (the environment decides before HIP initialises):

    default                     hipMalloc for the matrix and the block of visited-set slabs
    USEARCH_AMD_ALIGNED_MAP=1   one physical allocation each, mapped at a range aligned to its own size (≤ 1 GB)
    HSA_MAX_VA_ALIGN=18         the runtime's own allocator aligning ranges up to 1 GB

For each: the headline index (built once, image cached in /dev/shm between the processes of a session) restored `--copies` times
with the engine's draws OFF, each copy with its own workspace, the batch timed on each. Development tool (profiles/r05_fragments/)."""
import argparse
import os
import shutil
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
    p.add_argument("--tag", default="default")
    p.add_argument("--cache", default="/dev/shm/usearch_amd_fragment_study.img")
    p.add_argument("--hold", action="store_true", help="keep earlier copies resident (every copy lands elsewhere)")
    p.add_argument("--sleep", type=float, default=0.0, help="seconds to wait (device idle) between closing a copy and restoring the next")
    p.add_argument("--sleep-before-close", type=float, default=0.0, help="seconds of idleness BEFORE a copy is closed (none after)")
    p.add_argument("--settle", action="store_true",
                   help="after closing a copy, poll the device's free memory until it stops growing (the driver releases freed blocks "
                        "late: wipe on release), print the trajectory, then restore at once")
    args = p.parse_args()
    os.environ["USEARCH_AMD_PLACEMENT_DRAWS"] = "1"
    os.environ["USEARCH_AMD_SCRATCH_DRAWS"] = "1"
    os.environ["USEARCH_AMD_PLACEMENT_LOG"] = "1"
    metric = "hamming" if args.dtype == "b1" else "l2sq" if args.dtype == "i8" else "cos"

    import torch
    import usearch_amd

    device = torch.device("cuda", 0)
    print(f"=== {args.tag}: HSA_MAX_VA_ALIGN={os.environ.get('HSA_MAX_VA_ALIGN')} "
          f"USEARCH_AMD_ALIGNED_MAP={os.environ.get('USEARCH_AMD_ALIGNED_MAP')}", flush=True)
    if os.path.exists(args.cache):
        t0 = time.time()
        image = np.fromfile(args.cache, dtype=np.uint8)
        print(f"image of {image.nbytes / 1e9:.1f} GB read back from {args.cache} in {time.time() - t0:.1f}s", flush=True)
    else:
        data = bench.synthetic_vectors_device(args.n, args.dim, args.dtype, 42, device)
        t0 = time.time()
        built = usearch_amd.build(None, metric, args.dtype, device_pointer=data.data_ptr(), count=args.n, stride=data.stride(0),
                                  ndim=args.dim)
        print(f"GPU-built {args.n} in {time.time() - t0:.1f}s", flush=True)
        del data
        image = built.save_buffer()
        del built
        torch.cuda.empty_cache()
        try:
            if shutil.disk_usage(os.path.dirname(args.cache)).free > image.nbytes + (8 << 30):
                t0 = time.time()
                image.tofile(args.cache)
                print(f"image cached in {args.cache} in {time.time() - t0:.1f}s", flush=True)
        except OSError as error:
            print(f"no cache: {error}", flush=True)

    queries = bench.synthetic_vectors_device(args.queries, args.dim, args.dtype, 43, device)
    keys = torch.zeros((args.queries, 10), dtype=torch.int64, device=device)
    dists = torch.zeros((args.queries, 10), dtype=torch.float32, device=device)
    counts = torch.zeros(args.queries, dtype=torch.int64, device=device)
    visited = torch.zeros(args.queries, dtype=torch.int64, device=device)
    computed = torch.zeros(args.queries, dtype=torch.int64, device=device)

    def batch_ms(index):
        times = []
        for step in range(4):
            stats = index.search_device(queries.data_ptr(), args.queries, queries.stride(0), 10, args.ef, keys.data_ptr(),
                                        dists.data_ptr(), counts.data_ptr(), visited.data_ptr(), computed.data_ptr(), timed=True)
            if step:
                times.append(round(stats.kernel_ms, 3))
        return times

    held = []
    results = []
    for copy in range(args.copies):
        t0 = time.time()
        index = usearch_amd.Index.restore(image)
        load_s = time.time() - t0
        times = batch_ms(index)
        results.append(min(times))
        print(f"{args.tag} copy {copy}: load {load_s:.2f}s  batch {times} ms  gather {index.gather_probe():.0f} GB/s", flush=True)
        if args.sleep_before_close:
            torch.cuda.synchronize()
            time.sleep(args.sleep_before_close)
        if args.hold:
            held.append(index)
        else:
            index.close()
            del index
        if args.settle:
            torch.cuda.synchronize()
            t0 = time.time()
            trajectory, stable = [], 0
            while time.time() - t0 < 5.0 and stable < 20:
                free = torch.cuda.mem_get_info(device)[0]
                if trajectory and free == trajectory[-1][1]:
                    stable += 1
                else:
                    stable = 0
                    trajectory.append((round(time.time() - t0, 3), free))
                time.sleep(0.01)
            print("   free memory after close: " + " ".join(f"{t:.2f}s:{free / 1e9:.2f}GB" for t, free in trajectory), flush=True)
        if args.sleep:
            torch.cuda.synchronize()
            time.sleep(args.sleep)
    print(f"=== {args.tag}: batch ms per copy {results}  min {min(results):.3f}  max {max(results):.3f}", flush=True)


if __name__ == "__main__":
    main()
