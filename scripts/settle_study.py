#!/usr/bin/env python3
"""scripts/settle_study.py — round 6: the level a SETTLED placement lands on is reproducible inside a process (the same image loaded
twice: 45.10 / 45.06 ms, 48.31 / 48.28 ms) but differs between processes (45.1 against 48.3 ms for the headline batch). Is it the
state the device's frame allocator was left in by whoever ran before — and can a process put it into the good state itself?

One process = one policy (`--condition`), the headline image cached in /dev/shm between the processes of a session:

    none     restore (the loader waits out the settle window), time the batch; `--copies` times, closing each copy first
    giant    before every restore: ONE allocation of all free device memory but `--spare-gb`, freed at once, then the settle window
    pieces   the same memory in 2-GB pieces, freed in reverse order
    `--perturb-gb G`: before anything else allocate G GB in 1-GB pieces and free every other one first, then the rest (a process that
    leaves the free lists shuffled, like a test suite before a bench run)"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--condition", default="none", choices=["none", "giant", "pieces"])
    p.add_argument("--copies", type=int, default=3)
    p.add_argument("--spare-gb", type=float, default=6.0)
    p.add_argument("--perturb-gb", type=float, default=0.0)
    p.add_argument("--cache", default="/dev/shm/usearch_amd_settle_study.img")
    p.add_argument("--ef", type=int, default=608)
    p.add_argument("--tag", default="")
    p.add_argument("--history", default="none", choices=["none", "datagen", "build", "build-small"],
                   help="what this process does BEFORE it loads the image: nothing; generate the 15.4 GB of synthetic vectors with torch "
                        "and free them; generate them and build the index on the device (then drop it); build a 1M-vector index")
    args = p.parse_args()
    os.environ["USEARCH_AMD_PLACEMENT_LOG"] = "0"
    import torch
    import usearch_amd
    device = torch.device("cuda", 0)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]

    def malloc(nbytes):
        pointer = ctypes.c_void_p()
        return pointer if hip.hipMalloc(ctypes.byref(pointer), int(nbytes)) == 0 else None

    if os.path.exists(args.cache):
        image = np.fromfile(args.cache, dtype=np.uint8)
    else:
        data = bench.synthetic_vectors_device(10_000_000, 768, "f16", 42, device)
        built = usearch_amd.build(None, "cos", "f16", device_pointer=data.data_ptr(), count=10_000_000, stride=data.stride(0), ndim=768)
        del data
        image = built.save_buffer()
        built.close()
        del built
        torch.cuda.empty_cache()
        image.tofile(args.cache)
    if args.history != "none" and os.path.exists(args.cache):
        count = 1_000_000 if args.history == "build-small" else 10_000_000
        data = bench.synthetic_vectors_device(count, 768, "f16", 42, device)
        if args.history.startswith("build"):
            built = usearch_amd.build(None, "cos", "f16", device_pointer=data.data_ptr(), count=count, stride=data.stride(0), ndim=768)
            built.close()
            del built
        del data
        torch.cuda.empty_cache()
        usearch_amd.note_device_free()
    q = 10_000
    queries = bench.synthetic_vectors_device(q, 768, "f16", 43, device)
    outs = [torch.zeros((q, 10), dtype=torch.int64, device=device), torch.zeros((q, 10), dtype=torch.float32, device=device)] + \
           [torch.zeros(q, dtype=torch.int64, device=device) for _ in range(3)]
    torch.cuda.empty_cache()

    if args.perturb_gb:
        blocks = [malloc(1 << 30) for _ in range(int(args.perturb_gb))]
        for block in blocks[::2]:
            if block:
                hip.hipFree(block)
        for block in blocks[1::2]:
            if block:
                hip.hipFree(block)
        usearch_amd.note_device_free()

    def condition():
        if args.condition == "none":
            return 0.0
        t0 = time.time()
        free, _ = torch.cuda.mem_get_info(device)
        want = free - int(args.spare_gb * (1 << 30))
        if args.condition == "giant":
            block = malloc(want)
            if block is None:
                print("  (the giant allocation was refused)", flush=True)
            else:
                hip.hipFree(block)
        else:
            blocks = [malloc(2 << 30) for _ in range(want // (2 << 30))]
            for block in reversed(blocks):
                if block:
                    hip.hipFree(block)
        usearch_amd.note_device_free()
        return time.time() - t0

    results = []
    for copy in range(args.copies):
        conditioned = condition()
        t0 = time.time()
        index = usearch_amd.Index.restore(image)
        load_s = time.time() - t0
        times = []
        for step in range(12):
            stats = index.search_device(queries.data_ptr(), q, queries.stride(0), 10, args.ef, outs[0].data_ptr(), outs[1].data_ptr(),
                                        outs[2].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(), timed=True)
            if step >= 6:
                times.append(stats.kernel_ms)
        results.append(float(np.mean(times)))
        print(f"{args.tag or args.condition} [history {args.history}] copy {copy}: conditioning {conditioned:.2f}s, load {load_s:.2f}s (settled {index.placement['settle_ms']:.0f} ms), "
              f"batch {np.mean(times):.2f} ms", flush=True)
        index.close()
        del index
    print(f"{args.tag or args.condition}: {' '.join(f'{r:.2f}' for r in results)} ms", flush=True)


if __name__ == "__main__":
    main()
