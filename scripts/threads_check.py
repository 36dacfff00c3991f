#!/usr/bin/env python3
"""scripts/threads_check.py: T host threads, each looping single-query searches through the host API (what T goroutines calling
`usearch_search` look like to the device) on the headline index: queries per second at T = 1, 16, 64."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch

    import usearch_amd
    n, dim, dtype = int(os.environ.get("THREADS_N", 10_000_000)), 768, "f16"
    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
    built = usearch_amd.build(None, "cos", dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
    del data
    torch.cuda.empty_cache()
    index = built.index
    queries = bench.synthetic_vectors_device(4096, dim, dtype, 43, device).cpu().numpy().view(bench.NUMPY_STORAGE[dtype])
    for ef in (608, 64):
        index.expansion_search = ef
        for threads in (1, 16, 64):
            per_thread = 24 if ef == 608 else 96

            def work(t):
                for i in range(per_thread):
                    index.search(queries[(t * per_thread + i) % 4096][None, :], 10, dtype=dtype)

            work(0)
            pool = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
            t0 = time.perf_counter()
            for thread in pool:
                thread.start()
            for thread in pool:
                thread.join()
            seconds = time.perf_counter() - t0
            print(f"ef={ef} threads={threads}: {threads * per_thread / seconds:,.0f} single-query calls per second "
                  f"({seconds / per_thread * 1e3:.2f} ms per call and thread)", flush=True)


if __name__ == "__main__":
    main()
