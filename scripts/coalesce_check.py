#!/usr/bin/env python3
"""scripts/coalesce_check.py — T host threads looping `usearch_search` (ONE query per call, the C ABI's hot signature) on the headline
index through the drop-in library, with and without the call combiner (USEARCH_AMD_COALESCE, csrc/combiner.hpp: calls that arrive
while a launch is in flight go out together in the next one). Calls per second and milliseconds per call at T = 1, 16, 64.

    GPU_MAX_HW_QUEUES=16 python scripts/coalesce_check.py
"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Options(C.Structure):  # usearch_init_options_t, c/usearch.h:64-110
    _fields_ = [("metric_kind", C.c_int), ("metric", C.c_void_p), ("quantization", C.c_int), ("dimensions", C.c_size_t),
                ("connectivity", C.c_size_t), ("expansion_add", C.c_size_t), ("expansion_search", C.c_size_t), ("multi", C.c_bool)]


def main():
    import torch

    import usearch_amd
    n, dim, dtype = int(os.environ.get("THREADS_N", 10_000_000)), 768, "f16"
    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
    built = usearch_amd.build(None, "cos", dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
    del data
    image = built.save_buffer()
    built.close()
    torch.cuda.empty_cache()
    queries = bench.synthetic_vectors_device(4096, dim, dtype, 43, device).cpu().numpy().view(np.float16)
    L = C.CDLL(os.path.join(ROOT, "usearch_amd", "lib", "libusearch_c.so"))
    err_p = C.POINTER(C.c_char_p)
    L.usearch_init.restype = C.c_void_p
    L.usearch_init.argtypes = [C.POINTER(Options), err_p]
    L.usearch_view_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_p]
    L.usearch_change_expansion_search.argtypes = [C.c_void_p, C.c_size_t, err_p]
    L.usearch_change_threads_search.argtypes = [C.c_void_p, C.c_size_t, err_p]
    L.usearch_search.restype = C.c_size_t
    L.usearch_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, err_p]
    L.usearch_free.argtypes = [C.c_void_p, err_p]
    for coalesce in (0, 1):
        os.environ["USEARCH_AMD_COALESCE"] = str(coalesce)  # read when the index is created
        err = C.c_char_p()
        options = Options(1, None, 3, dim, 16, 128, 64, False)  # cos, f16
        index = L.usearch_init(C.byref(options), C.byref(err))
        L.usearch_view_buffer(index, C.c_void_p(image.ctypes.data), image.size, C.byref(err))
        assert not err.value, err.value
        L.usearch_change_threads_search(index, 64, C.byref(err))
        for ef in (608, 64):
            L.usearch_change_expansion_search(index, ef, C.byref(err))
            for threads in (1, 16, 64):
                per_thread = 24 if ef == 608 else 96
                failures = []

                def work(t):
                    keys, distances, e = np.zeros(10, dtype=np.uint64), np.zeros(10, dtype=np.float32), C.c_char_p()
                    for i in range(per_thread):
                        q = queries[(t * per_thread + i) % 4096]
                        L.usearch_search(index, C.c_void_p(q.ctypes.data), 3, 10, C.c_void_p(keys.ctypes.data),
                                         C.c_void_p(distances.ctypes.data), C.byref(e))
                        if e.value:
                            failures.append(e.value)
                            return

                work(0)
                pool = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
                t0 = time.perf_counter()
                for thread in pool:
                    thread.start()
                for thread in pool:
                    thread.join()
                seconds = time.perf_counter() - t0
                assert not failures, failures[:1]
                print(f"coalesce={coalesce} ef={ef} threads={threads}: {threads * per_thread / seconds:,.0f} calls per second "
                      f"({seconds / per_thread * 1e3:.2f} ms per call and thread)", flush=True)
        L.usearch_free(index, C.byref(err))


if __name__ == "__main__":
    main()
