#!/usr/bin/env python3
"""scripts/coalesce_check.py — T host threads looping `usearch_search` (ONE query per call, the C ABI's hot signature) on the headline
index through the drop-in library (native threads: scripts/callers_loop.c), with and without the call combiner (USEARCH_AMD_COALESCE, csrc/combiner.hpp: calls that arrive
while a launch is in flight go out together in the next one; with a window, the launcher waits that long at most for the callers of
the launch that just finished to call again). Calls per second and milliseconds per call at T = 1, 16, 64.

    GPU_MAX_HW_QUEUES=16 python scripts/coalesce_check.py
"""
import ctypes as C
import os
import subprocess
import sys

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
    helper = "/tmp/usearch_amd_callers_loop.so"  # native caller threads: Python's would pass the interpreter lock around between calls
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(ROOT, "scripts", "callers_loop.c"), "-o", helper])
    native = C.CDLL(helper)
    native.callers_loop.restype = C.c_double
    native.callers_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                    C.POINTER(C.c_int)]
    search_pointer = C.cast(L.usearch_search, C.c_void_p)
    for coalesce, window in ((0, 0), (1, 0), (1, 200), (1, 500)):
        os.environ["USEARCH_AMD_COALESCE"] = str(coalesce)  # both read when the index is created
        os.environ["USEARCH_AMD_COALESCE_WINDOW_US"] = str(window)
        err = C.c_char_p()
        options = Options(1, None, 3, dim, 16, 128, 64, False)  # cos, f16
        index = L.usearch_init(C.byref(options), C.byref(err))
        L.usearch_view_buffer(index, C.c_void_p(image.ctypes.data), image.size, C.byref(err))
        assert not err.value, err.value
        L.usearch_change_threads_search(index, 64, C.byref(err))
        for ef in (608, 64):
            L.usearch_change_expansion_search(index, ef, C.byref(err))
            for threads in (1, 16, 64):
                per_thread = 48 if ef == 608 else 192
                failures = C.c_int(0)
                native.callers_loop(search_pointer, index, C.c_void_p(queries.ctypes.data), dim * 2, 4096, 3, 10, 1, 1, C.byref(failures))
                seconds = native.callers_loop(search_pointer, index, C.c_void_p(queries.ctypes.data), dim * 2, 4096, 3, 10, threads, per_thread,
                                              C.byref(failures))
                failures = [failures.value] if failures.value else []
                assert not failures, failures[:1]
                print(f"coalesce={coalesce} window={window}us ef={ef} threads={threads}: {threads * per_thread / seconds:,.0f} calls per second "
                      f"({seconds / per_thread * 1e3:.2f} ms per call and thread)", flush=True)
        L.usearch_free(index, C.byref(err))


if __name__ == "__main__":
    main()
