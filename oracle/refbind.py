"""ctypes binding over `oracle/_ref/libusearch_ref.so` — the REAL reference compiled by `oracle/Makefile`.

TEST INFRASTRUCTURE. Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product (`usearch_amd/`) never does. The library is the reference's own `c/lib.cpp` (38 `usearch_*`
symbols, `c/usearch.h:116-481`) plus the batch driver `oracle/ref_ext.cpp` (`uref_*`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB_PATH = os.path.join(HERE, "_ref", "libusearch_ref.so")

# c/usearch.h:40-62
METRIC = {"cos": 1, "ip": 2, "l2sq": 3, "haversine": 4, "divergence": 5, "pearson": 6, "jaccard": 7, "hamming": 8,
          "tanimoto": 9, "sorensen": 10}
SCALAR = {"f32": 1, "f64": 2, "f16": 3, "i8": 4, "b1": 5, "bf16": 6}
NP_DTYPE = {"f32": np.float32, "f64": np.float64, "f16": np.float16, "i8": np.int8, "b1": np.uint8,
            "bf16": np.uint16}  # numpy has no bfloat16: bf16 rows travel as their uint16 bit patterns


class InitOptions(C.Structure):
    # c/usearch.h:64-110
    _fields_ = [
        ("metric_kind", C.c_int),
        ("metric", C.c_void_p),
        ("quantization", C.c_int),
        ("dimensions", C.c_size_t),
        ("connectivity", C.c_size_t),
        ("expansion_add", C.c_size_t),
        ("expansion_search", C.c_size_t),
        ("multi", C.c_bool),
    ]


FILTER_T = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_void_p)  # int (*)(usearch_key_t key, void* state), c/usearch.h:392-395


def available() -> bool:
    return os.path.exists(REF_LIB_PATH)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{REF_LIB_PATH} missing: run `make -C oracle ref` where /root/reference is mounted")
        _lib = C.CDLL(REF_LIB_PATH, mode=os.RTLD_LOCAL)
        L = _lib
        err_p = C.POINTER(C.c_char_p)
        L.usearch_init.restype = C.c_void_p
        L.usearch_init.argtypes = [C.POINTER(InitOptions), err_p]
        L.usearch_free.argtypes = [C.c_void_p, err_p]
        for name in ("usearch_size", "usearch_capacity", "usearch_dimensions", "usearch_connectivity",
                     "usearch_expansion_add", "usearch_expansion_search", "usearch_serialized_length",
                     "usearch_memory_usage"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_void_p, err_p]
        L.usearch_reserve.argtypes = [C.c_void_p, C.c_size_t, err_p]
        L.usearch_change_expansion_search.argtypes = [C.c_void_p, C.c_size_t, err_p]
        L.usearch_change_threads_search.argtypes = [C.c_void_p, C.c_size_t, err_p]
        L.usearch_change_threads_add.argtypes = [C.c_void_p, C.c_size_t, err_p]
        L.usearch_save.argtypes = [C.c_void_p, C.c_char_p, err_p]
        L.usearch_load.argtypes = [C.c_void_p, C.c_char_p, err_p]
        L.usearch_view.argtypes = [C.c_void_p, C.c_char_p, err_p]
        L.usearch_save_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_p]
        L.usearch_load_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_p]
        L.usearch_view_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_p]
        L.usearch_add.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, err_p]
        L.usearch_remove.restype = C.c_size_t
        L.usearch_remove.argtypes = [C.c_void_p, C.c_uint64, err_p]
        L.usearch_search.restype = C.c_size_t
        L.usearch_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, err_p]
        L.usearch_distance.restype = C.c_float
        L.usearch_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, err_p]
        L.usearch_exact_search.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                           C.c_int, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p,
                                           C.c_size_t, C.c_void_p, C.c_size_t, err_p]
        L.uref_max_threads.restype = C.c_int
        L.uref_cluster_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.uref_add_many.restype = C.c_size_t
        L.uref_add_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
        L.uref_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                       C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.uref_graph_shape.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    return _lib


def _check(err: C.c_char_p, what: str) -> None:
    if err.value:
        raise RuntimeError(f"reference {what}: {err.value.decode()}")


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class RefIndex:
    """The reference `index_dense_t` behind its own C ABI (`c/lib.cpp:136-507`)."""

    def __init__(self, ndim: int = 0, metric: str = "cos", dtype: str = "f32", connectivity: int = 16,
                 expansion_add: int = 128, expansion_search: int = 64, multi: bool = False, empty: bool = False):
        L = lib()
        err = C.c_char_p()
        if empty:
            self.handle = L.usearch_init(None, C.byref(err))
        else:
            opts = InitOptions(METRIC[metric], None, SCALAR[dtype], ndim, connectivity, expansion_add,
                               expansion_search, multi)
            self.handle = L.usearch_init(C.byref(opts), C.byref(err))
        _check(err, "usearch_init")
        self.dtype = dtype
        self._keepalive = None

    def close(self) -> None:
        if getattr(self, "handle", None):
            err = C.c_char_p()
            lib().usearch_free(self.handle, C.byref(err))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        err = C.c_char_p()
        return lib().usearch_size(self.handle, C.byref(err))

    @property
    def ndim(self) -> int:
        err = C.c_char_p()
        return lib().usearch_dimensions(self.handle, C.byref(err))

    @property
    def connectivity(self) -> int:
        err = C.c_char_p()
        return lib().usearch_connectivity(self.handle, C.byref(err))

    @property
    def expansion_search(self) -> int:
        err = C.c_char_p()
        return lib().usearch_expansion_search(self.handle, C.byref(err))

    @expansion_search.setter
    def expansion_search(self, ef: int) -> None:
        err = C.c_char_p()
        lib().usearch_change_expansion_search(self.handle, ef, C.byref(err))
        _check(err, "change_expansion_search")

    def reserve(self, n: int) -> None:
        err = C.c_char_p()
        lib().usearch_reserve(self.handle, n, C.byref(err))
        _check(err, "usearch_reserve")

    def add(self, keys: np.ndarray, vectors: np.ndarray, dtype: Optional[str] = None, threads: int = 0) -> int:
        """Batch add (OpenMP, `cpp/bench.cpp:329-350` loop shape). threads=1 gives a deterministic graph."""
        dtype = dtype or self.dtype
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        vectors = np.ascontiguousarray(vectors)
        assert vectors.ndim == 2 and len(keys) == len(vectors)
        self.reserve(len(self) + len(keys))
        return lib().uref_add_many(self.handle, _ptr(keys), _ptr(vectors), SCALAR[dtype], len(keys),
                                   vectors.strides[0], threads)

    def remove(self, key: int) -> int:
        err = C.c_char_p()
        n = lib().usearch_remove(self.handle, key, C.byref(err))
        _check(err, "usearch_remove")
        return n

    def search(self, queries: np.ndarray, k: int, dtype: Optional[str] = None, exact: bool = False,
               threads: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """Batched search → (keys[Q,k] u64, distances[Q,k] f32, counts[Q], visited[Q], computed[Q])."""
        dtype = dtype or self.dtype
        queries = np.ascontiguousarray(queries)
        if queries.ndim == 1:
            queries = queries[None, :]
        q = len(queries)
        keys = np.zeros((q, k), dtype=np.uint64)
        dists = np.zeros((q, k), dtype=np.float32)
        counts = np.zeros(q, dtype=np.uint64)
        visited = np.zeros(q, dtype=np.uint64)
        computed = np.zeros(q, dtype=np.uint64)
        if q:
            lib().uref_search_many(self.handle, _ptr(queries), SCALAR[dtype], q, queries.strides[0], k, int(exact),
                                   threads, _ptr(keys), _ptr(dists), _ptr(counts), _ptr(visited), _ptr(computed))
        return keys, dists, counts, visited, computed

    def cluster(self, queries: np.ndarray, level: int, dtype: Optional[str] = None, threads: int = 0):
        """Batched `index_dense_gt::cluster(query, level)` → (keys[Q], distances[Q], visited[Q], computed[Q])."""
        dtype = dtype or self.dtype
        queries = np.ascontiguousarray(queries)
        q = len(queries)
        keys = np.zeros(q, dtype=np.uint64)
        dists = np.zeros(q, dtype=np.float32)
        visited = np.zeros(q, dtype=np.uint64)
        computed = np.zeros(q, dtype=np.uint64)
        if q:
            lib().uref_cluster_many(self.handle, _ptr(queries), SCALAR[dtype], q, queries.strides[0], level, threads,
                                    _ptr(keys), _ptr(dists), _ptr(visited), _ptr(computed))
        return keys, dists, visited, computed

    def search_one(self, query: np.ndarray, k: int, dtype: Optional[str] = None):
        """The C-ABI hot signature itself: `usearch_search` (`c/usearch.h:371-374`)."""
        dtype = dtype or self.dtype
        query = np.ascontiguousarray(query)
        keys = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        err = C.c_char_p()
        found = lib().usearch_search(self.handle, _ptr(query), SCALAR[dtype], k, _ptr(keys), _ptr(dists),
                                     C.byref(err))
        _check(err, "usearch_search")
        return found, keys, dists

    def filtered_search(self, query: np.ndarray, k: int, predicate, dtype: Optional[str] = None):
        """`usearch_filtered_search` (`c/usearch.h:392-395`): `predicate(key) -> bool` decides what may be returned."""
        dtype = dtype or self.dtype
        query = np.ascontiguousarray(query)
        keys = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        callback = FILTER_T(lambda key, _state: int(bool(predicate(int(key)))))
        err = C.c_char_p()
        call = lib().usearch_filtered_search
        call.restype = C.c_size_t
        call.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, FILTER_T, C.c_void_p, C.c_void_p, C.c_void_p,
                         C.POINTER(C.c_char_p)]
        found = call(self.handle, _ptr(query), SCALAR[dtype], k, callback, None, _ptr(keys), _ptr(dists), C.byref(err))
        _check(err, "usearch_filtered_search")
        return int(found), keys, dists

    def save_buffer(self) -> np.ndarray:
        err = C.c_char_p()
        n = lib().usearch_serialized_length(self.handle, C.byref(err))
        buf = np.zeros(n, dtype=np.uint8)
        lib().usearch_save_buffer(self.handle, _ptr(buf), n, C.byref(err))
        _check(err, "usearch_save_buffer")
        return buf

    def save(self, path: str) -> None:
        err = C.c_char_p()
        lib().usearch_save(self.handle, path.encode(), C.byref(err))
        _check(err, "usearch_save")

    @classmethod
    def from_buffer(cls, buf: np.ndarray, view: bool = False, dtype: str = "f32") -> "RefIndex":
        self = cls(empty=True)
        self.dtype = dtype
        err = C.c_char_p()
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if view:
            self._keepalive = buf
            lib().usearch_view_buffer(self.handle, _ptr(buf), len(buf), C.byref(err))
        else:
            lib().usearch_load_buffer(self.handle, _ptr(buf), len(buf), C.byref(err))
        _check(err, "usearch_load_buffer")
        return self

    def graph_shape(self):
        ml, m, m0 = C.c_uint64(), C.c_uint64(), C.c_uint64()
        per_level = np.zeros(32, dtype=np.uint64)
        lib().uref_graph_shape(self.handle, C.byref(ml), C.byref(m), C.byref(m0), _ptr(per_level))
        return ml.value, m.value, m0.value, per_level[: ml.value + 1].copy()


def distance(a: np.ndarray, b: np.ndarray, metric: str, dtype: str, ndim: int) -> float:
    """`usearch_distance` (`c/lib.cpp:458-466`) → `metric_punned_t::operator()`."""
    err = C.c_char_p()
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return float(lib().usearch_distance(_ptr(a), _ptr(b), SCALAR[dtype], ndim, METRIC[metric], C.byref(err)))


def metadata(buffer: np.ndarray) -> dict:
    """`usearch_metadata_buffer` (c/usearch.h:223-224): what the REFERENCE reads from the head of a serialized image."""
    options = InitOptions()
    err = C.c_char_p()
    buffer = np.ascontiguousarray(buffer, dtype=np.uint8)
    L = lib()
    L.usearch_metadata_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(InitOptions), C.POINTER(C.c_char_p)]
    L.usearch_metadata_buffer(_ptr(buffer), len(buffer), C.byref(options), C.byref(err))
    _check(err, "usearch_metadata_buffer")
    return {"metric_kind": options.metric_kind, "quantization": options.quantization, "dimensions": options.dimensions,
            "connectivity": options.connectivity, "multi": bool(options.multi)}


def max_threads() -> int:
    return int(lib().uref_max_threads())
