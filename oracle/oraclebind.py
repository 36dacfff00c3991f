"""ctypes binding over `oracle/_build/libusearch_oracle.so` — the plain-C restatement (`oracle/usearch_oracle.c`).

TEST INFRASTRUCTURE. Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB_PATH = os.path.join(HERE, "_build", "libusearch_oracle.so")

# on-disk enum values (index_plugins.hpp:113-159)
METRIC = {"ip": ord("i"), "cos": ord("c"), "l2sq": ord("e"), "hamming": ord("b"), "pearson": ord("p"),
          "haversine": ord("h"), "divergence": ord("d"), "jaccard": ord("j"), "tanimoto": ord("t"), "sorensen": ord("s")}
SCALAR = {"b1": 1, "bf16": 4, "f64": 10, "f32": 11, "f16": 12, "i8": 23}
SCALAR_NAME = {v: k for k, v in SCALAR.items()}
METRIC_NAME = {v: k for k, v in METRIC.items()}


class _Index(C.Structure):
    _fields_ = [
        ("image", C.c_void_p), ("image_length", C.c_size_t),
        ("rows", C.c_uint64), ("cols", C.c_uint64), ("vectors", C.c_void_p),
        ("version", C.c_uint16 * 3),
        ("metric_kind", C.c_uint8), ("scalar_kind", C.c_uint8), ("key_kind", C.c_uint8), ("slot_kind", C.c_uint8),
        ("count_present", C.c_uint64), ("count_deleted", C.c_uint64), ("dimensions", C.c_uint64),
        ("multi", C.c_uint8),
        ("size", C.c_uint64), ("connectivity", C.c_uint64), ("connectivity_base", C.c_uint64),
        ("max_level", C.c_uint64), ("entry_slot", C.c_uint64),
        ("levels", C.c_void_p), ("node_offsets", C.c_void_p),
    ]


def build() -> str:
    """Compile the restatement with gcc (seconds). Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return ORACLE_LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "usearch_oracle.c")
        if not os.path.exists(ORACLE_LIB_PATH) or os.path.getmtime(ORACLE_LIB_PATH) < os.path.getmtime(src):
            build()
        _lib = C.CDLL(ORACLE_LIB_PATH, mode=os.RTLD_LOCAL)
        L = _lib
        L.uo_open.restype = C.c_int
        L.uo_open.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p)]
        L.uo_close.argtypes = [C.POINTER(_Index)]
        L.uo_bytes_per_vector.restype = C.c_size_t
        L.uo_bytes_per_vector.argtypes = [C.c_uint8, C.c_uint64]
        L.uo_neighbors.restype = C.c_uint32
        L.uo_neighbors.argtypes = [C.POINTER(_Index), C.c_uint64, C.c_int, C.c_void_p, C.c_uint32]
        L.uo_key.restype = C.c_uint64
        L.uo_key.argtypes = [C.POINTER(_Index), C.c_uint64]
        L.uo_level.restype = C.c_int
        L.uo_level.argtypes = [C.POINTER(_Index), C.c_uint64]
        L.uo_distance.restype = C.c_float
        L.uo_distance.argtypes = [C.c_uint8, C.c_uint8, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.uo_cast.restype = C.c_int
        L.uo_cast.argtypes = [C.c_uint8, C.c_uint8, C.c_void_p, C.c_uint64, C.c_void_p]
        L.uo_search_many.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_uint8, C.c_size_t, C.c_size_t, C.c_size_t,
                                     C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        L.uo_search.restype = C.c_size_t
        L.uo_search.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_uint8, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.uo_cluster_many.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_uint8, C.c_size_t, C.c_size_t, C.c_size_t,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.uo_set_frontier_in_top.argtypes = [C.c_int]
        L.uo_merge_into.restype = C.c_size_t
        L.uo_merge_into.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_size_t]
    return _lib


FILTER_T = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_void_p)


def _ptr(a: Optional[np.ndarray]):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class OracleIndex:
    """Parsed view of a serialized `.usearch` v2 image; searches it with the C restatement."""

    def __init__(self, image: np.ndarray):
        self.image = np.ascontiguousarray(image, dtype=np.uint8)
        self.ix = _Index()
        err = C.c_char_p()
        if lib().uo_open(C.byref(self.ix), _ptr(self.image), len(self.image), C.byref(err)):
            raise RuntimeError(f"oracle uo_open: {err.value.decode()}")

    def __del__(self):
        try:
            lib().uo_close(C.byref(self.ix))
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self.ix.size)

    @property
    def ndim(self) -> int:
        return int(self.ix.dimensions)

    @property
    def dtype(self) -> str:
        return SCALAR_NAME[self.ix.scalar_kind]

    @property
    def metric(self) -> str:
        return METRIC_NAME[self.ix.metric_kind]

    def neighbors(self, slot: int, level: int = 0) -> np.ndarray:
        out = np.zeros(4096, dtype=np.uint32)
        n = lib().uo_neighbors(C.byref(self.ix), slot, level, _ptr(out), 4096)
        return out[:n].copy()

    def key(self, slot: int) -> int:
        return int(lib().uo_key(C.byref(self.ix), slot))

    def level(self, slot: int) -> int:
        return int(lib().uo_level(C.byref(self.ix), slot))

    def search(self, queries: np.ndarray, k: int, dtype: Optional[str] = None, expansion: int = 64,
               exact: bool = False, lanes: int = 0, frontier_in_top: bool = False):
        """→ (keys[Q,k], distances[Q,k], counts[Q], visited[Q], computed[Q]) like RefIndex.search.
        `frontier_in_top`: the device kernels' heap-less frontier (applied where the engine applies it, `uo_set_frontier_in_top`);
        the default is the reference's heap."""
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
            lib().uo_set_frontier_in_top(int(frontier_in_top))
            try:
                lib().uo_search_many(C.byref(self.ix), _ptr(queries), SCALAR[dtype], q, queries.strides[0], k, expansion,
                                     int(exact), lanes, _ptr(keys), _ptr(dists), _ptr(counts), _ptr(visited),
                                     _ptr(computed))
            finally:
                lib().uo_set_frontier_in_top(0)
        return keys, dists, counts, visited, computed

    def cluster(self, queries: np.ndarray, level: int, dtype: Optional[str] = None, lanes: int = 0):
        """`index_dense_gt::cluster(query, level)` for a batch → (keys[Q], distances[Q], visited[Q], computed[Q])."""
        dtype = dtype or self.dtype
        queries = np.ascontiguousarray(queries)
        q = len(queries)
        keys = np.zeros(q, dtype=np.uint64)
        dists = np.zeros(q, dtype=np.float32)
        visited = np.zeros(q, dtype=np.uint64)
        computed = np.zeros(q, dtype=np.uint64)
        if q:
            lib().uo_cluster_many(C.byref(self.ix), _ptr(queries), SCALAR[dtype], q, queries.strides[0], level, lanes,
                                  _ptr(keys), _ptr(dists), _ptr(visited), _ptr(computed))
        return keys, dists, visited, computed

    def filtered_search(self, query: np.ndarray, k: int, predicate, dtype: Optional[str] = None,
                        expansion: int = 64, lanes: int = 0, exact: bool = False, counters: bool = False):
        """→ (found, keys, distances) [+ (visited_members, computed_distances) with `counters`]; `exact`: the brute-force scan
        under the predicate (index.hpp:4252-4268)."""
        dtype = dtype or self.dtype
        query = np.ascontiguousarray(query)
        keys = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        visited, computed = C.c_uint64(0), C.c_uint64(0)
        cb = FILTER_T(lambda key, _state: int(bool(predicate(int(key)))))
        n = lib().uo_search(C.byref(self.ix), _ptr(query), SCALAR[dtype], k, expansion, int(exact), lanes,
                            C.cast(cb, C.c_void_p), None, _ptr(keys), _ptr(dists), C.byref(visited), C.byref(computed))
        if counters:
            return int(n), keys, dists, int(visited.value), int(computed.value)
        return int(n), keys, dists


def distance(a: np.ndarray, b: np.ndarray, metric: str, dtype: str, ndim: int, lanes: int = 0) -> float:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return float(lib().uo_distance(METRIC[metric], SCALAR[dtype], _ptr(a), _ptr(b), ndim, lanes))


def cast(vector: np.ndarray, from_dtype: str, to_dtype: str, ndim: int) -> Optional[np.ndarray]:
    vector = np.ascontiguousarray(vector)
    nbytes = lib().uo_bytes_per_vector(SCALAR[to_dtype], ndim)
    out = np.zeros(nbytes, dtype=np.uint8)
    done = lib().uo_cast(SCALAR[from_dtype], SCALAR[to_dtype], _ptr(vector), ndim, _ptr(out))
    return out if done else None


def merge_into(keys: np.ndarray, dists: np.ndarray, old_count: int, new_keys: np.ndarray, new_dists: np.ndarray,
               new_count: int) -> int:
    assert keys.dtype == np.uint64 and dists.dtype == np.float32
    new_keys = np.ascontiguousarray(new_keys, dtype=np.uint64)
    new_dists = np.ascontiguousarray(new_dists, dtype=np.float32)
    return int(lib().uo_merge_into(_ptr(keys), _ptr(dists), old_count, len(keys), _ptr(new_keys), _ptr(new_dists),
                                   new_count))
