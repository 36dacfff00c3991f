"""Python host side of the MI355X search engine: a thin ctypes binding over the C ABI of `include/usearch_amd.h`.

It mirrors the *search* surface of the reference's Python package (`/root/reference/python/usearch/index.py`):
`Index.restore(path_or_buffer)` (index.py:574-630) gives an object whose `search(vectors, count, ...)` has the argument
meaning of `Index.search` (index.py:700-748) and returns `Matches` / `BatchMatches` shaped like index.py:291-385 —
row-major `keys[Q, k]` (u64), `distances[Q, k]` (f32), `counts[Q]`, plus the two traversal counters. Everything else of
that package (add / remove / cluster / join …) stays with the reference: this engine consumes the index files it writes.

There is no CPU fallback: without `libusearch_amd.so` or without a HIP device, constructing an `Index` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Union

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# USEARCH_AMD_LIBRARY names an experimental build of the same library (scripts/ A/B runs); the product is the in-tree one
LIBRARY_PATH = os.environ.get("USEARCH_AMD_LIBRARY") or os.path.join(HERE, "lib", "libusearch_amd.so")

# `usearch_scalar_kind_t` / `usearch_metric_kind_t` of c/usearch.h:40-62
SCALAR_KINDS = {"f32": 1, "f64": 2, "f16": 3, "i8": 4, "b1": 5, "bf16": 6}
SCALAR_NAMES = {v: k for k, v in SCALAR_KINDS.items()}
METRIC_NAMES = {1: "cos", 2: "ip", 3: "l2sq", 4: "haversine", 5: "divergence", 6: "pearson", 7: "jaccard",
                8: "hamming", 9: "tanimoto", 10: "sorensen"}
NUMPY_DTYPES = {"f32": np.float32, "f64": np.float64, "f16": np.float16, "i8": np.int8, "b1": np.uint8,
                "bf16": np.uint16}  # numpy has no bfloat16: bf16 rows are handed over as their uint16 bit patterns


def _infer_dtype(array: np.ndarray) -> str:
    """Scalar kind of a numpy array when the caller did not name it: `uint8` means bit-packed `b1` rows; `bf16` has no
    numpy dtype and must always be named (`dtype="bf16"` with a `uint16` array)."""
    dtype = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64", np.dtype(np.float16): "f16",
             np.dtype(np.int8): "i8", np.dtype(np.uint8): "b1"}.get(array.dtype)
    if dtype is None:
        raise ValueError(f"Unsupported dtype {array.dtype}")
    return dtype


class Tuning(C.Structure):
    """`usearch_amd_tuning_t`."""
    _fields_ = [("hash_cap", C.c_uint32), ("next_cap", C.c_uint32), ("variant", C.c_uint32), ("mode", C.c_uint32),
                ("waves_per_cu", C.c_uint32), ("frontier", C.c_uint32), ("wave_clock", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    """`usearch_amd_stats_t`."""
    _fields_ = [("passes", C.c_uint32), ("retried_lds", C.c_uint32), ("retried_global", C.c_uint32),
                ("kernel_ms", C.c_float), ("mode", C.c_uint32), ("grid", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("frontier", C.c_uint32), ("variant", C.c_uint32), ("tail_idle", C.c_float), ("span_ms", C.c_float),
                ("top_cells", C.c_uint32), ("probe_mode", C.c_uint32), ("seen_cells", C.c_uint32), ("claim_bits", C.c_uint32),
                ("early_rows", C.c_uint32), ("plain", C.c_uint32), ("aside_cells", C.c_uint32)]


class Arrays(C.Structure):
    """`usearch_amd_arrays_t`: device pointers and shapes of a snapshot's HBM arrays."""
    _fields_ = [("vectors", C.c_void_p), ("level0", C.c_void_p), ("keys", C.c_void_p), ("size", C.c_uint64),
                ("row_stride", C.c_uint32), ("level0_cells", C.c_uint32), ("device", C.c_int), ("reserved", C.c_uint32)]


class BuildConfig(C.Structure):
    """`usearch_amd_build_config_t`; zeros = the reference's defaults."""
    _fields_ = [("connectivity", C.c_uint32), ("connectivity_base", C.c_uint32), ("expansion_add", C.c_uint32),
                ("batch_divisor", C.c_uint32), ("max_batch", C.c_uint32), ("reserved", C.c_uint32),
                ("seed", C.c_uint64)]


class BuildStats(C.Structure):
    """`usearch_amd_build_stats_t`."""
    _fields_ = [("batches", C.c_uint64), ("passes", C.c_uint64), ("search_distances", C.c_uint64),
                ("search_hops", C.c_uint64), ("select_distances", C.c_uint64), ("reverse_distances", C.c_uint64),
                ("repruned_lists", C.c_uint64), ("dropped_requests", C.c_uint64), ("seconds_total", C.c_double),
                ("seconds_search", C.c_double), ("seconds_link", C.c_double), ("seconds_upload", C.c_double),
                ("max_level", C.c_uint32), ("reserved", C.c_uint32), ("refiled_requests", C.c_uint64)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


_library = None

EXPORTED_SYMBOLS = [
    "usearch_amd_device_count", "usearch_amd_snapshot_from_buffer", "usearch_amd_snapshot_from_file",
    "usearch_amd_snapshot_from_parts",
    "usearch_amd_snapshot_free", "usearch_amd_snapshot_size", "usearch_amd_snapshot_dimensions",
    "usearch_amd_snapshot_connectivity", "usearch_amd_snapshot_max_level", "usearch_amd_snapshot_bytes_per_vector",
    "usearch_amd_snapshot_row_stride", "usearch_amd_snapshot_device_bytes", "usearch_amd_snapshot_placement", "usearch_amd_snapshot_arrays",
    "usearch_amd_note_device_free", "usearch_amd_settle", "usearch_amd_snapshot_settle_ms", "usearch_amd_snapshot_tune",
    "usearch_amd_snapshot_scalar_kind",
    "usearch_amd_snapshot_metric_kind", "usearch_amd_snapshot_lanes_per_row", "usearch_amd_snapshot_inline_rows",
    "usearch_amd_search_many",
    "usearch_amd_search_many_device", "usearch_amd_last_peaks", "usearch_amd_distances",
    "usearch_amd_last_distances_ms", "usearch_amd_merge_many", "usearch_amd_merge_many_device",
    "usearch_amd_exact_search_many", "usearch_amd_exact_search_many_tiled", "usearch_amd_exact_search_dataset",
    "usearch_amd_exact_search_many_device",
    "usearch_amd_cluster_many",
    "usearch_amd_filter_from_key_range", "usearch_amd_filter_from_keys", "usearch_amd_filter_from_bits",
    "usearch_amd_filter_allowed", "usearch_amd_filter_device_bits", "usearch_amd_filter_free",
    "usearch_amd_filtered_search_many", "usearch_amd_filtered_search_many_device", "usearch_amd_filtered_exact_search_many",
    "usearch_amd_cast",
    "usearch_amd_build", "usearch_amd_build_free", "usearch_amd_build_snapshot",
    "usearch_amd_build_serialized_length", "usearch_amd_build_save_buffer", "usearch_amd_build_stats",
    # sharded search across GPUs (usearch_amd/sharded.py binds these)
    "usearch_amd_comm_unique_id", "usearch_amd_comm_init_rccl", "usearch_amd_comm_init_custom", "usearch_amd_comm_free",
    "usearch_amd_comm_rank", "usearch_amd_comm_world", "usearch_amd_comm_broadcast", "usearch_amd_sharded_search_many",
]


def _share_hip_runtime() -> None:
    """One HIP runtime per process. PyTorch wheels bundle their own `libamdhip64.so` and `libtorch_hip.so` asks for it by that
    unversioned file name, so the loader does not recognise a system runtime that is already mapped (its SONAME is
    `libamdhip64.so.7`) and maps torch's copy as a second runtime — which then finds no device, the first one holding the
    driver. Mapping torch's copy first makes both sides resolve to it: this library asks for the SONAME, and torch's copy
    carries it. Only the path is looked up, torch is not imported; without torch there is nothing to share."""
    try:
        with open("/proc/self/maps") as maps:
            if "libamdhip64" in maps.read():
                return  # a runtime is mapped already (torch imported first, or the caller's own)
    except OSError:
        pass
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=os.RTLD_GLOBAL)


def library() -> C.CDLL:
    """Loads `usearch_amd/lib/libusearch_amd.so` (built by `make -C usearch_amd/csrc` / `__graft_entry__.build()`)."""
    global _library
    if _library is not None:
        return _library
    if not os.path.exists(LIBRARY_PATH):
        raise RuntimeError(f"{LIBRARY_PATH} is missing: build it with `make -C usearch_amd/csrc` "
                           "(hipcc, gfx950). The engine has no CPU fallback.")
    _share_hip_runtime()
    L = C.CDLL(LIBRARY_PATH, mode=os.RTLD_LOCAL)
    err_p = C.POINTER(C.c_char_p)
    L.usearch_amd_device_count.restype = C.c_int
    L.usearch_amd_device_count.argtypes = [err_p]
    L.usearch_amd_snapshot_from_buffer.restype = C.c_void_p
    L.usearch_amd_snapshot_from_buffer.argtypes = [C.c_void_p, C.c_size_t, C.c_int, err_p]
    L.usearch_amd_snapshot_from_parts.restype = C.c_void_p
    L.usearch_amd_snapshot_from_parts.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, err_p]
    L.usearch_amd_snapshot_from_file.restype = C.c_void_p
    L.usearch_amd_snapshot_from_file.argtypes = [C.c_char_p, C.c_int, err_p]
    L.usearch_amd_snapshot_free.argtypes = [C.c_void_p, err_p]
    for name in ("size", "dimensions", "connectivity", "max_level", "bytes_per_vector", "row_stride", "device_bytes",
                 "lanes_per_row", "inline_rows"):
        f = getattr(L, f"usearch_amd_snapshot_{name}")
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p]
    L.usearch_amd_note_device_free.restype = None
    L.usearch_amd_note_device_free.argtypes = []
    L.usearch_amd_snapshot_tune.restype = C.c_uint32
    L.usearch_amd_snapshot_tune.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, err_p]
    L.usearch_amd_settle.restype = C.c_float
    L.usearch_amd_settle.argtypes = []
    L.usearch_amd_snapshot_settle_ms.restype = C.c_float
    L.usearch_amd_snapshot_settle_ms.argtypes = [C.c_void_p]
    L.usearch_amd_snapshot_placement.restype = None
    L.usearch_amd_snapshot_placement.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                                 C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.usearch_amd_snapshot_arrays.restype = None
    L.usearch_amd_snapshot_arrays.argtypes = [C.c_void_p, C.POINTER(Arrays)]
    for name in ("scalar_kind", "metric_kind"):
        f = getattr(L, f"usearch_amd_snapshot_{name}")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
    L.usearch_amd_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                          C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(Tuning), C.POINTER(Stats), err_p]
    L.usearch_amd_search_many_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                                 C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.POINTER(Tuning), C.c_int,
                                                 C.POINTER(Stats), err_p]
    L.usearch_amd_merge_many_device.argtypes = [C.c_void_p] * 3 + [C.c_size_t] * 3 + [C.c_void_p] * 4 + [err_p]
    L.usearch_amd_merge_many.argtypes = [C.c_void_p] * 3 + [C.c_size_t] * 3 + [C.c_void_p] * 3 + [err_p]
    L.usearch_amd_cluster_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, err_p]
    L.usearch_amd_exact_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), err_p]
    L.usearch_amd_exact_search_many_tiled.argtypes = L.usearch_amd_exact_search_many.argtypes
    L.usearch_amd_exact_search_many_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float),
                                                       C.POINTER(C.c_char_p)]
    L.usearch_amd_exact_search_dataset.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                   C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p,
                                                   C.c_size_t, C.c_void_p, C.c_size_t, err_p]
    L.usearch_amd_filter_from_key_range.restype = C.c_void_p
    L.usearch_amd_filter_from_key_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, err_p]
    L.usearch_amd_filter_from_keys.restype = C.c_void_p
    L.usearch_amd_filter_from_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, err_p]
    L.usearch_amd_filter_from_bits.restype = C.c_void_p
    L.usearch_amd_filter_from_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_p]
    L.usearch_amd_filter_allowed.restype = C.c_size_t
    L.usearch_amd_filter_allowed.argtypes = [C.c_void_p]
    L.usearch_amd_filter_device_bits.restype = C.c_void_p
    L.usearch_amd_filter_device_bits.argtypes = [C.c_void_p]
    L.usearch_amd_filter_free.argtypes = [C.c_void_p, err_p]
    L.usearch_amd_filtered_search_many.argtypes = [C.c_void_p, C.c_void_p] + L.usearch_amd_search_many.argtypes[1:]
    L.usearch_amd_filtered_search_many_device.argtypes = [C.c_void_p, C.c_void_p] + L.usearch_amd_search_many_device.argtypes[1:]
    L.usearch_amd_filtered_exact_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t,
                                                         C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                         C.POINTER(C.c_float), err_p]
    L.usearch_amd_last_peaks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_p]
    L.usearch_amd_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, err_p]
    L.usearch_amd_last_distances_ms.restype = C.c_float
    L.usearch_amd_last_distances_ms.argtypes = [C.c_void_p]
    L.usearch_amd_cast.restype = C.c_int
    L.usearch_amd_cast.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.usearch_amd_build.restype = C.c_void_p
    L.usearch_amd_build.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_void_p,
                                    C.POINTER(BuildConfig), C.c_int, C.c_int, err_p]
    L.usearch_amd_build_free.argtypes = [C.c_void_p, err_p]
    L.usearch_amd_build_snapshot.restype = C.c_void_p
    L.usearch_amd_build_snapshot.argtypes = [C.c_void_p]
    L.usearch_amd_build_serialized_length.restype = C.c_size_t
    L.usearch_amd_build_serialized_length.argtypes = [C.c_void_p]
    L.usearch_amd_build_save_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_p]
    L.usearch_amd_build_stats.argtypes = [C.c_void_p, C.POINTER(BuildStats)]
    _library = L
    return L


def _raise(err: C.c_char_p, what: str) -> None:
    if err.value:
        raise RuntimeError(f"{what}: {err.value.decode()}")


def _pointer(array: Optional[np.ndarray]):
    return C.c_void_p(array.ctypes.data) if array is not None else None


@dataclass
class Matches:
    """Results of one query — `usearch.index.Matches` (index.py:303-331)."""
    keys: np.ndarray
    distances: np.ndarray
    visited_members: int = 0
    computed_distances: int = 0

    def __len__(self) -> int:
        return len(self.keys)

    def to_list(self):
        return [(int(k), float(d)) for k, d in zip(self.keys, self.distances)]


@dataclass
class BatchMatches:
    """Results of a batch — `usearch.index.BatchMatches` (index.py:334-385); counters kept per query as well."""
    keys: np.ndarray
    distances: np.ndarray
    counts: np.ndarray
    visited_members: int = 0
    computed_distances: int = 0
    visited_per_query: Optional[np.ndarray] = None
    computed_per_query: Optional[np.ndarray] = None
    stats: Optional[Stats] = None

    def __len__(self) -> int:
        return len(self.counts)

    def __getitem__(self, index: int) -> Matches:
        if isinstance(index, int) and index < len(self):
            n = int(self.counts[index])
            return Matches(self.keys[index, :n], self.distances[index, :n],
                           int(self.visited_per_query[index]), int(self.computed_per_query[index]))
        raise IndexError(f"`index` must be an integer under {len(self)}")

    def count_matches(self, expected: np.ndarray, count: Optional[int] = None) -> int:
        assert len(expected) == len(self)
        count = self.keys.shape[1] if count is None else count
        if count == 1:
            return int(np.sum(self.keys[:, 0] == expected))
        return int(sum(expected[i] in self.keys[i, :count] for i in range(len(self))))

    def mean_recall(self, expected: np.ndarray, count: Optional[int] = None) -> float:
        return self.count_matches(expected, count=count) / len(expected)


class Filter:
    """A predicate over members as an HBM-resident bitmap (`usearch_amd_filter_t`): made once, reused by any number of batches.
    Stands in for the `predicate(key)` callable of `index_dense_gt::filtered_search` (index_dense.hpp:774-779)."""

    def __init__(self, handle: int, index: "Index"):
        self._handle = handle
        self._index = index  # keeps the snapshot alive

    @property
    def allowed(self) -> int:
        return int(library().usearch_amd_filter_allowed(self._handle))

    @property
    def device_bits(self) -> int:
        return int(library().usearch_amd_filter_device_bits(self._handle) or 0)

    def close(self) -> None:
        if getattr(self, "_handle", None):
            err = C.c_char_p()
            library().usearch_amd_filter_free(self._handle, C.byref(err))
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """An immutable HBM-resident snapshot of a serialized USearch index, searchable in batches on one MI355X."""

    def __init__(self, handle: int, expansion_search: int = 0, owner=None):
        self._handle = handle
        self._owner = owner  # a `BuiltIndex` whose builder owns the snapshot: keeps it alive, frees it
        self.expansion_search = expansion_search  # 0 = the reference's default of 64 (index.hpp:3029-3030)

    @classmethod
    def restore(cls, source: Union[str, os.PathLike, bytes, bytearray, memoryview, np.ndarray], device: int = 0,
                expansion_search: int = 0, vectors: Optional[np.ndarray] = None) -> "Index":
        """Uploads a `.usearch` file or an in-memory image (`usearch_save` / `usearch_save_buffer` output). `vectors`: the
        matrix the caller kept for an image saved with `exclude_vectors` (one row per member, in slot order)."""
        L = library()
        err = C.c_char_p()
        if vectors is not None:
            if isinstance(source, (str, os.PathLike)):  # the graph alone sits in the file: map it for the call
                source = np.memmap(os.fspath(source), dtype=np.uint8, mode="r")
            image = np.ascontiguousarray(np.frombuffer(source, dtype=np.uint8)
                                         if not isinstance(source, np.ndarray) else source, dtype=np.uint8)
            vectors = np.asarray(vectors)
            if vectors.ndim != 2 or vectors.strides[1] != vectors.itemsize:
                vectors = np.ascontiguousarray(vectors)
            handle = L.usearch_amd_snapshot_from_parts(_pointer(image), image.size, _pointer(vectors), vectors.strides[0],
                                                       device, C.byref(err))
        elif isinstance(source, (str, os.PathLike)):
            handle = L.usearch_amd_snapshot_from_file(os.fspath(source).encode(), device, C.byref(err))
        else:
            image = np.ascontiguousarray(np.frombuffer(source, dtype=np.uint8)
                                         if not isinstance(source, np.ndarray) else source, dtype=np.uint8)
            handle = L.usearch_amd_snapshot_from_buffer(_pointer(image), image.size, device, C.byref(err))
        _raise(err, "usearch_amd snapshot")
        if not handle:
            raise RuntimeError("usearch_amd snapshot: failed without a message")
        return cls(handle, expansion_search)

    @classmethod
    def restore_placed(cls, source, probe, draws: int = 8, device: int = 0, expansion_search: int = 0,
                       vectors: Optional[np.ndarray] = None, first: Optional["Index"] = None,
                       good_enough: Optional[float] = None, free_bytes=None):
        """`restore`, but choosing WHERE in HBM the index lands. The same index bytes at other physical addresses walk at one
        of two speeds (headline batch: 45.4 or 51.7 ms; stable for the life of an allocation, profiles/r02_placement.log), and
        the allocator decides which. So: upload the image up to `draws` times, time `probe(index) -> milliseconds` (a batch of
        the caller's own queries) on each, keep the fastest. Candidates that lost stay allocated until the end — the next one
        must land somewhere else — as long as `free_bytes()` (device memory still free) leaves room for one more; they are
        released before returning. `good_enough` (milliseconds): stop drawing once a candidate is at least that fast (a deployment
        that knows its fast speed). `first`: an index that is already resident (the builder's own) takes part as draw 0; it is
        never closed here. Returns (index, {"probe_ms": [...], "kept": i}) — `index is first` when nothing beat it."""
        candidates, timings = [], []
        if first is not None:
            candidates.append(first)
            timings.append(float(probe(first)))
        try:
            while len(timings) < max(1, draws):
                if timings and good_enough is not None and min(timings) <= good_enough:
                    break
                if candidates and free_bytes is not None and free_bytes() < 1.25 * candidates[0].memory_usage:
                    # no room next to the ones held: let go of every loser, keep drawing
                    keep = int(np.argmin(timings))
                    for i, candidate in enumerate(candidates):
                        if i != keep and candidate is not None and candidate is not first:
                            candidate.close()
                            candidates[i] = None
                    if free_bytes() < 1.25 * candidates[keep].memory_usage:
                        break
                try:
                    candidate = cls.restore(source, device, expansion_search, vectors)
                except RuntimeError:
                    if not candidates:
                        raise
                    break  # out of device memory next to the ones held: the best so far it is
                candidates.append(candidate)
                timings.append(float("inf"))  # in the list before the probe runs: a failing probe still gets it released
                timings[-1] = float(probe(candidate))
        except BaseException:
            for candidate in candidates:  # nothing is handed out on this path: every copy made here goes, `first` stays the caller's
                if candidate is not None and candidate is not first:
                    candidate.close()
            raise
        kept = int(np.argmin(timings))
        for i, candidate in enumerate(candidates):
            if i != kept and candidate is not None and candidate is not first:
                candidate.close()
        return candidates[kept], {"probe_ms": [round(t, 3) for t in timings], "kept": kept}

    def close(self) -> None:
        if getattr(self, "_handle", None):
            if getattr(self, "_owner", None) is None:
                err = C.c_char_p()
                library().usearch_amd_snapshot_free(self._handle, C.byref(err))
            self._handle = None
            self._owner = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def placement(self) -> dict:
        """How the engine placed the matrix of stored rows in HBM (csrc/placement.hpp): the trials made so far — each a fresh copy
        judged against the incumbent on a chip-filling launch's own first queries — how many of them moved the matrix, and both
        times of every trial; `draws == 0` before the first such launch and for arrays too small to bother."""
        draws, kept, probe_ms = C.c_uint32(), C.c_uint32(), C.c_float()
        rates, incumbents = (C.c_float * 8)(), (C.c_float * 8)()
        library().usearch_amd_snapshot_placement(self._handle, C.byref(draws), C.byref(kept), rates, incumbents, C.byref(probe_ms))
        shown = min(int(draws.value), 8)
        return {"settle_ms": round(float(library().usearch_amd_snapshot_settle_ms(self._handle)), 1),
                "draws": int(draws.value), "kept": int(kept.value), "probe_ms": round(float(probe_ms.value), 2),
                "judge_ms": [round(float(rates[i]), 3) for i in range(shown)],
                "incumbent_ms": [round(float(incumbents[i]), 3) for i in range(shown)]}

    def tune_device(self, queries_ptr: int, queries_count: int, queries_stride: int, count: int, expansion: int, max_trials: int = 8) -> int:
        """`usearch_amd_snapshot_tune`: places the matrix of stored rows by trial on a SAMPLE of the batches to come (device-resident
        queries in the storage kind, as for `search_device`): up to `max_trials` fresh copies timed against the incumbent on the
        sample's first queries at `expansion`, the faster stays. Explicit, synchronous, before serving; returns the trials made
        (`placement` has their times)."""
        err = C.c_char_p()
        made = library().usearch_amd_snapshot_tune(self._handle, C.c_void_p(queries_ptr), queries_count, queries_stride, count, expansion,
                                                   max_trials, C.byref(err))
        _raise(err, "usearch_amd_snapshot_tune")
        return int(made)

    @property
    def arrays(self) -> "Arrays":
        """The HBM-resident arrays (`usearch_amd_snapshot_arrays`): device pointers of the padded matrix, the level-0 lists and the
        keys, with their shapes — read-only, for a host's own kernels."""
        out = Arrays()
        library().usearch_amd_snapshot_arrays(self._handle, C.byref(out))
        return out

    # ---- diagnostics of the placement studies (scripts/placement_study.py, fragment_study.py): the probes live in the test-hooks
    #      library, over the arrays above
    def latency_probe(self, lists: bool = False) -> float:
        """Nanoseconds per DEPENDENT read of a random stored row (or, `lists`, of a random level-0 neighbour list)."""
        err, a = C.c_char_p(), self.arrays
        base, row = (a.level0, a.level0_cells * 4) if lists else (a.vectors, a.row_stride)
        value = test_hooks().usearch_amd_test_latency_probe(C.c_void_p(base), a.size * row, row, C.byref(err))
        _raise(err, "usearch_amd_test_latency_probe")
        return float(value)

    def translation_probe(self) -> float:
        """Million random 4-KB pages of the resident matrix touched per second (16 bytes each): the address-translation path."""
        err, a = C.c_char_p(), self.arrays
        rate = test_hooks().usearch_amd_test_translation_probe(C.c_void_p(a.vectors), a.size * a.row_stride, C.byref(err))
        _raise(err, "usearch_amd_test_translation_probe")
        return float(rate)

    def gather_probe(self, first_row: int = 0, rows: int = 0) -> float:
        """GB/s of a dependency-free gather of random stored rows among rows [first_row, first_row + rows) (`rows` = 0: to the end)."""
        err, a = C.c_char_p(), self.arrays
        if first_row >= a.size:
            return 0.0
        rows = a.size - first_row if not rows or first_row + rows > a.size else rows
        rate = test_hooks().usearch_amd_test_gather_probe(C.c_void_p((a.vectors or 0) + first_row * a.row_stride), rows * a.row_stride,
                                                          a.row_stride, C.byref(err))
        _raise(err, "usearch_amd_test_gather_probe")
        return float(rate)

    # ---- introspection (names of usearch.index.Index properties, index.py:1180-1300)
    def __len__(self) -> int:
        return library().usearch_amd_snapshot_size(self._handle)

    @property
    def size(self) -> int:
        return len(self)

    @property
    def ndim(self) -> int:
        return library().usearch_amd_snapshot_dimensions(self._handle)

    @property
    def connectivity(self) -> int:
        return library().usearch_amd_snapshot_connectivity(self._handle)

    @property
    def max_level(self) -> int:
        return library().usearch_amd_snapshot_max_level(self._handle)

    @property
    def dtype(self) -> str:
        return SCALAR_NAMES[library().usearch_amd_snapshot_scalar_kind(self._handle)]

    @property
    def metric_kind(self) -> str:
        return METRIC_NAMES[library().usearch_amd_snapshot_metric_kind(self._handle)]

    @property
    def bytes_per_vector(self) -> int:
        return library().usearch_amd_snapshot_bytes_per_vector(self._handle)

    @property
    def row_stride(self) -> int:
        return library().usearch_amd_snapshot_row_stride(self._handle)

    @property
    def memory_usage(self) -> int:
        return library().usearch_amd_snapshot_device_bytes(self._handle)

    @property
    def lanes_per_row(self) -> int:
        return library().usearch_amd_snapshot_lanes_per_row(self._handle)

    @property
    def inline_rows(self) -> bool:
        """Rows of ≤ 16 bytes are stored next to the neighbour lists (one contiguous read per hop)."""
        return bool(library().usearch_amd_snapshot_inline_rows(self._handle))

    @property
    def hardware_acceleration(self) -> str:
        return "gfx950"

    # ---- filters
    def filter_key_range(self, first: int, last: int) -> Filter:
        """Members whose key lies in [first, last]."""
        err = C.c_char_p()
        handle = library().usearch_amd_filter_from_key_range(self._handle, first, last, C.byref(err))
        _raise(err, "usearch_amd_filter_from_key_range")
        return Filter(handle, self)

    def filter_keys(self, keys, allow: bool = True) -> Filter:
        """Members whose key is (allow) / is not (deny list) among `keys`."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        err = C.c_char_p()
        handle = library().usearch_amd_filter_from_keys(self._handle, _pointer(keys), len(keys), int(allow), C.byref(err))
        _raise(err, "usearch_amd_filter_from_keys")
        return Filter(handle, self)

    def filter_bits(self, bits: np.ndarray) -> Filter:
        """The caller's bitmap: bit `s & 31` of word `s >> 5` = the member in slot `s` passes. A boolean array of one entry per
        slot is packed first."""
        bits = np.asarray(bits)
        if bits.dtype == np.bool_:
            padded = np.zeros((len(bits) + 31) // 32 * 32, dtype=np.uint8)
            padded[:len(bits)] = bits
            bits = np.packbits(padded.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint32)
        bits = np.ascontiguousarray(bits, dtype=np.uint32)
        err = C.c_char_p()
        handle = library().usearch_amd_filter_from_bits(self._handle, _pointer(bits), len(bits), C.byref(err))
        _raise(err, "usearch_amd_filter_from_bits")
        return Filter(handle, self)

    # ---- search
    def search(self, vectors: np.ndarray, count: int = 10, *, expansion: Optional[int] = None,
               dtype: Optional[str] = None, tuning: Optional[Tuning] = None,
               exact: Union[bool, str] = False, filter: Optional[Filter] = None) -> Union[Matches, BatchMatches]:
        """`Index.search` (index.py:700-748): one vector → `Matches`, a 2-D batch → `BatchMatches`.

        `dtype` names the scalar kind of `vectors` when numpy cannot tell (bit-packed `b1` rows are `uint8`);
        by default it is derived from the array's dtype, `uint8` meaning bits."""
        vectors = np.asarray(vectors)
        single = vectors.ndim == 1
        if single:
            vectors = vectors[None, :]
        if vectors.ndim != 2:
            raise ValueError("Expects a matrix or a vector")
        if dtype is None:
            dtype = _infer_dtype(vectors)
        expected_columns = (self.ndim + 7) // 8 if dtype == "b1" else self.ndim
        if vectors.shape[1] != expected_columns:
            raise ValueError(f"The number of columns {vectors.shape[1]} must match the index ({expected_columns})")
        if vectors.strides[1] != vectors.itemsize:  # rows may be strided (python/lib.cpp:415-461), scalars may not
            vectors = np.ascontiguousarray(vectors)
        q = vectors.shape[0]
        keys = np.zeros((q, count), dtype=np.uint64)
        distances = np.zeros((q, count), dtype=np.float32)
        counts = np.zeros(q, dtype=np.uint64)
        visited = np.zeros(q, dtype=np.uint64)
        computed = np.zeros(q, dtype=np.uint64)
        stats = Stats()
        err = C.c_char_p()
        if exact and filter is not None:  # brute force over the members the filter lets through (index.hpp:4260-4263)
            kernel_ms = C.c_float()
            library().usearch_amd_filtered_exact_search_many(
                self._handle, filter._handle, _pointer(vectors), SCALAR_KINDS[dtype], q,
                vectors.shape[1] * vectors.itemsize if q <= 1 else vectors.strides[0], count, _pointer(keys), _pointer(distances),
                _pointer(counts), int(exact == "tiled"), C.byref(kernel_ms), C.byref(err))
            _raise(err, "usearch_amd_filtered_exact_search_many")
            stats.kernel_ms = kernel_ms.value
            batch = BatchMatches(keys, distances, counts, 0, 0, visited, computed, stats)
            return batch[0] if single else batch
        if exact:  # brute force over every stored vector (Index.search(..., exact=True), index.py:700-748)
            kernel_ms = C.c_float()
            # exact="tiled": the matrix-unit kernel (f16 / bf16 cos, ip within float tolerance; i8 bit-identical)
            entry = (library().usearch_amd_exact_search_many_tiled if exact == "tiled"
                     else library().usearch_amd_exact_search_many)
            entry(self._handle, _pointer(vectors), SCALAR_KINDS[dtype], q,
                                                    vectors.shape[1] * vectors.itemsize if q <= 1 else vectors.strides[0],
                                                    count, _pointer(keys), _pointer(distances), _pointer(counts),
                                                    C.byref(kernel_ms), C.byref(err))
            _raise(err, "usearch_amd_exact_search_many")
            stats.kernel_ms = kernel_ms.value
            computed[:] = len(self)
            batch = BatchMatches(keys, distances, counts, 0, int(computed.sum()), visited, computed, stats)
            return batch[0] if single else batch
        ef = self.expansion_search if expansion is None else expansion
        if filter is not None:
            library().usearch_amd_filtered_search_many(self._handle, filter._handle, _pointer(vectors), SCALAR_KINDS[dtype], q,
                                                       vectors.shape[1] * vectors.itemsize if q <= 1 else vectors.strides[0],
                                                       count, ef, _pointer(keys), _pointer(distances), _pointer(counts),
                                                       _pointer(visited), _pointer(computed),
                                                       C.byref(tuning) if tuning is not None else None, C.byref(stats), C.byref(err))
            _raise(err, "usearch_amd_filtered_search_many")
            batch = BatchMatches(keys, distances, counts, int(visited.sum()), int(computed.sum()), visited, computed, stats)
            return batch[0] if single else batch
        library().usearch_amd_search_many(self._handle, _pointer(vectors), SCALAR_KINDS[dtype], q,
                                          vectors.shape[1] * vectors.itemsize if q <= 1 else vectors.strides[0],
                                          count, ef, _pointer(keys),
                                          _pointer(distances), _pointer(counts), _pointer(visited),
                                          _pointer(computed), C.byref(tuning) if tuning is not None else None,
                                          C.byref(stats), C.byref(err))
        _raise(err, "usearch_amd_search_many")
        batch = BatchMatches(keys, distances, counts, int(visited.sum()), int(computed.sum()), visited, computed,
                             stats)
        return batch[0] if single else batch

    def search_device(self, queries_ptr: int, queries_count: int, queries_stride: int, count: int, expansion: int,
                      keys_ptr: int, distances_ptr: int, counts_ptr: int, visited_ptr: int, computed_ptr: int,
                      stream: int = 0, timed: bool = False, tuning: Optional[Tuning] = None,
                      filter: Optional[Filter] = None) -> Stats:
        """HBM-resident batch: raw device addresses in and out (e.g. `torch.Tensor.data_ptr()`), storage scalar kind."""
        stats = Stats()
        err = C.c_char_p()
        if filter is not None:
            library().usearch_amd_filtered_search_many_device(
                self._handle, filter._handle, C.c_void_p(queries_ptr), queries_count, queries_stride, count, expansion,
                C.c_void_p(keys_ptr), C.c_void_p(distances_ptr), C.c_void_p(counts_ptr), C.c_void_p(visited_ptr),
                C.c_void_p(computed_ptr), C.c_void_p(stream), C.byref(tuning) if tuning is not None else None, int(timed),
                C.byref(stats), C.byref(err))
            _raise(err, "usearch_amd_filtered_search_many_device")
            return stats
        library().usearch_amd_search_many_device(self._handle, C.c_void_p(queries_ptr), queries_count, queries_stride,
                                                 count, expansion, C.c_void_p(keys_ptr), C.c_void_p(distances_ptr),
                                                 C.c_void_p(counts_ptr), C.c_void_p(visited_ptr),
                                                 C.c_void_p(computed_ptr), C.c_void_p(stream),
                                                 C.byref(tuning) if tuning is not None else None, int(timed),
                                                 C.byref(stats), C.byref(err))
        _raise(err, "usearch_amd_search_many_device")
        return stats

    def exact_search_device(self, queries_ptr: int, queries_count: int, queries_stride: int, count: int, keys_ptr: int,
                            distances_ptr: int, counts_ptr: int, stream: int = 0, tiled: bool = True) -> float:
        """Exact search of an HBM-resident batch into HBM-resident results (`search(…, exact = true)` of the class for a batch);
        returns the kernels' HIP-event time in ms. `tiled`: the matrix-unit kernel (exact_tiled.hip)."""
        kernel_ms = C.c_float(0)
        err = C.c_char_p()
        library().usearch_amd_exact_search_many_device(self._handle, C.c_void_p(queries_ptr), queries_count, queries_stride,
                                                       count, C.c_void_p(keys_ptr), C.c_void_p(distances_ptr),
                                                       C.c_void_p(counts_ptr), C.c_void_p(stream), int(tiled),
                                                       C.byref(kernel_ms), C.byref(err))
        _raise(err, "usearch_amd_exact_search_many_device")
        return float(kernel_ms.value)

    @property
    def last_distances_ms(self) -> float:
        return float(library().usearch_amd_last_distances_ms(self._handle))

    def last_peaks(self, queries_count: int) -> np.ndarray:
        """[Q, 2] = {peak frontier size, visited-set size} of the most recent search (scratch-sizing telemetry)."""
        out = np.zeros((queries_count, 2), dtype=np.uint32)
        err = C.c_char_p()
        library().usearch_amd_last_peaks(self._handle, _pointer(out), queries_count, C.byref(err))
        _raise(err, "usearch_amd_last_peaks")
        return out

    def cluster(self, vectors: np.ndarray, level: int = 1, dtype: Optional[str] = None):
        """`index_dense_gt::cluster(query, level)` for a batch (index_dense.hpp:788-793 → index.hpp:3089-3125):
        → (keys[Q], distances[Q], visited[Q], computed[Q]) — per query the member the greedy descent reaches on `level`."""
        vectors = np.asarray(vectors)
        if vectors.ndim == 1:
            vectors = vectors[None, :]
        if dtype is None:
            dtype = _infer_dtype(vectors)
        if vectors.strides[1] != vectors.itemsize:
            vectors = np.ascontiguousarray(vectors)
        q = len(vectors)
        keys = np.zeros(q, dtype=np.uint64)
        distances = np.zeros(q, dtype=np.float32)
        visited = np.zeros(q, dtype=np.uint64)
        computed = np.zeros(q, dtype=np.uint64)
        err = C.c_char_p()
        library().usearch_amd_cluster_many(self._handle, _pointer(vectors), SCALAR_KINDS[dtype], q,
                                           vectors.shape[1] * vectors.itemsize if q <= 1 else vectors.strides[0], level,
                                           _pointer(keys), _pointer(distances), _pointer(visited), _pointer(computed),
                                           C.byref(err))
        _raise(err, "usearch_amd_cluster_many")
        return keys, distances, visited, computed

    def distances(self, queries: np.ndarray, slots: np.ndarray) -> np.ndarray:
        """out[q, j] = metric(queries[q], stored vector of slot slots[q, j]); queries in the storage kind."""
        queries = np.ascontiguousarray(queries)
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        assert queries.ndim == 2 and slots.ndim == 2 and len(queries) == len(slots)
        out = np.zeros(slots.shape, dtype=np.float32)
        err = C.c_char_p()
        library().usearch_amd_distances(self._handle, _pointer(queries), len(queries), queries.strides[0],
                                        _pointer(slots), slots.shape[1], _pointer(out), C.byref(err))
        _raise(err, "usearch_amd_distances")
        return out


class BuiltIndex:
    """An index constructed on the MI355X (`usearch_amd_build`): a searchable `Index` (`.index`) plus the reference's
    serialized form (`.save_buffer()` / `.save(path)`), so the reference — or anything else that reads `.usearch` files —
    can load what was built. Mirrors the `Index(...)` + `add(keys, vectors)` + `save` sequence of python/usearch/index.py
    (index.py:400-520, 640-700, 1060-1100) collapsed into one call."""

    def __init__(self, handle: int):
        self._builder = handle
        self.index = Index(library().usearch_amd_build_snapshot(handle), owner=self)

    @property
    def stats(self) -> BuildStats:
        out = BuildStats()
        library().usearch_amd_build_stats(self._builder, C.byref(out))
        return out

    @property
    def serialized_length(self) -> int:
        return library().usearch_amd_build_serialized_length(self._builder)

    def save_buffer(self) -> np.ndarray:
        image = np.empty(self.serialized_length, dtype=np.uint8)
        err = C.c_char_p()
        library().usearch_amd_build_save_buffer(self._builder, _pointer(image), image.size, C.byref(err))
        _raise(err, "usearch_amd_build_save_buffer")
        return image

    def save(self, path) -> None:
        self.save_buffer().tofile(os.fspath(path))

    def close(self) -> None:
        if getattr(self, "_builder", None):
            self.index._handle = None
            self.index._owner = None
            err = C.c_char_p()
            library().usearch_amd_build_free(self._builder, C.byref(err))
            self._builder = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build(vectors, metric: str = "cos", dtype: Optional[str] = None, *, keys: Optional[np.ndarray] = None,
          connectivity: int = 16, expansion_add: int = 128, connectivity_base: int = 0, device: int = 0,
          max_batch: int = 0, batch_divisor: int = 0, seed: int = 0, ndim: Optional[int] = None,
          device_pointer: Optional[int] = None, count: Optional[int] = None, stride: Optional[int] = None) -> BuiltIndex:
    """Builds an HNSW index on the device. `vectors` is a [N, d] numpy matrix in the storage scalar kind (bit-packed
    `uint8` rows for `b1`); or pass `device_pointer` / `count` / `stride` / `ndim` / `dtype` for rows already in HBM."""
    err = C.c_char_p()
    config = BuildConfig(connectivity, connectivity_base, expansion_add, batch_divisor, max_batch, 0, seed)
    metric_kind = {name: kind for kind, name in METRIC_NAMES.items()}[metric]
    if keys is not None:
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
    if device_pointer is not None:
        assert dtype and ndim and count and stride
        handle = library().usearch_amd_build(C.c_void_p(device_pointer), count, stride, SCALAR_KINDS[dtype], ndim,
                                             metric_kind, _pointer(keys), C.byref(config), device, 1, C.byref(err))
    else:
        vectors = np.asarray(vectors)
        if dtype is None:
            dtype = _infer_dtype(vectors)
        if vectors.ndim != 2 or vectors.strides[1] != vectors.itemsize:
            vectors = np.ascontiguousarray(vectors)
        if ndim is None:
            ndim = vectors.shape[1] * 8 if dtype == "b1" else vectors.shape[1]
        handle = library().usearch_amd_build(_pointer(vectors), len(vectors), vectors.strides[0], SCALAR_KINDS[dtype],
                                             ndim, metric_kind, _pointer(keys), C.byref(config), device, 0,
                                             C.byref(err))
    _raise(err, "usearch_amd_build")
    if not handle:
        raise RuntimeError("usearch_amd_build: failed without a message")
    return BuiltIndex(handle)


def exact_search(dataset: np.ndarray, queries: np.ndarray, count: int, metric: str = "cos",
                 dtype: Optional[str] = None):
    """`usearch.index.search(dataset, queries, count, exact=True)` (index.py:1608-1700) → (keys = row offsets [Q, k],
    distances [Q, k]); both matrices in the same scalar kind, rows may be strided."""
    dataset, queries = np.asarray(dataset), np.asarray(queries)
    if dtype is None:
        dtype = _infer_dtype(dataset)
    ndim = dataset.shape[1] * 8 if dtype == "b1" else dataset.shape[1]
    metric_kind = {name: kind for kind, name in METRIC_NAMES.items()}[metric]
    keys = np.zeros((len(queries), count), dtype=np.uint64)
    distances = np.zeros((len(queries), count), dtype=np.float32)
    err = C.c_char_p()
    library().usearch_amd_exact_search_dataset(_pointer(dataset), len(dataset), dataset.strides[0], _pointer(queries),
                                               len(queries), queries.strides[0], SCALAR_KINDS[dtype], ndim,
                                               metric_kind, count, _pointer(keys), keys.strides[0],
                                               _pointer(distances), distances.strides[0], C.byref(err))
    _raise(err, "usearch_amd_exact_search_dataset")
    return keys, distances


def merge_many(distances: np.ndarray, keys: np.ndarray, counts: np.ndarray):
    """Host-buffer form of the shard merge: [P, Q, k] distances/keys + [P, Q] counts → ([Q, k] keys, distances, [Q] counts)."""
    distances = np.ascontiguousarray(distances, dtype=np.float32)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    shards, queries, wanted = distances.shape
    out_d = np.zeros((queries, wanted), dtype=np.float32)
    out_k = np.zeros((queries, wanted), dtype=np.uint64)
    out_c = np.zeros(queries, dtype=np.uint64)
    err = C.c_char_p()
    library().usearch_amd_merge_many(_pointer(distances), _pointer(keys), _pointer(counts), shards, queries, wanted,
                                     _pointer(out_d), _pointer(out_k), _pointer(out_c), C.byref(err))
    _raise(err, "usearch_amd_merge_many")
    return out_k, out_d, out_c


def merge_many_device(distances_ptr: int, keys_ptr: int, counts_ptr: int, shards: int, queries: int, wanted: int,
                      out_distances_ptr: int, out_keys_ptr: int, out_counts_ptr: int, stream: int = 0) -> None:
    err = C.c_char_p()
    library().usearch_amd_merge_many_device(C.c_void_p(distances_ptr), C.c_void_p(keys_ptr), C.c_void_p(counts_ptr),
                                            shards, queries, wanted, C.c_void_p(out_distances_ptr),
                                            C.c_void_p(out_keys_ptr), C.c_void_p(out_counts_ptr), C.c_void_p(stream),
                                            C.byref(err))
    _raise(err, "usearch_amd_merge_many_device")


def note_device_free() -> None:
    """Tells the engine that the host just released device memory through another allocator (`torch.cuda.empty_cache()`): the next
    loader or builder waits out the driver's settle window before it places its matrix (csrc/placement.hpp)."""
    library().usearch_amd_note_device_free()


def settle() -> float:
    """Waits out what is left of the settle window (at most USEARCH_AMD_SETTLE_MS = 1000); returns the milliseconds waited."""
    return float(library().usearch_amd_settle())


def device_count() -> int:
    err = C.c_char_p()
    n = library().usearch_amd_device_count(C.byref(err))
    return n


def cast(vector: np.ndarray, from_dtype: str, to_dtype: str, ndim: int) -> Optional[np.ndarray]:
    """Host-side query cast (`usearch_amd_cast`); None when the kinds are equal."""
    vector = np.ascontiguousarray(vector)
    nbytes = {"b1": (ndim + 7) // 8, "i8": ndim, "f16": 2 * ndim, "bf16": 2 * ndim, "f32": 4 * ndim,
              "f64": 8 * ndim}[to_dtype]
    out = np.zeros(nbytes, dtype=np.uint8)
    done = library().usearch_amd_cast(SCALAR_KINDS[from_dtype], SCALAR_KINDS[to_dtype], _pointer(vector), ndim,
                                      _pointer(out))
    return out if done else None


TEST_HOOKS_PATH = os.path.join(os.path.dirname(LIBRARY_PATH), "libusearch_amd_testhooks.so")
_test_hooks = None


def test_hooks() -> C.CDLL:
    """The self-test / micro-benchmark entry points of the device containers: their own library (csrc/test_hooks.hip), which the
    product library does not carry."""
    global _test_hooks
    if _test_hooks is None:
        library()  # one HIP runtime per process, mapped by the product library's loader
        if not os.path.exists(TEST_HOOKS_PATH):
            raise RuntimeError(f"{TEST_HOOKS_PATH} is missing: build it with `make -C usearch_amd/csrc`")
        _test_hooks = C.CDLL(TEST_HOOKS_PATH, mode=os.RTLD_LOCAL)
        for name, arguments in (("gather_probe", [C.c_void_p, C.c_size_t, C.c_size_t]), ("translation_probe", [C.c_void_p, C.c_size_t]),
                                ("latency_probe", [C.c_void_p, C.c_size_t, C.c_size_t])):
            probe = getattr(_test_hooks, f"usearch_amd_test_{name}")
            probe.restype = C.c_float
            probe.argtypes = arguments + [C.POINTER(C.c_char_p)]
        _test_hooks.usearch_amd_test_containers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                            C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                                                            C.POINTER(C.c_char_p)]
    return _test_hooks


def test_containers(kinds: np.ndarray, keys: np.ndarray, slots: np.ndarray, limit: int):
    """Device container self-test hook → (popped[(key, slot)], top[(distance, slot)])."""
    kinds = np.ascontiguousarray(kinds, dtype=np.uint32)
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    slots = np.ascontiguousarray(slots, dtype=np.uint32)
    n = len(kinds)
    popped = np.zeros(n + 1, dtype=np.uint64)
    top = np.zeros(limit + 1, dtype=np.uint64)
    popped_count, top_count = C.c_size_t(), C.c_size_t()
    err = C.c_char_p()
    test_hooks().usearch_amd_test_containers(_pointer(kinds), _pointer(keys), _pointer(slots), n, limit, _pointer(popped),
                                          C.byref(popped_count), _pointer(top), C.byref(top_count), C.byref(err))
    _raise(err, "usearch_amd_test_containers")

    def unpack(a):
        bits = (a & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        return bits.view(np.float32), (a >> np.uint64(32)).astype(np.uint32)

    return unpack(popped[: popped_count.value]), unpack(top[: top_count.value])
