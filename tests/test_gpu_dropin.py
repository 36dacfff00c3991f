"""The drop-in `libusearch_c.so` (include/usearch_c_dropin.h): the reference's own C test program, restated call for
call through ctypes — /root/reference/c/test.c:52-368 (`test_init`, `test_add_vector`, `test_find_vector`,
`test_get_vector`, `test_remove_vector`, `test_save_load`, `test_view`, sizes {11, 512} x dimensions {83, 2} as in its
`main`, c/test.c:370-393) — plus the exact-value checks the other bindings' tests hold for this ABI
(golang/lib_test.go:835-877 distances, cpp/test.cpp:1105-1145 filtered search) and a side-by-side run against the real
reference library on the same calls (the save / load case with c/test.c:246-250's own pearson / f64 configuration)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import refbind
from tests import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBRARY = os.path.join(ROOT, "usearch_amd", "lib", "libusearch_c.so")

METRIC = {"cos": 1, "ip": 2, "l2sq": 3, "haversine": 4, "hamming": 8, "pearson": 6}
SCALAR = {"f32": 1, "f64": 2, "f16": 3, "i8": 4, "b1": 5}


class Options(C.Structure):  # usearch_init_options_t, c/usearch.h:64-110
    _fields_ = [("metric_kind", C.c_int), ("metric", C.c_void_p), ("quantization", C.c_int), ("dimensions", C.c_size_t),
                ("connectivity", C.c_size_t), ("expansion_add", C.c_size_t), ("expansion_search", C.c_size_t),
                ("multi", C.c_bool)]


FILTER = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(LIBRARY)
    err = C.POINTER(C.c_char_p)
    L.usearch_version.restype = C.c_char_p
    L.usearch_init.restype = C.c_void_p
    L.usearch_init.argtypes = [C.POINTER(Options), err]
    L.usearch_hardware_acceleration.restype = C.c_char_p
    L.usearch_hardware_acceleration.argtypes = [C.c_void_p, err]
    for name in ("size", "capacity", "dimensions", "connectivity", "memory_usage", "serialized_length",
                 "expansion_add", "expansion_search"):
        f = getattr(L, f"usearch_{name}")
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p, err]
    L.usearch_free.argtypes = [C.c_void_p, err]
    L.usearch_clear.argtypes = [C.c_void_p, err]
    L.usearch_reserve.argtypes = [C.c_void_p, C.c_size_t, err]
    L.usearch_change_threads_search.argtypes = [C.c_void_p, C.c_size_t, err]
    L.usearch_change_expansion_search.argtypes = [C.c_void_p, C.c_size_t, err]
    L.usearch_add.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, err]
    L.usearch_contains.restype = C.c_bool
    L.usearch_contains.argtypes = [C.c_void_p, C.c_uint64, err]
    L.usearch_count.restype = C.c_size_t
    L.usearch_count.argtypes = [C.c_void_p, C.c_uint64, err]
    L.usearch_search.restype = C.c_size_t
    L.usearch_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, err]
    L.usearch_filtered_search.restype = C.c_size_t
    L.usearch_filtered_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, FILTER, C.c_void_p, C.c_void_p,
                                          C.c_void_p, err]
    L.usearch_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, err]
    L.usearch_cluster_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                       C.c_void_p, err]
    L.usearch_get.restype = C.c_size_t
    L.usearch_get.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, err]
    L.usearch_remove.restype = C.c_size_t
    L.usearch_remove.argtypes = [C.c_void_p, C.c_uint64, err]
    L.usearch_rename.restype = C.c_size_t
    L.usearch_rename.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, err]
    for name in ("save", "load", "view"):
        getattr(L, f"usearch_{name}").argtypes = [C.c_void_p, C.c_char_p, err]
    for name in ("save_buffer", "load_buffer", "view_buffer"):
        getattr(L, f"usearch_{name}").argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err]
    L.usearch_metadata.argtypes = [C.c_char_p, C.POINTER(Options), err]
    L.usearch_distance.restype = C.c_float
    L.usearch_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, err]
    L.usearch_exact_search.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                       C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_size_t, err]
    L.usearch_filter_from_key_range.restype = C.c_void_p
    L.usearch_filter_from_key_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, err]
    L.usearch_filter_from_keys.restype = C.c_void_p
    L.usearch_filter_from_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_bool, err]
    L.usearch_filter_from_callback.restype = C.c_void_p
    L.usearch_filter_from_callback.argtypes = [C.c_void_p, FILTER, C.c_void_p, err]
    L.usearch_filter_allowed.restype = C.c_size_t
    L.usearch_filter_allowed.argtypes = [C.c_void_p, err]
    L.usearch_filter_free.argtypes = [C.c_void_p, err]
    L.usearch_filtered_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                               C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                               err]
    L.usearch_filtered_search_exact_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t,
                                                     C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, err]
    return L


def ok(err):
    assert not err.value, err.value


def ptr(a):
    return C.c_void_p(a.ctypes.data)


def create_options(dimensions, **changes):  # c/test.c:32-43
    o = Options(METRIC["ip"], None, SCALAR["f32"], dimensions, 3, 40, 16, False)
    for name, value in changes.items():
        setattr(o, name, value)
    return o


def create_vectors(count, dimensions, seed=0):  # c/test.c:25-31: uniform [0, 1)
    return np.random.default_rng(seed).random((count, dimensions), dtype=np.float32)


def filled_index(lib, count, dimensions, options=None, data=None, keys=None):
    err = C.c_char_p()
    options = options or create_options(dimensions)
    index = lib.usearch_init(C.byref(options), C.byref(err))
    ok(err)
    lib.usearch_reserve(index, count, C.byref(err))
    data = create_vectors(count, dimensions) if data is None else data
    for i in range(count):
        lib.usearch_add(index, int(i if keys is None else keys[i]), ptr(data[i]), SCALAR["f32"], C.byref(err))
        ok(err)
    return index, data


SIZES = [(11, 83), (11, 2), (512, 83), (512, 2)]  # c/test.c:374-375


@pytest.mark.parametrize("count,dimensions", SIZES)
def test_c_test_program(lib, count, dimensions, tmp_path):
    err = C.c_char_p()
    assert lib.usearch_version() == b"2.21.0"
    # test_init, c/test.c:52-86
    options = create_options(dimensions)
    index = lib.usearch_init(C.byref(options), C.byref(err))
    ok(err)
    lib.usearch_free(index, C.byref(err))
    index = lib.usearch_init(C.byref(options), C.byref(err))
    assert lib.usearch_size(index, C.byref(err)) == 0 and lib.usearch_capacity(index, C.byref(err)) == 0
    assert lib.usearch_dimensions(index, C.byref(err)) == dimensions
    assert lib.usearch_connectivity(index, C.byref(err)) == 3
    lib.usearch_reserve(index, count, C.byref(err))
    ok(err)
    assert lib.usearch_size(index, C.byref(err)) == 0 and lib.usearch_capacity(index, C.byref(err)) >= count
    assert lib.usearch_hardware_acceleration(index, C.byref(err)) == b"gfx950"
    lib.usearch_free(index, C.byref(err))

    # test_add_vector, c/test.c:93-122
    index, data = filled_index(lib, count, dimensions)
    assert lib.usearch_size(index, C.byref(err)) == count and lib.usearch_capacity(index, C.byref(err)) >= count
    assert all(lib.usearch_contains(index, i, C.byref(err)) for i in range(count))
    assert not lib.usearch_contains(index, 2 ** 64 - 1, C.byref(err))
    assert lib.usearch_memory_usage(index, C.byref(err)) > 0

    # test_find_vector, c/test.c:129-162: every vector is found, asking for `collection_size` results
    keys = np.zeros(count, dtype=np.uint64)
    distances = np.zeros(count, dtype=np.float32)
    for i in range(count):
        found = lib.usearch_search(index, ptr(data[i]), SCALAR["f32"], count, ptr(keys), ptr(distances), C.byref(err))
        ok(err)
        assert 1 <= found <= count
        assert np.all(np.diff(distances[:found]) >= 0)  # cpp/test.cpp:499-503: ascending
    lib.usearch_free(index, C.byref(err))

    # test_get_vector, c/test.c:170-199: a multi-index returns every vector stored under one key
    options = create_options(dimensions, multi=True)
    index = lib.usearch_init(C.byref(options), C.byref(err))
    lib.usearch_reserve(index, count, C.byref(err))  # c/test.c:177
    for i in range(count):
        lib.usearch_add(index, 1, ptr(data[i]), SCALAR["f32"], C.byref(err))
        ok(err)
    out = np.zeros((count, dimensions), dtype=np.float32)
    assert lib.usearch_get(index, 1, count, ptr(out), SCALAR["f32"], C.byref(err)) == count
    assert np.array_equal(out, data)
    lib.usearch_free(index, C.byref(err))

    # test_remove_vector, c/test.c:207-233
    index, _ = filled_index(lib, count, dimensions, data=data)
    for i in range(count):
        lib.usearch_remove(index, i, C.byref(err))
        ok(err)
    assert lib.usearch_size(index, C.byref(err)) == 0
    found = lib.usearch_search(index, ptr(data[0]), SCALAR["f32"], 5, ptr(keys), ptr(distances), C.byref(err))
    assert found == 0  # everything is a tombstone now
    lib.usearch_free(index, C.byref(err))

    # test_save_load, c/test.c:241-328 (odd connectivity / expansions survive the round trip through a shell index)
    path = str(tmp_path / "tmp.usearch").encode()
    weird = create_options(dimensions, connectivity=11, expansion_add=15, expansion_search=19,
                           metric_kind=METRIC["pearson"], quantization=SCALAR["f64"])
    index, _ = filled_index(lib, count, dimensions, options=weird, data=data)
    lib.usearch_save(index, path, C.byref(err))
    ok(err)
    lib.usearch_free(index, C.byref(err))
    meta = Options()
    lib.usearch_metadata(path, C.byref(meta), C.byref(err))
    ok(err)
    assert (meta.metric_kind, meta.quantization, meta.dimensions, meta.connectivity) == (6, 2, dimensions, 11)
    index = lib.usearch_init(None, C.byref(err))
    ok(err)
    lib.usearch_load(index, path, C.byref(err))
    ok(err)
    assert lib.usearch_size(index, C.byref(err)) == count and lib.usearch_capacity(index, C.byref(err)) == count
    assert lib.usearch_dimensions(index, C.byref(err)) == dimensions and lib.usearch_connectivity(index, C.byref(err)) == 11
    assert all(lib.usearch_contains(index, i, C.byref(err)) for i in range(count))
    lib.usearch_change_threads_search(index, 1, C.byref(err))
    for i in range(0, count, 7):
        found = lib.usearch_search(index, ptr(data[i]), SCALAR["f32"], count, ptr(keys), ptr(distances), C.byref(err))
        ok(err)
        assert 1 <= found <= count
    # the REAL reference loads what the drop-in saved and answers the same queries with the same labels
    theirs = refbind.RefIndex.from_buffer(np.fromfile(path.decode(), dtype=np.uint8), dtype="f64")
    assert len(theirs) == count
    theirs.expansion_search = 64
    lib.usearch_change_expansion_search(index, 64, C.byref(err))
    their_keys, their_distances, *_ = theirs.search(data[:8], 3, dtype="f32")
    for i in range(8):
        lib.usearch_search(index, ptr(data[i]), SCALAR["f32"], 3, ptr(keys), ptr(distances), C.byref(err))
        # 2-d Pearson is all cancellation (every distance is 0, 1 or 2 up to rounding that the two builds fuse differently)
        assert np.allclose(distances[:3], their_distances[i], atol=1e-5 if dimensions > 2 else 1e-3)
        if dimensions > 2:  # the Pearson distance of 2-d vectors is 0, 1 or 2: ties everywhere
            assert np.array_equal(keys[:3], their_keys[i])
    lib.usearch_free(index, C.byref(err))

    # test_view, c/test.c:335-368
    index = lib.usearch_init(C.byref(create_options(dimensions)), C.byref(err))
    lib.usearch_view(index, path, C.byref(err))
    ok(err)
    assert lib.usearch_size(index, C.byref(err)) == count
    lib.usearch_free(index, C.byref(err))


def test_known_distances(lib):
    """golang/lib_test.go:835-877: cos(e1, e2) = 1, l2sq(e1, e2) = 2, i8 l2sq(10 e1, 10 e2) = 200; rust/lib.rs:1897-1924 bits."""
    err = C.c_char_p()
    e1, e2 = np.array([1, 0, 0], dtype=np.float32), np.array([0, 1, 0], dtype=np.float32)
    assert abs(lib.usearch_distance(ptr(e1), ptr(e2), SCALAR["f32"], 3, METRIC["cos"], C.byref(err)) - 1.0) < 0.01
    assert abs(lib.usearch_distance(ptr(e1), ptr(e2), SCALAR["f32"], 3, METRIC["l2sq"], C.byref(err)) - 2.0) < 0.01
    a, b = np.array([10, 0, 0], dtype=np.int8), np.array([0, 10, 0], dtype=np.int8)
    assert lib.usearch_distance(ptr(a), ptr(b), SCALAR["i8"], 3, METRIC["l2sq"], C.byref(err)) == 200.0
    q, x, y = (np.array([v], dtype=np.uint8) for v in (0b01111000, 0b11110000, 0b00001111))
    assert lib.usearch_distance(ptr(q), ptr(x), SCALAR["b1"], 8, METRIC["hamming"], C.byref(err)) == 2.0
    assert lib.usearch_distance(ptr(q), ptr(y), SCALAR["b1"], 8, METRIC["hamming"], C.byref(err)) == 6.0
    ok(err)
    # pearson: e1 vs e2 in 3-d → correlation -1/2 → distance 1.5 (metric_pearson_gt, index_plugins.hpp:1478-1520)
    assert abs(lib.usearch_distance(ptr(e1), ptr(e2), SCALAR["f32"], 3, METRIC["pearson"], C.byref(err)) - 1.5) < 1e-6
    ok(err)
    # what the reference's dispatch table does not hold has no kernel either (haversine over half floats,
    # index_plugins.hpp:1981-1982): refused by name, never silently computed elsewhere
    half = np.zeros(2, dtype=np.float16)
    lib.usearch_distance(ptr(half), ptr(half), SCALAR["f16"], 2, METRIC["haversine"], C.byref(err))
    assert err.value and b"kernel" in err.value


def test_made_filters_answer_like_the_callback(lib):
    """`usearch_filter_from_*` + `usearch_filtered_search_many`: the predicate evaluated once (or never on the host: ranges and key
    sets are built by a kernel over the keys in HBM) must answer exactly like `usearch_filtered_search` with the callback, query by
    query — and a filter must refuse to outlive a change of the index."""
    err = C.c_char_p()
    n, dims, k = 3000, 48, 10
    data = util.make_vectors(n, dims, "f32", seed=5)
    options = Options(METRIC["cos"], None, SCALAR["f32"], dims, 16, 128, 64, False)
    index, _ = filled_index(lib, n, dims, options=options, data=data)
    queries = util.make_vectors(32, dims, "f32", seed=6)
    listed = np.array([key for key in range(n) if key % 7 == 3], dtype=np.uint64)
    cases = {
        "range": (lambda key: 500 <= key <= 1999, lambda: lib.usearch_filter_from_key_range(index, 500, 1999, C.byref(err))),
        "keys": (lambda key: key % 7 == 3, lambda: lib.usearch_filter_from_keys(index, ptr(listed), len(listed), True, C.byref(err))),
        "deny": (lambda key: key % 7 != 3, lambda: lib.usearch_filter_from_keys(index, ptr(listed), len(listed), False, C.byref(err))),
    }
    calls = [0]
    def counting(key, state):
        calls[0] += 1
        return int(key % 5 == 0)
    counted = FILTER(counting)
    cases["callback"] = (lambda key: key % 5 == 0, lambda: lib.usearch_filter_from_callback(index, counted, None, C.byref(err)))
    for name, (predicate, make) in cases.items():
        made = make()
        ok(err)
        assert made, name
        assert lib.usearch_filter_allowed(made, C.byref(err)) == sum(1 for key in range(n) if predicate(key)), name
        keys = np.zeros((len(queries), k), dtype=np.uint64)
        distances = np.zeros((len(queries), k), dtype=np.float32)
        counts = np.zeros(len(queries), dtype=np.uint64)
        before = calls[0]
        lib.usearch_filtered_search_many(index, made, ptr(queries), SCALAR["f32"], len(queries), queries.strides[0], k, ptr(keys),
                                         keys.strides[0], ptr(distances), distances.strides[0], ptr(counts), None, None, C.byref(err))
        ok(err)
        assert calls[0] == before, "a made filter costs no callback at search time"
        callback = FILTER(lambda key, state, predicate=predicate: int(predicate(key)))
        one_keys, one_distances = np.zeros(k, dtype=np.uint64), np.zeros(k, dtype=np.float32)
        for q in range(len(queries)):
            found = lib.usearch_filtered_search(index, ptr(queries[q]), SCALAR["f32"], k, callback, None, ptr(one_keys),
                                                ptr(one_distances), C.byref(err))
            ok(err)
            assert found == counts[q], (name, q)
            assert np.array_equal(one_keys[:found], keys[q, :found]) and util.same_float_bits(one_distances[:found], distances[q, :found])
            assert all(predicate(int(key)) for key in keys[q, :found])
        # brute force under the same filter: nothing the predicate rejects, and never farther than the graph's answer
        exact_keys = np.zeros((len(queries), k), dtype=np.uint64)
        exact_distances = np.zeros((len(queries), k), dtype=np.float32)
        exact_counts = np.zeros(len(queries), dtype=np.uint64)
        lib.usearch_filtered_search_exact_many(index, made, ptr(queries), SCALAR["f32"], len(queries), queries.strides[0], k,
                                               ptr(exact_keys), exact_keys.strides[0], ptr(exact_distances),
                                               exact_distances.strides[0], ptr(exact_counts), C.byref(err))
        ok(err)
        assert np.all(exact_counts == k) and all(predicate(int(key)) for key in exact_keys.reshape(-1))
        assert np.all(exact_distances[:, 0] <= distances[:, 0] + 1e-6)
        if name == "callback":
            assert calls[0] == n, "the callback ran once per member, when the filter was made"
            # the index changes: the filter describes members that are no longer the whole story
            fresh = util.make_vectors(1, dims, "f32", seed=9)
            lib.usearch_reserve(index, n + 8, C.byref(err))
            lib.usearch_add(index, 900000, ptr(fresh[0]), SCALAR["f32"], C.byref(err))
            ok(err)
            lib.usearch_filtered_search_many(index, made, ptr(queries), SCALAR["f32"], len(queries), queries.strides[0], k, ptr(keys),
                                             keys.strides[0], ptr(distances), distances.strides[0], ptr(counts), None, None, C.byref(err))
            assert err.value and b"changed since the filter was made" in err.value
            err = C.c_char_p()
        lib.usearch_filter_free(made, C.byref(err))
    lib.usearch_free(index, C.byref(err))


def test_removed_slots_are_recycled(lib, reference):
    """`usearch_add` after `usearch_remove` takes the freed slot, oldest first, and links the new member in place (index_dense.hpp:
    1479-1511 → index_gt::update, index.hpp:2916-2999) instead of growing the index: same size, same serialized length as the
    reference after the same calls, the new members findable, the removed ones gone — before and after the first device build."""
    from oracle import refbind
    err = C.c_char_p()
    n, dims, k = 3000, 32, 10
    data = util.make_vectors(n + 600, dims, "f32", seed=7)
    options = Options(METRIC["cos"], None, SCALAR["f32"], dims, 16, 128, 64, False)
    index, _ = filled_index(lib, n, dims, options=options, data=data[:n])
    ref = refbind.RefIndex(dims, "cos", "f32", connectivity=16, expansion_add=128)
    ref.add(np.arange(n, dtype=np.uint64), data[:n], threads=1)
    keys = np.zeros(k, dtype=np.uint64)
    distances = np.zeros(k, dtype=np.float32)
    lib.usearch_search(index, ptr(data[0]), SCALAR["f32"], k, ptr(keys), ptr(distances), C.byref(err))  # the first device build
    ok(err)
    length_before = lib.usearch_serialized_length(index, C.byref(err))
    reference_length_before = len(ref.save_buffer())
    removed = list(range(100, 400))
    for round_, (first_new, count) in enumerate([(n, 300), (n + 300, 300)]):
        victims = removed if round_ == 0 else list(range(n, n + 300))  # second round: remove what the first round added
        for key in victims:
            assert lib.usearch_remove(index, key, C.byref(err)) == 1
            ok(err)
            ref.remove(key)
        assert lib.usearch_size(index, C.byref(err)) == n - 300 == len(ref)
        for i in range(count):
            lib.usearch_add(index, first_new + i, ptr(data[first_new + i]), SCALAR["f32"], C.byref(err))
            ok(err)  # no `usearch_reserve` beyond n: the freed slots are the room
        ref.add(np.arange(first_new, first_new + count, dtype=np.uint64), data[first_new:first_new + count], threads=1)
        assert lib.usearch_size(index, C.byref(err)) == n == len(ref)
        assert lib.usearch_serialized_length(index, C.byref(err)) == length_before, "the index grew"
        assert len(ref.save_buffer()) == reference_length_before, "the reference's did not either (levels stay with the slots)"
        gone = set(victims)
        found_self = 0
        for i in range(count):
            found = lib.usearch_search(index, ptr(data[first_new + i]), SCALAR["f32"], k, ptr(keys), ptr(distances), C.byref(err))
            ok(err)
            assert found == k and not (set(keys.tolist()) & gone)
            found_self += int(keys[0] == first_new + i and distances[0] < 1e-4)
        assert found_self >= 0.98 * count, f"only {found_self} of {count} recycled members find themselves"
        for key in victims[:50]:
            assert not lib.usearch_contains(index, key, C.byref(err))
        # the rest of the index still answers like a graph: recall against brute force over the same members
        probes = data[500:700]
        exact_keys = np.zeros((len(probes), k), dtype=np.uint64)
        exact_distances = np.zeros((len(probes), k), dtype=np.float32)
        exact_counts = np.zeros(len(probes), dtype=np.uint64)
        lib.usearch_search_exact_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                                  C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_char_p)]
        lib.usearch_search_exact_many(index, ptr(probes), SCALAR["f32"], len(probes), probes.strides[0], k, ptr(exact_keys),
                                      exact_keys.strides[0], ptr(exact_distances), exact_distances.strides[0], ptr(exact_counts),
                                      C.byref(err))
        ok(err)
        hits = 0
        for i in range(len(probes)):
            lib.usearch_search(index, ptr(probes[i]), SCALAR["f32"], k, ptr(keys), ptr(distances), C.byref(err))
            hits += len(set(keys.tolist()) & set(exact_keys[i].tolist()))
        assert hits / (len(probes) * k) > 0.9
    lib.usearch_free(index, C.byref(err))


def test_a_loaded_index_at_capacity_takes_additions_into_its_tombstones(lib):
    """After a load the reference lists the image's freed slots (`reindex_keys_`, index_dense.hpp:2162-2200), so an `add` into a
    full index with tombstones needs no `reserve` (the capacity test used to run before the slots were listed)."""
    err = C.c_char_p()
    n, dims, k = 400, 16, 5
    data = util.make_vectors(n + 3, dims, "f32", seed=17)
    index, _ = filled_index(lib, n, dims, options=Options(METRIC["cos"], None, SCALAR["f32"], dims, 16, 128, 64, False), data=data[:n])
    for key in (7, 8, 9):
        assert lib.usearch_remove(index, key, C.byref(err)) == 1
        ok(err)
    length = lib.usearch_serialized_length(index, C.byref(err))
    buffer = np.zeros(length, dtype=np.uint8)
    lib.usearch_save_buffer(index, ptr(buffer), length, C.byref(err))
    ok(err)
    loaded = lib.usearch_init(None, C.byref(err))
    lib.usearch_load_buffer(loaded, ptr(buffer), length, C.byref(err))
    ok(err)
    assert lib.usearch_size(loaded, C.byref(err)) == n - 3
    assert lib.usearch_capacity(loaded, C.byref(err)) == n, "the test wants an index with no spare capacity"
    keys = np.zeros(k, dtype=np.uint64)
    distances = np.zeros(k, dtype=np.float32)
    for i in range(3):  # exactly as many as there are tombstones …
        lib.usearch_add(loaded, n + i, ptr(data[n + i]), SCALAR["f32"], C.byref(err))
        ok(err)
    assert lib.usearch_size(loaded, C.byref(err)) == n
    lib.usearch_add(loaded, n + 3, ptr(data[0]), SCALAR["f32"], C.byref(err))
    assert err.value and b"Reserve capacity" in err.value  # … and not one more
    err = C.c_char_p()
    for i in range(3):
        assert lib.usearch_search(loaded, ptr(data[n + i]), SCALAR["f32"], k, ptr(keys), ptr(distances), C.byref(err)) == k
        ok(err)
        assert keys[0] == n + i and not (set(keys.tolist()) & {7, 8, 9})
    lib.usearch_free(index, C.byref(err))
    lib.usearch_free(loaded, C.byref(err))


def test_filtered_search_and_rename(lib):
    """cpp/test.cpp:1105-1145: predicate key != 0 → 10 results none 0; `false` → 0; key == 10 → exactly [10]."""
    err = C.c_char_p()
    data = util.make_vectors(2048, 32, "f32", seed=3)
    options = Options(METRIC["cos"], None, SCALAR["f32"], 32, 16, 128, 64, False)
    index, _ = filled_index(lib, 2048, 32, options=options, data=data)
    keys = np.zeros(10, dtype=np.uint64)
    distances = np.zeros(10, dtype=np.float32)
    not_zero = FILTER(lambda key, state: int(key != 0))
    found = lib.usearch_filtered_search(index, ptr(data[0]), SCALAR["f32"], 10, not_zero, None, ptr(keys), ptr(distances),
                                        C.byref(err))
    ok(err)
    assert found == 10 and 0 not in keys.tolist()
    never = FILTER(lambda key, state: 0)
    assert lib.usearch_filtered_search(index, ptr(data[0]), SCALAR["f32"], 10, never, None, ptr(keys), ptr(distances),
                                       C.byref(err)) == 0
    only_ten = FILTER(lambda key, state: int(key == 10))
    found = lib.usearch_filtered_search(index, ptr(data[0]), SCALAR["f32"], 10, only_ten, None, ptr(keys), ptr(distances),
                                        C.byref(err))
    assert found == 1 and keys[0] == 10
    # rename: the member answers to its new key only
    assert lib.usearch_rename(index, 10, 777777, C.byref(err)) == 1
    ok(err)
    assert lib.usearch_contains(index, 777777, C.byref(err)) and not lib.usearch_contains(index, 10, C.byref(err))
    found = lib.usearch_search(index, ptr(data[10]), SCALAR["f32"], 1, ptr(keys), ptr(distances), C.byref(err))
    assert found == 1 and keys[0] == 777777
    lib.usearch_free(index, C.byref(err))


def test_filtered_search_remembers_a_pure_predicate_when_told_to(lib, monkeypatch):
    """`usearch_filtered_search` has to run the caller's callback over every member per call (the reference runs it on the few
    thousand members a walk meets: c/lib.cpp:413-429). With USEARCH_AMD_FILTER_MEMO=1 — read at `usearch_init`, for callers whose
    predicate is a pure function of the key while its state pointer stays the same — the bitmap is kept per (callback, state,
    index version): the second call makes no callback at all, answers the same, and any mutation of the index forgets it."""
    err = C.c_char_p()
    n, dims, k = 3000, 32, 10
    data = util.make_vectors(n, dims, "f32", seed=23)
    options = Options(METRIC["cos"], None, SCALAR["f32"], dims, 16, 128, 64, False)
    calls = [0]

    def even(key, state):
        calls[0] += 1
        return int(key % 2 == 0)

    def third(key, state):
        calls[0] += 1
        return int(key % 3 == 0)

    even_c, third_c = FILTER(even), FILTER(third)
    keys, distances = np.zeros(k, dtype=np.uint64), np.zeros(k, dtype=np.float32)
    answers = {}
    for memo in ("0", "1", "lazy"):
        monkeypatch.setenv("USEARCH_AMD_FILTER_MEMO", "1" if memo == "1" else "0")
        monkeypatch.setenv("USEARCH_AMD_FILTER_LAZY", "1" if memo == "lazy" else "0")
        index, _ = filled_index(lib, n, dims, options=options, data=data)
        per_call = []
        for callback in (even_c, even_c, third_c, even_c):
            calls[0] = 0
            found = lib.usearch_filtered_search(index, ptr(data[1]), SCALAR["f32"], k, callback, None, ptr(keys), ptr(distances),
                                                C.byref(err))
            ok(err)
            assert found == k
            per_call.append((calls[0], keys.copy(), distances.copy()))
        answers[memo] = per_call
        if memo == "1":
            assert [c for c, _, _ in per_call] == [n, 0, n, 0], "a remembered predicate is not evaluated again"
            # a mutation forgets: the new member must be asked about
            lib.usearch_reserve(index, n + 1, C.byref(err))
            lib.usearch_add(index, n, ptr(data[1]), SCALAR["f32"], C.byref(err))  # key 3000: even, a twin of the query
            ok(err)
            calls[0] = 0
            found = lib.usearch_filtered_search(index, ptr(data[1]), SCALAR["f32"], k, even_c, None, ptr(keys), ptr(distances),
                                                C.byref(err))
            ok(err)
            assert calls[0] == n + 1 and found == k and n in keys.tolist()
        elif memo == "0":
            assert [c for c, _, _ in per_call] == [n, n, n, n]
        else:  # the default: only the members the walk wants to admit are asked about, afresh in every call
            assert all(0 < c < n // 3 for c, _, _ in per_call), [c for c, _, _ in per_call]
        lib.usearch_free(index, C.byref(err))
    for other in ("1", "lazy"):
        for (_, keys_off, distances_off), (_, keys_on, distances_on) in zip(answers["0"], answers[other]):
            assert np.array_equal(keys_off, keys_on) and np.array_equal(distances_off, distances_on)
    assert all(key % 2 == 0 for key in answers["1"][1][1]) and all(key % 3 == 0 for key in answers["1"][2][1])


@pytest.mark.parametrize("metric,dtype", [("l2sq", "i8"), ("cos", "f32")])
def test_filtered_search_calls_back_like_the_reference(lib, reference, metric, dtype):
    """c/lib.cpp:413-429 → index.hpp:4200-4205, 4236-4240: the reference calls the predicate for the members its traversal is about
    to admit to `top`, a few hundred per query. The drop-in (no environment switch) evaluates the callback lazily — provisional runs
    that ask the host about what they met, until a run asks nothing (dropin.hip `lazy_predicate_t`): the same results as the
    reference's `usearch_filtered_search` on the same image, and no more than twice its callbacks, at every selectivity (the
    integer-valued pair: keys and distance bits identical)."""
    err = C.c_char_p()
    n, dims, k = 20_000, 96, 10
    image, vectors, theirs = util.build_image(n, dims, metric, dtype, seed=31)
    queries = util.make_vectors(12, dims, dtype, seed=32)
    index = lib.usearch_init(None, C.byref(err))
    lib.usearch_view_buffer(index, ptr(image), image.size, C.byref(err))
    ok(err)
    keys, distances = np.zeros(k, dtype=np.uint64), np.zeros(k, dtype=np.float32)
    for modulus in (1, 2, 5, 20):  # selectivity 1, 1/2, 1/5, 1/20
        ours, reference_calls = [0], [0]

        def allowed(key, state, modulus=modulus):
            ours[0] += 1
            return int(key % modulus == 0)

        def reference_allowed(key, modulus=modulus):
            reference_calls[0] += 1
            return key % modulus == 0
        callback = FILTER(allowed)
        ours_total = reference_total = 0
        for query in queries:
            ours[0] = reference_calls[0] = 0
            found = lib.usearch_filtered_search(index, ptr(query), SCALAR[dtype], k, callback, None, ptr(keys), ptr(distances), C.byref(err))
            ok(err)
            rfound, rkeys, rdistances = theirs.filtered_search(query, k, reference_allowed, dtype=dtype)
            assert found == rfound and all(key % modulus == 0 for key in keys[:found].tolist())
            if util.exact_pair(metric, dtype):
                assert np.array_equal(keys[:found], rkeys[:found]) and util.same_float_bits(distances[:found], rdistances[:found])
            else:
                assert np.allclose(distances[:found], rdistances[:found], rtol=0, atol=util.tolerance(dtype))
            assert 0 < ours[0] < n // 4, (modulus, ours[0], reference_calls[0], "the predicate must not be evaluated for every member")
            ours_total, reference_total = ours_total + ours[0], reference_total + reference_calls[0]
        print(f"[lazy predicate] {metric} {dtype}, one member in {modulus} allowed: {ours_total} callbacks for {len(queries)} queries, "
              f"the reference {reference_total}")
        assert ours_total <= 2 * reference_total, (modulus, ours_total, reference_total)
    lib.usearch_free(index, C.byref(err))


def test_batch_and_buffers_match_the_reference(lib, reference):
    """`usearch_search_many` over an image the REAL reference built equals the reference's own per-query loop
    (integer-valued metric: bit-exact keys, distances, counts and both counters)."""
    err = C.c_char_p()
    image, vectors, theirs = util.build_image(3000, 96, "l2sq", "i8", seed=21)
    queries = util.make_vectors(200, 96, "i8", seed=22)
    index = lib.usearch_init(None, C.byref(err))
    lib.usearch_view_buffer(index, ptr(image), image.size, C.byref(err))
    ok(err)
    keys = np.zeros((200, 10), dtype=np.uint64)
    distances = np.zeros((200, 10), dtype=np.float32)
    counts = np.zeros(200, dtype=np.uint64)
    visited, computed = C.c_size_t(), C.c_size_t()
    lib.usearch_search_many(index, ptr(queries), SCALAR["i8"], 200, queries.strides[0], 10, ptr(keys), keys.strides[0],
                            ptr(distances), distances.strides[0], ptr(counts), C.byref(visited), C.byref(computed),
                            C.byref(err))
    ok(err)
    rkeys, rdistances, rcounts, rvisited, rcomputed = theirs.search(queries, 10, dtype="i8")
    assert np.array_equal(keys, rkeys) and util.same_float_bits(distances, rdistances)
    assert np.array_equal(counts, rcounts)
    assert visited.value == int(np.sum(rvisited)) and computed.value == int(np.sum(rcomputed))
    # the additive `usearch_cluster_many`: index_dense_gt::cluster(query, level) of the reference, level by level
    for level in (0, 1, 2, 5):
        cluster_keys, cluster_distances = np.zeros(200, dtype=np.uint64), np.zeros(200, dtype=np.float32)
        lib.usearch_cluster_many(index, ptr(queries), SCALAR["i8"], 200, queries.strides[0], level, ptr(cluster_keys),
                                 ptr(cluster_distances), C.byref(err))
        ok(err)
        their_keys, their_distances, *_ = theirs.cluster(queries, level, dtype="i8", threads=1)
        assert np.array_equal(cluster_keys, their_keys) and util.same_float_bits(cluster_distances, their_distances)
    # save_buffer of an untouched image returns it byte for byte
    length = lib.usearch_serialized_length(index, C.byref(err))
    assert length == image.size
    copy = np.zeros(length, dtype=np.uint8)
    lib.usearch_save_buffer(index, ptr(copy), length, C.byref(err))
    ok(err)
    assert np.array_equal(copy, image)
    lib.usearch_free(index, C.byref(err))


# ---------------------------------------------------------------------------------------------------------------------
#  Behaviour a long-lived caller depends on: incremental adds, in-place removal, capacity, the multi flag, bad files,
#  concurrent searches
# ---------------------------------------------------------------------------------------------------------------------

def search_one(lib, index, query, k):
    err = C.c_char_p()
    keys, distances = np.zeros(k, dtype=np.uint64), np.zeros(k, dtype=np.float32)
    found = lib.usearch_search(index, ptr(query), SCALAR["f32"], k, ptr(keys), ptr(distances), C.byref(err))
    ok(err)
    return found, keys, distances


def test_add_after_a_search_links_only_the_new_members(lib):
    """index_gt::add (index.hpp:2780-2879) on a built index: the graph is extended, not rebuilt — every old and new member is
    still found first by its own vector, and an interleaved add / search loop stays cheap."""
    import time
    err = C.c_char_p()
    dimensions, first, more = 32, 3000, 1200
    options = create_options(dimensions, metric_kind=METRIC["l2sq"], connectivity=16, expansion_add=128, expansion_search=64)
    data = create_vectors(first + more + 64, dimensions, seed=5)
    index = lib.usearch_init(C.byref(options), C.byref(err))
    ok(err)
    lib.usearch_reserve(index, first + more + 64, C.byref(err))
    for i in range(first):
        lib.usearch_add(index, i, ptr(data[i]), SCALAR["f32"], C.byref(err))
        ok(err)
    assert search_one(lib, index, data[7], 1)[1][0] == 7  # builds the first 3000
    for i in range(first, first + more):
        lib.usearch_add(index, i, ptr(data[i]), SCALAR["f32"], C.byref(err))
        ok(err)
    t0 = time.perf_counter()
    found, keys, distances = search_one(lib, index, data[first + 5], 3)  # links the 1200 new members only
    extend_seconds = time.perf_counter() - t0
    assert keys[0] == first + 5 and distances[0] == 0.0
    missed = sum(search_one(lib, index, data[i], 1)[1][0] != i for i in range(0, first + more, 37))
    assert missed == 0
    assert lib.usearch_size(index, C.byref(err)) == first + more
    # one member at a time, searched right away: each step links one member
    t0 = time.perf_counter()
    for i in range(first + more, first + more + 64):
        lib.usearch_add(index, i, ptr(data[i]), SCALAR["f32"], C.byref(err))
        ok(err)
        assert search_one(lib, index, data[i], 1)[1][0] == i
    single_seconds = (time.perf_counter() - t0) / 64
    print(f"extend by {more}: {extend_seconds * 1e3:.1f} ms; add + search one at a time: {single_seconds * 1e3:.2f} ms each")
    assert single_seconds < 0.25
    # what was extended saves and loads like anything else, and the reference reads it
    length = lib.usearch_serialized_length(index, C.byref(err))
    buffer = np.zeros(length, dtype=np.uint8)
    lib.usearch_save_buffer(index, ptr(buffer), length, C.byref(err))
    ok(err)
    reference_index = refbind.RefIndex.from_buffer(buffer, view=True, dtype="f32")
    rkeys, *_ = reference_index.search(data[: first + more + 64: 41], 1, dtype="f32", threads=1)
    assert np.array_equal(rkeys[:, 0], np.arange(0, first + more + 64, 41, dtype=np.uint64))
    lib.usearch_free(index, C.byref(err))


@pytest.mark.parametrize("immediate", ["auto", "0", "1"])
def test_readers_racing_a_writer_find_what_was_added(lib, monkeypatch, immediate):
    """index.hpp:2780-2879: a member is findable the moment `add` returns. The drop-in links staged members inside `usearch_add`
    whenever searches interleave with adds (the default, "auto": readers are about), always (USEARCH_AMD_IMMEDIATE_ADD=1) or at the
    next search (=0; read at `usearch_init`): every way four reader threads racing one writer find every member whose `add` has
    returned, first and at distance zero."""
    import threading
    if immediate == "auto":
        monkeypatch.delenv("USEARCH_AMD_IMMEDIATE_ADD", raising=False)
    else:
        monkeypatch.setenv("USEARCH_AMD_IMMEDIATE_ADD", immediate)
    err = C.c_char_p()
    dimensions, first, more = 24, 1500, 120
    options = create_options(dimensions, metric_kind=METRIC["l2sq"], connectivity=16, expansion_add=128, expansion_search=64)
    data = create_vectors(first + more, dimensions, seed=29)
    index = lib.usearch_init(C.byref(options), C.byref(err))
    ok(err)
    lib.usearch_reserve(index, first + more, C.byref(err))
    for i in range(first):
        lib.usearch_add(index, i, ptr(data[i]), SCALAR["f32"], C.byref(err))
        ok(err)
    added = [first]  # members 0 … added[0] - 1 are in: their `add` has returned
    failures = []

    def reader(seed):
        rng = np.random.default_rng(seed)
        failure = C.c_char_p()
        keys = np.zeros(1, dtype=np.uint64)
        distances = np.zeros(1, dtype=np.float32)
        while added[0] < first + more and not failures:
            upto = added[0]
            member = int(rng.integers(max(0, upto - 8), upto))  # mostly the newest ones
            found = lib.usearch_search(index, ptr(data[member]), SCALAR["f32"], 1, ptr(keys), ptr(distances), C.byref(failure))
            if failure.value or found != 1 or keys[0] != member or distances[0] != 0.0:
                failures.append((member, upto, failure.value, int(keys[0]), float(distances[0])))

    readers = [threading.Thread(target=reader, args=(s,)) for s in range(4)]
    for thread in readers:
        thread.start()
    for i in range(first, first + more):
        lib.usearch_add(index, i, ptr(data[i]), SCALAR["f32"], C.byref(err))
        ok(err)
        added[0] = i + 1
    for thread in readers:
        thread.join()
    assert not failures, failures[:3]
    assert lib.usearch_size(index, C.byref(err)) == first + more
    lib.usearch_free(index, C.byref(err))


def test_remove_and_rename_do_not_relink(lib):
    """index_dense_gt::remove (index_dense.hpp:1479-1511) leaves the member in the graph as a tombstone; rename rewrites the
    key. Neither touches a list: the serialized graph before and after differs in the keys only."""
    err = C.c_char_p()
    index, data = filled_index(lib, 400, 24, options=create_options(24, metric_kind=METRIC["l2sq"], connectivity=8))
    assert search_one(lib, index, data[10], 1)[1][0] == 10

    def serialized():
        length = lib.usearch_serialized_length(index, C.byref(err))
        buffer = np.zeros(length, dtype=np.uint8)
        lib.usearch_save_buffer(index, ptr(buffer), length, C.byref(err))
        ok(err)
        return buffer
    before = serialized()
    assert lib.usearch_remove(index, 10, C.byref(err)) == 1
    assert lib.usearch_rename(index, 11, 900011, C.byref(err)) == 1
    found, keys, _ = search_one(lib, index, data[10], 5)
    assert 10 not in keys[:found] and found == 5
    assert search_one(lib, index, data[11], 1)[1][0] == 900011
    after = serialized()
    assert len(before) == len(after)
    changed = np.nonzero(before != after)[0]
    assert 0 < len(changed) <= 16 + 16, "only the two keys and the present / deleted counts of the head may differ"
    lib.usearch_free(index, C.byref(err))


def test_capacity_multi_and_damaged_files(lib):
    err = C.c_char_p()
    # index.hpp:2812-2818: no room, no insertion
    options = create_options(8)
    index = lib.usearch_init(C.byref(options), C.byref(err))
    vector = np.ones(8, dtype=np.float32)
    lib.usearch_add(index, 1, ptr(vector), SCALAR["f32"], C.byref(err))
    assert err.value and b"Reserve capacity" in err.value
    err = C.c_char_p()
    lib.usearch_reserve(index, 2, C.byref(err))
    lib.usearch_add(index, 1, ptr(vector), SCALAR["f32"], C.byref(err))
    ok(err)
    lib.usearch_free(index, C.byref(err))
    # the multi flag survives save → load (head.multi, index_dense.hpp:1046) and the reference sees it too
    options = create_options(8, multi=True)
    index = lib.usearch_init(C.byref(options), C.byref(err))
    lib.usearch_reserve(index, 8, C.byref(err))
    for i in range(4):
        lib.usearch_add(index, 42, ptr(np.full(8, i, dtype=np.float32)), SCALAR["f32"], C.byref(err))
        ok(err)
    assert lib.usearch_count(index, 42, C.byref(err)) == 4
    length = lib.usearch_serialized_length(index, C.byref(err))
    buffer = np.zeros(length, dtype=np.uint8)
    lib.usearch_save_buffer(index, ptr(buffer), length, C.byref(err))
    ok(err)
    loaded = lib.usearch_init(None, C.byref(err))
    lib.usearch_load_buffer(loaded, ptr(buffer), length, C.byref(err))
    ok(err)
    lib.usearch_reserve(loaded, 8, C.byref(err))
    lib.usearch_add(loaded, 42, ptr(np.full(8, 9, dtype=np.float32)), SCALAR["f32"], C.byref(err))
    ok(err)  # a duplicate key is fine in a multi-index
    assert lib.usearch_count(loaded, 42, C.byref(err)) == 5
    assert refbind.metadata(buffer)["multi"]
    # a truncated image is refused when it is loaded, not when a later call walks its tapes
    for cut in (len(buffer) - 1, len(buffer) - 40, len(buffer) // 2):
        broken = lib.usearch_init(None, C.byref(err))
        failure = C.c_char_p()
        lib.usearch_load_buffer(broken, ptr(buffer[:cut].copy()), cut, C.byref(failure))
        assert failure.value, f"a file cut at {cut} of {len(buffer)} bytes was accepted"
        assert lib.usearch_size(broken, C.byref(err)) == 0 and not lib.usearch_contains(broken, 42, C.byref(err))
        lib.usearch_free(broken, C.byref(err))
    lib.usearch_free(index, C.byref(err))
    lib.usearch_free(loaded, C.byref(err))


@pytest.mark.parametrize("coalesce", [False, True])
def test_concurrent_callers_share_the_index(lib, monkeypatch, coalesce):
    """The reference leases a context per thread (index_dense.hpp:1984-2000); here every call in flight leases a workspace — or,
    with `USEARCH_AMD_COALESCE=1` (read when the index is created), calls in flight share a launch (csrc/combiner.hpp).
    Eight threads looping `usearch_search` get the answers a single thread gets."""
    import threading
    err = C.c_char_p()
    if coalesce:
        monkeypatch.setenv("USEARCH_AMD_COALESCE", "1")
    index, data = filled_index(lib, 2000, 48, options=create_options(48, metric_kind=METRIC["cos"], connectivity=16,
                                                                     expansion_add=64, expansion_search=64))
    lib.usearch_change_threads_search(index, 8, C.byref(err))
    expected = [search_one(lib, index, data[i], 5)[1].copy() for i in range(200)]
    failures = []

    def worker(offset):
        for i in range(offset, 200, 8):
            for _ in range(3):
                if not np.array_equal(search_one(lib, index, data[i], 5)[1], expected[i]):
                    failures.append(i)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures
    lib.usearch_free(index, C.byref(err))


def test_the_references_own_c_test_program_passes_against_the_drop_in():
    """/root/reference/c/test.c itself, compiled by `__graft_entry__.build()` where the reference is mounted and linked against
    usearch_amd/lib/libusearch_c.so instead of the reference's library (oracle/Makefile `dropin_test` → oracle/_ref/)."""
    import subprocess
    binary = os.path.join(ROOT, "oracle", "_ref", "reference_test_c")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/reference_test_c was not built (`make -C oracle dropin_test` needs /root/reference)")
    environment = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "usearch_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    result = subprocess.run([binary], capture_output=True, text=True, timeout=600, env=environment)
    assert result.returncode == 0, result.stdout[-2000:] + result.stderr[-2000:]
