"""CPU: the product library loads and exports every symbol its public header declares; host-side logic that needs no
GPU (query casts) matches the oracle; and without a GPU the engine fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oraclebind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"USEARCH_AMD_EXPORT[^;(]*?\b(usearch\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import usearch_amd
    library = usearch_amd.library()
    names = declared_symbols("usearch_amd.h")
    assert len(names) >= 20
    for name in names:
        assert hasattr(library, name), f"{name} is declared in include/usearch_amd.h but not exported"
    assert sorted(usearch_amd.EXPORTED_SYMBOLS) == names


@pytest.mark.parametrize("source,target", [("f32", "f16"), ("f32", "i8"), ("f32", "b1"), ("f16", "f32"), ("f16", "i8"),
                                           ("i8", "f32"), ("i8", "f16"), ("b1", "f32"), ("b1", "i8"), ("f64", "f16"),
                                           ("f64", "b1"), ("i8", "b1"), ("f32", "f32"), ("f32", "bf16"), ("bf16", "f32"),
                                           ("bf16", "i8"), ("i8", "bf16"), ("b1", "bf16"), ("bf16", "b1"), ("f64", "bf16"),
                                           ("bf16", "f16"), ("f32", "f64"), ("i8", "f64"), ("bf16", "bf16")])
def test_query_casts_match_oracle(source, target):
    import usearch_amd
    rng = np.random.default_rng(3)
    for ndim in (8, 64, 96):
        if source == "b1":
            vector = rng.integers(0, 256, ndim // 8, dtype=np.uint8)
        elif source == "i8":
            vector = rng.integers(-127, 128, ndim).astype(np.int8)
        elif source == "bf16":  # brain floats travel as bit patterns: the upper halves of f32 values
            vector = ((rng.standard_normal(ndim) * 3).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        else:
            vector = (rng.standard_normal(ndim) * 3).astype({"f32": np.float32, "f16": np.float16, "f64": np.float64}[source])
        ours = usearch_amd.cast(vector, source, target, ndim)
        theirs = oraclebind.cast(vector, source, target, ndim)
        assert (ours is None) == (theirs is None)
        if ours is not None:
            assert np.array_equal(ours, theirs), f"{source}->{target} ndim={ndim}"


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import usearch_amd
    if usearch_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    image = np.load(os.path.join(ROOT, "tests", "golden", "l2sq_f32_3.npz"))["image"]
    with pytest.raises(RuntimeError):
        usearch_amd.Index.restore(image)


def test_corrupt_images_are_rejected():
    """Parsing happens before any device work, so the reference's error wording is observable without a GPU."""
    import usearch_amd
    library = usearch_amd.library()
    image = np.load(os.path.join(ROOT, "tests", "golden", "l2sq_f32_3.npz"))["image"].copy()
    cases = {"truncated": image[: len(image) // 2], "no magic": None, "tiny": image[:4]}
    broken = image.copy()
    rows, cols = np.frombuffer(image[:8].tobytes(), dtype=np.uint32)
    broken[8 + int(rows) * int(cols)] ^= 0xFF
    cases["no magic"] = broken
    from tests import util
    cases["saved without vectors"] = util.without_vectors(image)  # index_dense.hpp:1004: nothing to search in there
    # the variable-length part: levels and node tapes are validated when the image is opened (a later key lookup or `get`
    # walks the tapes without looking back)
    head = 8 + int(rows) * int(cols) + 64
    size, max_level = (int(v) for v in np.frombuffer(image[head:head + 40].tobytes(), dtype=np.uint64)[[0, 3]])
    levels_at = head + 40
    negative = image.copy()
    negative[levels_at:levels_at + 2] = np.frombuffer(np.int16(-3).tobytes(), dtype=np.uint8)
    cases["negative level"] = negative
    taller = image.copy()
    taller[levels_at + 2:levels_at + 4] = np.frombuffer(np.int16(max_level + 5).tobytes(), dtype=np.uint8)
    cases["level above the index's"] = taller
    cases["128-bit keys"] = util.with_uuid_keys_announced(image)  # index_dense_big_t's uuid_t: refused by name
    wide = util.with_40_bit_slots(image)  # uint40_t slots load (tests/test_gpu_dropin.py); cut short they fail like any tape
    cases["40-bit slots, last tape cut"] = wide[: len(wide) - 3]
    cases["last tape cut"] = image[: len(image) - 7]
    cases["tapes missing"] = image[: levels_at + 2 * size + 5]
    expected = {"128-bit keys": b"128-bit keys", "40-bit slots, last tape cut": b"nodes", "negative level": b"nodes", "level above the index's": b"nodes", "last tape cut": b"nodes", "tapes missing": b"nodes"}
    for name, data in cases.items():
        data = np.ascontiguousarray(data)
        err = C.c_char_p()
        handle = library.usearch_amd_snapshot_from_buffer(C.c_void_p(data.ctypes.data), data.size, 0, C.byref(err))
        assert not handle and err.value, name
        if name == "saved without vectors":
            assert b"exclude_vectors" in err.value
        if name in expected:  # the reference's wording: "Failed to pull nodes from the stream"
            assert expected[name] in err.value, (name, err.value)


REFERENCE_C_ABI = [  # the 38 entry points of /root/reference/c/usearch.h:116-481 (verified with `nm -D`)
    "version", "init", "free", "memory_usage", "hardware_acceleration", "serialized_length", "save", "load", "view",
    "metadata", "save_buffer", "load_buffer", "view_buffer", "metadata_buffer", "size", "capacity", "dimensions",
    "connectivity", "reserve", "expansion_add", "expansion_search", "change_expansion_add", "change_expansion_search",
    "change_threads_add", "change_threads_search", "change_metric_kind", "change_metric", "add", "contains", "count",
    "search", "filtered_search", "get", "remove", "rename", "distance", "exact_search", "clear"]


def test_drop_in_library_exports_the_reference_c_abi():
    """`libusearch_c.so` carries every symbol a program linked against the reference's `libusearch_c` resolves, plus the
    additive batch entry points its header declares. (No compute calls: there is no GPU here.)"""
    library = C.CDLL(os.path.join(ROOT, "usearch_amd", "lib", "libusearch_c.so"))
    assert len(REFERENCE_C_ABI) == 38
    for name in REFERENCE_C_ABI:
        assert hasattr(library, f"usearch_{name}"), f"usearch_{name} is missing from the drop-in"
    header = open(os.path.join(ROOT, "include", "usearch_c_dropin.h")).read()
    declared = sorted(set(re.findall(r"USEARCH_EXPORT[^;(]*?\b(usearch_\w+)\s*\(", header)))
    # the 38 + search_many, cluster_many, search_exact_many, threads_search, gpu_sync, gpu_release, c_api + the seven filter entry points
    assert len(declared) == 52, declared
    for name in declared:
        assert hasattr(library, name), f"{name} is declared in include/usearch_c_dropin.h but not exported"
    reference_header = "/root/reference/c/usearch.h"
    if os.path.exists(reference_header):  # this container only: the list above is the reference's, not ours
        theirs = sorted(set(re.findall(r"USEARCH_EXPORT[^;(]*?\b(usearch_\w+)\s*\(", open(reference_header).read()))
                        - {"usearch_distance_t"})  # a return type caught by the pattern, not a function
        assert theirs == sorted(f"usearch_{name}" for name in REFERENCE_C_ABI)
    library.usearch_version.restype = C.c_char_p
    assert library.usearch_version() == b"2.21.0"


def test_an_allocation_failure_is_an_error_string_not_an_exception_across_the_c_abi():
    """Every drop-in entry point that can allocate runs inside `guarded` (dropin.hip): an impossible reservation comes back as
    the reference's out-of-memory wording and the index stays usable. (Host-side only: no device is touched.)"""

    class Options(C.Structure):  # usearch_init_options_t, c/usearch.h:64-110
        _fields_ = [("metric_kind", C.c_int), ("metric", C.c_void_p), ("quantization", C.c_int),
                    ("dimensions", C.c_size_t), ("connectivity", C.c_size_t), ("expansion_add", C.c_size_t),
                    ("expansion_search", C.c_size_t), ("multi", C.c_bool)]

    library = C.CDLL(os.path.join(ROOT, "usearch_amd", "lib", "libusearch_c.so"))
    error_t = C.POINTER(C.c_char_p)
    library.usearch_init.restype = C.c_void_p
    library.usearch_init.argtypes = [C.POINTER(Options), error_t]
    library.usearch_reserve.argtypes = [C.c_void_p, C.c_size_t, error_t]
    library.usearch_size.restype = C.c_size_t
    library.usearch_size.argtypes = [C.c_void_p, error_t]
    library.usearch_contains.restype = C.c_bool
    library.usearch_contains.argtypes = [C.c_void_p, C.c_uint64, error_t]
    library.usearch_free.argtypes = [C.c_void_p, error_t]
    options = Options(metric_kind=1, metric=None, quantization=1, dimensions=16, connectivity=16, expansion_add=128,
                      expansion_search=64, multi=False)
    error = C.c_char_p()
    index = library.usearch_init(C.byref(options), C.byref(error))
    assert index and not error.value
    library.usearch_reserve(index, 1 << 60, C.byref(error))  # 2^60 keys of 8 bytes: beyond any allocator
    assert error.value in (b"Out of memory!", b"Unexpected failure inside the index")
    error = C.c_char_p()
    library.usearch_capacity.restype = C.c_size_t
    library.usearch_capacity.argtypes = [C.c_void_p, error_t]
    assert library.usearch_capacity(index, C.byref(error)) == 0  # the failed reservation left nothing behind
    assert library.usearch_size(index, C.byref(error)) == 0 and not error.value
    assert library.usearch_contains(index, 42, C.byref(error)) is False and not error.value
    library.usearch_free(index, C.byref(error))


def test_one_hip_runtime_per_process_whichever_side_loads_first():
    """`usearch_amd.index.library()` before `import torch` must not leave two HIP runtimes mapped (the second one would find
    no device): the engine maps torch's bundled copy first when torch is installed (index.py `_share_hip_runtime`)."""
    import subprocess
    import sys
    script = ("import sys; sys.path.insert(0, %r)\n"
              "import usearch_amd.index as ix\n"
              "ix.library()\n"
              "import torch\n"
              "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "1"


def test_placement_draws_keep_the_fastest_and_release_the_rest():
    """`Index.restore_placed` (index.py): uploads `draws` times (or until one is `good_enough`), holds the losers while it draws
    (so the next one lands elsewhere) when memory allows, keeps the fastest, releases every loser, never closes `first`."""
    from usearch_amd.index import Index

    class Stub(Index):
        made, closed = [], []
        memory_usage = 100

        def __init__(self, number):
            self.number, self._handle = number, None

        @classmethod
        def restore(cls, source, device=0, expansion_search=0, vectors=None):
            stub = cls(len(cls.made))
            cls.made.append(stub)
            return stub

        def close(self):
            Stub.closed.append(self.number)

        def __del__(self):
            pass

    def run(speeds, **options):
        Stub.made, Stub.closed = [], []
        return Stub.restore_placed(b"", lambda index: speeds[index.number], **options)

    best, report = run([51.7, 51.8, 48.9, 45.4, 51.7], draws=5)
    assert best.number == 3 and report == {"probe_ms": [51.7, 51.8, 48.9, 45.4, 51.7], "kept": 3}
    assert sorted(Stub.closed) == [0, 1, 2, 4]
    best, report = run([51.7, 51.8, 48.9, 45.4, 51.7], draws=5, good_enough=49.0)  # a caller that knows what fast is
    assert best.number == 2 and report["probe_ms"] == [51.7, 51.8, 48.9] and sorted(Stub.closed) == [0, 1]
    best, report = run([51.7, 51.8, 51.75, 51.72], draws=4)  # one speed only: all draws spent, the best of them kept
    assert best.number == 0 and len(report["probe_ms"]) == 4 and sorted(Stub.closed) == [1, 2, 3]
    best, report = run([51.7, 51.8], draws=4, free_bytes=lambda: 110)  # no room for a second one next to the first
    assert best.number == 0 and report["probe_ms"] == [51.7] and Stub.closed == []
    already_there = Stub(99)
    best, report = run({99: 51.7, 0: 45.0}, draws=2, first=already_there)  # the resident index takes part and is never closed
    assert best.number == 0 and report["kept"] == 1 and 99 not in Stub.closed
    best, report = run({99: 45.0, 0: 51.7}, draws=2, first=already_there)
    assert best is already_there and Stub.closed == [0]


def test_the_bench_hash_names_files_that_exist():
    """`config.sources` of the bench line (and of every traffic.json) hashes the sources the walk is compiled from: a renamed file must
    not silently fall out of it."""
    import bench
    directory = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "usearch_amd", "csrc")
    for name in bench.WALK_SOURCES:
        assert os.path.isfile(os.path.join(directory, name)), name
    pairs = [name for name in os.listdir(directory) if name.startswith("search_") and name.endswith(".hip")]
    assert len(pairs) == 29  # one translation unit per (metric, scalar) pair of the reference's dispatch table
    assert len(bench.source_hash()) == 16


def test_calls_in_flight_share_a_launch():
    """The drop-in's call combiner (usearch_amd/csrc/combiner.hpp: pure host logic) under 32 threads with a mock launch
    (tests/cpp/combiner_test.cpp): every call gets its own answer, callers that queue up during a launch go out together,
    groups never mix query kinds or result counts, an exception inside a launch becomes every call's error string."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = "/tmp/usearch_amd_combiner_test"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread",
                           os.path.join(root, "tests", "cpp", "combiner_test.cpp"), "-o", binary])
    out = subprocess.run([binary], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "PASSED" in out.stdout, out.stdout + out.stderr



def test_calls_in_flight_share_a_launch_under_the_thread_sanitizer():
    """The same program under ThreadSanitizer (clang's runtime: it knows `pthread_cond_clockwait`, which the launcher's timed wait
    for returning callers uses): no data race, no lock-order report in the combiner."""
    import subprocess
    compiler = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(compiler):
        pytest.skip("no clang++ with a ThreadSanitizer runtime here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = "/tmp/usearch_amd_combiner_test_tsan"
    built = subprocess.run([compiler, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
                            os.path.join(root, "tests", "cpp", "combiner_test.cpp"), "-o", binary], capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("the ThreadSanitizer runtime does not link here: " + built.stderr[-300:])
    out = subprocess.run([binary], capture_output=True, text=True, timeout=300, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    assert out.returncode == 0 and "PASSED" in out.stdout, out.stdout + out.stderr[-1000:]


def test_ctypes_mirrors_follow_the_header_field_by_field():
    """The Python host side mirrors the plain structs of include/usearch_amd.h by hand; an out-struct that grew in the header and not in
    the mirror would be written past the Python buffer. Scalar structs are compared name by name, type by type, in order."""
    from usearch_amd import index as host
    header = open(os.path.join(ROOT, "include", "usearch_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    ctype = {"uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "float": C.c_float, "double": C.c_double, "int": C.c_int,
             "size_t": C.c_size_t, "int32_t": C.c_int32, "int64_t": C.c_int64}
    checked = 0
    for c_name, mirror in (("usearch_amd_tuning_t", host.Tuning), ("usearch_amd_stats_t", host.Stats),
                           ("usearch_amd_build_config_t", host.BuildConfig), ("usearch_amd_build_stats_t", host.BuildStats)):
        found = re.search(r"typedef struct " + c_name + r"\s*\{(.*?)\}\s*" + c_name + r"\s*;", header, flags=re.S)
        if not found:
            continue
        declared = []
        for kind, names in re.findall(r"^\s*(\w+)\s+(\w+(?:\s*,\s*\w+)*)\s*;", found.group(1), flags=re.M):
            declared += [(name.strip(), ctype[kind]) for name in names.split(",")]
        assert declared, c_name
        assert [(name, kind) for name, kind in mirror._fields_] == declared, f"{c_name} and its ctypes mirror differ"
        checked += 1
    assert checked >= 2, "the header's struct names changed: teach this test the new ones"
