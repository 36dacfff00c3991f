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
                                           ("f64", "b1"), ("i8", "b1"), ("f32", "f32")])
def test_query_casts_match_oracle(source, target):
    import usearch_amd
    rng = np.random.default_rng(3)
    for ndim in (8, 64, 96):
        if source == "b1":
            vector = rng.integers(0, 256, ndim // 8, dtype=np.uint8)
        elif source == "i8":
            vector = rng.integers(-127, 128, ndim).astype(np.int8)
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
    for name, data in cases.items():
        data = np.ascontiguousarray(data)
        err = C.c_char_p()
        handle = library.usearch_amd_snapshot_from_buffer(C.c_void_p(data.ctypes.data), data.size, 0, C.byref(err))
        assert not handle and err.value, name
