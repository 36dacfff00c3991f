"""CPU: the oracle against the REAL reference compiled from /root/reference (oracle/_ref), side by side on seeded inputs.
Skipped where neither the built library nor the reference sources are available."""
import numpy as np
import pytest

from tests import util

CASES = [
    ("cos", "f32", 32, 1500, 16, 10, 64), ("l2sq", "f32", 7, 700, 5, 4, 16), ("ip", "f16", 48, 900, 16, 10, 64),
    ("cos", "f16", 128, 1200, 16, 10, 100), ("l2sq", "i8", 64, 2000, 16, 10, 64), ("cos", "i8", 40, 800, 16, 10, 64),
    ("ip", "i8", 24, 600, 8, 5, 32), ("hamming", "b1", 64, 3000, 16, 10, 64), ("hamming", "b1", 256, 1000, 3, 20, 20),
]


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity,k,expansion", CASES)
def test_search_side_by_side(reference, metric, dtype, ndim, n, connectivity, k, expansion):
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=5, connectivity=connectivity)
    queries = util.make_vectors(120, ndim, dtype, seed=6)
    queries[:20] = vectors[:20]
    ref_index.expansion_search = expansion
    rkeys, rdists, rcounts, rvisited, rcomputed = ref_index.search(queries, k, dtype=dtype, threads=1)
    keys, dists, counts, visited, computed = util.oracle_search(image, queries, k, dtype, expansion, lanes=0)
    assert np.array_equal(counts, rcounts)
    if dtype in ("i8", "b1") and metric != "cos":
        assert np.array_equal(keys, rkeys) and util.same_float_bits(dists, rdists)
        assert np.array_equal(visited, rvisited) and np.array_equal(computed, rcomputed)
    else:
        found = np.arange(k)[None, :] < rcounts[:, None]
        tolerance = 2e-3 if dtype == "f16" else 1e-5
        reference_d = np.where(found, rdists, 0)
        assert np.all(np.abs(np.where(found, dists, 0) - reference_d) <= tolerance * np.maximum(1, np.abs(reference_d)))
        assert ((keys == rkeys) | ~found).mean() > 0.99
        assert np.abs(computed.astype(float).mean() / rcomputed.astype(float).mean() - 1) < 0.02


def test_casts_side_by_side(reference):
    """Queries in a foreign scalar kind: the oracle's casts (index_plugins.hpp:1105-1224) must route like the reference's."""
    for dtype, metric in (("f16", "cos"), ("i8", "cos"), ("b1", "hamming"), ("f32", "l2sq")):
        image, _, ref_index = util.build_image(1000, 64, metric, dtype, seed=8)
        for query_dtype in ("f32", "f16"):
            if query_dtype == dtype:
                continue
            queries = util.make_vectors(50, 64, query_dtype, seed=9)
            rkeys, rdists, *_ = ref_index.search(queries, 10, dtype=query_dtype, threads=1)
            keys, dists, *_ = util.oracle_search(image, queries, 10, query_dtype, 64)
            if dtype in ("i8", "b1"):
                assert np.array_equal(keys, rkeys)
            else:
                assert (keys == rkeys).mean() > 0.99


def test_tombstones_side_by_side(reference):
    removed = np.arange(0, 900, 4) + 1000
    image, _, ref_index = util.build_image(900, 16, "l2sq", "i8", seed=10, remove=removed)
    queries = util.make_vectors(80, 16, "i8", seed=11)
    rkeys, rdists, rcounts, *_ = ref_index.search(queries, 10, threads=1)
    keys, dists, counts, *_ = util.oracle_search(image, queries, 10, "i8", 64)
    assert np.array_equal(keys, rkeys) and np.array_equal(counts, rcounts)
    assert not np.isin(keys, removed).any()
