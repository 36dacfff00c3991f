"""CPU: the oracle against the REAL reference compiled from /root/reference (oracle/_ref), side by side on seeded inputs.
Skipped where neither the built library nor the reference sources are available."""
import numpy as np
import pytest

from oracle import oraclebind
from tests import util

CASES = [
    ("cos", "f32", 32, 1500, 16, 10, 64), ("l2sq", "f32", 7, 700, 5, 4, 16), ("ip", "f16", 48, 900, 16, 10, 64),
    ("cos", "f16", 128, 1200, 16, 10, 100), ("l2sq", "i8", 64, 2000, 16, 10, 64), ("cos", "i8", 40, 800, 16, 10, 64),
    ("ip", "i8", 24, 600, 8, 5, 32), ("hamming", "b1", 64, 3000, 16, 10, 64), ("hamming", "b1", 256, 1000, 3, 20, 20),
    # the rest of the reference's dispatch table (index_plugins.hpp:1930-2008)
    ("cos", "bf16", 48, 900, 16, 10, 64), ("l2sq", "bf16", 20, 600, 8, 5, 32), ("ip", "f64", 32, 700, 16, 10, 64),
    ("cos", "f64", 24, 700, 16, 10, 64), ("pearson", "f32", 32, 900, 16, 10, 64), ("pearson", "f16", 48, 700, 16, 10, 64),
    ("pearson", "i8", 40, 800, 16, 10, 64), ("pearson", "f64", 16, 600, 8, 5, 32), ("pearson", "bf16", 24, 600, 8, 5, 32),
    ("divergence", "f32", 32, 700, 16, 10, 64), ("divergence", "f16", 24, 600, 16, 10, 64),
    ("divergence", "f64", 16, 500, 8, 5, 32), ("haversine", "f32", 2, 1500, 16, 10, 64), ("haversine", "f64", 2, 800, 8, 5, 32),
    ("tanimoto", "b1", 128, 2000, 16, 10, 64), ("jaccard", "b1", 64, 800, 8, 5, 32), ("sorensen", "b1", 96, 1500, 16, 10, 64),
]


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity,k,expansion", CASES)
def test_search_side_by_side(reference, metric, dtype, ndim, n, connectivity, k, expansion):
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=5, connectivity=connectivity)
    queries = util.make_vectors(120, ndim, dtype, seed=6, metric=metric)
    queries[:20] = vectors[:20]
    ref_index.expansion_search = expansion
    rkeys, rdists, rcounts, rvisited, rcomputed = ref_index.search(queries, k, dtype=dtype, threads=1)
    keys, dists, counts, visited, computed = util.oracle_search(image, queries, k, dtype, expansion, lanes=0)
    assert np.array_equal(counts, rcounts)
    if util.exact_pair(metric, dtype):
        assert np.array_equal(keys, rkeys) and util.same_float_bits(dists, rdists)
        assert np.array_equal(visited, rvisited) and np.array_equal(computed, rcomputed)
    else:
        found = np.arange(k)[None, :] < rcounts[:, None]
        tolerance = util.tolerance(dtype)
        reference_d = np.where(found, rdists, 0)
        assert np.all(np.abs(np.where(found, dists, 0) - reference_d) <= tolerance * np.maximum(1, np.abs(reference_d)))
        assert ((keys == rkeys) | ~found).mean() > 0.99
        assert np.abs(computed.astype(float).mean() / rcomputed.astype(float).mean() - 1) < 0.02


def test_casts_side_by_side(reference):
    """Queries in a foreign scalar kind: the oracle's casts (index_plugins.hpp:1105-1224) must route like the reference's."""
    for dtype, metric in (("f16", "cos"), ("i8", "cos"), ("b1", "hamming"), ("f32", "l2sq"), ("bf16", "cos"), ("f64", "l2sq"),
                          ("b1", "tanimoto")):
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


DISTANCE_PAIRS = ([(m, d) for m in ("ip", "cos", "l2sq", "pearson") for d in ("bf16", "i8", "f16", "f32", "f64")]
                  + [("divergence", d) for d in ("bf16", "f16", "f32", "f64")] + [("haversine", "f32"), ("haversine", "f64")]
                  + [(m, "b1") for m in ("hamming", "tanimoto", "jaccard", "sorensen")])


@pytest.mark.parametrize("metric,dtype", DISTANCE_PAIRS)
def test_every_dispatched_pair_at_the_distance_level(reference, metric, dtype):
    """`usearch_distance` of the compiled reference against the restatement, for every (metric, scalar) pair of
    `configure_with_autovec` (index_plugins.hpp:1930-2008), in the reference's loop order and in the kernels' layouts."""
    from oracle import oraclebind
    for ndim in ((2,) if metric == "haversine" else (3, 17, 96, 257)):
        a = util.make_vectors(12, ndim, dtype, 1, clustered=False, metric=metric)
        b = util.make_vectors(12, ndim, dtype, 2, clustered=False, metric=metric)
        for i in range(12):
            want = reference.distance(a[i], b[i], metric, dtype, ndim)
            for lanes in (0, 1, 2, 8):
                got = oraclebind.distance(a[i], b[i], metric, dtype, ndim, lanes)
                if util.exact_pair(metric, dtype) or (dtype == "i8" and metric == "cos"):
                    assert got == want, (ndim, lanes)
                else:
                    assert abs(got - want) <= 5e-6 * max(1.0, abs(want)), (ndim, lanes, got, want)


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity", [
    ("l2sq", "i8", 32, 6000, 4), ("hamming", "b1", 64, 5000, 3), ("cos", "f32", 24, 4000, 4), ("tanimoto", "b1", 128, 3000, 2),
])
def test_cluster_side_by_side(reference, metric, dtype, ndim, n, connectivity):
    """`index_dense_gt::cluster(query, level)`: the restated greedy descent against the reference's, level by level."""
    from oracle import oraclebind
    image, _, ref_index = util.build_image(n, ndim, metric, dtype, seed=3, connectivity=connectivity)
    queries = util.make_vectors(60, ndim, dtype, seed=4, metric=metric)
    oracle = oraclebind.OracleIndex(image)
    max_level = int(ref_index.graph_shape()[0])
    for level in (0, 1, 2, 3, max_level, max_level + 2):
        rkeys, rdistances, rvisited, rcomputed = ref_index.cluster(queries, level, threads=1)
        keys, distances, visited, computed = oracle.cluster(queries, level, dtype=dtype)
        assert np.array_equal(keys, rkeys) and np.array_equal(visited, rvisited) and np.array_equal(computed, rcomputed)
        if util.exact_pair(metric, dtype):
            assert util.same_float_bits(distances, rdistances)
        else:
            assert np.all(np.abs(distances - rdistances) <= 1e-5 * np.maximum(1, np.abs(rdistances)))


@pytest.mark.parametrize("metric,dtype,ndim,n", [("l2sq", "i8", 48, 2500), ("hamming", "b1", 128, 3000), ("cos", "f32", 32, 2000)])
def test_filtered_search_side_by_side(reference, metric, dtype, ndim, n):
    """`usearch_filtered_search` (c/usearch.h:392-395 → index_dense.hpp:2053-2085 with a predicate → the two `allow` tests of
    the traversal, index.hpp:4200-4205 and 4236-4240) against the oracle's restatement, for predicates of every selectivity:
    members that fail the test still route, they just never enter the result."""
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=41)
    queries = util.make_vectors(40, ndim, dtype, seed=42, metric=metric)
    queries[:5] = vectors[:5]
    index = oraclebind.OracleIndex(image)
    ref_index.expansion_search = 64
    predicates = {"every third": lambda key: key % 3 == 0, "one in fifty": lambda key: key % 50 == 7,
                  "the upper half": lambda key: key >= n // 2, "all": lambda key: True, "none": lambda key: False}
    for name, predicate in predicates.items():
        for query in queries:
            rfound, rkeys, rdists = ref_index.filtered_search(query, 10, predicate, dtype=dtype)
            found, keys, dists = index.filtered_search(query, 10, predicate, dtype=dtype, expansion=64)
            assert found == rfound, name
            assert all(predicate(int(key)) for key in rkeys[:rfound]), name
            if util.exact_pair(metric, dtype):
                assert np.array_equal(keys[:found], rkeys[:found]), name
                assert util.same_float_bits(dists[:found], rdists[:found]), name
            else:
                assert np.all(np.abs(dists[:found] - rdists[:found]) <= util.tolerance(dtype) * np.maximum(1, np.abs(rdists[:found])))
                assert (keys[:found] == rkeys[:found]).mean() > 0.9 if found else True
