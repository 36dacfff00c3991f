"""GPU: the distance kernels alone — bit-exact against the oracle in the kernels' summation layout (`lanes = G`),
within tolerance against the oracle's reference loop order, and exact for the integer-valued metrics."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

CASES = [
    ("cos", "f32", 128), ("ip", "f32", 97), ("l2sq", "f32", 3), ("cos", "f32", 768), ("l2sq", "f32", 20),
    ("cos", "f16", 768), ("ip", "f16", 64), ("l2sq", "f16", 100), ("cos", "f16", 7),
    ("l2sq", "i8", 96), ("cos", "i8", 96), ("ip", "i8", 33), ("l2sq", "i8", 256),
    ("hamming", "b1", 128), ("hamming", "b1", 8), ("hamming", "b1", 1024), ("hamming", "b1", 72),
    # the rest of the reference's dispatch table (index_plugins.hpp:1930-2008)
    ("cos", "bf16", 768), ("ip", "bf16", 64), ("l2sq", "bf16", 100), ("pearson", "bf16", 40),
    ("cos", "f64", 96), ("ip", "f64", 200), ("l2sq", "f64", 3), ("pearson", "f64", 33),
    ("pearson", "f32", 128), ("pearson", "f32", 1), ("pearson", "f16", 100), ("pearson", "i8", 96), ("pearson", "i8", 257),
    ("divergence", "f32", 64), ("divergence", "f16", 48), ("divergence", "bf16", 32), ("divergence", "f64", 24),
    ("haversine", "f32", 2), ("haversine", "f64", 2),
    ("tanimoto", "b1", 128), ("jaccard", "b1", 72), ("sorensen", "b1", 1024), ("sorensen", "b1", 8),
]


@pytest.mark.parametrize("metric,dtype,ndim", CASES)
def test_distances_against_oracle(reference, metric, dtype, ndim):
    from oracle import oraclebind
    from usearch_amd import Index
    n, q, per = 200, 9, 37
    image, vectors, _ = util.build_image(n, ndim, metric, dtype, seed=3, clustered=False)
    queries = util.make_vectors(q, ndim, dtype, seed=4, clustered=False, metric=metric)
    if metric == "cos" and dtype != "b1":
        queries[0] = 0  # zero-norm branch of metric_cos_gt (index_plugins.hpp:1353-1358)
    index = Index.restore(image)
    lanes = index.lanes_per_row
    rng = np.random.default_rng(5)
    slots = rng.integers(0, n, size=(q, per)).astype(np.uint32)
    got = index.distances(queries, slots)
    want_layout = np.zeros_like(got)
    want_loop = np.zeros_like(got)
    for i in range(q):
        for j in range(per):
            want_layout[i, j] = oraclebind.distance(queries[i], vectors[slots[i, j]], metric, dtype, ndim, lanes)
            want_loop[i, j] = oraclebind.distance(queries[i], vectors[slots[i, j]], metric, dtype, ndim, 0)
    if util.layout_exact(metric):
        assert util.same_float_bits(got, want_layout), f"lanes={lanes} max diff {np.abs(got - want_layout).max()}"
    if dtype in ("i8", "b1") and metric != "pearson":
        assert util.same_float_bits(got, want_loop)
    else:
        tolerance = util.tolerance(dtype)
        assert np.all(np.abs(got - want_loop) <= tolerance * np.maximum(1.0, np.abs(want_loop)))
