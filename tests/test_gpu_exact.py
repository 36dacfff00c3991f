"""GPU: exact (brute-force) search — `search(…, exact=True)` on a snapshot and `usearch_exact_search` on a raw dataset —
against the oracle's restatement of `search_exact_` (index.hpp:4252-4268), the golden answers of the real reference, and
plain numpy."""
import glob
import json
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("metric,dtype,ndim,n,k", [("cos", "f32", 64, 3000, 10), ("l2sq", "f16", 96, 2500, 25),
                                                   ("hamming", "b1", 128, 6000, 10), ("l2sq", "i8", 40, 3000, 100),
                                                   ("ip", "f32", 7, 300, 3), ("cos", "f16", 768, 5000, 10),
                                                   ("hamming", "b1", 64, 100, 300)])
def test_exact_matches_oracle_bit_for_bit(reference, metric, dtype, ndim, n, k):
    from usearch_amd import Index
    removed = np.arange(3, n, 11) + 1000 if metric == "hamming" else ()
    image, _, _ = util.build_image(n, ndim, metric, dtype, seed=91, remove=removed)
    queries = util.make_vectors(70, ndim, dtype, seed=92)
    index = Index.restore(image)
    got = index.search(queries, k, exact=True)
    keys, distances, counts, *_ = util.oracle_search(image, queries, k, dtype, lanes=index.lanes_per_row, exact=True)
    assert np.array_equal(got.counts, counts)
    assert np.array_equal(got.keys, keys), "ties must resolve like lower_bound insertion in slot order"
    assert util.same_float_bits(got.distances, distances)
    assert not np.isin(got.keys, removed).any()


def test_exact_matches_reference_golden():
    from usearch_amd import Index
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        data = np.load(path)
        meta = json.loads(str(data["meta"]))
        got = Index.restore(data["image"]).search(data["queries"], meta["k"], dtype=meta["dtype"], exact=True)
        assert np.array_equal(got.counts, data["exact_counts"])
        if util.exact_pair(meta["metric"], meta["dtype"]):
            assert np.array_equal(got.keys, data["exact_keys"]), path
            assert util.same_float_bits(got.distances, data["exact_distances"])
        else:
            assert (got.keys == data["exact_keys"]).mean() > 0.99


def test_exact_search_of_a_raw_dataset():
    """`usearch_exact_search`: keys are row offsets; top-1 of a row against its own dataset is itself
    (python/scripts/test_tooling.py:71-103, cpp/test.cpp:878-897)."""
    import usearch_amd
    rng = np.random.default_rng(5)
    dataset = rng.standard_normal((4000, 48)).astype(np.float32)
    queries = dataset[:300] + 1e-4 * rng.standard_normal((300, 48)).astype(np.float32)
    keys, distances = usearch_amd.exact_search(dataset, queries, 10, metric="l2sq")
    assert np.array_equal(keys[:, 0], np.arange(300))
    brute = ((queries[:, None, :] - dataset[None, :, :]) ** 2).sum(-1)
    order = np.argsort(brute, axis=1)[:, :10]
    assert (keys == order).mean() > 0.999
    assert np.allclose(distances, np.take_along_axis(brute, order, axis=1), rtol=1e-4, atol=1e-5)
    assert np.all(np.diff(distances, axis=1) >= 0)
    # strided rows and bit vectors
    wide = np.zeros((1000, 24), dtype=np.uint8)
    wide[:, :16] = rng.integers(0, 256, (1000, 16), dtype=np.uint8)
    keys, distances = usearch_amd.exact_search(wide[:, :16], wide[:50, :16], 5, metric="hamming")
    assert np.array_equal(keys[:, 0] == np.arange(50), distances[:, 0] == 0) and np.all(distances[:, 0] == 0)
