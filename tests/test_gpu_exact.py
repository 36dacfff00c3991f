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


@pytest.mark.parametrize("tile", [64, 256])
@pytest.mark.parametrize("metric,dtype,ndim,n,k", [("l2sq", "i8", 96, 20011, 10), ("cos", "i8", 40, 9000, 64),
                                                   ("ip", "i8", 130, 5003, 7), ("cos", "i8", 200, 9000, 16)])
def test_tiled_exact_search_is_bit_identical_for_i8(reference, monkeypatch, metric, dtype, ndim, n, k, tile):
    """The matrix-unit kernels (exact_tiled.hip: 64 queries per workgroup, and the wide tile of 256 for batches that fill the
    chip with it — forced here either way) sum integers exactly and close with the same arithmetic as the wave-per-query
    kernel, so keys, distance bits and counts are identical — ties resolved as lower_bound insertion in slot order does —
    tombstones skipped, ragged query counts, ragged row tiles, ragged dimensions."""
    from usearch_amd import Index
    monkeypatch.setenv("USEARCH_AMD_EXACT_TILE", str(tile))
    removed = np.arange(5, n, 13) + 1000
    image, vectors, _ = util.build_image(n, ndim, metric, dtype, seed=93, remove=removed[:200], expansion_add=16, connectivity=4)
    queries = util.make_vectors(131 if tile == 64 else 700, ndim, dtype, seed=94)
    queries[:20] = vectors[:20]
    index = Index.restore(image)
    exact = index.search(queries, k, exact=True)
    tiled = index.search(queries, k, exact="tiled")
    assert np.array_equal(exact.counts, tiled.counts)
    assert np.array_equal(exact.keys, tiled.keys)
    assert util.same_float_bits(exact.distances, tiled.distances)
    assert not np.isin(tiled.keys, removed[:200]).any()


@pytest.mark.parametrize("tile", [64, 256])
@pytest.mark.parametrize("metric,dtype,ndim,n,k", [("cos", "f16", 768, 12001, 10), ("ip", "f16", 100, 9000, 32), ("ip", "f16", 256, 9000, 10),
                                                   ("cos", "bf16", 96, 7000, 10), ("ip", "bf16", 768, 5000, 5),
                                                   ("l2sq", "f16", 96, 6000, 10), ("l2sq", "bf16", 200, 5000, 8)])
def test_tiled_exact_search_of_float_pairs_is_within_tolerance(reference, monkeypatch, metric, dtype, ndim, n, k, tile):
    """f16 / bf16: products are exact, the matrix unit accumulates them in f32 in its own order — every distance within the
    float tolerance of the bit-exact kernel's, the same neighbours wherever distances are separated by more than that."""
    from usearch_amd import Index
    monkeypatch.setenv("USEARCH_AMD_EXACT_TILE", str(tile))
    image, vectors, _ = util.build_image(n, ndim, metric, dtype, seed=95, expansion_add=16, connectivity=4)
    queries = util.make_vectors(70 if tile == 64 else 300, ndim, dtype, seed=96)
    queries[:10] = vectors[:10]
    index = Index.restore(image)
    exact = index.search(queries, k, exact=True, dtype=dtype)
    tiled = index.search(queries, k, exact="tiled", dtype=dtype)
    assert np.array_equal(exact.counts, tiled.counts)
    scale = np.maximum(1.0, np.abs(exact.distances))
    assert np.all(np.abs(exact.distances - tiled.distances) <= util.tolerance(dtype) * scale)
    assert (exact.keys == tiled.keys).mean() > 0.97
    assert np.all(np.diff(tiled.distances, axis=1) >= 0)
    with pytest.raises(RuntimeError):  # no matrix-unit kernel for this pair: said so, not silently rerouted
        image32, _, _ = util.build_image(500, 16, "l2sq", "f32", seed=1)
        Index.restore(image32).search(np.zeros((2, 16), dtype=np.float32), 3, exact="tiled")


@pytest.mark.parametrize("metric", ["cos", "ip"])
def test_wide_tile_thresholds_inside_the_sums_at_their_corners(reference, monkeypatch, metric):
    """The wide tile's fold for f16 cos / ip keeps Σab − threshold·√Σb² in the accumulators (exact_tiled.hip: `fold_tile_fused`):
    thresholds of either sign (queries that point AWAY from every row: the k-th best similarity is negative), rows whose norm f16
    cannot carry (zero rows, rows of 1e-6), a query of zero norm (no finite threshold: the general fold takes its wave's tiles),
    many equal rows (ties at the threshold). Same neighbours and distances as the bit-exact kernel within the float tolerance —
    and, with the general fold forced (knock-out 32), the very same bits from the same kernel."""
    from usearch_amd import Index
    monkeypatch.setenv("USEARCH_AMD_EXACT_TILE", "256")
    rng = np.random.default_rng(123)
    n, ndim, k = 9000, 128, 10
    direction = rng.standard_normal(ndim).astype(np.float32)
    vectors = (direction[None, :] + 0.3 * rng.standard_normal((n, ndim))).astype(np.float16)
    vectors[100:110] = 0                       # zero rows
    vectors[200:220] = (vectors[200:220].astype(np.float32) * 1e-6).astype(np.float16)  # norms below what f16 carries
    vectors[300:340] = vectors[300]            # forty equal rows
    image, _, _ = util.build_image(n, ndim, metric, "f16", vectors=vectors, expansion_add=16, connectivity=4)
    queries = np.concatenate([(-direction[None, :] + 0.3 * rng.standard_normal((150, ndim))),   # every similarity negative
                              (direction[None, :] + 0.3 * rng.standard_normal((149, ndim))),
                              np.zeros((1, ndim))]).astype(np.float16)                          # a query of zero norm
    queries[7] = vectors[300]
    index = Index.restore(image)
    exact = index.search(queries, k, exact=True, dtype="f16")
    tiled = index.search(queries, k, exact="tiled", dtype="f16")
    assert np.array_equal(exact.counts, tiled.counts)
    scale = np.maximum(1.0, np.abs(exact.distances))
    assert np.all(np.abs(exact.distances - tiled.distances) <= util.tolerance("f16") * scale)
    assert (exact.keys == tiled.keys).mean() > 0.9  # ties (forty equal rows, zero rows at distance 1) may be named in another order
    assert np.all(np.diff(tiled.distances, axis=1) >= 0)
    monkeypatch.setenv("USEARCH_AMD_EXACT_KNOCKOUT", "32")
    general = index.search(queries, k, exact="tiled", dtype="f16")
    monkeypatch.delenv("USEARCH_AMD_EXACT_KNOCKOUT")
    assert np.array_equal(general.counts, tiled.counts)
    assert np.all(np.abs(general.distances - tiled.distances) <= 2e-6 * scale), "the two folds close the same sums"
    assert (general.keys == tiled.keys).mean() > 0.97


def test_exact_search_of_a_raw_i8_dataset_takes_the_matrix_units(monkeypatch):
    """`usearch_exact_search` over i8 rows routes to the tiled kernel on its own (bit-identical); the environment switch brings
    the wave-per-query kernel back for the comparison."""
    import usearch_amd
    rng = np.random.default_rng(9)
    dataset = rng.integers(-100, 100, (30000, 64)).astype(np.int8)
    queries = rng.integers(-100, 100, (200, 64)).astype(np.int8)
    keys, distances = usearch_amd.exact_search(dataset, queries, 10, metric="l2sq")
    monkeypatch.setenv("USEARCH_AMD_NO_TILED_EXACT", "1")
    plain_keys, plain_distances = usearch_amd.exact_search(dataset, queries, 10, metric="l2sq")
    assert np.array_equal(keys, plain_keys) and util.same_float_bits(distances, plain_distances)
    brute = ((queries[:, None, :].astype(np.int32) - dataset[None, :6000, :].astype(np.int32)) ** 2).sum(-1)
    assert np.all(distances[:, 0] <= brute.min(axis=1))


def test_tiled_exact_search_over_more_rows_than_one_launch_has_threads(monkeypatch):
    """70M rows: a wave per row would be 4.5·10⁹ threads, past the 2³² a launch may have — the helper kernels walk rows with a
    grid stride instead (a 100M-row index once got its ground truth from norms that were never computed)."""
    import usearch_amd
    rng = np.random.default_rng(11)
    rows = 70_000_000
    dataset = rng.integers(-120, 120, (rows, 16), dtype=np.int8)
    queries = dataset[rng.integers(0, rows, 64)] + rng.integers(-2, 3, (64, 16)).astype(np.int8)
    keys, distances = usearch_amd.exact_search(dataset, queries, 10, metric="l2sq")
    monkeypatch.setenv("USEARCH_AMD_NO_TILED_EXACT", "1")
    plain_keys, plain_distances = usearch_amd.exact_search(dataset, queries, 10, metric="l2sq")
    assert np.array_equal(keys, plain_keys) and util.same_float_bits(distances, plain_distances)
    assert np.all(distances >= 0) and np.all(distances[:, 0] <= 16 * 4)
