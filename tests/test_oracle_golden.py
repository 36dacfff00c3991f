"""CPU: the oracle (oracle/usearch_oracle.c) pinned against the committed golden fixtures — answers of the REAL reference
(tests/golden/make_golden.py) — and against the literal known-answer vectors of the reference's own tests."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import oraclebind
from tests import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def load(path):
    data = np.load(path)
    return data, json.loads(str(data["meta"]))


def test_fixtures_exist():
    assert len(FIXTURES) >= 8 and os.path.exists(os.path.join(GOLDEN, "kat.json"))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_reference_answers(path):
    data, meta = load(path)
    index = oraclebind.OracleIndex(data["image"])
    assert len(index) == meta["n"] and index.ndim == meta["ndim"] and index.dtype == meta["dtype"]
    keys, distances, counts, visited, computed = index.search(data["queries"], meta["k"], dtype=meta["dtype"],
                                                              expansion=meta["expansion"], lanes=0)
    assert np.array_equal(counts, data["counts"])
    found = np.arange(meta["k"])[None, :] < counts[:, None]
    if util.exact_pair(meta["metric"], meta["dtype"]):
        # integer-valued distances: everything is exact, ties included (container semantics restated literally)
        assert np.array_equal(keys, data["keys"])
        assert util.same_float_bits(distances, data["distances"])
        assert np.array_equal(visited, data["visited"]) and np.array_equal(computed, data["computed"])
    else:
        tolerance = util.tolerance(meta["dtype"])
        reference = np.where(found, data["distances"], 0)
        assert np.all(np.abs(np.where(found, distances, 0) - reference) <= tolerance * np.maximum(1, np.abs(reference)))
        assert ((keys == data["keys"]) | ~found).mean() > 0.99
    assert np.all(keys[~found] == 0) and np.all(np.isnan(distances[~found]))  # padding of index.hpp:2707-2722


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_exact_search(path):
    data, meta = load(path)
    index = oraclebind.OracleIndex(data["image"])
    keys, distances, counts, *_ = index.search(data["queries"], meta["k"], dtype=meta["dtype"], exact=True)
    assert np.array_equal(counts, data["exact_counts"])
    if util.exact_pair(meta["metric"], meta["dtype"]):
        assert np.array_equal(keys, data["exact_keys"]) and util.same_float_bits(distances, data["exact_distances"])
    else:
        assert (keys == data["exact_keys"]).mean() > 0.99


def test_summation_layouts_agree_within_tolerance():
    """`lanes = G` (the kernels' layout) only reorders float additions: same neighbours, tiny distance differences."""
    data, meta = load(os.path.join(GOLDEN, "cos_f16_96.npz"))
    index = oraclebind.OracleIndex(data["image"])
    base = index.search(data["queries"], meta["k"], dtype="f16", expansion=64, lanes=0)
    for lanes in (1, 2, 4, 8):
        other = index.search(data["queries"], meta["k"], dtype="f16", expansion=64, lanes=lanes)
        assert (other[0] == base[0]).mean() > 0.99
        assert np.nanmax(np.abs(other[1] - base[1])) < 1e-3


def test_known_answer_vectors():
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    for case in kat["distance"]:
        dtype = util.NP_DTYPE[case["dtype"]]
        a, b = np.array(case["a"], dtype=dtype), np.array(case["b"], dtype=dtype)
        for lanes in (0, 1):
            d = oraclebind.distance(a, b, case["metric"], case["dtype"], case["ndim"], lanes)
            assert abs(d - case["expected"]) <= case["tolerance"], case["source"]


def test_merge_into_places_later_equals_first():
    """search_result_t::merge_into (index.hpp:2650-2670): lower_bound ⇒ a later-merged equal distance goes BEFORE."""
    keys = np.zeros(4, dtype=np.uint64)
    dists = np.zeros(4, dtype=np.float32)
    count = oraclebind.merge_into(keys, dists, 0, np.array([1, 2, 3]), np.array([1.0, 2.0, 3.0]), 3)
    count = oraclebind.merge_into(keys, dists, count, np.array([7, 8]), np.array([2.0, 2.0]), 2)
    assert count == 4
    assert keys.tolist() == [1, 8, 7, 2] and dists.tolist() == [1.0, 2.0, 2.0, 2.0]


def test_images_with_64_bit_matrix_dimensions():
    """`use_64_bit_dimensions` (index_dense.hpp:1006-1024) changes the first bytes only; sniffed like index_dense.hpp:321-371."""
    data, meta = load(os.path.join(GOLDEN, "l2sq_i8_96.npz"))
    wide = oraclebind.OracleIndex(util.with_64_bit_dimensions(data["image"]))
    assert len(wide) == meta["n"] and wide.ndim == meta["ndim"]
    keys, distances, counts, visited, computed = wide.search(data["queries"], meta["k"], dtype=meta["dtype"],
                                                             expansion=meta["expansion"])
    assert np.array_equal(keys, data["keys"]) and util.same_float_bits(distances, data["distances"])
    assert np.array_equal(computed, data["computed"])


@pytest.mark.parametrize("metric,dtype", [(m, d) for m in ("cos", "l2sq", "divergence", "pearson") for d in ("f32", "f16", "bf16", "i8")
                                          if (m, d) != ("divergence", "i8")] + [(m, "b1") for m in ("hamming", "tanimoto", "sorensen")])
def test_distance_to_itself_is_zero_and_to_another_is_not(metric, dtype):
    """python/scripts/test_distances.py:61-109 (`test_distances_continuous`, `test_distances_sparse`): 1024 dimensions,
    d(x, x) = 0 and d(x, y) != 0 at 1e-2, for every metric x quantization the reference's Python test sweeps."""
    ndim = 1024
    x, y = util.make_vectors(2, ndim, dtype, seed=7, clustered=False, metric=metric)
    for lanes in (0, 8):
        assert abs(oraclebind.distance(x, x, metric, dtype, ndim, lanes)) <= 1e-2
        assert abs(oraclebind.distance(y, y, metric, dtype, ndim, lanes)) <= 1e-2
        assert abs(oraclebind.distance(x, y, metric, dtype, ndim, lanes)) > 1e-2
