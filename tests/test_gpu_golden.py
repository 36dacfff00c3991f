"""GPU: the committed golden fixtures (answers of the REAL reference) through the C ABI — what travels to a box that has
no /root/reference."""
import glob
import json
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_gpu_reproduces_reference_answers(path):
    from usearch_amd import Index
    data = np.load(path)
    meta = json.loads(str(data["meta"]))
    index = Index.restore(data["image"])
    got = index.search(data["queries"], meta["k"], expansion=meta["expansion"], dtype=meta["dtype"])
    assert np.array_equal(got.counts, data["counts"])
    found = np.arange(meta["k"])[None, :] < data["counts"][:, None]
    if util.exact_pair(meta["metric"], meta["dtype"]):
        assert np.array_equal(got.keys, data["keys"])
        assert util.same_float_bits(got.distances, data["distances"])
        assert np.array_equal(got.visited_per_query, data["visited"])
        assert np.array_equal(got.computed_per_query, data["computed"])
    else:
        tolerance = util.tolerance(meta["dtype"])
        reference = np.where(found, data["distances"], 0)
        assert np.all(np.abs(np.where(found, got.distances, 0) - reference) <= tolerance * np.maximum(1, np.abs(reference)))
        assert ((got.keys == data["keys"]) | ~found).mean() > 0.99


def test_known_answer_searches():
    """rust/lib.rs:1897-1924 (Hamming order [43, 42], distances 2 and 6) and cpp/test.cpp:1045-1100 (42, 43, 44)."""
    from oracle import refbind
    from usearch_amd import Index
    if not refbind.available():
        pytest.skip("needs oracle/_ref to build the tiny index")
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    for case in kat["search"]:
        dtype = util.NP_DTYPE[case["dtype"]]
        ref = refbind.RefIndex(case["ndim"], case["metric"], case["dtype"])
        ref.add(np.array(case["keys"], dtype=np.uint64), np.array(case["vectors"], dtype=dtype), threads=1)
        index = Index.restore(ref.save_buffer())
        got = index.search(np.array(case["query"], dtype=dtype), case["k"], dtype=case["dtype"])
        assert got.keys.tolist() == case["expected_keys"], case["source"]
        if case["expected_distances"]:
            assert got.distances.tolist() == case["expected_distances"]


def test_images_with_64_bit_matrix_dimensions():
    """`serialization_config_t::use_64_bit_dimensions` (index_dense.hpp:1006-1024): same index, two u64 in front."""
    from usearch_amd import Index
    data = np.load(os.path.join(GOLDEN, "l2sq_i8_96.npz"))
    meta = json.loads(str(data["meta"]))
    index = Index.restore(util.with_64_bit_dimensions(data["image"]))
    assert len(index) == meta["n"]
    got = index.search(data["queries"], meta["k"], expansion=meta["expansion"], dtype=meta["dtype"])
    assert np.array_equal(got.keys, data["keys"]) and util.same_float_bits(got.distances, data["distances"])
    assert np.array_equal(got.computed_per_query, data["computed"])
