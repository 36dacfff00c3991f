"""Filtered search on the device: the caller's predicate as an HBM-resident bitmap, tested where the reference
calls the predicate (index.hpp:4200-4205, 4236-4240; brute force: index.hpp:4260-4263). The HIP path against the oracle (bit for
bit, traversal counters included, in the kernels' summation layout) and against the REAL reference's `usearch_filtered_search`,
for predicates of every selectivity — the sweep tests/test_oracle_vs_reference.py::test_filtered_search_side_by_side holds the
oracle to."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

# (metric, dtype, ndim, n)
SHAPES = [("l2sq", "i8", 96, 3000), ("hamming", "b1", 128, 4000), ("cos", "f16", 768, 1200), ("cos", "f32", 64, 2000),
          ("ip", "bf16", 64, 1000), ("tanimoto", "b1", 256, 1500)]


def predicates(n):
    return {"every third": lambda key: key % 3 == 0, "one in fifty": lambda key: key % 50 == 7,
            "the upper half": lambda key: key >= n // 2, "all": lambda key: True, "none": lambda key: False,
            "one member": lambda key: key == 11}


def make_filters(index, n):
    """The same predicates, each through the entry point that suits it: a range of keys, a list of keys, a deny list, a bitmap."""
    keys = np.arange(n, dtype=np.uint64)
    return {"every third": index.filter_keys(keys[keys % 3 == 0]),
            "one in fifty": index.filter_bits(keys % 50 == 7),
            "the upper half": index.filter_key_range(n // 2, 2**64 - 2),
            "all": index.filter_keys(np.zeros(0, dtype=np.uint64), allow=False),
            "none": index.filter_keys(np.zeros(0, dtype=np.uint64), allow=True),
            "one member": index.filter_key_range(11, 11)}


@pytest.mark.parametrize("metric,dtype,ndim,n", SHAPES)
def test_filtered_search_matches_oracle_and_reference(reference, metric, dtype, ndim, n):
    from oracle import oraclebind
    from usearch_amd import Index
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=41, keys=np.arange(n, dtype=np.uint64))
    queries = util.make_vectors(48, ndim, dtype, seed=42, metric=metric)
    queries[:6] = vectors[:6]
    index = Index.restore(image)
    oracle = oraclebind.OracleIndex(image)
    ref_index.expansion_search = 64
    k = 10
    filters = make_filters(index, n)
    for name, predicate in predicates(n).items():
        expected_allowed = sum(1 for key in range(n) if predicate(key))
        assert filters[name].allowed == expected_allowed, name
        got = index.search(queries, k, expansion=64, dtype=dtype, filter=filters[name])
        assert got.stats.frontier == 1, "a filtered walk keeps the reference's heap (rejected members route without entering `top`)"
        for q, query in enumerate(queries):
            found, keys, dists, visited, computed = oracle.filtered_search(query, k, predicate, dtype=dtype, expansion=64,
                                                                            lanes=index.lanes_per_row, counters=True)
            assert int(got.counts[q]) == found, (name, q)
            assert all(predicate(int(key)) for key in got.keys[q, :found]), name
            # bit for bit against the oracle in the kernels' layout: keys, distance bits, padding, both counters
            assert np.array_equal(got.keys[q], keys), (name, q, got.keys[q], keys)
            assert util.same_float_bits(got.distances[q], dists), (name, q)
            assert int(got.visited_per_query[q]) == visited and int(got.computed_per_query[q]) == computed, (name, q)
            # and against the real reference: exact for the integer-valued pairs, the float tolerance otherwise
            rfound, rkeys, rdists = ref_index.filtered_search(query, k, predicate, dtype=dtype)
            assert rfound == found, (name, q)
            if util.exact_pair(metric, dtype):
                assert np.array_equal(got.keys[q, :found], rkeys[:found]), (name, q)
                assert util.same_float_bits(got.distances[q, :found], rdists[:found]), (name, q)
            else:
                scale = np.maximum(1, np.abs(rdists[:found]))
                assert np.all(np.abs(got.distances[q, :found] - rdists[:found]) <= util.tolerance(dtype) * scale), (name, q)


@pytest.mark.parametrize("metric,dtype,ndim,n", [("l2sq", "i8", 96, 2500), ("cos", "f16", 256, 1500), ("hamming", "b1", 128, 3000)])
def test_filtered_exact_search_matches_oracle(metric, dtype, ndim, n):
    """`filtered_search(…, exact = true)`: the brute-force scan skips what the predicate rejects (index.hpp:4260-4263) — the
    wave-per-query kernel bit for bit, the matrix-unit kernel bit for bit on i8 and within the float tolerance on f16."""
    from oracle import oraclebind
    from usearch_amd import Index
    image, vectors, _ = util.build_image(n, ndim, metric, dtype, seed=51, keys=np.arange(n, dtype=np.uint64))
    queries = util.make_vectors(40, ndim, dtype, seed=52, metric=metric)
    index = Index.restore(image)
    oracle = oraclebind.OracleIndex(image)
    filters = make_filters(index, n)
    k = 7
    for name, predicate in predicates(n).items():
        got = index.search(queries, k, dtype=dtype, exact=True, filter=filters[name])
        tiled = index.search(queries, k, dtype=dtype, exact="tiled", filter=filters[name]) if dtype in ("i8", "f16") else None
        for q, query in enumerate(queries):
            found, keys, dists = oracle.filtered_search(query, k, predicate, dtype=dtype, lanes=index.lanes_per_row, exact=True)
            assert int(got.counts[q]) == found == min(k, filters[name].allowed), (name, q)
            assert np.array_equal(got.keys[q], keys) and util.same_float_bits(got.distances[q], dists), (name, q)
            if tiled is not None:
                assert int(tiled.counts[q]) == found
                if dtype == "i8":
                    assert np.array_equal(tiled.keys[q], keys) and util.same_float_bits(tiled.distances[q], dists), (name, q)
                else:
                    assert np.all(np.abs(tiled.distances[q, :found] - dists[:found]) <= util.tolerance(dtype)), (name, q)
                    assert all(predicate(int(key)) for key in tiled.keys[q, :found])


def test_filter_follows_tombstones_and_checks_its_snapshot():
    """Tombstones never pass a filter; a filter serves the snapshot it was made for and no other."""
    from usearch_amd import Index
    n = 2000
    image, vectors, _ = util.build_image(n, 64, "cos", "f32", seed=61, keys=np.arange(n, dtype=np.uint64), remove=[5, 17, 300])
    index = Index.restore(image)
    everything = index.filter_key_range(0, 2**64 - 1)
    assert everything.allowed == n - 3
    assert index.filter_bits(np.ones(n, dtype=bool)).allowed == n - 3
    other = Index.restore(image)
    with pytest.raises(RuntimeError, match="another index"):
        other.search(vectors[:4], 5, filter=everything)
    with pytest.raises(RuntimeError, match="does not cover"):
        index.filter_bits(np.zeros(3, dtype=np.uint32))
    # a device-resident batch under a filter: the same answers as the host-buffer route
    torch = pytest.importorskip("torch")
    upper = index.filter_key_range(n // 2, n)
    q = torch.from_numpy(vectors[:64].copy()).cuda()
    out = [torch.zeros((64, 5), dtype=torch.int64, device="cuda"), torch.zeros((64, 5), dtype=torch.float32, device="cuda")] + \
          [torch.zeros(64, dtype=torch.int64, device="cuda") for _ in range(3)]
    index.search_device(q.data_ptr(), 64, q.stride(0) * 4, 5, 64, *[t.data_ptr() for t in out], filter=upper)
    host = index.search(vectors[:64], 5, expansion=64, filter=upper)
    assert np.array_equal(out[0].cpu().numpy().astype(np.uint64), host.keys)
    assert np.all(host.keys[host.counts > 0][:, 0] >= n // 2)
