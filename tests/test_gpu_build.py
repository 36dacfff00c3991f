"""Index construction on the MI355X (`usearch_amd_build`): the graph it leaves must be (1) a well-formed HNSW index the
REAL reference loads and searches, (2) of the same quality as a reference-built one, and (3) searched by the GPU path with
the usual parity against the oracle / the reference — i.e. a GPU-built image is just another `.usearch` image.

Reference behaviour being reproduced: `index_gt::add` (index.hpp:2759-2879) = insertion search (4011-4079) +
`form_links_to_closest_` (3825-3845) + `form_reverse_links_` (3848-3893), both pruning with `refine_` (4276-4318)."""
import numpy as np
import pytest

import usearch_amd
from oracle import oraclebind, refbind
from tests import util

pytestmark = pytest.mark.gpu


def check_structure(image: np.ndarray, n: int, connectivity: int):
    """Every list: within capacity, no self loop, no duplicate, every neighbour exists on that level."""
    oracle = oraclebind.OracleIndex(image)
    assert len(oracle) == n
    levels = np.array([oracle.level(s) for s in range(n)])
    linked = 0
    for slot in range(n):
        for level in range(levels[slot] + 1):
            neighbours = oracle.neighbors(slot, level)
            assert len(neighbours) <= (2 * connectivity if level == 0 else connectivity)
            assert slot not in neighbours
            assert len(set(neighbours.tolist())) == len(neighbours)
            assert np.all(neighbours < n)
            assert np.all(levels[neighbours] >= level)
            linked += len(neighbours)
    return levels, linked


def recall_at(found: np.ndarray, truth: np.ndarray) -> float:
    k = truth.shape[1]
    return float(np.mean([len(np.intersect1d(found[i], truth[i])) / k for i in range(len(truth))]))


CASES = [
    # metric, dtype, ndim, n, connectivity, expansion_add
    ("cos", "f32", 64, 6000, 16, 128),
    ("cos", "f16", 768, 3000, 16, 128),
    ("l2sq", "f16", 96, 5000, 16, 128),
    ("l2sq", "i8", 96, 6000, 16, 128),
    ("hamming", "b1", 128, 6000, 16, 128),
    ("ip", "f32", 40, 4000, 8, 64),
    ("cos", "f32", 24, 3000, 3, 40),
]


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity,expansion_add", CASES)
def test_built_graph_is_valid_and_as_good_as_the_references(reference, metric, dtype, ndim, n, connectivity,
                                                            expansion_add):
    vectors = util.make_vectors(n, ndim, dtype, seed=11)
    if metric == "ip":
        vectors = (vectors / np.linalg.norm(vectors, axis=1, keepdims=True)).astype(vectors.dtype)
    queries = util.make_vectors(300, ndim, dtype, seed=12)
    keys = np.arange(n, dtype=np.uint64) * 3 + 7
    built = usearch_amd.build(vectors, metric, dtype, keys=keys, connectivity=connectivity,
                              expansion_add=expansion_add, max_batch=512)
    stats = built.stats
    assert stats.dropped_requests == 0
    image = built.save_buffer()
    levels, linked = check_structure(image, n, connectivity)
    assert linked > n  # a connected graph at the very least has more links than nodes

    # the reference loads the image and finds every vector by itself (cpp/test.cpp:232-236 style)
    theirs = refbind.RefIndex.from_buffer(image, view=False, dtype=dtype)
    assert len(theirs) == n
    self_keys, self_distances, *_ = theirs.search(vectors[:500], 1, dtype=dtype)
    if metric != "hamming" and dtype != "i8":
        assert np.mean(self_keys[:, 0] == keys[:500]) >= 0.99

    # quality: recall@10 of the REFERENCE searching our graph vs searching its own, same data, same parameters
    truth, *_ = theirs.search(queries, 10, dtype=dtype, exact=True)
    ours_found, *_ = theirs.search(queries, 10, dtype=dtype)
    own = refbind.RefIndex(ndim, metric, dtype, connectivity=connectivity, expansion_add=expansion_add)
    own.add(keys, vectors, threads=1)
    own_found, *_ = own.search(queries, 10, dtype=dtype)
    if metric in ("hamming",) or dtype == "i8":  # ties between equal integer distances make key-recall noisy: compare distances
        ours_d = theirs.search(queries, 10, dtype=dtype)[1]
        own_d = own.search(queries, 10, dtype=dtype)[1]
        truth_d = theirs.search(queries, 10, dtype=dtype, exact=True)[1]
        ours_recall = float(np.mean(ours_d[:, -1] <= truth_d[:, -1]))
        own_recall = float(np.mean(own_d[:, -1] <= truth_d[:, -1]))
    else:
        ours_recall, own_recall = recall_at(ours_found, truth), recall_at(own_found, truth)
    assert ours_recall >= own_recall - (0.03 if connectivity >= 8 else 0.08), (ours_recall, own_recall)

    # the GPU searches the graph it built exactly as it searches any other image: bit parity with the oracle
    got = built.index.search(queries, 10, dtype=dtype)
    okeys, odistances, ocounts, ovisited, ocomputed = util.oracle_search(image, queries, 10, dtype, expansion=64,
                                                                         lanes=built.index.lanes_per_row)
    assert np.array_equal(got.keys, okeys)
    assert util.same_float_bits(got.distances, odistances)
    assert np.array_equal(got.counts, ocounts)
    assert np.array_equal(got.visited_per_query, ovisited) and np.array_equal(got.computed_per_query, ocomputed)

    # and an `Index.restore` of the saved image behaves like the snapshot the builder kept
    again = usearch_amd.Index.restore(image)
    redo = again.search(queries, 10, dtype=dtype)
    assert np.array_equal(redo.keys, got.keys) and util.same_float_bits(redo.distances, got.distances)


OTHER_PAIRS = [
    # the rest of the reference's metric x scalar dispatch table: metric, dtype, ndim, n
    ("cos", "bf16", 64, 3000), ("l2sq", "f64", 24, 3000), ("pearson", "f32", 48, 3000), ("pearson", "i8", 64, 3000),
    ("divergence", "f32", 32, 2500), ("haversine", "f32", 2, 3000), ("tanimoto", "b1", 128, 4000),
    ("sorensen", "b1", 256, 3000),
]


@pytest.mark.parametrize("metric,dtype,ndim,n", OTHER_PAIRS)
def test_build_with_the_other_metric_scalar_pairs(reference, metric, dtype, ndim, n):
    """Construction runs the same distance loop as the search, so every pair with a search kernel builds too."""
    vectors = util.make_vectors(n, ndim, dtype, seed=21, metric=metric)
    queries = util.make_vectors(200, ndim, dtype, seed=22, metric=metric)
    built = usearch_amd.build(vectors, metric, dtype, connectivity=16, expansion_add=128, max_batch=512)
    assert built.stats.dropped_requests == 0
    image = built.save_buffer()
    check_structure(image, n, 16)
    theirs = refbind.RefIndex.from_buffer(image, view=False, dtype=dtype)
    assert len(theirs) == n
    own = refbind.RefIndex(ndim, metric, dtype, connectivity=16, expansion_add=128)
    own.add(np.arange(n, dtype=np.uint64), vectors, threads=1)
    truth_d = theirs.search(queries, 10, dtype=dtype, exact=True)[1]
    ours_d, own_d = theirs.search(queries, 10, dtype=dtype)[1], own.search(queries, 10, dtype=dtype)[1]
    slack = 1e-6 * np.maximum(1.0, np.abs(truth_d[:, -1]))  # recall by distance: robust to ties between equal distances
    ours_recall = float(np.mean(ours_d[:, -1] <= truth_d[:, -1] + slack))
    own_recall = float(np.mean(own_d[:, -1] <= truth_d[:, -1] + slack))
    assert ours_recall >= own_recall - 0.05, (ours_recall, own_recall)
    got = built.index.search(queries, 10, dtype=dtype)
    okeys, odistances, ocounts, ovisited, ocomputed = util.oracle_search(image, queries, 10, dtype, expansion=64,
                                                                         lanes=built.index.lanes_per_row)
    assert np.array_equal(got.counts, ocounts)
    if util.layout_exact(metric):
        assert np.array_equal(got.keys, okeys) and util.same_float_bits(got.distances, odistances)
        assert np.array_equal(got.visited_per_query, ovisited) and np.array_equal(got.computed_per_query, ocomputed)
    else:
        assert (got.keys == okeys).mean() > 0.98


def test_build_is_reproducible_and_takes_device_vectors():
    import ctypes
    vectors = util.make_vectors(4000, 96, "f16", seed=5)
    first = usearch_amd.build(vectors, "cos", "f16", max_batch=256, seed=99).save_buffer()
    second = usearch_amd.build(vectors, "cos", "f16", max_batch=256, seed=99).save_buffer()
    assert np.array_equal(first, second)
    # rows already in HBM, with a pitch: plain HIP runtime calls (the runtime the engine itself has loaded)
    hip = ctypes.CDLL(util.mapped_hip_runtime())
    pitch = 208
    padded = np.zeros((len(vectors), pitch), dtype=np.uint8)
    padded[:, :192] = vectors.view(np.uint8).reshape(len(vectors), -1)
    pointer = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(pointer), ctypes.c_size_t(padded.nbytes)) == 0
    assert hip.hipMemcpy(pointer, ctypes.c_void_p(padded.ctypes.data), ctypes.c_size_t(padded.nbytes), 1) == 0
    third = usearch_amd.build(None, "cos", "f16", device_pointer=pointer.value, count=len(vectors), stride=pitch,
                              ndim=96, max_batch=256, seed=99).save_buffer()
    assert hip.hipFree(pointer) == 0
    assert np.array_equal(first, third)


def test_hubs_lose_no_reverse_link(reference):
    """A few rows that every other row picks as a neighbour (long vectors under the inner product): far more reverse-link
    requests per pass than a hub's inbox holds. The reference takes any number (index.hpp:3848-3893, one at a time under the
    node's lock); here the surplus waits for the next re-filing round — counted, and never dropped."""
    n, ndim = 6000, 32
    vectors = np.abs(util.make_vectors(n, ndim, "f32", seed=71))
    vectors[:4] *= 25.0
    built = usearch_amd.build(vectors, "ip", "f32", connectivity=16, expansion_add=128, max_batch=4096)
    assert built.stats.dropped_requests == 0
    assert built.stats.refiled_requests > 0
    image = built.save_buffer()
    check_structure(image, n, 16)
    theirs = refbind.RefIndex.from_buffer(image, view=False, dtype="f32")
    queries = np.abs(util.make_vectors(200, ndim, "f32", seed=72))
    found = theirs.search(queries, 4, dtype="f32")[0]
    assert np.mean(np.sort(found, axis=1) == np.arange(4)[None, :]) > 0.99  # the hubs are everybody's nearest


@pytest.mark.parametrize("connectivity,connectivity_base", [(30, 0), (16, 63), (31, 62), (32, 0), (48, 0), (64, 0), (20, 100)])
def test_wide_base_lists_build(reference, connectivity, connectivity_base):
    """Base lists of 57 ... 63 neighbours leave 7 ... 1 places in the wave that re-prunes a list: more rounds, same graph rules.
    Lists of 64 … 128 cells (connectivity 32 — HNSW's other popular choice — 48, 64) go through LDS, three candidates per lane."""
    n, ndim = 4000, 48
    vectors = util.make_vectors(n, ndim, "f32", seed=73)
    queries = util.make_vectors(200, ndim, "f32", seed=74)
    built = usearch_amd.build(vectors, "cos", "f32", connectivity=connectivity, connectivity_base=connectivity_base,
                              expansion_add=128, max_batch=512)
    assert built.stats.dropped_requests == 0
    image = built.save_buffer()
    theirs = refbind.RefIndex.from_buffer(image, view=False, dtype="f32")
    assert len(theirs) == n and theirs.connectivity == connectivity
    own = refbind.RefIndex(ndim, "cos", "f32", connectivity=connectivity, expansion_add=128)
    own.add(np.arange(n, dtype=np.uint64), vectors, threads=1)
    truth = theirs.search(queries, 10, dtype="f32", exact=True)[0]
    ours_recall = recall_at(theirs.search(queries, 10, dtype="f32")[0], truth)
    own_recall = recall_at(own.search(queries, 10, dtype="f32")[0], truth)
    assert ours_recall >= own_recall - 0.03, (ours_recall, own_recall)
    got = built.index.search(queries, 10, dtype="f32")
    okeys, odistances, *_ = util.oracle_search(image, queries, 10, "f32", expansion=64, lanes=built.index.lanes_per_row)
    assert np.array_equal(got.keys, okeys) and util.same_float_bits(got.distances, odistances)


@pytest.mark.parametrize("expansion_add", [300, 1000])
def test_wide_insertion_beams_build(reference, expansion_add):
    """expansion_add above 256: the candidates' bitmap of `refine_` lives in LDS, any beam up to 1 024 links (the reference has
    no bound, index.hpp:1359-1394); the graph is at least as good as the reference's at the same setting."""
    n, ndim = 3000, 32
    vectors = util.make_vectors(n, ndim, "f32", seed=75)
    queries = util.make_vectors(200, ndim, "f32", seed=76)
    built = usearch_amd.build(vectors, "l2sq", "f32", connectivity=8, expansion_add=expansion_add, max_batch=256)
    assert built.stats.dropped_requests == 0
    image = built.save_buffer()
    check_structure(image, n, 8)
    theirs = refbind.RefIndex.from_buffer(image, view=False, dtype="f32")
    own = refbind.RefIndex(ndim, "l2sq", "f32", connectivity=8, expansion_add=expansion_add)
    own.add(np.arange(n, dtype=np.uint64), vectors, threads=1)
    truth = theirs.search(queries, 10, dtype="f32", exact=True)[0]
    ours_recall = recall_at(theirs.search(queries, 10, dtype="f32")[0], truth)
    own_recall = recall_at(own.search(queries, 10, dtype="f32")[0], truth)
    assert ours_recall >= own_recall - 0.03, (ours_recall, own_recall)
    with pytest.raises(RuntimeError):
        usearch_amd.build(vectors[:100], "l2sq", "f32", connectivity=8, expansion_add=1025)


def test_build_edge_cases():
    one = usearch_amd.build(util.make_vectors(1, 16, "f32", seed=1), "cos", "f32")
    assert len(one.index) == 1
    got = one.index.search(util.make_vectors(3, 16, "f32", seed=2), 5)
    assert np.all(got.counts == 1)
    few = usearch_amd.build(util.make_vectors(7, 16, "f32", seed=1), "l2sq", "f32", connectivity=4)
    got = few.index.search(util.make_vectors(3, 16, "f32", seed=2), 10)
    assert np.all(got.counts == 7)
    with pytest.raises(RuntimeError):
        usearch_amd.build(util.make_vectors(10, 16, "f32", seed=1), "cos", "f32", connectivity=65)
    with pytest.raises(RuntimeError):
        usearch_amd.build(util.make_vectors(10, 16, "f32", seed=1), "cos", "f32", connectivity=16, connectivity_base=129)
