"""The heap-less frontier of the device kernels (usearch_amd/csrc/kernels.hpp, frontier_top_k; restated in
oracle/usearch_oracle.c) against the reference-shaped traversal: identical keys, distance bits, counts and BOTH traversal
counters on every float-valued pair whenever the distances meeting in the frontier are distinct — which pins the GPU's
default mode to the reference through the oracle — and a documented, bounded difference when they are not."""
import numpy as np
import pytest

from oracle import oraclebind
from tests import util

CONFIGS = [  # (metric, dtype, ndim, n, connectivity, k, expansion)
    ("cos", "f32", 128, 3000, 16, 10, 64), ("cos", "f16", 768, 1200, 16, 10, 64), ("l2sq", "f32", 24, 2000, 16, 10, 256),
    ("ip", "bf16", 64, 1500, 16, 5, 32), ("l2sq", "f16", 100, 1200, 13, 19, 19), ("cos", "f32", 32, 2000, 50, 10, 64),
    ("pearson", "f32", 64, 1500, 16, 10, 64), ("cos", "f64", 96, 1000, 16, 10, 700), ("l2sq", "f32", 3, 500, 3, 3, 8),
    ("divergence", "f32", 32, 800, 16, 10, 64), ("haversine", "f32", 2, 2000, 16, 10, 64),
]


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity,k,expansion", CONFIGS)
def test_frontier_in_top_equals_the_reference_heap(reference, metric, dtype, ndim, n, connectivity, k, expansion):
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=17, connectivity=connectivity)
    queries = util.make_vectors(150, ndim, dtype, seed=18, metric=metric)
    queries[:30] = vectors[:30]
    index = oraclebind.OracleIndex(image)
    for lanes in (0, 8):
        heap = index.search(queries, k, dtype=dtype, expansion=expansion, lanes=lanes)
        in_top = index.search(queries, k, dtype=dtype, expansion=expansion, lanes=lanes, frontier_in_top=True)
        assert np.array_equal(heap[0], in_top[0]), "keys"
        assert util.same_float_bits(heap[1], in_top[1]), "distances"
        for a, b in zip(heap[2:], in_top[2:]):
            assert np.array_equal(a, b), "counts / visited_members / computed_distances"
    # and the reference-shaped oracle is the compiled reference itself (lanes = 0 is its loop order)
    ref_index.expansion_search = expansion
    rkeys, rdists, rcounts, rvisited, rcomputed = ref_index.search(queries, k, dtype=dtype, threads=1)
    heap = index.search(queries, k, dtype=dtype, expansion=expansion, lanes=0, frontier_in_top=True)
    assert np.array_equal(heap[2], rcounts)
    if util.layout_exact(metric):
        assert (heap[0] == rkeys).mean() > 0.98


def test_integer_valued_pairs_and_filters_keep_the_heap(reference):
    """The mode is only ever applied where the engine applies it: b1 / i8 (ties are the norm), predicates, tombstones and
    expansions beyond the register `top` fall back to the reference's heap, so asking for it changes nothing there."""
    for metric, dtype, ndim in (("hamming", "b1", 64), ("l2sq", "i8", 32)):
        image, _, _ = util.build_image(2000, ndim, metric, dtype, seed=19)
        queries = util.make_vectors(60, ndim, dtype, seed=20)
        index = oraclebind.OracleIndex(image)
        heap = index.search(queries, 10, dtype=dtype, expansion=64)
        asked = index.search(queries, 10, dtype=dtype, expansion=64, frontier_in_top=True)
        for a, b in zip(heap, asked):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    removed = np.arange(0, 1500, 3) + 1000
    image, _, _ = util.build_image(1500, 32, "cos", "f32", seed=31, remove=removed)
    index = oraclebind.OracleIndex(image)
    queries = util.make_vectors(60, 32, "f32", seed=32)
    heap = index.search(queries, 10, expansion=64, lanes=8)
    asked = index.search(queries, 10, expansion=64, lanes=8, frontier_in_top=True)
    assert np.array_equal(heap[0], asked[0]) and np.array_equal(heap[3], asked[3]) and np.array_equal(heap[4], asked[4])
    image, _, _ = util.build_image(3000, 24, "l2sq", "f32", seed=33)
    index = oraclebind.OracleIndex(image)
    queries = util.make_vectors(20, 24, "f32", seed=34)
    heap = index.search(queries, 10, expansion=1500)
    asked = index.search(queries, 10, expansion=1500, frontier_in_top=True)
    assert np.array_equal(heap[0], asked[0]) and np.array_equal(heap[4], asked[4])


def test_exact_ties_are_where_the_two_may_differ(reference):
    """Every vector stored three times: frontier candidates at exactly equal distances. The reference's heap pops equal keys
    in the order its sift rules produce, the open cells of `top` in `top`'s own order (newest equal first) — so which twin is
    expanded first, and hence the hop / distance counters, may differ; a member evicted from `top` at exactly the radius is
    still expanded by the reference (strict `>` of index.hpp:4210) and not by the heap-less frontier. The result DISTANCES
    are the same."""
    from oracle import refbind
    base = util.make_vectors(700, 48, "f32", seed=91)
    vectors = np.concatenate([base, base, base])
    reference_index = refbind.RefIndex(48, "l2sq", "f32", connectivity=16, expansion_add=128)
    reference_index.add(np.arange(len(vectors), dtype=np.uint64) + 1000, vectors, threads=1)
    index = oraclebind.OracleIndex(reference_index.save_buffer())
    queries = util.make_vectors(120, 48, "f32", seed=92)
    heap = index.search(queries, 10, expansion=64, lanes=2)
    in_top = index.search(queries, 10, expansion=64, lanes=2, frontier_in_top=True)
    assert util.same_float_bits(heap[1], in_top[1])
    assert np.array_equal(heap[2], in_top[2])
    differing = int((heap[3] != in_top[3]).sum())
    print(f"queries whose hop counter differs under exact ties: {differing} of {len(queries)}")
    assert abs(in_top[4].astype(float).mean() / heap[4].astype(float).mean() - 1) < 0.05
