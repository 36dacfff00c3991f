"""GPU parity proper: batched HNSW search through the C ABI against the oracle (bit-exact, the oracle run in the
kernels' summation layout) and against the real reference compiled from /root/reference (exact for integer-valued
metrics, tolerance for float ones), on seeded inputs small enough for the CPU to finish in seconds."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

# (metric, dtype, ndim, n, connectivity, k, expansion, queries)
CONFIGS = [
    ("cos", "f32", 128, 3000, 16, 10, 64, 200),       # BASELINE config 1 shape, scaled
    ("cos", "f16", 768, 1500, 16, 10, 64, 100),       # headline shape (config 3), scaled
    ("cos", "f32", 768, 800, 16, 10, 64, 50),         # config 2 shape, scaled
    ("l2sq", "i8", 96, 4000, 16, 10, 64, 200),        # config 4 shape, scaled
    ("hamming", "b1", 128, 5000, 16, 10, 64, 300),    # config 5 shape, scaled: ties everywhere
    ("ip", "f32", 64, 1000, 16, 5, 32, 100),
    ("l2sq", "f32", 3, 500, 3, 3, 8, 100),            # cpp/test.cpp:820-865 "absurd" corners: tiny dims/connectivity
    ("l2sq", "f16", 100, 1200, 13, 19, 19, 80),       # cpp/test.cpp:652-766 connectivity 13, ragged dims
    ("cos", "i8", 96, 1500, 16, 10, 128, 80),
    ("ip", "i8", 40, 800, 8, 7, 64, 60),
    ("cos", "f32", 32, 2000, 50, 10, 64, 80),         # connectivity 50 ⇒ M0 = 100 ⇒ two neighbour tiles per hop
    ("hamming", "b1", 1024, 1500, 16, 10, 256, 60),
    ("hamming", "b1", 8, 300, 2, 1, 1, 60),
    ("l2sq", "f16", 16, 64, 16, 100, 64, 20),         # k > n
    # the rest of the reference's metric x scalar dispatch table (index_plugins.hpp:1930-2008)
    ("cos", "bf16", 768, 900, 16, 10, 64, 60),
    ("l2sq", "bf16", 100, 1200, 13, 10, 64, 60),
    ("ip", "bf16", 64, 800, 16, 5, 32, 60),
    ("cos", "f64", 96, 1200, 16, 10, 64, 60),
    ("l2sq", "f64", 7, 600, 5, 4, 16, 60),
    ("ip", "f64", 200, 800, 16, 10, 64, 40),         # 1600-byte rows: 8 lanes per row, 12.5 chunks per lane
    ("pearson", "f32", 128, 1500, 16, 10, 64, 80),
    ("pearson", "f16", 96, 1000, 16, 10, 64, 60),
    ("pearson", "bf16", 40, 800, 8, 5, 32, 60),
    ("pearson", "f64", 32, 800, 16, 10, 64, 60),
    ("pearson", "i8", 96, 1500, 16, 10, 64, 80),
    ("divergence", "f32", 64, 1200, 16, 10, 64, 60),
    ("divergence", "f16", 48, 800, 16, 10, 64, 60),
    ("divergence", "bf16", 32, 800, 16, 10, 64, 40),
    ("divergence", "f64", 24, 600, 8, 5, 32, 40),
    ("haversine", "f32", 2, 3000, 16, 10, 64, 100),
    ("haversine", "f64", 2, 1000, 8, 5, 32, 60),
    ("tanimoto", "b1", 128, 3000, 16, 10, 64, 200),
    ("jaccard", "b1", 256, 1500, 16, 10, 64, 80),    # served by the tanimoto kernel (index_plugins.hpp:2003-2004)
    ("sorensen", "b1", 1024, 1200, 16, 10, 128, 60),
    ("sorensen", "b1", 72, 800, 4, 3, 8, 60),
]


def check_against_oracle(index, image, queries, k, dtype, expansion, **search_kwargs):
    got = index.search(queries, k, expansion=expansion, dtype=dtype, **search_kwargs)
    # the oracle restates whichever frontier the engine ran: the reference's heap, or the open cells of `top`
    # (tests/test_oracle_frontier.py pins the latter to the former wherever no two frontier distances coincide)
    keys, dists, counts, visited, computed = util.oracle_search(image, queries, k, dtype, expansion,
                                                                lanes=index.lanes_per_row,
                                                                frontier_in_top=got.stats.frontier == 2)
    if not util.layout_exact(index.metric_kind):
        # log / sin / cos / asin come from two different math libraries (ocml on the device, libm in the oracle): the
        # float tolerance of north_star instead of bit equality, labels wherever neighbouring distances are separated
        assert np.array_equal(got.counts, counts)
        found = np.arange(k)[None, :] < counts[:, None]
        scale = np.maximum(1.0, np.abs(np.where(found, dists, 0)))
        assert np.all(np.abs(np.where(found, got.distances - dists, 0)) <= util.tolerance(dtype) * scale)
        assert ((got.keys == keys) | ~found).mean() > 0.98
        assert abs(got.computed_per_query.astype(float).mean() / computed.astype(float).mean() - 1) < 0.02
        return got
    bad = np.nonzero((got.keys != keys).any(axis=1))[0]
    assert len(bad) == 0, (f"{len(bad)}/{len(queries)} queries differ in keys; first {bad[0]}: "
                           f"gpu {got.keys[bad[0]]} {got.distances[bad[0]]} oracle {keys[bad[0]]} {dists[bad[0]]}")
    assert np.array_equal(got.counts, counts)
    assert util.same_float_bits(got.distances, dists)
    assert np.array_equal(got.visited_per_query, visited), "visited_members differ"
    assert np.array_equal(got.computed_per_query, computed), "computed_distances differ"
    return got


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity,k,expansion,nq", CONFIGS)
def test_search_matches_oracle_and_reference(reference, metric, dtype, ndim, n, connectivity, k, expansion, nq):
    """Small batches over long rows run four waves per query (the team build) on their own; `test_team_and_one_wave_agree`
    holds that against the one-wave kernel, which every batch that fills the chip uses."""
    from usearch_amd import Index
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=11, connectivity=connectivity)
    queries = util.make_vectors(nq, ndim, dtype, seed=12, metric=metric)
    queries[: nq // 4] = vectors[: nq // 4]  # some in-sample queries: self must come back first (cpp/test.cpp:232-236)
    index = Index.restore(image)
    assert len(index) == n and index.ndim == ndim and index.connectivity == connectivity
    got = check_against_oracle(index, image, queries, k, dtype, expansion)
    assert got.stats.passes == 1, "default scratch sizing should not overflow on these shapes"
    # float-valued pairs run without a frontier heap by default; the reference-shaped heap stays one switch away and must
    # give the same answer on these (tie-free) inputs — keys, distance bits, counters
    float_valued = dtype not in ("b1", "i8")
    assert got.stats.frontier == (2 if float_valued and max(expansion, k) <= 1024 else 1)
    if float_valued:
        from usearch_amd import Tuning
        heap = check_against_oracle(index, image, queries, k, dtype, expansion, tuning=Tuning(frontier=1))
        assert heap.stats.frontier == 1
        if util.layout_exact(metric):
            assert np.array_equal(heap.keys, got.keys) and util.same_float_bits(heap.distances, got.distances)
            assert np.array_equal(heap.visited_per_query, got.visited_per_query)
            assert np.array_equal(heap.computed_per_query, got.computed_per_query)

    # the real reference, same image, same queries
    ref_index.expansion_search = expansion
    rkeys, rdists, rcounts, rvisited, rcomputed = ref_index.search(queries, k, dtype=dtype, threads=1)
    assert np.array_equal(got.counts, rcounts)
    if util.exact_pair(metric, dtype):
        assert np.array_equal(got.keys, rkeys)
        assert util.same_float_bits(got.distances, rdists)
        assert np.array_equal(got.visited_per_query, rvisited)
        assert np.array_equal(got.computed_per_query, rcomputed)
    else:
        # every distance within the stated tolerance, labels IDENTICAL wherever the reference's neighbouring
        # distances (among its k + 1 nearest) are farther apart than both sides' rounding can bridge
        separated, agreement = util.assert_float_parity(
            got.keys, got.distances, got.counts, lambda batch, wanted: ref_index.search(batch, wanted, dtype=dtype, threads=1),
            queries, k, dtype, what=f"{metric}/{dtype}")
        # (the 16-bit kinds' tolerance of 2e-3 leaves a fifth to a half of these clustered rows separated; f32 / f64 nearly all)
        assert separated > (0.1 if dtype in ("f16", "bf16") else 0.5), f"only {separated:.2f} of the positions are separated"
        assert agreement > 0.98, f"label agreement with the reference over ALL positions {agreement:.4f}"
    # monotone distances (cpp/test.cpp:499-503) and self-hit for in-sample queries
    for qi in range(nq):
        c = int(got.counts[qi])
        assert np.all(np.diff(got.distances[qi, :c]) >= 0)
        assert np.all(got.keys[qi, c:] == 0) and np.all(np.isnan(got.distances[qi, c:]))


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity", [("cos", "f16", 96, 3000, 16), ("hamming", "b1", 128, 2500, 7)])
def test_images_with_40_bit_slots_load_and_search_the_same(reference, metric, dtype, ndim, n, connectivity):
    """`uint40_t` compressed slots on the node tapes (index.hpp:969-1031; connectivity 7 makes every list and most nodes an odd
    number of bytes): the device flattener reads them byte by byte and the index it builds is the one the 32-bit image gives."""
    from usearch_amd import Index
    image, vectors, _ = util.build_image(n, ndim, metric, dtype, seed=21, connectivity=connectivity)
    queries = util.make_vectors(120, ndim, dtype, seed=22, metric=metric)
    narrow, wide = Index.restore(image), Index.restore(util.with_40_bit_slots(image))
    assert len(wide) == n and wide.connectivity == connectivity
    a, b = narrow.search(queries, 10, dtype=dtype), wide.search(queries, 10, dtype=dtype)
    assert np.array_equal(a.keys, b.keys) and util.same_float_bits(a.distances, b.distances)
    assert np.array_equal(a.computed_per_query, b.computed_per_query) and np.array_equal(a.visited_per_query, b.visited_per_query)


def test_empty_index_and_zero_wanted(reference):
    from usearch_amd import Index
    image, _, _ = util.build_image(0, 16, "cos", "f32")
    index = Index.restore(image)
    assert len(index) == 0
    got = index.search(np.ones((3, 16), dtype=np.float32), 5)
    assert np.all(got.counts == 0) and np.all(got.keys == 0) and np.all(np.isnan(got.distances))
    image, _, _ = util.build_image(10, 16, "cos", "f32")
    index = Index.restore(image)
    got = index.search(np.ones((3, 16), dtype=np.float32), 0)
    assert got.keys.shape == (3, 0)


def test_single_vector_and_single_query(reference):
    from usearch_amd import Index
    image, vectors, _ = util.build_image(1, 24, "l2sq", "f32", seed=5)
    index = Index.restore(image)
    one = index.search(vectors[0], 3)
    assert len(one) == 1 and one.keys[0] == 1000 and one.distances[0] == 0.0


def test_query_casts_match_reference(reference):
    """Queries handed over in a kind other than the storage kind are cast first (index_dense.hpp:2058-2064)."""
    from usearch_amd import Index
    n, ndim = 1200, 64
    for dtype, metric in (("f16", "cos"), ("i8", "cos"), ("b1", "hamming")):
        image, _, ref_index = util.build_image(n, ndim, metric, dtype, seed=21)
        index = Index.restore(image)
        queries = util.make_vectors(40, ndim, "f32", seed=22)
        got = index.search(queries, 10, dtype="f32")
        keys, dists, counts, visited, computed = util.oracle_search(image, queries, 10, "f32", 64,
                                                                    lanes=index.lanes_per_row,
                                                                    frontier_in_top=got.stats.frontier == 2)
        assert np.array_equal(got.keys, keys) and util.same_float_bits(got.distances, dists)
        assert np.array_equal(got.computed_per_query, computed)
        rkeys, rdists, *_ = ref_index.search(queries, 10, dtype="f32", threads=1)
        if dtype != "f16":
            assert np.array_equal(got.keys, rkeys)


def test_tombstones_are_skipped(reference):
    """Removed entries keep routing the search but never reach the results (index_dense.hpp:2071-2081)."""
    from usearch_amd import Index
    n, ndim = 1500, 32
    removed = np.arange(0, n, 3) + 1000
    image, vectors, ref_index = util.build_image(n, ndim, "cos", "f32", seed=31, remove=removed)
    index = Index.restore(image)
    queries = util.make_vectors(100, ndim, "f32", seed=32)
    got = check_against_oracle(index, image, queries, 10, "f32", 64)
    assert not np.isin(got.keys, removed).any()
    rkeys, *_ = ref_index.search(queries, 10, threads=1)
    assert (got.keys == rkeys).mean() > 0.98


def test_scratch_overflow_retries_give_identical_results(reference):
    """Tiny scratch forces the retry ladder (bigger scratch with a global visited hash, then all-global memory); every
    scratch placement must give the very same results."""
    from usearch_amd import Index, Tuning
    for metric, dtype, ndim in (("cos", "f32", 64), ("hamming", "b1", 64)):
        image, _, _ = util.build_image(3000, ndim, metric, dtype, seed=41)
        index = Index.restore(image)
        queries = util.make_vectors(64, ndim, dtype, seed=42)
        base = check_against_oracle(index, image, queries, 10, dtype, 64)
        in_lds = check_against_oracle(index, image, queries, 10, dtype, 64, tuning=Tuning(mode=1))
        assert in_lds.stats.mode == 1 and in_lds.stats.passes == 1
        hashed = check_against_oracle(index, image, queries, 10, dtype, 64, tuning=Tuning(mode=2))
        assert hashed.stats.mode == 2 and hashed.stats.passes == 1
        small = check_against_oracle(index, image, queries, 10, dtype, 64, tuning=Tuning(hash_cap=256, next_cap=96, mode=1))
        assert small.stats.passes >= 2 and small.stats.retried_lds > 0
        tiny = check_against_oracle(index, image, queries, 10, dtype, 64, tuning=Tuning(hash_cap=64, next_cap=40, mode=2))
        assert tiny.stats.passes >= 3 and tiny.stats.retried_global > 0
        forced = check_against_oracle(index, image, queries, 10, dtype, 64, tuning=Tuning(mode=3))
        assert forced.stats.retried_global == len(queries)
        few_waves = check_against_oracle(index, image, queries, 10, dtype, 64, tuning=Tuning(mode=2, waves_per_cu=1))
        for other in (in_lds, hashed, small, tiny, forced, few_waves):
            assert np.array_equal(base.keys, other.keys)


def test_persistent_waves_cover_large_batches(reference):
    """More queries than resident waves: the ticket queue must hand every query to exactly one wave."""
    from usearch_amd import Index, Tuning
    image, _, _ = util.build_image(2000, 32, "l2sq", "i8", seed=61)
    index = Index.restore(image)
    queries = util.make_vectors(9000, 32, "i8", seed=62)
    for mode in (1, 2):
        got = check_against_oracle(index, image, queries, 5, "i8", 16, tuning=Tuning(mode=mode, waves_per_cu=2))
        assert got.stats.grid <= 2 * 256 < len(queries)
        peaks = index.last_peaks(len(queries))
        assert peaks[:, 0].max() <= 4 * 64 and np.all(peaks[:, 1] >= 1)


def test_every_kernel_build_agrees(reference):
    """`top` in registers (1 / 4 / 8 entries per lane) or in LDS, times the three unroll/occupancy builds."""
    from usearch_amd import Index, Tuning
    image, _, _ = util.build_image(2500, 768, "cos", "f16", seed=71)
    index = Index.restore(image)
    assert index.lanes_per_row == 8
    queries = util.make_vectors(48, 768, "f16", seed=72)
    for expansion in (64, 128, 300, 600):
        for variant in (1, 2, 3, 4):  # 4: two rows per lane group per round — the registers only the heap-less frontier has
            got = check_against_oracle(index, image, queries, 10, "f16", expansion, tuning=Tuning(variant=variant))
            assert got.stats.passes == 1 and got.stats.variant == variant and got.stats.frontier == 2
        for variant in (1, 2, 3):
            got = check_against_oracle(index, image, queries, 10, "f16", expansion, tuning=Tuning(variant=variant, frontier=1))
            assert got.stats.passes == 1 and got.stats.variant == variant and got.stats.frontier == 1
    with pytest.raises(RuntimeError):  # the tight builds do not exist with the heap
        index.search(queries, 10, expansion=64, tuning=Tuning(variant=4, frontier=1))


def test_exact_float_ties_between_frontier_candidates(reference):
    """Where the two frontiers may part ways, and only there: members at EXACTLY the same distance from the query (here: every
    vector stored three times). The heap-less frontier (default for float-valued pairs) is bit-exact against its restatement
    in the oracle; the reference-shaped heap (`Tuning(frontier=1)`) is bit-exact against the reference-shaped oracle; both
    return the same DISTANCES for every query (the same members up to which of three identical twins is named), and the
    traversal counters differ by little. With distinct distances the two are identical (every other test of this file)."""
    from usearch_amd import Index, Tuning
    base = util.make_vectors(700, 48, "f32", seed=91)
    vectors = np.concatenate([base, base, base])
    from oracle import refbind
    reference_index = refbind.RefIndex(48, "l2sq", "f32", connectivity=16, expansion_add=128)
    reference_index.add(np.arange(len(vectors), dtype=np.uint64) + 1000, vectors, threads=1)
    image = reference_index.save_buffer()
    index = Index.restore(image)
    queries = util.make_vectors(120, 48, "f32", seed=92)
    queries[:30] = base[:30]
    in_top = check_against_oracle(index, image, queries, 10, "f32", 64)
    heap = check_against_oracle(index, image, queries, 10, "f32", 64, tuning=Tuning(frontier=1))
    assert in_top.stats.frontier == 2 and heap.stats.frontier == 1
    assert util.same_float_bits(in_top.distances, heap.distances), "the same neighbourhoods, whichever twin is named"
    hops_in_top, hops_heap = in_top.visited_per_query.astype(float), heap.visited_per_query.astype(float)
    assert abs(hops_in_top.mean() / hops_heap.mean() - 1) < 0.05


@pytest.mark.parametrize("metric,dtype,ndim,expansion", [("cos", "f16", 768, 64), ("cos", "f32", 256, 300), ("l2sq", "i8", 1024, 128),
                                                         ("ip", "bf16", 512, 600), ("l2sq", "f32", 300, 1000)])
def test_team_and_one_wave_agree(reference, monkeypatch, metric, dtype, ndim, expansion):
    """Rows of at least 128 bytes, batches of at most two queries per CU: five waves share a query (team_search_kernel) — wave 0
    walks and commits, the others measure the hop's rows (over the reference's heap all five measure). Keys, distance bits and both
    counters equal the one-wave kernel's and the oracle's; a single query, a handful, and the largest batch that still takes the
    team build."""
    from usearch_amd import Index
    image, vectors, _ = util.build_image(5000, ndim, metric, dtype, seed=57)
    index = Index.restore(image)
    queries = util.make_vectors(600, ndim, dtype, seed=58)
    for count in (1, 7, 512, 600):
        monkeypatch.delenv("USEARCH_AMD_NO_TEAM", raising=False)
        team = check_against_oracle(index, image, queries[:count], 10, dtype, expansion)
        monkeypatch.setenv("USEARCH_AMD_NO_TEAM", "1")
        plain = index.search(queries[:count], 10, expansion=expansion, dtype=dtype)
        assert team.stats.variant == (5 if count <= 512 else plain.stats.variant) and plain.stats.variant != 5
        assert np.array_equal(team.keys, plain.keys) and util.same_float_bits(team.distances, plain.distances)
        assert np.array_equal(team.visited_per_query, plain.visited_per_query)
        assert np.array_equal(team.computed_per_query, plain.computed_per_query)


@pytest.mark.parametrize("metric,dtype,ndim,expansion", [("cos", "f16", 768, 96), ("l2sq", "f32", 256, 700)])
def test_team_settles_ties_like_one_wave(reference, monkeypatch, metric, dtype, ndim, expansion):
    """The team's leader names the next member to expand before it commits the hop just measured — exact unless distances tie, and
    then it commits first and looks again. An index in which every vector occurs three times (and queries that ARE stored vectors)
    ties at every hop: keys, distance bits and counters still equal the one-wave kernel's and the oracle's."""
    from usearch_amd import Index
    distinct = util.make_vectors(1700, ndim, dtype, seed=61, metric=metric)
    order = np.random.default_rng(62).permutation(5100)
    vectors = np.ascontiguousarray(np.concatenate([distinct, distinct, distinct])[order])
    image, _, _ = util.build_image(5100, ndim, metric, dtype, vectors=vectors)
    index = Index.restore(image)
    queries = np.ascontiguousarray(np.concatenate([distinct[:40], util.make_vectors(40, ndim, dtype, seed=63, metric=metric)]))
    for count in (1, 80):
        monkeypatch.delenv("USEARCH_AMD_NO_TEAM", raising=False)
        team = check_against_oracle(index, image, queries[:count], 10, dtype, expansion)
        monkeypatch.setenv("USEARCH_AMD_NO_TEAM", "1")
        plain = index.search(queries[:count], 10, expansion=expansion, dtype=dtype)
        assert team.stats.variant == 5 and plain.stats.variant != 5
        assert np.array_equal(team.keys, plain.keys) and util.same_float_bits(team.distances, plain.distances)
        assert np.array_equal(team.visited_per_query, plain.visited_per_query)
        assert np.array_equal(team.computed_per_query, plain.computed_per_query)


def test_team_and_one_wave_agree_when_every_distance_is_a_nan(reference, monkeypatch):
    """A query with a NaN component measures NaN against every member: no newcomer equals the smallest landing distance (the
    minimum of NaNs is +inf), so the pipelined team has nobody to name early — it must commit first and look again, as for a tie
    (it used to read a lane that does not exist). Same keys, bits and counters as the one-wave kernel."""
    from usearch_amd import Index
    ndim, dtype, metric, expansion = 768, "f32", "cos", 400
    vectors = util.make_vectors(3000, ndim, dtype, seed=71, metric=metric)
    image, _, _ = util.build_image(3000, ndim, metric, dtype, vectors=vectors)
    index = Index.restore(image)
    queries = util.make_vectors(6, ndim, dtype, seed=72, metric=metric).copy()
    queries[0, :] = np.nan
    queries[3, 5] = np.nan
    monkeypatch.delenv("USEARCH_AMD_NO_TEAM", raising=False)
    team = index.search(queries, 10, expansion=expansion, dtype=dtype)
    monkeypatch.setenv("USEARCH_AMD_NO_TEAM", "1")
    plain = index.search(queries, 10, expansion=expansion, dtype=dtype)
    assert team.stats.variant == 5 and plain.stats.variant != 5
    assert np.array_equal(team.keys, plain.keys) and np.array_equal(team.counts, plain.counts)
    assert np.array_equal(team.distances.view(np.uint32), plain.distances.view(np.uint32))
    assert np.array_equal(team.visited_per_query, plain.visited_per_query)
    assert np.array_equal(team.computed_per_query, plain.computed_per_query)


def test_the_benchmarked_instantiation_matches_the_oracle(reference):
    """The kernel every BASELINE line of `bench.py` times, held against the oracle ITSELF (not piece by piece): rows of 768 f16
    (8 lanes per row, 12 chunks per lane), expansion 608 (16 `top` cells per lane, frontier in `top`), a batch that fills the chip
    and then some (more queries than the 2 048 persistent waves, so tickets are drawn more than once) ⇒ one wave per query,
    `search_kernel<cos, f16, 8, u12x2, global-hash, 16, in-top>`, the visited sets in per-wave slabs of global memory at 65 536
    cells. Keys, distance bits, counts and BOTH traversal counters of every query equal the oracle's in the kernels' summation
    layout; the reference-shaped heap over the same batch equals its oracle too. (cpp/test.cpp:499-503)"""
    import torch

    import bench
    import usearch_amd
    from usearch_amd import Tuning
    import os
    n, dim, dtype, count, expansion = 20_000, 768, "f16", 2_560, 608
    host_threads = max(1, min(16, os.cpu_count() or 1))
    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
    built = usearch_amd.build(None, "cos", dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
    queries = bench.synthetic_vectors_device(count, dim, dtype, 43, device).cpu().numpy().view(np.float16)
    del data
    image = built.save_buffer()
    index = usearch_amd.Index.restore(image)
    got = index.search(queries, 10, expansion=expansion, dtype=dtype)
    assert got.stats.passes == 1
    assert (got.stats.mode, got.stats.variant, got.stats.frontier, got.stats.top_cells) == (2, 4, 2, 16), \
        "not the instantiation the bench times"
    assert got.stats.grid >= 2048 and count > got.stats.grid
    keys, dists, counts, visited, computed = util.oracle_search(image, queries, 10, dtype, expansion, lanes=index.lanes_per_row,
                                                                frontier_in_top=True, threads=host_threads)
    assert np.array_equal(got.keys, keys) and np.array_equal(got.counts, counts)
    assert util.same_float_bits(got.distances, dists)
    assert np.array_equal(got.visited_per_query, visited) and np.array_equal(got.computed_per_query, computed)
    # the 16-waves-per-CU build that batches of >= 40 000 queries take, on the same batch: nothing but the schedule differs
    wide = index.search(queries, 10, expansion=expansion, dtype=dtype, tuning=Tuning(variant=1))
    assert (wide.stats.mode, wide.stats.variant, wide.stats.frontier) == (2, 1, 2)
    assert np.array_equal(wide.keys, keys) and util.same_float_bits(wide.distances, dists)
    assert np.array_equal(wide.visited_per_query, visited) and np.array_equal(wide.computed_per_query, computed)
    # the reference's heap, same batch, against the reference-shaped oracle; on this index the two frontiers name the same members
    heap = index.search(queries, 10, expansion=expansion, dtype=dtype, tuning=Tuning(frontier=1))
    assert heap.stats.frontier == 1 and heap.stats.mode == 2
    hkeys, hdists, hcounts, hvisited, hcomputed = util.oracle_search(image, queries, 10, dtype, expansion, lanes=index.lanes_per_row,
                                                                     threads=host_threads)
    assert np.array_equal(heap.keys, hkeys) and util.same_float_bits(heap.distances, hdists)
    assert np.array_equal(heap.visited_per_query, hvisited) and np.array_equal(heap.computed_per_query, hcomputed)
    assert util.same_float_bits(heap.distances, got.distances)


def test_large_expansion(reference):
    from usearch_amd import Index
    image, _, _ = util.build_image(4000, 48, "l2sq", "f32", seed=51)
    index = Index.restore(image)
    queries = util.make_vectors(3000, 48, "f32", seed=52)
    for expansion in (256, 1000):
        got = check_against_oracle(index, image, queries, 10, "f32", expansion)
        assert got.stats.mode == 2  # the visited set no longer fits LDS next to 8 waves per CU
        # a batch so small that every query has a wave of its own even at the LDS residency keeps the visited set in LDS
        few = check_against_oracle(index, image, queries[:40], 10, "f32", expansion)
        assert few.stats.mode == 1
        assert np.array_equal(few.keys, got.keys[:40]) and util.same_float_bits(few.distances, got.distances[:40])


@pytest.mark.parametrize("metric,dtype,ndim", [("hamming", "b1", 64), ("cos", "f16", 48)])
def test_every_result_buffer_shape(reference, metric, dtype, ndim):
    """The result buffer lives in 1 / 4 / 8 / 16 registers per lane or in LDS depending on the expansion: every shape, the
    boundaries between them, and `wanted == expansion` (the whole buffer is dumped, across lanes) — on a tie-heavy metric
    (insertion order among equal distances is the reference's lower_bound rule, index.hpp:928-939) and a float one."""
    from usearch_amd import Index
    image, _, ref_index = util.build_image(6000, ndim, metric, dtype, seed=31)
    index = Index.restore(image)
    queries = util.make_vectors(24, ndim, dtype, seed=32)
    for expansion in (1, 10, 63, 64, 65, 255, 256, 257, 512, 513, 700, 1024, 1025, 1500):
        for k in {1, 10, min(expansion, 1100)}:
            if k > expansion:
                continue
            got = check_against_oracle(index, image, queries, k, dtype, expansion)
            if metric == "hamming":  # integer distances: the real reference agrees bit for bit, ties included
                ref_index.expansion_search = expansion
                rkeys, rdists, rcounts, *_ = ref_index.search(queries, k, dtype=dtype, threads=1)
                assert np.array_equal(got.keys, rkeys) and util.same_float_bits(got.distances, rdists)
                assert np.array_equal(got.counts, rcounts)


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity", [
    ("l2sq", "i8", 32, 6000, 4), ("hamming", "b1", 64, 5000, 3), ("cos", "f16", 96, 4000, 4), ("cos", "f32", 24, 4000, 4),
    ("tanimoto", "b1", 128, 3000, 2), ("pearson", "f64", 16, 2000, 3),
])
def test_cluster_matches_oracle_and_reference(reference, metric, dtype, ndim, n, connectivity):
    """`index_dense_gt::cluster(query, level)` (index_dense.hpp:788-793 → index.hpp:3089-3125): the greedy descent alone.
    Small connectivities make tall hierarchies, so several levels are exercised."""
    from oracle import oraclebind
    from usearch_amd import Index
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=31, connectivity=connectivity)
    queries = util.make_vectors(120, ndim, dtype, seed=32, metric=metric)
    index = Index.restore(image)
    oracle = oraclebind.OracleIndex(image)
    max_level = int(ref_index.graph_shape()[0])
    assert max_level >= 3
    for level in (0, 1, 2, max_level, max_level + 3):
        keys, distances, visited, computed = index.cluster(queries, level, dtype=dtype)
        okeys, odistances, ovisited, ocomputed = oracle.cluster(queries, level, dtype=dtype, lanes=index.lanes_per_row)
        assert np.array_equal(keys, okeys), level
        assert util.same_float_bits(distances, odistances)
        assert np.array_equal(visited, ovisited) and np.array_equal(computed, ocomputed)
        rkeys, rdistances, rvisited, rcomputed = ref_index.cluster(queries, level, dtype=dtype, threads=1)
        if util.exact_pair(metric, dtype):
            assert np.array_equal(keys, rkeys) and util.same_float_bits(distances, rdistances)
            assert np.array_equal(visited, rvisited) and np.array_equal(computed, rcomputed)
        else:
            assert (keys == rkeys).mean() > 0.97
            assert np.all(np.abs(distances - rdistances)[keys == rkeys] <= util.tolerance(dtype) * np.maximum(1, np.abs(rdistances[keys == rkeys])))
    # a query in a foreign scalar kind is cast first, like `search`
    if dtype == "f16":
        as_f32 = queries.astype(np.float32)
        keys, *_ = index.cluster(as_f32, 1, dtype="f32")
        okeys, *_ = oracle.cluster(as_f32, 1, dtype="f32", lanes=index.lanes_per_row)
        assert np.array_equal(keys, okeys)


def test_cluster_on_an_empty_index(reference):
    from usearch_amd import Index
    image, _, _ = util.build_image(0, 16, "cos", "f32")
    keys, distances, visited, computed = Index.restore(image).cluster(np.ones((3, 16), dtype=np.float32), 1)
    assert np.all(keys == 0) and np.all(np.isnan(distances)) and np.all(computed == 0)


def test_graph_only_image_with_the_callers_vectors(reference):
    """`serialization_config_t::exclude_vectors` (index_dense.hpp:1004): the file holds the graph alone, the vectors stayed with
    the caller. Loaded together they are the index that was saved — strided rows included."""
    from usearch_amd import Index
    image, vectors, _ = util.build_image(2500, 48, "cos", "f16", seed=81)
    queries = util.make_vectors(60, 48, "f16", seed=82)
    whole = Index.restore(image).search(queries, 10)
    graph_only = util.without_vectors(image)
    with pytest.raises(RuntimeError, match="exclude_vectors"):
        Index.restore(graph_only)
    padded = np.zeros((len(vectors), 64), dtype=np.float16)
    padded[:, :48] = vectors
    for matrix in (vectors, padded[:, :48]):
        parts = Index.restore(graph_only, vectors=matrix).search(queries, 10)
        assert np.array_equal(parts.keys, whole.keys) and util.same_float_bits(parts.distances, whole.distances)
        assert np.array_equal(parts.computed_per_query, whole.computed_per_query)


# (metric, dtype, ndim, n, connectivity, k, expansion, queries, forced tuning, the plain build exists)
PLAIN_CONFIGS = [
    ("hamming", "b1", 128, 5000, 16, 10, 64, 300, {}, True),      # config 5's kernel: rows inline with the lists, one `top` cell per lane
    ("hamming", "b1", 128, 5000, 16, 10, 72, 200, {}, True),      # two `top` cells per lane
    ("hamming", "b1", 128, 5000, 16, 10, 100, 200, {}, True),
    ("hamming", "b1", 96, 3000, 5, 7, 32, 200, {}, True),         # lists of 10 cells
    ("l2sq", "i8", 96, 4000, 16, 10, 64, 200, {}, True),          # config 4's rows (G = 2): gathered next to the probe
    ("l2sq", "i8", 96, 4000, 16, 10, 80, 200, {}, True),          # config 4's kernel
    ("cos", "i8", 96, 1500, 16, 10, 112, 80, {}, True),           # two `top` cells, the widest frontier that leaves LDS for `aside`
    ("cos", "i8", 96, 1500, 16, 10, 128, 80, {}, False),          # … and one that does not: 512 cells more would cost a resident wave
    ("ip", "i8", 16, 1200, 16, 10, 64, 100, {}, True),            # 16-byte rows of another pair inline
    ("l2sq", "f16", 48, 1500, 16, 10, 64, 100, {"frontier": 1}, True),   # float pairs only when the heap is asked for
    ("cos", "f32", 24, 1500, 16, 10, 64, 100, {"frontier": 1}, True),
    ("l2sq", "f16", 48, 1500, 16, 10, 64, 100, {}, False),        # their default frontier rides in `top`: the general build
    ("l2sq", "i8", 96, 1500, 40, 10, 64, 100, {}, False),         # lists of 80 cells: two tiles per hop
    ("hamming", "b1", 128, 3000, 16, 10, 200, 100, {}, False),    # four `top` cells per lane
    ("tanimoto", "b1", 128, 3000, 16, 10, 64, 100, {}, False),    # outside the common pairs
]


@pytest.mark.parametrize("metric,dtype,ndim,n,connectivity,k,expansion,nq,forced,exists", PLAIN_CONFIGS)
def test_plain_build_of_the_short_row_walk(reference, monkeypatch, metric, dtype, ndim, n, connectivity, k, expansion, nq, forced, exists):
    """Rows of ≤ 128 bytes over the global hash: a plain `search` batch runs the kernel build cut for it (kernels.hpp `plain_ak`:
    level 0, no predicate / tombstones, lists of one tile, `seen` cells, rows inline or gathered next to the probe) wherever that
    build exists, and the general build everywhere else. Both against the oracle bit for bit incl. both counters, against each
    other, and against the compiled reference."""
    from usearch_amd import Index, Tuning
    image, vectors, ref_index = util.build_image(n, ndim, metric, dtype, seed=31, connectivity=connectivity)
    queries = util.make_vectors(nq, ndim, dtype, seed=32, metric=metric)
    queries[: nq // 4] = vectors[: nq // 4]
    index = Index.restore(image)
    assert index.lanes_per_row <= 2
    monkeypatch.delenv("USEARCH_AMD_NO_PLAIN", raising=False)
    first = check_against_oracle(index, image, queries, k, dtype, expansion, tuning=Tuning(mode=2, **forced))
    assert first.stats.mode == 2 and first.stats.passes == 1
    assert first.stats.plain == (1 if exists else 0)
    # gathered rows (G = 2): the cut probes the slab at the home cell only and sets collided members aside in LDS
    assert (first.stats.aside_cells > 0) == (bool(exists) and index.lanes_per_row == 2)
    monkeypatch.setenv("USEARCH_AMD_NO_PLAIN", "1")
    general = check_against_oracle(index, image, queries, k, dtype, expansion, tuning=Tuning(mode=2, **forced))
    assert general.stats.plain == 0 and general.stats.mode == 2
    monkeypatch.delenv("USEARCH_AMD_NO_PLAIN")
    assert np.array_equal(first.keys, general.keys) and util.same_float_bits(first.distances, general.distances)
    assert np.array_equal(first.counts, general.counts)
    assert np.array_equal(first.visited_per_query, general.visited_per_query)
    assert np.array_equal(first.computed_per_query, general.computed_per_query)
    if util.exact_pair(metric, dtype):
        ref_index.expansion_search = expansion
        rkeys, rdists, rcounts, rvisited, rcomputed = ref_index.search(queries, k, dtype=dtype, threads=1)
        assert np.array_equal(first.keys, rkeys) and util.same_float_bits(first.distances, rdists)
        assert np.array_equal(first.counts, rcounts)
        assert np.array_equal(first.visited_per_query, rvisited) and np.array_equal(first.computed_per_query, rcomputed)


def test_plain_build_outgrows_its_room(reference, monkeypatch):
    """Gathered rows (G = 2): the plain build probes the slab at a member's home cell only and sets what collides aside in LDS. A
    table far too small for the walk (forced) fills up: the query is abandoned and run again by the retry ladder with more room —
    same results, bit for bit, as the first try of a well-sized launch."""
    from usearch_amd import Index, Tuning
    for expansion in (64, 80):  # one and two `top` cells per lane
        image, vectors, _ = util.build_image(6000, 96, "l2sq", "i8", seed=35)
        queries = util.make_vectors(200, 96, "i8", seed=36)
        index = Index.restore(image)
        monkeypatch.delenv("USEARCH_AMD_ASIDE_CELLS", raising=False)
        base = check_against_oracle(index, image, queries, 10, "i8", expansion, tuning=Tuning(mode=2))
        assert base.stats.plain == 1 and base.stats.aside_cells == 512 and base.stats.passes == 1
        monkeypatch.setenv("USEARCH_AMD_PLAIN_WHATEVER_THE_ROOM", "1")
        monkeypatch.setenv("USEARCH_AMD_ASIDE_CELLS", "64")
        tight = check_against_oracle(index, image, queries, 10, "i8", expansion, tuning=Tuning(mode=2))
        assert tight.stats.passes >= 2, "64 cells (48 usable) were expected to overflow with the collisions of these walks"
        assert np.array_equal(base.keys, tight.keys) and util.same_float_bits(base.distances, tight.distances)
        assert np.array_equal(base.visited_per_query, tight.visited_per_query)
        assert np.array_equal(base.computed_per_query, tight.computed_per_query)
        monkeypatch.delenv("USEARCH_AMD_PLAIN_WHATEVER_THE_ROOM")


def test_plain_build_steps_aside(reference):
    """What the plain build takes for granted is checked per launch: a predicate, tombstones or a member's own row to leave out send the
    same index through the general build (tests/test_gpu_filtered.py and test_gpu_build.py hold those against the oracle)."""
    from usearch_amd import Index, Tuning
    image, vectors, _ = util.build_image(4000, 128, "hamming", "b1", seed=33)
    queries = util.make_vectors(128, 128, "b1", seed=34)
    index = Index.restore(image)
    assert index.search(queries, 10, expansion=64, dtype="b1", tuning=Tuning(mode=2)).stats.plain == 1
    keys = np.arange(4000, dtype=np.uint64)
    every_third = index.filter_keys(keys[keys % 3 == 0])
    filtered = index.search(queries, 10, expansion=64, dtype="b1", tuning=Tuning(mode=2), filter=every_third)
    assert filtered.stats.plain == 0 and filtered.stats.mode == 2
    assert np.all(filtered.keys[np.arange(10)[None, :] < filtered.counts[:, None]] % 3 == 0)
