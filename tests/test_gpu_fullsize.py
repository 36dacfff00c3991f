"""Parity at the BASELINE configurations' OWN sizes, under the driver: config 2 (1M x 768 f32 cosine), config 3 — the headline —
(10M x 768 f16 cosine at the headline's expansion 608: the `u12x2` build over the global visited set that bench.py times),
config 4 (100M x 96 i8 L2 at its expansion 80) and one shard of config 5 (125M x 128 b1 Hamming), plus 2M-vector slices of
the short-row shapes — index built on the device, saved, and handed to the REAL reference
(`oracle/_ref`, `usearch_view_buffer`); the same >= 512 queries searched by both (more than two per compute unit: the one-wave
kernel the bench lines time, not the team build of small batches). Integer-valued pairs: keys, distance bits,
counts and both traversal counters identical, ties included. Float pair: the oracle in the kernels' summation layout bit for
bit, the reference within the stated tolerance with IDENTICAL labels at every position whose neighbouring reference distances
are farther apart than twice that tolerance (tests/util.py `assert_float_parity`; the reference's own bar at this scale
is cpp/test.cpp:499-503: counts within the index size, distances non-decreasing — asserted too). The full-size rows take about
a minute each (build 4 … 50 s, image 3 … 25 GB on the host); USEARCH_AMD_SKIP_FULL_SIZE=1 leaves them out of a quick run."""
import os
import sys

import numpy as np
import pytest

from oracle import oraclebind, refbind
from tests import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FULL = pytest.mark.skipif(os.environ.get("USEARCH_AMD_SKIP_FULL_SIZE") == "1", reason="USEARCH_AMD_SKIP_FULL_SIZE=1")
CONFIGS = [  # (vectors, dimensions, dtype, metric, queries, k, expansion)
    (1_000_000, 768, "f32", "cos", 640, 10, 64),   # > 2 queries per CU: the one-wave kernel every BASELINE line times
    (2_000_000, 96, "i8", "l2sq", 512, 10, 64),
    (2_000_000, 128, "b1", "hamming", 512, 10, 64),
    pytest.param(10_000_000, 768, "f16", "cos", 640, 10, 608, marks=FULL, id="headline-10M-768-f16-ef608"),
    pytest.param(100_000_000, 96, "i8", "l2sq", 1024, 10, 80, marks=FULL, id="config4-100M-96-i8-ef80"),
    pytest.param(125_000_000, 128, "b1", "hamming", 1024, 10, 64, marks=FULL, id="config5-shard-125M-128-b1-ef64"),
]


@pytest.mark.parametrize("n,dim,dtype,metric,queries,k,expansion", CONFIGS)
def test_baseline_shapes_at_scale_match_the_reference(reference, n, dim, dtype, metric, queries, k, expansion):
    import torch

    import bench
    import usearch_amd
    device = torch.device("cuda", 0)
    data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
    built = usearch_amd.build(None, metric, dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
    del data
    torch.cuda.empty_cache()
    index = built.index
    assert len(index) == n
    batch = bench.synthetic_vectors_device(queries, dim, dtype, 43, device).cpu().numpy().view(bench.NUMPY_STORAGE[dtype])
    image = built.save_buffer()
    reference_index = refbind.RefIndex.from_buffer(image, view=True, dtype=dtype)
    reference_index.expansion_search = expansion
    got = index.search(batch, k, expansion=expansion, dtype=dtype)
    assert got.stats.passes == 1, "the default scratch must hold these traversals"
    assert got.stats.variant != 5, "a batch of this size must walk with one wave per query (the benchmarked kernel), not the team build"
    if (n, dtype, expansion) == (10_000_000, "f16", 608):  # the instantiation bench.py times: search_kernel<cos, f16, 8, u12x2, global hash, 16, in-top>
        assert (got.stats.mode, got.stats.variant, got.stats.frontier, got.stats.top_cells) == (2, 4, 2, 16), \
            (got.stats.mode, got.stats.variant, got.stats.frontier, got.stats.top_cells)
    rkeys, rdists, rcounts, rvisited, rcomputed = reference_index.search(batch, k, dtype=dtype, threads=0)
    assert np.array_equal(got.counts, rcounts)
    # the reference's own bar at scale (cpp/test.cpp:499-503): no more results than members, distances non-decreasing
    found = got.counts.astype(np.int64)
    assert np.all(found <= min(n, k)) and np.all(np.diff(got.distances, axis=1)[np.arange(k - 1)[None, :] < found[:, None] - 1] >= 0)
    if util.exact_pair(metric, dtype):
        assert np.array_equal(got.keys, rkeys)
        assert util.same_float_bits(got.distances, rdists)
        assert np.array_equal(got.visited_per_query, rvisited)
        assert np.array_equal(got.computed_per_query, rcomputed)
    else:
        separated, agreement = util.assert_float_parity(
            got.keys, got.distances, got.counts,
            lambda queries_, wanted: reference_index.search(queries_, wanted, dtype=dtype, threads=0), batch, k, dtype,
            what=f"{n} x {dim} {dtype}")
        print(f"[fullsize] {n} x {dim} {dtype} ef={expansion}: {separated:.3f} of the positions separated in the reference's distances, "
              f"label agreement {agreement:.4f} on all positions")
        # f16 rows at 10M: neighbouring distances crowd inside the f16 tolerance (2e-3), fewer positions are separated
        assert (separated > 0.5 and agreement > 0.98) if dtype == "f32" else (separated > 0.02 and agreement > 0.5)
        okeys, odists, ocounts, ovisited, ocomputed = util.oracle_search(
            image, batch, k, dtype, expansion=expansion, lanes=index.lanes_per_row, frontier_in_top=got.stats.frontier == 2,
            threads=min(8, os.cpu_count() or 1))
        assert np.array_equal(got.keys, okeys) and util.same_float_bits(got.distances, odists)
        assert np.array_equal(got.visited_per_query, ovisited) and np.array_equal(got.computed_per_query, ocomputed)
    # what the device loader makes of the saved image is the index that was built
    if n >= 50_000_000:  # one copy of a 25 … 85 GB index in HBM at a time
        del index, reference_index
        built.close()
    restored = usearch_amd.Index.restore(image)
    again = restored.search(batch, k, expansion=expansion, dtype=dtype)
    assert np.array_equal(again.keys, got.keys) and util.same_float_bits(again.distances, got.distances)
    assert np.array_equal(again.computed_per_query, got.computed_per_query)


def test_tuning_the_placement_moves_nothing_but_the_matrix(reference):
    """`usearch_amd_snapshot_tune`: the host hands a sample batch, the engine times fresh copies of the matrix of
    stored rows against the incumbent on the sample's first queries and keeps the fastest. Whatever it decides, the index answers as
    before — keys, distance bits and both counters — every trial is reported, nothing is tried for arrays under 1 GiB or samples that do
    not fill the chip, and no later search call runs a trial on its own."""
    import torch

    import bench
    import usearch_amd
    device = torch.device("cuda", 0)
    n, dim, dtype, queries, k, expansion = 1_000_000, 768, "f32", 8192, 10, 64
    data = bench.synthetic_vectors_device(n, dim, dtype, 42, device)
    built = usearch_amd.build(None, "cos", dtype, device_pointer=data.data_ptr(), count=n, stride=data.stride(0), ndim=dim)
    del data
    torch.cuda.empty_cache()
    index = usearch_amd.Index.restore(built.save_buffer())
    built.close()
    batch_device = bench.synthetic_vectors_device(queries, dim, dtype, 43, device)
    batch = batch_device.cpu().numpy().view(bench.NUMPY_STORAGE[dtype])
    before = index.search(batch, k, expansion=expansion, dtype=dtype)
    assert index.placement["draws"] == 0, "no search call runs a placement trial on its own"
    trials = index.tune_device(batch_device.data_ptr(), queries, batch_device.stride(0), k, expansion, max_trials=4)
    report = index.placement
    assert 1 <= trials <= 4 and report["draws"] == trials and len(report["judge_ms"]) == trials and 0 <= report["kept"] <= trials
    assert all(ms > 0 for ms in report["judge_ms"] + report["incumbent_ms"])
    after = index.search(batch, k, expansion=expansion, dtype=dtype)
    assert np.array_equal(after.keys, before.keys) and util.same_float_bits(after.distances, before.distances)
    assert np.array_equal(after.computed_per_query, before.computed_per_query) and np.array_equal(after.visited_per_query, before.visited_per_query)
    assert index.placement["draws"] == trials, "… nor afterwards"
    # a sample that does not fill the chip twice over is no judge: nothing is tried
    assert index.tune_device(batch_device.data_ptr(), 256, batch_device.stride(0), k, expansion) == 0
    # nor is anything tried for a matrix under 1 GiB
    small_image, _, _ = util.build_image(3000, 96, "cos", "f16", seed=5)
    small = usearch_amd.Index.restore(small_image)
    small_batch = bench.synthetic_vectors_device(8192, 96, "f16", 44, device)
    assert small.tune_device(small_batch.data_ptr(), 8192, small_batch.stride(0), k, expansion) == 0 and small.placement["draws"] == 0
