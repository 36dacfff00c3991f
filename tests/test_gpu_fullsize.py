"""Parity at the BASELINE configurations' own scale, under the driver: config 2 at full size (1M x 768 f32 cosine) and 2M-vector
slices of configs 4 and 5 (96-d i8 L2, 128-bit Hamming) — index built on the device, saved, and handed to the REAL reference
(`oracle/_ref`, `usearch_view_buffer`); the same >= 512 queries searched by both (more than two per compute unit: the one-wave
kernel the bench lines time, not the team build of small batches). Integer-valued pairs: keys, distance bits,
counts and both traversal counters identical, ties included. Float pair: the oracle in the kernels' summation layout bit for
bit, the reference within the stated tolerance with IDENTICAL labels at every position whose neighbouring reference distances
are farther apart than twice that tolerance (tests/util.py `assert_float_parity`). (The 10M / 100M / 125M configurations themselves
carry the same comparison inside bench.py: "label agreement with the GPU" in every line.)"""
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

CONFIGS = [  # (vectors, dimensions, dtype, metric, queries, k, expansion)
    (1_000_000, 768, "f32", "cos", 640, 10, 64),   # > 2 queries per CU: the one-wave kernel every BASELINE line times
    (2_000_000, 96, "i8", "l2sq", 512, 10, 64),
    (2_000_000, 128, "b1", "hamming", 512, 10, 64),
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
    rkeys, rdists, rcounts, rvisited, rcomputed = reference_index.search(batch, k, dtype=dtype, threads=0)
    assert np.array_equal(got.counts, rcounts)
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
        assert separated > 0.5 and agreement > 0.98
        okeys, odists, ocounts, ovisited, ocomputed = oraclebind.OracleIndex(image).search(
            batch, k, dtype=dtype, expansion=expansion, lanes=index.lanes_per_row, frontier_in_top=got.stats.frontier == 2)
        assert np.array_equal(got.keys, okeys) and util.same_float_bits(got.distances, odists)
        assert np.array_equal(got.visited_per_query, ovisited) and np.array_equal(got.computed_per_query, ocomputed)
    # what the device loader makes of the saved image is the index that was built
    restored = usearch_amd.Index.restore(image)
    again = restored.search(batch, k, expansion=expansion, dtype=dtype)
    assert np.array_equal(again.keys, got.keys) and util.same_float_bits(again.distances, got.distances)
    assert np.array_equal(again.computed_per_query, got.computed_per_query)
