"""GPU: the shard-merge kernel (`usearch_amd_merge_many`) against the oracle's `merge_into` applied shard by shard, on
tie-heavy inputs, and the whole sharded pipeline on one GPU (several shard indexes searched, stacked as an all-gather
would, merged) against the oracle doing the same on the CPU."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def oracle_merge(distances, keys, counts):
    from oracle import oraclebind
    shards, q, k = distances.shape
    out_k = np.zeros((q, k), dtype=np.uint64)
    out_d = np.zeros((q, k), dtype=np.float32)
    out_c = np.zeros(q, dtype=np.uint64)
    for i in range(q):
        merged = 0
        for shard in range(shards):
            n = int(counts[shard, i])
            merged = oraclebind.merge_into(out_k[i], out_d[i], merged, keys[shard, i, :n], distances[shard, i, :n], n)
        out_c[i] = merged
    return out_k, out_d, out_c


@pytest.mark.parametrize("shards,k,levels", [(2, 10, 3), (8, 10, 4), (8, 1, 2), (3, 100, 50), (5, 7, 1000)])
def test_merge_kernel_matches_merge_into(shards, k, levels):
    from usearch_amd import index as ua
    rng = np.random.default_rng(shards * 100 + k)
    q = 200
    distances = np.sort(rng.integers(0, levels, size=(shards, q, k)).astype(np.float32) * 0.25, axis=2)
    keys = rng.integers(1, 1 << 40, size=(shards, q, k)).astype(np.uint64)
    counts = rng.integers(0, k + 1, size=(shards, q)).astype(np.uint64)
    counts[:, :5] = 0  # queries nobody answers
    counts[:, 5:10] = k
    got_k, got_d, got_c = ua.merge_many(distances, keys, counts)
    want_k, want_d, want_c = oracle_merge(distances, keys, counts)
    assert np.array_equal(got_c, want_c)
    for i in range(q):
        n = int(want_c[i])
        assert np.array_equal(got_k[i, :n], want_k[i, :n]) and np.array_equal(got_d[i, :n], want_d[i, :n]), i
        assert np.all(got_k[i, n:] == 0) and np.all(np.isnan(got_d[i, n:]))


def test_sharded_pipeline_on_one_gpu(reference):
    from usearch_amd import Index
    from usearch_amd import index as ua
    shards, per_shard, ndim, k = 3, 1500, 128, 10
    queries = util.make_vectors(64, ndim, "b1", seed=99)
    gathered, oracle_parts = [], []
    for shard in range(shards):
        keys = np.arange(per_shard, dtype=np.uint64) + shard * per_shard
        image, _, _ = util.build_image(per_shard, ndim, "hamming", "b1", seed=80 + shard, keys=keys)
        got = Index.restore(image).search(queries, k)
        gathered.append((got.distances, got.keys, got.counts))
        oracle_parts.append(util.oracle_search(image, queries, k, "b1", 64))
    distances = np.stack([g[0] for g in gathered])
    keys = np.stack([g[1] for g in gathered])
    counts = np.stack([g[2] for g in gathered])
    got_k, got_d, got_c = ua.merge_many(distances, keys, counts)
    want_k, want_d, want_c = oracle_merge(np.stack([p[1] for p in oracle_parts]), np.stack([p[0] for p in oracle_parts]),
                                          np.stack([p[2] for p in oracle_parts]))
    assert np.array_equal(got_k, want_k) and np.array_equal(got_d, want_d) and np.array_equal(got_c, want_c)
    assert len(np.unique(got_k // per_shard)) == shards  # results really come from every shard
