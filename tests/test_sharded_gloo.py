"""CPU, world_size 2, gloo: the sharded-search protocol of usearch_amd/sharded.py — query broadcast, one all-gather per
tensor, merge in rank order with the `merge_into` tie rule (index.hpp:2650-2670). The local search and the merge are
injected (a seeded fake per rank, the oracle's merge_into), so that no GPU is needed; the GPU bindings of the same class
are covered by tests/test_gpu_merge.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2
Q, K = 37, 6


def fake_shard_results(rank: int, queries: torch.Tensor):
    """Seeded, tie-heavy, ascending per-query results that depend on the (broadcast) queries."""
    seed = int(queries.sum().item()) % 1000 + 17 * rank
    rng = np.random.default_rng(seed)
    distances = np.sort(rng.integers(0, 5, size=(Q, K)).astype(np.float32), axis=1)
    keys = (rng.integers(0, 10_000, size=(Q, K)) * WORLD + rank).astype(np.int64)
    counts = rng.integers(0, K + 1, size=Q).astype(np.int64)
    return torch.from_numpy(keys), torch.from_numpy(distances), torch.from_numpy(counts)


def oracle_merge(all_distances, all_keys, all_counts):
    from oracle import oraclebind
    shards, q, k = all_distances.shape
    keys = np.zeros((q, k), dtype=np.uint64)
    distances = np.zeros((q, k), dtype=np.float32)
    counts = np.zeros(q, dtype=np.int64)
    for i in range(q):
        merged = 0
        for shard in range(shards):
            n = int(all_counts[shard, i])
            merged = oraclebind.merge_into(keys[i], distances[i], merged, all_keys[shard, i, :n].numpy().astype(np.uint64),
                                           all_distances[shard, i, :n].numpy(), n)
        counts[i] = merged
    return torch.from_numpy(keys.astype(np.int64)), torch.from_numpy(distances), torch.from_numpy(counts)


def worker(rank: int, port: int, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from usearch_amd.sharded import ShardedSearcher
        seen = {}

        def local_search(queries, k, expansion):
            seen["checksum"] = int(queries.sum().item())
            return fake_shard_results(rank, queries)

        searcher = ShardedSearcher(local_search, oracle_merge)
        queries = torch.full((Q, 8), 3 if rank == 0 else 99, dtype=torch.int32)  # only rank 0's batch counts
        keys, distances, counts = searcher.search(queries, K, 64)
        assert seen["checksum"] == Q * 8 * 3, "the batch was not broadcast from rank 0"
        # expectation, computed locally from both ranks' deterministic fakes
        reference_queries = torch.full((Q, 8), 3, dtype=torch.int32)
        parts = [fake_shard_results(r, reference_queries) for r in range(WORLD)]
        expected = oracle_merge(torch.stack([p[1] for p in parts]), torch.stack([p[0] for p in parts]),
                                torch.stack([p[2] for p in parts]))
        assert torch.equal(counts, expected[2])
        for i in range(Q):
            n = int(counts[i])
            assert torch.equal(keys[i, :n], expected[0][i, :n]) and torch.equal(distances[i, :n], expected[1][i, :n])
            assert n == min(K, int(parts[0][2][i] + parts[1][2][i]))
        results[rank] = True
    finally:
        dist.destroy_process_group()


def test_sharded_protocol_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(worker, args=(port, results), nprocs=WORLD, join=True)
    assert dict(results) == {0: True, 1: True}
