"""CPU, world_size 2, gloo: the sharded-search step through the SAME native entry point the GPUs use
(`usearch_amd_sharded_search_many`, usearch_amd/csrc/sharded.hip) — query broadcast, ONE all-gather of the packed block
{distances | keys | counts | flags}, merge in rank order with the `merge_into` tie rule (index.hpp:2650-2670). Only the
device search is a stand-in (a seeded fake per rank, handed in through the transport), the collectives are gloo's; packing,
exchange and merge are the product's own code running in host memory. The expectation is the oracle's `merge_into`.
The GPU bindings of the same step are covered by tests/test_gpu_merge.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2
Q, K, STRIDE = 37, 6, 32


def fake_shard_results(rank: int, checksum: int):
    """Seeded, tie-heavy, ascending per-query results that depend on the (broadcast) queries."""
    rng = np.random.default_rng(checksum % 1000 + 17 * rank)
    distances = np.sort(rng.integers(0, 5, size=(Q, K)).astype(np.float32), axis=1)
    keys = (rng.integers(0, 10_000, size=(Q, K)) * WORLD + rank).astype(np.uint64)
    counts = rng.integers(0, K + 1, size=Q).astype(np.uint64)
    return keys, distances, counts


def oracle_merge(parts):
    from oracle import oraclebind
    keys = np.zeros((Q, K), dtype=np.uint64)
    distances = np.zeros((Q, K), dtype=np.float32)
    counts = np.zeros(Q, dtype=np.uint64)
    for i in range(Q):
        merged = 0
        for shard_keys, shard_distances, shard_counts in parts:  # shards in rank order
            n = int(shard_counts[i])
            merged = oraclebind.merge_into(keys[i], distances[i], merged, shard_keys[i, :n], shard_distances[i, :n], n)
        counts[i] = merged
    return keys, distances, counts


def worker(rank: int, port: int, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # both ranks are here: never the interface the hostname resolves to
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from usearch_amd.sharded import Communicator
        seen = {}

        def all_gather(send: np.ndarray, receive: np.ndarray):
            dist.all_gather_into_tensor(torch.from_numpy(receive), torch.from_numpy(send))

        def broadcast(buffer: np.ndarray, root: int):
            dist.broadcast(torch.from_numpy(buffer), src=root)

        def local_search(queries: np.ndarray, count: int, wanted: int, expansion: int):
            assert (count, wanted, expansion) == (Q, K, 64) and queries.shape == (Q, STRIDE)
            seen["checksum"] = int(queries.astype(np.int64).sum())
            return fake_shard_results(rank, seen["checksum"])

        communicator = Communicator.on_host(rank, WORLD, all_gather, broadcast, local_search)
        assert (communicator.rank, communicator.world) == (rank, WORLD)
        queries = np.full((Q, STRIDE), 3 if rank == 0 else 99, dtype=np.uint8)  # only rank 0's batch counts
        keys = np.zeros((Q, K), dtype=np.uint64)
        distances = np.zeros((Q, K), dtype=np.float32)
        counts = np.zeros(Q, dtype=np.uint64)
        stats, step = communicator.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, 0, keys.ctypes.data,
                                              distances.ctypes.data, counts.ctypes.data, 0, 0)
        assert seen["checksum"] == Q * STRIDE * 3, "the batch was not broadcast from rank 0"
        assert step.block_bytes == ((Q * K * 4 + 7) // 8 * 8) + Q * K * 8 + Q * 8 + 8
        assert step.gathered_bytes == WORLD * step.block_bytes
        parts = [fake_shard_results(r, Q * STRIDE * 3) for r in range(WORLD)]
        expected_keys, expected_distances, expected_counts = oracle_merge(parts)
        assert np.array_equal(counts, expected_counts)
        for i in range(Q):
            n = int(counts[i])
            assert n == min(K, int(parts[0][2][i] + parts[1][2][i]))
            assert np.array_equal(keys[i, :n], expected_keys[i, :n])
            assert np.array_equal(distances[i, :n], expected_distances[i, :n])
            assert np.all(keys[i, n:] == 0) and np.all(np.isnan(distances[i, n:]))  # padding of index.hpp:2707-2722
        # without a broadcast every rank searches what it was handed
        communicator.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, -1, keys.ctypes.data, distances.ctypes.data,
                                counts.ctypes.data, 0, 0)
        assert seen["checksum"] == Q * STRIDE * 3  # rank 1's buffer was overwritten by the first step's broadcast

        # a rank whose local search FAILS still enters the collective (abort bit in its block's flag word): both ranks leave
        # the step with an error — the failing one with its own message, the other naming it — and nobody hangs
        failing = {"rank": 1}

        def flaky_search(queries: np.ndarray, count: int, wanted: int, expansion: int):
            if rank == failing["rank"]:
                raise MemoryError("injected: this shard ran out of scratch")
            return fake_shard_results(rank, 1)

        flaky = Communicator.on_host(rank, WORLD, all_gather, broadcast, flaky_search)
        for bad_rank in (1, 0):
            failing["rank"] = bad_rank
            try:
                flaky.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, 0, keys.ctypes.data, distances.ctypes.data,
                                 counts.ctypes.data, 0, 0)
                raise AssertionError("the step succeeded although a rank failed")
            except RuntimeError as error:
                text = str(error)
                if rank == bad_rank:
                    assert "injected" in text, text
                else:
                    assert f"aborted by rank {bad_rank} of {WORLD}" in text and "local search" in text, text
        # and the communicator is still usable afterwards: the next step is a normal one
        failing["rank"] = -1
        stats, step = flaky.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, 0, keys.ctypes.data,
                                       distances.ctypes.data, counts.ctypes.data, 0, 0)
        assert step.exchanges == 1
        results[rank] = True
    finally:
        dist.destroy_process_group()


def test_sharded_step_two_ranks_gloo_through_the_c_abi():
    manager = mp.Manager()
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        results = manager.dict()
        try:
            mp.spawn(worker, args=(port, results), nprocs=WORLD, join=True)
            break
        except Exception as error:  # noqa: BLE001
            # one more try on a fresh port when the rendezvous did not come up; what a worker asserted stays fatal
            if attempt or "AssertionError" in str(error):
                raise
            print(f"two-rank rendezvous failed, trying once more: {str(error)[-2000:]}", flush=True)
    assert dict(results) == {0: True, 1: True}


def test_single_rank_step_needs_no_exchange():
    """world = 1: the same entry point, the merge still pads and counts."""
    from usearch_amd.sharded import Communicator

    def refuse(*_):
        raise AssertionError("a single rank has nothing to exchange")

    parts = [fake_shard_results(0, 5)]
    communicator = Communicator.on_host(0, 1, refuse, None, lambda queries, count, wanted, expansion: parts[0])
    queries = np.zeros((Q, STRIDE), dtype=np.uint8)
    keys = np.zeros((Q, K), dtype=np.uint64)
    distances = np.zeros((Q, K), dtype=np.float32)
    counts = np.zeros(Q, dtype=np.uint64)
    communicator.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, 0, keys.ctypes.data, distances.ctypes.data,
                            counts.ctypes.data, 0, 0)
    expected_keys, expected_distances, expected_counts = oracle_merge(parts)
    assert np.array_equal(counts, expected_counts)
    for i in range(Q):
        n = int(counts[i])
        assert np.array_equal(keys[i, :n], expected_keys[i, :n]) and np.array_equal(distances[i, :n], expected_distances[i, :n])


# ---- world = 8: the shape of BASELINE config 5 (eight shards, one per GPU), uneven shards, fewer than k results in most of them,
#      a failing rank in the middle

WORLD8 = 8


def uneven_shard_results(rank: int, checksum: int):
    """Shards of very different sizes: rank r can return at most r results per query (rank 0 none at all), so that most queries
    collect their k results from several shards and some end up with fewer than k in total."""
    rng = np.random.default_rng(checksum % 1000 + 31 * rank)
    most = min(K, rank)
    distances = np.sort(rng.integers(0, 4, size=(Q, K)).astype(np.float32), axis=1)
    keys = (rng.integers(0, 10_000, size=(Q, K)) * WORLD8 + rank).astype(np.uint64)
    counts = rng.integers(0, most + 1, size=Q).astype(np.uint64)
    if rank == 5:
        counts[: Q // 2] = 0  # half of the queries find nothing in this shard
    return keys, distances, counts


def worker8(rank: int, port: int, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD8)
    try:
        from usearch_amd.sharded import Communicator
        failing = {"rank": -1}

        def all_gather(send: np.ndarray, receive: np.ndarray):
            dist.all_gather_into_tensor(torch.from_numpy(receive), torch.from_numpy(send))

        def broadcast(buffer: np.ndarray, root: int):
            dist.broadcast(torch.from_numpy(buffer), src=root)

        def local_search(queries: np.ndarray, count: int, wanted: int, expansion: int):
            if rank == failing["rank"]:
                raise MemoryError("injected: shard 3 lost its device")
            return uneven_shard_results(rank, int(queries.astype(np.int64).sum()))

        communicator = Communicator.on_host(rank, WORLD8, all_gather, broadcast, local_search)
        queries = np.full((Q, STRIDE), 7 if rank == 2 else 1, dtype=np.uint8)  # the batch comes from rank 2 this time
        keys = np.zeros((Q, K), dtype=np.uint64)
        distances = np.zeros((Q, K), dtype=np.float32)
        counts = np.zeros(Q, dtype=np.uint64)
        stats, step = communicator.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, 2, keys.ctypes.data,
                                              distances.ctypes.data, counts.ctypes.data, 0, 0)
        assert step.gathered_bytes == WORLD8 * step.block_bytes and step.exchanges == 1
        parts = [uneven_shard_results(r, Q * STRIDE * 7) for r in range(WORLD8)]
        expected_keys, expected_distances, expected_counts = oracle_merge(parts)
        assert np.array_equal(counts, expected_counts)
        assert (counts < K).any() and (counts == K).any(), "the case is meant to hold both short and full lists"
        for i in range(Q):
            n = int(counts[i])
            assert n == min(K, sum(int(p[2][i]) for p in parts))
            assert np.array_equal(keys[i, :n], expected_keys[i, :n]) and np.array_equal(distances[i, :n], expected_distances[i, :n])
            assert np.all(keys[i, n:] == 0) and np.all(np.isnan(distances[i, n:]))
        # a rank in the MIDDLE fails: all eight leave the step with an error, seven of them naming rank 3; then a clean step
        failing["rank"] = 3
        try:
            communicator.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, 2, keys.ctypes.data, distances.ctypes.data,
                                    counts.ctypes.data, 0, 0)
            raise AssertionError("the step succeeded although rank 3 failed")
        except RuntimeError as error:
            text = str(error)
            assert ("injected" in text) if rank == 3 else (f"aborted by rank 3 of {WORLD8}" in text), text
        failing["rank"] = -1
        communicator.search_raw(None, queries.ctypes.data, Q, STRIDE, K, 64, 2, keys.ctypes.data, distances.ctypes.data,
                                counts.ctypes.data, 0, 0)
        assert np.array_equal(counts, expected_counts)
        results[rank] = True
    finally:
        dist.destroy_process_group()


def test_sharded_step_eight_ranks_gloo_uneven_shards_and_a_failing_middle_rank():
    manager = mp.Manager()
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        results = manager.dict()
        try:
            mp.spawn(worker8, args=(port, results), nprocs=WORLD8, join=True)
            break
        except Exception as error:  # noqa: BLE001
            if attempt or "AssertionError" in str(error):
                raise
            print(f"eight-rank rendezvous failed, trying once more: {str(error)[-2000:]}", flush=True)
    assert dict(results) == {r: True for r in range(WORLD8)}
