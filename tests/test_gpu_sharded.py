"""GPU: the sharded step end to end with TWO ranks — two processes, one shard each, both on cuda:0 (RCCL refuses two ranks on
one device, so the collectives are gloo's over host memory: `usearch_amd_transport_t::buffers_on_host`; everything else —
broadcast of the batch, the search into the send block, the packed exchange, the overflow flags, the merge kernel — is the
production code of usearch_amd/csrc/sharded.hip). Expectation: every rank's own plain search, gathered and folded with the
oracle's `merge_into` in rank order. The single-rank RCCL communicator is exercised on the way."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

WORLD = 2
CASES = [("hamming", "b1", 128, 30000, 400, 10, 64), ("cos", "f16", 96, 20000, 300, 10, 96)]


def worker(rank: int, port: int, results, world: int = WORLD, transport: str = "host"):
    """`transport` "host": every rank on cuda:0, the collectives gloo's over host memory. "rccl": rank r on cuda:r, the native
    `ncclBroadcast` / `ncclAllGather` of csrc/sharded.hip (gloo only carries the unique id and the expectation)."""
    WORLD = world  # noqa: N806 — the body below reads the world size under this name
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # both ranks are here: never the interface the hostname resolves to
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC, or RCCL cannot share buffers between the processes
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import bench
        import usearch_amd
        from oracle import oraclebind
        from usearch_amd.sharded import Communicator, ShardedSearcher
        ordinal = rank if transport == "rccl" else 0
        device = torch.device("cuda", ordinal)
        torch.cuda.set_device(ordinal)

        def all_gather(send: np.ndarray, receive: np.ndarray):
            dist.all_gather_into_tensor(torch.from_numpy(receive), torch.from_numpy(send))

        def broadcast(buffer: np.ndarray, root: int):
            dist.broadcast(torch.from_numpy(buffer), src=root)

        def share_id(unique):
            box = [unique]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        if transport == "rccl":
            communicator = Communicator.rccl(rank, WORLD, ordinal, share_id)
            assert communicator.kind == "rccl-native" and communicator.world == WORLD and communicator.rank == rank
        else:
            communicator = Communicator.over_host_collectives(rank, WORLD, 0, all_gather, broadcast)
        for metric, dtype, dim, n, q, k, expansion in CASES:
            data = bench.synthetic_vectors_device(n, dim, dtype, 100 + rank, device)
            keys = np.arange(n, dtype=np.uint64) + rank * n
            built = usearch_amd.build(None, metric, dtype, keys=keys, device=ordinal, device_pointer=data.data_ptr(), count=n,
                                      stride=data.stride(0), ndim=dim)
            index = built.index
            queries = bench.synthetic_vectors_device(q, dim, dtype, 7 if rank == 0 else 8, device)  # rank 0's batch wins
            searcher = ShardedSearcher(index, communicator)
            merged_keys, merged_distances, merged_counts, stats = searcher.search(queries, k, expansion, broadcast_from=0)
            torch.cuda.synchronize()
            assert searcher.last_step.exchanges == 1 and searcher.last_step.gathered_bytes == WORLD * searcher.last_step.block_bytes
            # expectation: plain searches of the (now broadcast) batch on every shard, folded in rank order
            host_queries = queries.cpu().numpy().view(bench.NUMPY_STORAGE[dtype])
            local = index.search(host_queries, k, expansion=expansion, dtype=dtype)
            gathered = [None] * WORLD
            dist.all_gather_object(gathered, (local.keys, local.distances, local.counts))
            expected_keys = np.zeros((q, k), dtype=np.uint64)
            expected_distances = np.zeros((q, k), dtype=np.float32)
            for i in range(q):
                merged = 0
                for shard_keys, shard_distances, shard_counts in gathered:
                    count = int(shard_counts[i])
                    merged = oraclebind.merge_into(expected_keys[i], expected_distances[i], merged, shard_keys[i, :count],
                                                   shard_distances[i, :count], count)
                assert merged == int(merged_counts[i])
            assert np.array_equal(merged_keys.cpu().numpy().astype(np.uint64), expected_keys), (metric, dtype)
            assert np.array_equal(merged_distances.cpu().numpy().view(np.uint32), expected_distances.view(np.uint32))
            # a tiny visited set forces the scratch ladder on both ranks: the first exchange carried incomplete blocks, the
            # flags say so, every rank repeats the exchange — same answer
            tight = usearch_amd.Tuning(hash_cap=64, mode=2)
            again_keys, again_distances, _, _ = searcher.search(queries, k, expansion, broadcast_from=0, tuning=tight)
            torch.cuda.synchronize()
            assert searcher.last_step.exchanges == 2
            assert torch.equal(again_keys, merged_keys) and torch.equal(again_distances, merged_distances)
        results[rank] = True
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_shard_each_through_the_native_step():
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
            # a rendezvous that did not come up (the port was taken between the probe and the listen, a child that died while
            # starting) gets ONE more try on a fresh port; anything a worker asserted is a result and stays fatal
            if attempt or "AssertionError" in str(error):
                raise
            print(f"two-rank rendezvous failed, trying once more: {str(error)[-2000:]}", flush=True)
    assert dict(results) == {0: True, 1: True}


def test_rccl_transport_across_every_visible_device():
    """The production transport with more than one rank — `ncclCommInitRank`, `ncclBroadcast` of the batch, ONE `ncclAllGather` of
    the packed blocks, the merge kernel — one process per device over however many devices this box shows (up to 8), held to the
    same expectation as the two-rank host-collective test (every rank's plain search folded by the oracle's `merge_into`) and to
    the repeated exchange after a scratch overflow. Skipped on the one-GPU boxes of the test pool; wherever a node with more is
    leased, the first multi-rank `ncclAllGather` of this code is a test and not a benchmark (python/lib.cpp:321-402)."""
    devices = torch.cuda.device_count()
    if devices < 2:
        pytest.skip(f"{devices} device visible: RCCL refuses two ranks on one device (the two-rank test above runs the same step "
                    "over host collectives)")
    world = min(devices, 8)
    manager = mp.Manager()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    results = manager.dict()
    mp.spawn(worker, args=(port, results, world, "rccl"), nprocs=world, join=True)
    assert dict(results) == {rank: True for rank in range(world)}


def test_single_rank_rccl_communicator():
    """`ncclCommInitRank` with one rank, the packed block and the merge kernel on one stream: what `bench.py --sharded` runs."""
    import bench
    import usearch_amd
    from usearch_amd.sharded import Communicator, ShardedSearcher
    device = torch.device("cuda", 0)
    communicator = Communicator.rccl(0, 1, 0, lambda unique: unique)
    assert communicator.kind == "rccl-native" and communicator.world == 1
    data = bench.synthetic_vectors_device(20000, 128, "b1", 5, device)
    built = usearch_amd.build(None, "hamming", "b1", device=0, device_pointer=data.data_ptr(), count=20000,
                              stride=data.stride(0), ndim=128)
    queries = bench.synthetic_vectors_device(500, 128, "b1", 6, device)
    keys, distances, counts, stats = ShardedSearcher(built.index, communicator).search(queries, 10, 64)
    torch.cuda.synchronize()
    plain = built.index.search(queries.cpu().numpy(), 10, expansion=64, dtype="b1")
    # one shard folded into an empty buffer by merge_into: equal distances come out in reverse order, the distances themselves
    # and the sets of keys per distance are the plain search's
    assert np.array_equal(distances.cpu().numpy().view(np.uint32), plain.distances.view(np.uint32))
    assert np.array_equal(np.sort(keys.cpu().numpy().astype(np.uint64), axis=1), np.sort(plain.keys, axis=1))
