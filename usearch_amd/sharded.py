"""Sharded search across the GPUs of one node: one process per GPU, one HNSW per shard, one exchange step.

What the reference does on the CPU with `Indexes` (python/usearch/index.py:1473-1514 → python/lib.cpp:321-402: every shard
searches every query, per-query results are folded with `search_result_t::merge_into`, index.hpp:2650-2670) becomes:

    broadcast the batch (rank 0 → all)  →  local `usearch_amd_search_many_device` on this rank's shard
    →  all-gather of (distances f32, keys u64, counts u64)[Q][k] over RCCL  →  `usearch_amd_merge_many_device`

with shards merged in RANK ORDER (the reference's order is whatever its dynamic executor produces; ties between shards
are therefore only defined here). Message per rank = Q·k·12 bytes (+ Q·8), e.g. 100 k queries × k = 10 → 12.8 MB, gathered
96-102 MB on 8 GPUs: latency-, not bandwidth-bound on xGMI, hence ONE collective per batch per tensor, no bucketing.

The class is transport-agnostic on purpose: `local_search` and `merge` are injected callables so that the protocol is
covered on CPU with the `gloo` backend (tests/test_sharded_gloo.py) while production binds them to the GPU engine.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

SearchFn = Callable[[torch.Tensor, int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]
MergeFn = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]


class ShardedSearcher:
    """`local_search(queries, k, expansion) -> (keys i64[Q,k], distances f32[Q,k], counts i64[Q])` on this rank's shard;
    `merge(distances[P,Q,k], keys[P,Q,k], counts[P,Q]) -> (keys[Q,k], distances[Q,k], counts[Q])`."""

    def __init__(self, local_search: SearchFn, merge: MergeFn, group: Optional[dist.ProcessGroup] = None):
        self.local_search = local_search
        self.merge = merge
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def search(self, queries: torch.Tensor, k: int, expansion: int = 0, broadcast_from: Optional[int] = 0):
        """Every rank passes a tensor of the batch's shape; with `broadcast_from` set, that rank's content wins."""
        if self.world > 1 and broadcast_from is not None:
            dist.broadcast(queries, src=broadcast_from, group=self.group)
        keys, distances, counts = self.local_search(queries, k, expansion)
        if self.world == 1:
            return keys, distances, counts
        def gather(tensor: torch.Tensor) -> torch.Tensor:
            flat = tensor.contiguous().view(-1)  # flat in, flat out: the one form every backend agrees on
            out = torch.empty(self.world * flat.numel(), dtype=flat.dtype, device=flat.device)
            dist.all_gather_into_tensor(out, flat, group=self.group)
            return out.view((self.world,) + tuple(tensor.shape))

        all_distances, all_keys, all_counts = gather(distances), gather(keys), gather(counts)
        return self.merge(all_distances, all_keys, all_counts)


def gpu_searcher(index, group: Optional[dist.ProcessGroup] = None, stream: int = 0) -> ShardedSearcher:
    """Binds the protocol to the MI355X engine: `index` is this rank's `usearch_amd.Index` (its shard)."""
    from . import index as binding

    def local_search(queries: torch.Tensor, k: int, expansion: int):
        q = queries.shape[0]
        keys = torch.empty((q, k), dtype=torch.int64, device=queries.device)
        distances = torch.empty((q, k), dtype=torch.float32, device=queries.device)
        counts = torch.empty(q, dtype=torch.int64, device=queries.device)
        visited = torch.empty(q, dtype=torch.int64, device=queries.device)
        computed = torch.empty(q, dtype=torch.int64, device=queries.device)
        torch.cuda.current_stream(queries.device).synchronize()
        index.search_device(queries.data_ptr(), q, queries.stride(0) * queries.element_size(), k, expansion,
                            keys.data_ptr(), distances.data_ptr(), counts.data_ptr(), visited.data_ptr(),
                            computed.data_ptr(), stream=stream)
        local_search.last_visited, local_search.last_computed = visited, computed
        return keys, distances, counts

    def merge(all_distances: torch.Tensor, all_keys: torch.Tensor, all_counts: torch.Tensor):
        shards, q, k = all_distances.shape
        keys = torch.empty((q, k), dtype=torch.int64, device=all_keys.device)
        distances = torch.empty((q, k), dtype=torch.float32, device=all_keys.device)
        counts = torch.empty(q, dtype=torch.int64, device=all_keys.device)
        torch.cuda.current_stream(all_keys.device).synchronize()
        binding.merge_many_device(all_distances.data_ptr(), all_keys.data_ptr(), all_counts.data_ptr(), shards, q, k,
                                  distances.data_ptr(), keys.data_ptr(), counts.data_ptr(), stream)
        return keys, distances, counts

    return ShardedSearcher(local_search, merge, group)
