"""Sharded search across the GPUs of one node: one process per GPU, one HNSW per shard, one exchange step.

What the reference does on the CPU with `Indexes` (python/usearch/index.py:1473-1514 → python/lib.cpp:321-402: every shard
searches every query, per-query results are folded with `search_result_t::merge_into`, index.hpp:2650-2670) is ONE native
call here, `usearch_amd_sharded_search_many` (usearch_amd/csrc/sharded.hip): on every rank, on one stream, with one wait,

    [broadcast the batch]  →  search this rank's shard, results written straight into the send block
    →  ONE all-gather of the packed block {distances | keys | counts}  →  merge kernel (rank order, `merge_into` tie rule).

This module is the thin host mirror: it creates the communicator (RCCL over xGMI; the 128-byte unique id travels through
whatever process group the launcher set up) and forwards device pointers. `torch.distributed` is plumbing — rendezvous
and, when RCCL cannot be initialised natively, a fallback transport whose collectives are handed to the SAME native step as
callbacks. A third transport runs the step in host memory with a caller-supplied stand-in for the device search, which is
how the protocol is covered on machines without a GPU (tests/test_sharded_gloo.py).

Message per rank = Q·k·12 + Q·8 + 8 bytes, e.g. 100 k queries × k = 10 → 12.8 MB, gathered 102 MB on 8 GPUs: latency-, not
bandwidth-bound on xGMI, hence ONE collective per batch, no bucketing.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import index as binding

# the callbacks return `char const*`: NULL, or the address of a message that stays alive (see `_message`)
ALL_GATHER_T = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
BROADCAST_T = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
LOCAL_SEARCH_T = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                             C.c_void_p, C.c_void_p, C.c_void_p)
_messages = []


def _message(error: Exception) -> int:
    """An error string the native side may read after the callback has returned."""
    buffer = C.create_string_buffer(f"{type(error).__name__}: {error}".encode()[:240])
    _messages.append(buffer)
    del _messages[:-16]
    return C.addressof(buffer)


class Transport(C.Structure):
    """`usearch_amd_transport_t`."""
    _fields_ = [("context", C.c_void_p), ("all_gather", ALL_GATHER_T), ("broadcast", BROADCAST_T),
                ("buffers_on_host", C.c_int), ("local_search", LOCAL_SEARCH_T)]


class ShardedStats(C.Structure):
    """`usearch_amd_sharded_stats_t`."""
    _fields_ = [("block_bytes", C.c_uint64), ("gathered_bytes", C.c_uint64), ("exchange_ms", C.c_float),
                ("exchanges", C.c_uint32)]


def _bind(library: C.CDLL) -> C.CDLL:
    if getattr(library, "_sharded_bound", False):
        return library
    err_p = C.POINTER(C.c_char_p)
    library.usearch_amd_comm_unique_id.argtypes = [C.c_void_p, err_p]
    library.usearch_amd_comm_init_rccl.restype = C.c_void_p
    library.usearch_amd_comm_init_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, err_p]
    library.usearch_amd_comm_init_custom.restype = C.c_void_p
    library.usearch_amd_comm_init_custom.argtypes = [C.POINTER(Transport), C.c_int, C.c_int, C.c_int, err_p]
    library.usearch_amd_comm_free.argtypes = [C.c_void_p]
    library.usearch_amd_comm_rank.argtypes = [C.c_void_p]
    library.usearch_amd_comm_world.argtypes = [C.c_void_p]
    library.usearch_amd_comm_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, err_p]
    library.usearch_amd_sharded_search_many.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(binding.Tuning), C.c_int,
        C.POINTER(binding.Stats), C.POINTER(ShardedStats), err_p]
    library._sharded_bound = True
    return library


def _raise(err: C.c_char_p, where: str) -> None:
    if err.value:
        raise RuntimeError(f"{where}: {err.value.decode()}")


class Communicator:
    """Owns a `usearch_amd_comm_t`. Build one with `Communicator.rccl(...)`, `.over_torch(...)` or `.on_host(...)`."""

    def __init__(self, handle: int, kind: str, keep=()):
        self._handle = handle
        self.kind = kind
        self._keep = keep  # ctypes callbacks and their structure must outlive the native object

    def __del__(self):
        try:
            if self._handle:
                _bind(binding.library()).usearch_amd_comm_free(self._handle)
                self._handle = None
        except Exception:
            pass

    @property
    def rank(self) -> int:
        return int(_bind(binding.library()).usearch_amd_comm_rank(self._handle))

    @property
    def world(self) -> int:
        return int(_bind(binding.library()).usearch_amd_comm_world(self._handle))

    # ---- RCCL over xGMI, natively: the production transport
    @classmethod
    def rccl(cls, rank: int, world: int, device: int, share_id: Callable[[Optional[bytes]], bytes]) -> "Communicator":
        """`share_id(bytes on rank 0 / None elsewhere) -> the 128 bytes on every rank`: the launcher's side channel."""
        library = _bind(binding.library())
        err = C.c_char_p()
        unique, failure = None, None
        if rank == 0:
            block = (C.c_uint8 * 128)()
            library.usearch_amd_comm_unique_id(block, C.byref(err))
            if err.value:  # tell the other ranks instead of leaving them in the side channel
                failure = err.value.decode()
            else:
                unique = bytes(block)
        unique = share_id(unique)
        if not unique or len(unique) != 128:
            raise RuntimeError(f"usearch_amd_comm_unique_id: {failure or 'rank 0 could not create the RCCL unique id'}")
        err = C.c_char_p()
        block = (C.c_uint8 * 128).from_buffer_copy(unique)
        handle = library.usearch_amd_comm_init_rccl(block, rank, world, device, C.byref(err))
        _raise(err, "usearch_amd_comm_init_rccl")
        return cls(handle, "rccl-native")

    # ---- the caller's collectives on DEVICE buffers (torch.distributed with the nccl = RCCL backend): fallback
    @classmethod
    def over_torch(cls, rank: int, world: int, device: int, group=None) -> "Communicator":
        import torch
        import torch.distributed as dist

        def as_tensor(pointer: int, nbytes: int):
            # a uint8 view of raw device memory; torch only carries it to the collective
            holder = {"data": (pointer, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

            class Raw:
                __cuda_array_interface__ = holder
            return torch.as_tensor(Raw(), device=torch.device("cuda", device))

        def all_gather(_context, send, receive, nbytes, stream):
            try:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0, device=torch.device("cuda", device))):
                    dist.all_gather_into_tensor(as_tensor(receive, nbytes * world), as_tensor(send, nbytes), group=group)
                return None
            except Exception as error:  # surfaced through the C ABI as an error string
                return _message(error)

        def broadcast(_context, buffer, nbytes, root, stream):
            try:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0, device=torch.device("cuda", device))):
                    dist.broadcast(as_tensor(buffer, nbytes), src=root, group=group)
                return None
            except Exception as error:
                return _message(error)

        callbacks = (ALL_GATHER_T(all_gather), BROADCAST_T(broadcast))
        transport = Transport(None, callbacks[0], callbacks[1], 0, LOCAL_SEARCH_T())
        err = C.c_char_p()
        handle = _bind(binding.library()).usearch_amd_comm_init_custom(C.byref(transport), rank, world, device, C.byref(err))
        _raise(err, "usearch_amd_comm_init_custom")
        return cls(handle, "torch-distributed", keep=(callbacks, transport))

    # ---- the device step with HOST-side collectives (gloo, MPI …): blocks are staged through pinned memory
    @classmethod
    def over_host_collectives(cls, rank: int, world: int, device: int,
                              all_gather: Callable[[np.ndarray, np.ndarray], None],
                              broadcast: Optional[Callable[[np.ndarray, int], None]]) -> "Communicator":
        """`all_gather(send u8[n], receive u8[world·n])`, `broadcast(buffer u8[n], root)` over host memory; the search, the
        packing and the merge run on the device (`usearch_amd_transport_t::buffers_on_host = 1`)."""

        def view(pointer: int, nbytes: int) -> np.ndarray:
            return np.ctypeslib.as_array(C.cast(pointer, C.POINTER(C.c_uint8)), shape=(nbytes,))

        def gather_callback(_context, send, receive, nbytes, _stream):
            try:
                all_gather(view(send, nbytes), view(receive, nbytes * world))
                return None
            except Exception as error:
                return _message(error)

        def broadcast_callback(_context, buffer, nbytes, root, _stream):
            try:
                broadcast(view(buffer, nbytes), root)
                return None
            except Exception as error:
                return _message(error)

        callbacks = (ALL_GATHER_T(gather_callback), BROADCAST_T(broadcast_callback) if broadcast else BROADCAST_T())
        transport = Transport(None, callbacks[0], callbacks[1], 1, LOCAL_SEARCH_T())
        err = C.c_char_p()
        handle = _bind(binding.library()).usearch_amd_comm_init_custom(C.byref(transport), rank, world, device, C.byref(err))
        _raise(err, "usearch_amd_comm_init_custom")
        return cls(handle, "host-collectives", keep=(callbacks, transport))

    # ---- everything in host memory, the device search replaced by `local_search`: the protocol without a GPU
    @classmethod
    def on_host(cls, rank: int, world: int, all_gather: Callable[[np.ndarray, np.ndarray], None],
                broadcast: Optional[Callable[[np.ndarray, int], None]],
                local_search: Callable[[np.ndarray, int, int, int], tuple]) -> "Communicator":
        """`all_gather(send u8[n], receive u8[world·n])`, `broadcast(buffer u8[n], root)`,
        `local_search(queries u8[Q, stride], count, wanted, expansion) -> (keys u64[Q,k], distances f32[Q,k], counts u64[Q])`."""

        def view(pointer: int, nbytes: int) -> np.ndarray:
            return np.ctypeslib.as_array(C.cast(pointer, C.POINTER(C.c_uint8)), shape=(nbytes,))

        def gather_callback(_context, send, receive, nbytes, _stream):
            try:
                all_gather(view(send, nbytes), view(receive, nbytes * world))
                return None
            except Exception as error:
                return _message(error)

        def broadcast_callback(_context, buffer, nbytes, root, _stream):
            try:
                broadcast(view(buffer, nbytes), root)
                return None
            except Exception as error:
                return _message(error)

        def search_callback(_context, queries, count, stride, wanted, expansion, keys, distances, counts):
            try:
                found = local_search(view(queries, count * stride).reshape(count, stride), count, wanted, expansion)
                view(keys, count * wanted * 8)[:] = np.ascontiguousarray(found[0], dtype=np.uint64).view(np.uint8).ravel()
                view(distances, count * wanted * 4)[:] = np.ascontiguousarray(found[1], dtype=np.float32).view(np.uint8).ravel()
                view(counts, count * 8)[:] = np.ascontiguousarray(found[2], dtype=np.uint64).view(np.uint8).ravel()
                return None
            except Exception as error:
                return _message(error)

        callbacks = (ALL_GATHER_T(gather_callback), BROADCAST_T(broadcast_callback) if broadcast else BROADCAST_T(),
                     LOCAL_SEARCH_T(search_callback))
        transport = Transport(None, callbacks[0], callbacks[1], 1, callbacks[2])
        err = C.c_char_p()
        handle = _bind(binding.library()).usearch_amd_comm_init_custom(C.byref(transport), rank, world, 0, C.byref(err))
        _raise(err, "usearch_amd_comm_init_custom")
        return cls(handle, "host", keep=(callbacks, transport))

    def broadcast_device(self, pointer: int, nbytes: int, root: int, stream: int = 0) -> None:
        err = C.c_char_p()
        _bind(binding.library()).usearch_amd_comm_broadcast(self._handle, C.c_void_p(pointer), nbytes, root,
                                                            C.c_void_p(stream), C.byref(err))
        _raise(err, "usearch_amd_comm_broadcast")

    # ---- the step
    def search_raw(self, snapshot_handle, queries_ptr: int, count: int, stride: int, wanted: int, expansion: int,
                   broadcast_root: int, keys_ptr: int, distances_ptr: int, counts_ptr: int, visited_ptr: int,
                   computed_ptr: int, stream: int = 0, timed: bool = False,
                   tuning: Optional[binding.Tuning] = None):
        """`usearch_amd_sharded_search_many` on raw addresses (device memory, or host memory for `on_host`)."""
        stats, step, err = binding.Stats(), ShardedStats(), C.c_char_p()
        _bind(binding.library()).usearch_amd_sharded_search_many(
            snapshot_handle, self._handle, C.c_void_p(queries_ptr), count, stride, wanted, expansion, broadcast_root,
            C.c_void_p(keys_ptr), C.c_void_p(distances_ptr), C.c_void_p(counts_ptr), C.c_void_p(visited_ptr),
            C.c_void_p(computed_ptr), C.c_void_p(stream), C.byref(tuning) if tuning is not None else None, int(timed),
            C.byref(stats), C.byref(step), C.byref(err))
        _raise(err, "usearch_amd_sharded_search_many")
        return stats, step


class ShardedSearcher:
    """This rank's shard (`usearch_amd.Index`) + a communicator. `search` runs one step on torch CUDA tensors."""

    def __init__(self, index, communicator: Communicator, stream: int = 0):
        self.index = index
        self.communicator = communicator
        self.stream = stream
        self.last_visited = None
        self.last_computed = None
        self.last_step = None

    def search(self, queries, k: int, expansion: int = 0, broadcast_from: Optional[int] = 0, timed: bool = False,
               tuning: Optional[binding.Tuning] = None, out=None):
        """`queries`: uint8 CUDA tensor [Q, bytes] in the storage kind (every rank passes one; with `broadcast_from` set that
        rank's content wins). → (keys i64[Q,k], distances f32[Q,k], counts i64[Q]) merged over all shards, + engine stats.
        `out` = (keys, distances, counts, visited, computed) tensors to fill instead of fresh ones."""
        import torch
        q = queries.shape[0]
        if out is None:
            out = (torch.empty((q, k), dtype=torch.int64, device=queries.device),
                   torch.empty((q, k), dtype=torch.float32, device=queries.device),
                   torch.empty(q, dtype=torch.int64, device=queries.device),
                   torch.empty(q, dtype=torch.int64, device=queries.device),
                   torch.empty(q, dtype=torch.int64, device=queries.device))
            torch.cuda.current_stream(queries.device).synchronize()  # they exist before our stream touches them
        keys, distances, counts, visited, computed = out
        stats, step = self.communicator.search_raw(
            self.index._handle, queries.data_ptr(), q, queries.stride(0) * queries.element_size(), k, expansion,
            -1 if broadcast_from is None else broadcast_from, keys.data_ptr(), distances.data_ptr(), counts.data_ptr(),
            visited.data_ptr(), computed.data_ptr(), self.stream, timed, tuning)
        self.last_visited, self.last_computed, self.last_step = visited, computed, step
        return keys, distances, counts, stats


def gpu_searcher(index, rank: int, world: int, device: int, stream: int = 0, group=None,
                 prefer: str = "rccl") -> ShardedSearcher:
    """Production binding: native RCCL when it initialises, else the torch.distributed transport — say which in the logs
    (`ShardedSearcher.communicator.kind`). The unique id travels over the launcher's process group."""
    import torch.distributed as dist

    def share_id(unique: Optional[bytes]) -> bytes:
        if world == 1:
            return unique
        box = [unique]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]

    communicator = None
    if prefer == "gloo":  # host-side collectives of the launcher's (gloo) group: ranks may share a device — rehearsals, tests
        import torch

        def all_gather(send: np.ndarray, receive: np.ndarray):
            dist.all_gather_into_tensor(torch.from_numpy(receive), torch.from_numpy(send), group=group)

        def broadcast(buffer: np.ndarray, root: int):
            dist.broadcast(torch.from_numpy(buffer), src=root, group=group)

        return ShardedSearcher(index, Communicator.over_host_collectives(rank, world, device, all_gather, broadcast), stream)

    def agree(ok: bool) -> bool:
        """Every rank learns whether EVERY rank said yes: the choice of transport is collective, a rank that fell back on its
        own would sit in torch.distributed's all-gather while its peers sit in RCCL's."""
        if world == 1:
            return ok
        votes = [None] * world
        dist.all_gather_object(votes, bool(ok), group=group)
        return all(votes)

    if prefer == "rccl":
        import sys
        # 1. can every rank reach RCCL at all (the library resolves at run time)? Asked BEFORE the collective initialisation,
        #    which a rank without the library would never enter
        available, why = True, ""
        try:
            probe, err = (C.c_uint8 * 128)(), C.c_char_p()
            _bind(binding.library()).usearch_amd_comm_unique_id(probe, C.byref(err))
            available, why = not err.value, (err.value or b"").decode()
        except Exception as error:  # noqa: BLE001 — the binding itself is missing: same answer
            available, why = False, str(error)
        if not agree(available):
            print(f"[usearch_amd] native RCCL unavailable on some rank (rank {rank}: {why or 'ok'}); every rank uses "
                  f"torch.distributed", file=sys.stderr)
        else:
            # 2. the initialisation itself, then agreement again: one rank's failure moves everyone to the fallback
            try:
                communicator = Communicator.rccl(rank, world, device, share_id)
            except RuntimeError as error:
                print(f"[usearch_amd] native RCCL failed to initialise on rank {rank} ({error})", file=sys.stderr)
            if not agree(communicator is not None):
                communicator = None
                print(f"[usearch_amd] rank {rank}: every rank uses torch.distributed", file=sys.stderr)
    if communicator is None:
        communicator = Communicator.over_torch(rank, world, device, group)
    return ShardedSearcher(index, communicator, stream)
