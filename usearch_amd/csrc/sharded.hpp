/**
 *  usearch_amd/csrc/sharded.hpp — the communicator and the one-exchange sharded search step (sharded.hip).
 *  Mirrors, across GPUs, what the reference's `Indexes` does across sub-indexes on the CPU
 *  (/root/reference/python/lib.cpp:321-402 with `search_result_t::merge_into`, include/usearch/index.hpp:2650-2670).
 */
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>

#include "engine.hpp"

namespace usearch_amd {

/// Caller-supplied collectives (and, for machines without a device, a stand-in for the local search).
struct transport_t {
    void* context = nullptr;
    /// every rank contributes `bytes` bytes and receives world × bytes in rank order
    const char* (*all_gather)(void* context, const void* send, void* receive, std::size_t bytes, void* stream) = nullptr;
    const char* (*broadcast)(void* context, void* buffer, std::size_t bytes, int root, void* stream) = nullptr;
    int buffers_on_host = 0; ///< 1 = the collectives take host pointers (device blocks are staged through pinned memory)
    /// when set, replaces the device search and moves the whole step into host memory (no HIP call at all)
    const char* (*local_search)(void* context, const void* queries, std::size_t count, std::size_t stride,
                                std::size_t wanted, std::size_t expansion, std::uint64_t* keys, float* distances,
                                std::uint64_t* counts) = nullptr;
};

/// One rank's contribution to the exchange: distances f32[Q][k] | keys u64[Q][k] | counts u64[Q] | flags u64 — byte offsets.
struct block_layout_t {
    std::size_t distances = 0, keys = 0, counts = 0, flags = 0, bytes = 0;
};
block_layout_t block_layout(std::size_t queries, std::size_t wanted);

struct sharded_stats_t {
    std::uint64_t block_bytes = 0;    ///< what this rank sends
    std::uint64_t gathered_bytes = 0; ///< what it receives
    float exchange_ms = 0.f;          ///< all-gather + merge on the stream (HIP events; with `timed`)
    std::uint32_t exchanges = 0;      ///< 1, or 2 when some rank's scratch ladder ran after the first exchange
};

/// The flag word that closes a rank's block. Low half: how many of its queries outgrew their scratch (their results are not in
/// the block yet — every rank then repeats the exchange once the retry ladders have run). Top bit: the rank has FAILED — it
/// entered the collective only to say so, and every rank leaves the step with an error. Bits 32-47: where it failed.
constexpr std::uint64_t flag_abort_k = 1ull << 63;
constexpr std::uint64_t flag_overflow_mask_k = 0xFFFFFFFFull;
enum abort_stage_t : unsigned { stage_broadcast_k = 1, stage_search_k = 2, stage_ladder_k = 3 };

enum transport_kind_t : int { transport_none_k = 0, transport_rccl_k = 1, transport_custom_k = 2 };

class comm_t {
  public:
    comm_t() = default;
    ~comm_t();
    comm_t(const comm_t&) = delete;
    comm_t& operator=(const comm_t&) = delete;

    /// `ncclGetUniqueId`: rank 0 creates it, the launcher carries the 128 bytes to the other ranks.
    static const char* unique_id(void* out_128_bytes);
    /// `ncclCommInitRank` on `device` — collective across all ranks.
    const char* init_rccl(const void* unique_id, int rank, int world, int device);
    const char* init_custom(const transport_t& transport, int rank, int world, int device);

    int rank() const { return rank_; }
    int world() const { return world_; }
    transport_kind_t kind() const { return kind_; }

    /**
     *  One step. `queries` are in the storage kind (device memory; host memory in the no-device mode) and are overwritten
     *  on the other ranks when `broadcast_root >= 0`. keys / distances / counts receive the MERGED results (identical on
     *  every rank); visited / computed this rank's own traversal counters.
     */
    const char* search(snapshot_t* shard, void* queries, std::size_t count, std::size_t stride_bytes, std::size_t wanted,
                       std::size_t expansion, int broadcast_root, std::uint64_t* keys, float* distances,
                       std::uint64_t* counts, std::uint64_t* visited, std::uint64_t* computed, hipStream_t stream,
                       const search_tuning_t& tuning, bool timed, search_stats_t* stats, sharded_stats_t* step);

    /// The collectives alone, for callers that drive their own steps (bench.py's exact ground truth across shards).
    const char* broadcast(void* buffer, std::size_t bytes, int root, hipStream_t stream);

  private:
    const char* reserve(std::size_t block_bytes);
    const char* all_gather(std::size_t bytes, hipStream_t stream);
    const char* abort_message(const std::uint64_t* flags, const char* local) const;
    void abort_transport();

    int rank_ = 0, world_ = 1, device_ = 0;
    transport_kind_t kind_ = transport_none_k;
    bool on_device_ = true;
    bool broken_ = false; ///< a rank-local failure made entering a collective impossible: the transport was aborted
    void* rccl_comm_ = nullptr;
    transport_t transport_{};
    std::mutex mutex_;
    std::uint8_t *d_send_ = nullptr, *d_gathered_ = nullptr;
    std::uint8_t *h_send_ = nullptr, *h_gathered_ = nullptr, *h_flags_ = nullptr;
    std::size_t block_bytes_ = 0;
};

} // namespace usearch_amd
