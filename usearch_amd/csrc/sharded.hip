/**
 *  usearch_amd/csrc/sharded.hip — sharded search across the GPUs of one node: one process per GPU, one HNSW per shard,
 *  ONE exchange step per batch.
 *
 *  What the reference does on the CPU with `Indexes` (/root/reference/python/usearch/index.py:1473-1514 →
 *  python/lib.cpp:321-402: every shard searches every query, per-query results are folded with
 *  `search_result_t::merge_into`, include/usearch/index.hpp:2650-2670) becomes, on every rank, on one stream, with one wait:
 *
 *      [broadcast the batch]  →  search this rank's shard, results written straight into the send block
 *      →  ONE all-gather of the packed block { distances f32[Q][k] | keys u64[Q][k] | counts u64[Q] }
 *      →  merge kernel over the P gathered blocks (rank order = merge order, the `merge_into` tie rule)
 *
 *  Transports: RCCL over xGMI (`librccl.so.1`, resolved at run time so that single-GPU users carry no dependency), or
 *  caller-supplied collectives over device or host buffers (MPI, gloo, …). A transport may also replace the device search
 *  by a callback — then the whole step runs in host memory without a single HIP call, which is how the protocol (packing,
 *  exchange, merge order) is exercised on machines without a GPU (tests/test_sharded_gloo.py).
 */
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "engine.hpp"
#include "host_util.hpp"
#include "merge_core.hpp"
#include "sharded.hpp"

namespace usearch_amd {

// ---------------------------------------------------------------------------------------------------------------------
//  RCCL, resolved at run time. Only the handful of entry points the step needs; types as in <rccl/rccl.h>.
// ---------------------------------------------------------------------------------------------------------------------

namespace {

struct rccl_unique_id_t {
    char internal[128];
};
using rccl_comm_t = void*;
constexpr int rccl_uint8_k = 1; // ncclUint8

struct rccl_api_t {
    void* handle = nullptr;
    int (*get_unique_id)(rccl_unique_id_t*) = nullptr;
    int (*comm_init_rank)(rccl_comm_t*, int, rccl_unique_id_t, int) = nullptr;
    int (*comm_destroy)(rccl_comm_t) = nullptr;
    int (*comm_abort)(rccl_comm_t) = nullptr;
    int (*all_gather)(const void*, void*, std::size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*broadcast)(const void*, void*, std::size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
    std::string failure;
};

rccl_api_t& rccl() {
    static rccl_api_t api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already mapped into the process (PyTorch ships one) wins, so that both sides talk to one runtime
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (api.handle)
                break;
        }
        if (!api.handle)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (api.handle)
                    break;
            }
        if (!api.handle) {
            api.failure = std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "librccl.so.1 not found");
            return;
        }
        auto resolve = [&](const char* symbol) -> void* {
            void* address = dlsym(api.handle, symbol);
            if (!address && api.failure.empty())
                api.failure = std::string("RCCL lacks ") + symbol;
            return address;
        };
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(resolve("ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(resolve("ncclCommInitRank"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(resolve("ncclCommDestroy"));
        api.all_gather = reinterpret_cast<decltype(api.all_gather)>(resolve("ncclAllGather"));
        api.broadcast = reinterpret_cast<decltype(api.broadcast)>(resolve("ncclBroadcast"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(resolve("ncclGetErrorString"));
        api.comm_abort = reinterpret_cast<decltype(api.comm_abort)>(dlsym(api.handle, "ncclCommAbort")); // optional
    });
    return api;
}

/// Messages outlive the call that produced them (the C ABI hands out `char const*`): a small per-thread ring.
const char* keep_message(const std::string& text) {
    static thread_local std::string ring[8];
    static thread_local unsigned next = 0;
    std::string& cell = ring[next++ % 8];
    cell = text;
    return cell.c_str();
}

const char* rccl_message(int result) {
    if (!result)
        return nullptr;
    rccl_api_t& api = rccl();
    return keep_message(std::string("RCCL: ") + (api.error_string ? api.error_string(result) : "error"));
}

std::size_t pad8(std::size_t bytes) { return (bytes + 7) & ~(std::size_t)7; }

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
//  The communicator
// ---------------------------------------------------------------------------------------------------------------------

block_layout_t block_layout(std::size_t queries, std::size_t wanted) {
    block_layout_t l;
    l.distances = 0;
    l.keys = pad8(queries * wanted * 4);
    l.counts = l.keys + queries * wanted * 8;
    l.flags = l.counts + queries * 8;
    l.bytes = l.flags + 8;
    return l;
}

comm_t::~comm_t() {
    if (rccl_comm_ && rccl().comm_destroy)
        (void)rccl().comm_destroy(rccl_comm_);
    if (on_device_) {
        if (d_send_)
            (void)hipFree(d_send_);
        if (d_gathered_)
            (void)hipFree(d_gathered_);
        if (h_send_)
            (void)hipHostFree(h_send_);
        if (h_gathered_)
            (void)hipHostFree(h_gathered_);
        if (h_flags_)
            (void)hipHostFree(h_flags_);
    } else {
        std::free(h_send_);
        std::free(h_gathered_);
    }
}

/// Last resort for a rank that cannot enter a collective any more: make the peers' pending collectives fail rather than wait.
void comm_t::abort_transport() {
    if (kind_ == transport_rccl_k && rccl_comm_ && rccl().comm_abort) {
        (void)rccl().comm_abort(rccl_comm_);
        rccl_comm_ = nullptr;
    }
    broken_ = true; // a caller's transport has no such call: its peers time out by its own rules
}

const char* comm_t::unique_id(void* out) {
    rccl_api_t& api = rccl();
    if (!api.failure.empty())
        return keep_message(api.failure);
    rccl_unique_id_t id;
    if (const char* e = rccl_message(api.get_unique_id(&id)))
        return e;
    std::memcpy(out, &id, sizeof(id));
    return nullptr;
}

const char* comm_t::init_rccl(const void* unique_id, int rank, int world, int device) {
    if (world < 1 || rank < 0 || rank >= world)
        return "Rank outside the world";
    rccl_api_t& api = rccl();
    if (!api.failure.empty())
        return keep_message(api.failure);
    UA_HIP(hipSetDevice(device));
    rccl_unique_id_t id;
    std::memcpy(&id, unique_id, sizeof(id));
    if (const char* e = rccl_message(api.comm_init_rank(&rccl_comm_, world, id, rank)))
        return e;
    rank_ = rank, world_ = world, device_ = device, on_device_ = true, kind_ = transport_rccl_k;
    return nullptr;
}

const char* comm_t::init_custom(const transport_t& transport, int rank, int world, int device) {
    if (world < 1 || rank < 0 || rank >= world)
        return "Rank outside the world";
    if (!transport.all_gather)
        return "A transport needs an all-gather";
    transport_ = transport;
    rank_ = rank, world_ = world, device_ = device;
    on_device_ = transport.local_search == nullptr; // a search double moves the whole step into host memory
    kind_ = transport_custom_k;
    return nullptr;
}

const char* comm_t::reserve(std::size_t block_bytes) {
    if (block_bytes <= block_bytes_)
        return nullptr;
    const std::size_t room = std::max<std::size_t>(block_bytes, 4096);
    if (on_device_) {
        UA_HIP(hipSetDevice(device_));
        for (void* p : {(void*)d_send_, (void*)d_gathered_})
            if (p)
                (void)hipFree(p);
        for (void* p : {(void*)h_send_, (void*)h_gathered_})
            if (p)
                (void)hipHostFree(p);
        d_send_ = d_gathered_ = h_send_ = h_gathered_ = nullptr;
        block_bytes_ = 0;
        UA_HIP(hipMalloc((void**)&d_send_, room));
        UA_HIP(hipMalloc((void**)&d_gathered_, room * world_));
        if (!h_flags_)
            UA_HIP(hipHostMalloc((void**)&h_flags_, 8 * ((std::size_t)world_ + 1), hipHostMallocDefault)); // + this rank's staging word
        if (kind_ == transport_custom_k && transport_.buffers_on_host) {
            UA_HIP(hipHostMalloc((void**)&h_send_, room, hipHostMallocDefault));
            UA_HIP(hipHostMalloc((void**)&h_gathered_, room * world_, hipHostMallocDefault));
        }
    } else {
        std::free(h_send_);
        std::free(h_gathered_);
        block_bytes_ = 0;
        h_send_ = static_cast<std::uint8_t*>(std::calloc(room, 1));
        h_gathered_ = static_cast<std::uint8_t*>(std::calloc(room * world_, 1));
        if (!h_send_ || !h_gathered_)
            return "Out of memory";
    }
    block_bytes_ = room;
    return nullptr;
}

const char* comm_t::broadcast(void* buffer, std::size_t bytes, int root, hipStream_t stream) {
    if (world_ == 1 || !bytes)
        return nullptr;
    if (kind_ == transport_rccl_k)
        return rccl_message(rccl().broadcast(buffer, buffer, bytes, rccl_uint8_k, root, rccl_comm_, stream));
    if (!transport_.broadcast)
        return "This transport cannot broadcast: hand every rank the batch";
    if (!on_device_ || !transport_.buffers_on_host)
        return transport_.broadcast(transport_.context, buffer, bytes, root, on_device_ ? (void*)stream : nullptr);
    // device buffer, host transport: stage through a pinned block
    std::uint8_t* staged = nullptr;
    UA_HIP(hipHostMalloc((void**)&staged, bytes, hipHostMallocDefault));
    const char* error = nullptr;
    hipError_t e = hipSuccess;
    if (rank_ == root) {
        e = hipMemcpyAsync(staged, buffer, bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream);
    }
    if (e == hipSuccess)
        error = transport_.broadcast(transport_.context, staged, bytes, root, nullptr);
    if (e == hipSuccess && !error && rank_ != root) {
        e = hipMemcpyAsync(buffer, staged, bytes, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream);
    }
    (void)hipHostFree(staged);
    return e != hipSuccess ? hip_message(e) : error;
}

const char* comm_t::all_gather(std::size_t bytes, hipStream_t stream) {
    if (!on_device_)
        return transport_.all_gather(transport_.context, h_send_, h_gathered_, bytes, nullptr);
    if (kind_ == transport_rccl_k)
        return rccl_message(rccl().all_gather(d_send_, d_gathered_, bytes, rccl_uint8_k, rccl_comm_, stream));
    if (!transport_.buffers_on_host)
        return transport_.all_gather(transport_.context, d_send_, d_gathered_, bytes, (void*)stream);
    UA_HIP(hipMemcpyAsync(h_send_, d_send_, bytes, hipMemcpyDeviceToHost, stream));
    UA_HIP(hipStreamSynchronize(stream));
    if (const char* e = transport_.all_gather(transport_.context, h_send_, h_gathered_, bytes, nullptr))
        return e;
    UA_HIP(hipMemcpyAsync(d_gathered_, h_gathered_, bytes * world_, hipMemcpyHostToDevice, stream));
    return nullptr;
}

/// The merge of merge.hip's kernel, on the host, over the gathered blocks (the no-device mode).
static void merge_blocks_host(const std::uint8_t* gathered, std::size_t block_bytes, const block_layout_t& layout,
                              std::size_t shards, std::size_t queries, std::size_t wanted, std::uint64_t* out_keys,
                              float* out_distances, std::uint64_t* out_counts) {
    std::vector<float> pool(shards * wanted);
    std::vector<std::uint32_t> kept(shards);
    for (std::size_t q = 0; q < queries; ++q) {
        std::size_t available = 0;
        for (std::size_t shard = 0; shard < shards; ++shard) {
            const std::uint8_t* block = gathered + shard * block_bytes;
            std::uint64_t count;
            std::memcpy(&count, block + layout.counts + q * 8, 8);
            kept[shard] = (std::uint32_t)std::min<std::uint64_t>(count, wanted);
            available += kept[shard];
            std::memcpy(pool.data() + shard * wanted, block + layout.distances + q * wanted * 4, wanted * 4);
        }
        const std::size_t found = std::min(available, wanted);
        for (std::size_t shard = 0; shard < shards; ++shard)
            for (std::uint32_t position = 0; position < kept[shard]; ++position) {
                const std::uint32_t rank = merge_rank(pool.data(), kept.data(), (std::uint32_t)shards, (std::uint32_t)wanted,
                                                      (std::uint32_t)shard, position, true);
                if (rank >= wanted)
                    continue;
                out_distances[q * wanted + rank] = pool[shard * wanted + position];
                std::memcpy(out_keys + q * wanted + rank,
                            gathered + shard * block_bytes + layout.keys + (q * wanted + position) * 8, 8);
            }
        for (std::size_t i = found; i < wanted; ++i) { // padding of index.hpp:2707-2722
            out_keys[q * wanted + i] = 0;
            std::memcpy(out_distances + q * wanted + i, &signaling_nan_bits_k, 4);
        }
        out_counts[q] = found;
    }
}

/// Message for a step that some rank aborted: this rank's own failure if it is the one, else who it was and where.
const char* comm_t::abort_message(const std::uint64_t* flags, const char* local) const {
    for (int r = 0; r < world_; ++r)
        if (flags[r] & flag_abort_k) {
            if (r == rank_ && local)
                return local;
            const unsigned stage = (unsigned)((flags[r] >> 32) & 0xFFFFu);
            const char* where = stage == stage_broadcast_k ? "the broadcast of the batch"
                                : stage == stage_search_k  ? "its local search"
                                : stage == stage_ladder_k  ? "the retry of its outgrown queries"
                                                           : "an unknown stage";
            return keep_message("Sharded step aborted by rank " + std::to_string(r) + " of " + std::to_string(world_) + " in " +
                                where + (r == rank_ ? "" : " (that rank reports the cause)"));
        }
    return local;
}

const char* comm_t::search(snapshot_t* shard, void* queries, std::size_t count, std::size_t stride_bytes,
                           std::size_t wanted, std::size_t expansion, int broadcast_root, std::uint64_t* keys,
                           float* distances, std::uint64_t* counts, std::uint64_t* visited, std::uint64_t* computed,
                           hipStream_t stream, const search_tuning_t& tuning, bool timed, search_stats_t* stats,
                           sharded_stats_t* step) {
    if (stats)
        *stats = search_stats_t{};
    if (step)
        *step = sharded_stats_t{};
    if (!count || !wanted)
        return nullptr;
    std::lock_guard<std::mutex> lock(mutex_); // the blocks below are one batch deep
    if (broken_)
        return "This communicator was torn down after a rank could not enter a collective: create a new one";
    const block_layout_t layout = block_layout(count, wanted);
    // The blocks are the one thing a rank cannot fail on quietly: without them it cannot enter the collective at all, so the
    // communicator is torn down (RCCL: ncclCommAbort, the peers' pending collectives fail instead of waiting for ever).
    if (const char* e = reserve(layout.bytes)) {
        abort_transport();
        return e;
    }
    if (step)
        step->block_bytes = layout.bytes, step->gathered_bytes = layout.bytes * world_;

    // A rank-local failure from here on does NOT return before the exchange: the rank still enters the collective, with the
    // abort bit set in its block's flag word, so that every rank leaves the step with an error instead of waiting in
    // ncclAllGather for a peer that has gone home (`local` = this rank's message, `stage` = where it happened).
    const char* local = nullptr;
    unsigned stage = 0;
    auto abort_word = [&]() -> std::uint64_t { return flag_abort_k | ((std::uint64_t)stage << 32); };
    auto any = [&](const std::uint64_t* flags, std::uint64_t mask) {
        bool hit = false;
        for (int r = 0; r < world_; ++r)
            hit |= (flags[r] & mask) != 0;
        return hit;
    };

    // ---- no device: the search is the transport's double, everything lives in host memory
    if (!on_device_) {
        if (broadcast_root >= 0 && world_ > 1) {
            if (!transport_.broadcast)
                local = "This transport cannot broadcast: hand every rank the batch", stage = stage_broadcast_k;
            else if (const char* e = transport_.broadcast(transport_.context, queries, count * stride_bytes, broadcast_root, nullptr))
                local = keep_message(e), stage = stage_broadcast_k;
        }
        if (!local)
            if (const char* e = transport_.local_search(transport_.context, queries, count, stride_bytes, wanted, expansion,
                                                        reinterpret_cast<std::uint64_t*>(h_send_ + layout.keys),
                                                        reinterpret_cast<float*>(h_send_ + layout.distances),
                                                        reinterpret_cast<std::uint64_t*>(h_send_ + layout.counts)))
                local = keep_message(e), stage = stage_search_k;
        const std::uint64_t word = local ? abort_word() : 0;
        std::memcpy(h_send_ + layout.flags, &word, 8);
        if (world_ == 1)
            std::memcpy(h_gathered_, h_send_, layout.bytes);
        else if (const char* e = all_gather(layout.bytes, nullptr))
            return local ? local : e; // the transport itself failed: nothing left to agree over
        std::vector<std::uint64_t> flags((std::size_t)world_);
        for (int r = 0; r < world_; ++r)
            std::memcpy(&flags[(std::size_t)r], h_gathered_ + (std::size_t)r * layout.bytes + layout.flags, 8);
        if (step)
            step->exchanges = 1;
        if (any(flags.data(), flag_abort_k))
            return abort_message(flags.data(), local);
        merge_blocks_host(h_gathered_, layout.bytes, layout, (std::size_t)world_, count, wanted, keys, distances, counts);
        return nullptr;
    }

    // ---- device: one stream, one wait
    if (!shard)
        return "No shard to search"; // a caller's mistake, the same on every rank
    UA_HIP(hipSetDevice(shard->device()));
    if (!stream)
        stream = shard->stream();
    hipEvent_t begin = nullptr, end = nullptr;
    if (timed && step) {
        UA_HIP(hipEventCreate(&begin));
        UA_HIP(hipEventCreate(&end));
    }
    struct events_t {
        hipEvent_t &begin, &end;
        ~events_t() {
            if (begin)
                (void)hipEventDestroy(begin);
            if (end)
                (void)hipEventDestroy(end);
        }
    } events{begin, end};

    if (broadcast_root >= 0)
        if (const char* e = broadcast(queries, count * stride_bytes, broadcast_root, stream))
            local = e, stage = stage_broadcast_k;

    snapshot_t::search_call_t call;
    bool began = false;
    std::uint64_t* send_keys = reinterpret_cast<std::uint64_t*>(d_send_ + layout.keys);
    float* send_distances = reinterpret_cast<float*>(d_send_ + layout.distances);
    std::uint64_t* send_counts = reinterpret_cast<std::uint64_t*>(d_send_ + layout.counts);
    if (!local) {
        if (const char* e = shard->search_begin(call, queries, count, stride_bytes, wanted, expansion, send_keys, send_distances,
                                                send_counts, visited, computed, stream, tuning, timed)) {
            if (call.workspace)
                shard->give_back(call.workspace);
            call.workspace = nullptr;
            local = e, stage = stage_search_k;
        } else {
            began = true;
        }
    }
    // The block's last word: the low half tells the other ranks how many of this rank's queries outgrew their scratch (rare;
    // their results are not in the block yet) — copied device to device from the search's own counter, nobody waits — and the
    // top bit that this rank has failed. A device that cannot even take these copies cannot enter the collective: tear down.
    std::uint64_t* own_word = reinterpret_cast<std::uint64_t*>(h_flags_) + world_; // pinned staging word of this rank
    auto device_failed = [&](hipError_t e) -> const char* {
        if (began)
            (void)shard->search_finish(call, nullptr);
        abort_transport();
        return local ? local : hip_message(e);
    };
    auto post_own_word = [&](std::uint64_t word) -> hipError_t {
        *own_word = word;
        return hipMemcpyAsync(d_send_ + layout.flags, own_word, 8, hipMemcpyHostToDevice, stream);
    };
    if (hipError_t e = post_own_word(local ? abort_word() : 0); e != hipSuccess)
        return device_failed(e);
    if (began && call.workspace && !call.done)
        if (hipError_t e = hipMemcpyAsync(d_send_ + layout.flags, call.workspace->d_queue + 1, 4, hipMemcpyDeviceToDevice, stream);
            e != hipSuccess)
            return device_failed(e);
    const std::uint8_t* gathered = world_ == 1 ? d_send_ : d_gathered_;
    std::uint64_t* flags = reinterpret_cast<std::uint64_t*>(h_flags_);
    // all-gather → merge → every rank's flag word strided out of the gathered blocks. An error in here is the transport's or
    // the device's own: there is nothing left to agree over.
    auto exchange_and_merge = [&]() -> const char* {
        if (begin)
            UA_HIP(hipEventRecord(begin, stream));
        if (world_ > 1) // with one rank there is nothing to exchange; the merge still runs — one code path for any P
            if (const char* e = all_gather(layout.bytes, stream))
                return e;
        if (const char* e = merge_shards_enqueue(reinterpret_cast<const float*>(gathered + layout.distances),
                                                 reinterpret_cast<const std::uint64_t*>(gathered + layout.keys),
                                                 reinterpret_cast<const std::uint64_t*>(gathered + layout.counts),
                                                 layout.bytes / 4, layout.bytes / 8, layout.bytes / 8, (std::size_t)world_,
                                                 count, wanted, distances, keys, counts, stream, true))
            return e;
        if (end)
            UA_HIP(hipEventRecord(end, stream));
        UA_HIP(hipMemcpy2DAsync(flags, 8, gathered + layout.flags, layout.bytes, 8, (std::size_t)world_, hipMemcpyDeviceToHost,
                                stream));
        return nullptr;
    };
    const char* transport_error = exchange_and_merge();
    // ---- the one wait of the step (inside search_finish), plus this rank's scratch ladder if it had overflows
    search_stats_t local_stats;
    const char* finish_error = nullptr;
    if (began)
        finish_error = shard->search_finish(call, &local_stats);
    else if (hipError_t e = hipStreamSynchronize(stream); e != hipSuccess && !transport_error)
        transport_error = hip_message(e);
    if (stats)
        *stats = local_stats;
    if (transport_error)
        return local ? local : transport_error;
    if (step)
        step->exchanges = 1;
    if (any(flags, flag_abort_k))
        return abort_message(flags, local);
    // ---- rare: some rank's block was incomplete. All ranks saw the same flag words, so all of them repeat the exchange; by now
    //      every ladder has run and every block is whole — or that rank's ladder failed, which the second flag word says.
    if (any(flags, flag_overflow_mask_k)) {
        if (step)
            step->exchanges = 2;
        if (finish_error)
            local = finish_error, stage = stage_ladder_k;
        if (hipError_t e = post_own_word(finish_error ? abort_word() : 0); e != hipSuccess) {
            abort_transport();
            return local ? local : hip_message(e);
        }
        if (const char* e = exchange_and_merge())
            return local ? local : e;
        UA_HIP(hipStreamSynchronize(stream));
        if (any(flags, flag_abort_k))
            return abort_message(flags, local);
    } else if (finish_error) {
        return finish_error; // this rank's own trouble after a complete exchange (the peers hold whole results)
    }
    if (step && begin) {
        UA_HIP(hipEventSynchronize(end));
        UA_HIP(hipEventElapsedTime(&step->exchange_ms, begin, end));
    }
    return nullptr;
}

} // namespace usearch_amd
