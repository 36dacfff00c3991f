/**
 *  usearch_amd/csrc/merge.hip — per-query merge of per-shard top-k lists (the exchange step of sharded search).
 *
 *  Reproduces `search_result_t::merge_into` (/root/reference/include/usearch/index.hpp:2650-2670) applied to shards
 *  0, 1, …, P-1 in that order, as the reference's `Indexes` does per query (python/lib.cpp:321-402; there the shard order
 *  is whatever its dynamic executor yields — here it is fixed to the rank order, documented in DESIGN.md §6).
 *  `merge_into` inserts every incoming element at `lower_bound(distance)`, i.e. BEFORE all equal distances already
 *  merged, and a full buffer drops its last element. Unrolled over all shards that is a plain top-k under the total
 *  order  (distance ↑, shard ↓, position-within-shard ↓) — which is what this kernel ranks by, one wave per query.
 */
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "engine.hpp"

namespace usearch_amd {

__global__ __launch_bounds__(64) void merge_kernel(const float* distances, const std::uint64_t* keys,
                                                   const std::uint64_t* counts, std::uint32_t shards,
                                                   std::uint32_t queries, std::uint32_t wanted,
                                                   std::uint32_t later_position_first, float* out_distances,
                                                   std::uint64_t* out_keys, std::uint64_t* out_counts) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    const std::uint32_t q = blockIdx.x, lane = threadIdx.x;
    const std::uint32_t total = shards * wanted;
    float* pool_d = reinterpret_cast<float*>(lds); // [shards][wanted]
    std::uint32_t* pool_valid = reinterpret_cast<std::uint32_t*>(pool_d + total);

    std::uint32_t available = 0;
    for (std::uint32_t shard = 0; shard < shards; ++shard) {
        const std::uint64_t count = counts[(std::uint64_t)shard * queries + q];
        available += (std::uint32_t)(count < wanted ? count : wanted);
    }
    for (std::uint32_t i = lane; i < total; i += 64) {
        const std::uint32_t shard = i / wanted, position = i % wanted;
        const bool valid = position < counts[(std::uint64_t)shard * queries + q];
        pool_valid[i] = valid;
        pool_d[i] = valid ? distances[((std::uint64_t)shard * queries + q) * wanted + position] : 0.f;
    }
    __syncthreads();
    const std::uint32_t found = available < wanted ? available : wanted;
    for (std::uint32_t i = lane; i < total; i += 64) {
        if (!pool_valid[i])
            continue;
        const float mine = pool_d[i];
        std::uint32_t rank = 0; // how many candidates precede this one
        for (std::uint32_t j = 0; j < total; ++j) {
            const float other = pool_d[j];
            // ties: a later shard always goes first; inside one shard the later position does under `merge_into`, the
            // earlier one when folding the slot-ordered partitions of an exact search
            const bool same_shard = j / wanted == i / wanted;
            const bool tie_wins = same_shard && !later_position_first ? j < i : j > i;
            rank += pool_valid[j] && (other < mine || (other == mine && tie_wins));
        }
        if (rank < wanted) {
            const std::uint32_t shard = i / wanted, position = i % wanted;
            out_distances[(std::uint64_t)q * wanted + rank] = mine;
            out_keys[(std::uint64_t)q * wanted + rank] = keys[((std::uint64_t)shard * queries + q) * wanted + position];
        }
    }
    for (std::uint32_t i = found + lane; i < wanted; i += 64) { // padding of index.hpp:2707-2722
        out_keys[(std::uint64_t)q * wanted + i] = 0;
        reinterpret_cast<std::uint32_t*>(out_distances)[(std::uint64_t)q * wanted + i] = signaling_nan_bits_k;
    }
    if (lane == 0)
        out_counts[q] = found;
}

const char* merge_shards_device(const float* distances, const std::uint64_t* keys, const std::uint64_t* counts,
                                std::size_t shards, std::size_t queries, std::size_t wanted, float* out_distances,
                                std::uint64_t* out_keys, std::uint64_t* out_counts, hipStream_t stream,
                                bool later_position_first) {
    if (!queries || !wanted || !shards)
        return nullptr;
    const std::size_t lds = shards * wanted * 8;
    if (lds > 64 * 1024)
        return "Too many candidates per query for the merge kernel";
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)queries), dim3(64), lds, stream, distances, keys, counts,
                       (std::uint32_t)shards, (std::uint32_t)queries, (std::uint32_t)wanted,
                       later_position_first ? 1u : 0u, out_distances, out_keys, out_counts);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipStreamSynchronize(stream);
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

} // namespace usearch_amd
