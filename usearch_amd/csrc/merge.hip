/**
 *  usearch_amd/csrc/merge.hip — per-query merge of per-shard top-k lists (the exchange step of sharded search).
 *
 *  Reproduces `search_result_t::merge_into` (/root/reference/include/usearch/index.hpp:2650-2670) applied to shards
 *  0, 1, …, P-1 in that order, as the reference's `Indexes` does per query (python/lib.cpp:321-402; there the shard order
 *  is whatever its dynamic executor yields — here it is fixed to the rank order, documented in DESIGN.md §7).
 *  `merge_into` inserts every incoming element at `lower_bound(distance)`, i.e. BEFORE all equal distances already
 *  merged, and a full buffer drops its last element. Unrolled over all shards that is a plain top-k under the total
 *  order  (distance ↑, shard ↓, position-within-shard ↓) — which is what `merge_rank` computes for one element: every
 *  shard's list is ascending, so the elements of shard t that precede it are a prefix of t, found by binary search
 *  (P·log k steps per element instead of P·k). The same function serves the kernel (lists in LDS) and the host path of
 *  the sharded step (sharded.hip, transports without a device).
 */
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "engine.hpp"
#include "merge_core.hpp"

namespace usearch_amd {

__global__ __launch_bounds__(64) void merge_kernel(const float* distances, const std::uint64_t* keys,
                                                   const std::uint64_t* counts, std::uint64_t distances_stride,
                                                   std::uint64_t keys_stride, std::uint64_t counts_stride,
                                                   std::uint32_t shards, std::uint32_t queries, std::uint32_t wanted,
                                                   std::uint32_t later_position_first, float* out_distances,
                                                   std::uint64_t* out_keys, std::uint64_t* out_counts) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    const std::uint32_t q = blockIdx.x, lane = threadIdx.x;
    const std::uint32_t total = shards * wanted;
    float* pool_d = reinterpret_cast<float*>(lds);                              // [shards][wanted]
    std::uint32_t* pool_count = reinterpret_cast<std::uint32_t*>(pool_d + total); // [shards]

    for (std::uint32_t shard = lane; shard < shards; shard += 64) {
        const std::uint64_t count = counts[(std::uint64_t)shard * counts_stride + q];
        pool_count[shard] = (std::uint32_t)(count < wanted ? count : wanted);
    }
    __syncthreads();
    for (std::uint32_t i = lane; i < total; i += 64) {
        const std::uint32_t shard = i / wanted, position = i % wanted;
        pool_d[i] = position < pool_count[shard]
                        ? distances[(std::uint64_t)shard * distances_stride + (std::uint64_t)q * wanted + position]
                        : 0.f;
    }
    __syncthreads();
    std::uint32_t available = 0;
    for (std::uint32_t shard = 0; shard < shards; ++shard)
        available += pool_count[shard];
    const std::uint32_t found = available < wanted ? available : wanted;
    for (std::uint32_t i = lane; i < total; i += 64) {
        const std::uint32_t shard = i / wanted, position = i % wanted;
        if (position >= pool_count[shard])
            continue;
        const std::uint32_t rank = merge_rank(pool_d, pool_count, shards, wanted, shard, position, later_position_first != 0);
        if (rank < wanted) {
            out_distances[(std::uint64_t)q * wanted + rank] = pool_d[i];
            out_keys[(std::uint64_t)q * wanted + rank] =
                keys[(std::uint64_t)shard * keys_stride + (std::uint64_t)q * wanted + position];
        }
    }
    for (std::uint32_t i = found + lane; i < wanted; i += 64) { // padding of index.hpp:2707-2722
        out_keys[(std::uint64_t)q * wanted + i] = 0;
        reinterpret_cast<std::uint32_t*>(out_distances)[(std::uint64_t)q * wanted + i] = signaling_nan_bits_k;
    }
    if (lane == 0)
        out_counts[q] = found;
}

const char* merge_shards_enqueue(const float* distances, const std::uint64_t* keys, const std::uint64_t* counts,
                                 std::uint64_t distances_stride, std::uint64_t keys_stride, std::uint64_t counts_stride,
                                 std::size_t shards, std::size_t queries, std::size_t wanted, float* out_distances,
                                 std::uint64_t* out_keys, std::uint64_t* out_counts, hipStream_t stream,
                                 bool later_position_first) {
    if (!queries || !wanted || !shards)
        return nullptr;
    const std::size_t lds = shards * wanted * 4 + shards * 4;
    if (lds > 64 * 1024)
        return "Too many candidates per query for the merge kernel";
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)queries), dim3(64), lds, stream, distances, keys, counts,
                       distances_stride, keys_stride, counts_stride, (std::uint32_t)shards, (std::uint32_t)queries,
                       (std::uint32_t)wanted, later_position_first ? 1u : 0u, out_distances, out_keys, out_counts);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

const char* merge_shards_device(const float* distances, const std::uint64_t* keys, const std::uint64_t* counts,
                                std::size_t shards, std::size_t queries, std::size_t wanted, float* out_distances,
                                std::uint64_t* out_keys, std::uint64_t* out_counts, hipStream_t stream,
                                bool later_position_first) {
    if (const char* e = merge_shards_enqueue(distances, keys, counts, (std::uint64_t)queries * wanted,
                                             (std::uint64_t)queries * wanted, queries, shards, queries, wanted,
                                             out_distances, out_keys, out_counts, stream, later_position_first))
        return e;
    const hipError_t e = hipStreamSynchronize(stream);
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

} // namespace usearch_amd
