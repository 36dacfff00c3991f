/**
 *  usearch_amd/csrc/merge_core.hpp — where one element of one shard's result list lands in the merged list.
 *
 *  `search_result_t::merge_into` (/root/reference/include/usearch/index.hpp:2650-2670) over shards 0 … P-1 in order is a
 *  top-k under (distance ↑, shard ↓, position ↓): an incoming element is placed at `lower_bound(distance)`, in front of
 *  every equal distance merged before it. Lists are ascending, so what precedes element (shard, position) in shard t is a
 *  prefix of t's list: `upper_bound` for later shards (their equals go first), `lower_bound` for earlier ones.
 *  Host and device share this one function.
 */
#pragma once
#include <cstdint>

#ifndef __host__
#define __host__
#define __device__
#endif

namespace usearch_amd {

__host__ __device__ inline std::uint32_t merge_lower_bound(const float* list, std::uint32_t count, float d) {
    std::uint32_t lo = 0, hi = count;
    while (lo < hi) {
        const std::uint32_t mid = (lo + hi) / 2;
        if (list[mid] < d)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__host__ __device__ inline std::uint32_t merge_upper_bound(const float* list, std::uint32_t count, float d) {
    std::uint32_t lo = 0, hi = count;
    while (lo < hi) {
        const std::uint32_t mid = (lo + hi) / 2;
        if (d < list[mid])
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

/**
 *  Rank of element `position` of shard `shard` in the merged order. `pool` = [shards][wanted] distances (only the first
 *  `counts[t]` of shard t are valid). `later_position_first` = the `merge_into` rule inside one shard; false = the earlier
 *  position first (folding the slot-ordered partitions of an exact search).
 */
__host__ __device__ inline std::uint32_t merge_rank(const float* pool, const std::uint32_t* counts, std::uint32_t shards,
                                                    std::uint32_t wanted, std::uint32_t shard, std::uint32_t position,
                                                    bool later_position_first) {
    const float d = pool[(std::uint64_t)shard * wanted + position];
    std::uint32_t rank = 0;
    for (std::uint32_t t = 0; t < shards; ++t) {
        const float* list = pool + (std::uint64_t)t * wanted;
        if (t > shard)
            rank += merge_upper_bound(list, counts[t], d);
        else if (t < shard)
            rank += merge_lower_bound(list, counts[t], d);
        else if (later_position_first)
            rank += merge_lower_bound(list, counts[t], d) + (merge_upper_bound(list, counts[t], d) - 1 - position);
        else
            rank += position;
    }
    return rank;
}

} // namespace usearch_amd
