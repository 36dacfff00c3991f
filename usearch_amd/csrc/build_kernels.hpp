/**
 *  usearch_amd/csrc/build_kernels.hpp — CDNA4 device code of batched HNSW CONSTRUCTION: the two linking steps that
 *  follow the insertion search of every new node (reference `index_gt::add`,
 *  /root/reference/include/usearch/index.hpp:2855-2863):
 *
 *    build_select_kernel    `form_links_to_closest_` (index.hpp:3825-3845): the heuristic `refine_` (index.hpp:4276-4318)
 *                           picks ≤ M of the candidates the insertion search returned; they become the new node's list.
 *                           Every pick also files a reverse-link request in the picked node's inbox.
 *    build_reverse_kernel   `form_reverse_links_` (index.hpp:3848-3893): one wave per node that received requests in
 *                           this batch. Room left → append; otherwise existing ∪ incoming neighbours go through
 *                           `refine_` again with the list capacity as limit.
 *
 *  The insertion search itself (`search_to_insert_`, index.hpp:4011-4079) is the search kernel of kernels.hpp run on
 *  the level being linked (`search_args_t::beam_level`) with the stored vectors as queries.
 *
 *  A whole batch of nodes is inserted against the graph as it stood BEFORE the batch — the device-wide analogue of the
 *  reference's concurrent `add` calls, which do not see each other's half-built nodes either. Lists stay prefix-compact
 *  with `none_slot_k` padding (the snapshot layout of common.hpp), so a finished build IS a searchable snapshot.
 *
 *  Lists of up to 128 cells (base connectivity ≤ 128, connectivity ≤ 64): a list and what is filed against it fit one wave — a
 *  candidate per lane — while capacity + inbox ≤ 64; wider lists go through LDS, up to three candidates per lane.
 *
 *  `refine_` restated for a wave ("forward elimination"): candidates sorted ascending by distance to the centre; the
 *  first live candidate is accepted, then every later live candidate `c` with d(accepted, c) < d(c, centre) is struck
 *  out — the reference rejects exactly those when it reaches them (index.hpp:4297-4304). Same result, but each accepted
 *  node costs one batched row gather instead of one dependent distance per (candidate, accepted) pair.
 */
#pragma once
#include "kernels.hpp"

namespace usearch_amd {

constexpr std::uint32_t build_max_candidates_k = 1024; ///< insertion beam width the link kernels accept (ef_construction)
constexpr std::uint32_t build_max_capacity_k = 128;    ///< widest list the link kernels take (base connectivity)
constexpr std::uint32_t build_wide_inbox_k = 32;       ///< requests per target and round once a list no longer shares a wave with them
constexpr std::uint32_t build_selected_k = 192;        ///< cells of the accepted / staging arrays: capacity + inbox, whole 64s

/// One linking pass: the nodes of one batch that exist on `level`.
struct build_args_t {
    std::uint32_t* nbr0;            ///< mutable aliases of the snapshot's graph arrays
    std::uint32_t* upper;
    const std::uint32_t* upper_ref;
    std::uint32_t level;            ///< level being linked
    std::uint32_t needed;           ///< M: links a new node gets (index.hpp:3832 — `connectivity` on every level)
    std::uint32_t capacity;         ///< list capacity on this level: M0 on level 0, M above (index.hpp:3853)
    const std::uint32_t* nodes;     ///< [count] slots of the new nodes
    std::uint32_t count;
    const std::uint64_t* cand_slots;  ///< [count][ef] insertion-search results (slots, ascending by distance)
    const float* cand_distances;      ///< [count][ef]
    const std::uint64_t* cand_counts; ///< [count]
    std::uint32_t ef;
    std::uint32_t* inbox_count;     ///< [size] reverse-link requests filed against a node in this pass
    cand_t* inbox;                  ///< [size][inbox_cap] {distance, requesting slot}
    std::uint32_t inbox_cap;
    std::uint32_t* touched;         ///< nodes with a non-empty inbox
    std::uint32_t* touched_count;
    unsigned long long* counters;   ///< [0] distances in select, [1] distances in reverse, [2] re-pruned lists, [3] dropped requests
    // requests that found their target's inbox full wait here and are filed again once the reverse kernel has emptied the inboxes
    // (build_refile_kernel): nothing is dropped unless this list overflows too
    std::uint32_t* deferred_targets; ///< [deferred_cap]
    cand_t* deferred_requests;       ///< [deferred_cap] {distance, requesting slot}
    std::uint32_t* deferred_count;   ///< how many are waiting (may exceed the capacity: the excess was dropped and counted)
    std::uint32_t deferred_cap;
    std::uint32_t candidate_cap;     ///< candidates per node the LDS carve-up is cut for: `ef` rounded up to whole 64s (≥ 64)
};

/// Files reverse-link request {`request`} against `target`; false = the inbox is full (the caller defers or drops it).
UA_DEVICE bool build_file_request(const build_args_t& b, std::uint32_t target, cand_t request) {
    const std::uint32_t position = atomicAdd(b.inbox_count + target, 1u);
    if (position == 0)
        b.touched[atomicAdd(b.touched_count, 1u)] = target;
    if (position >= b.inbox_cap)
        return false;
    b.inbox[(std::uint64_t)target * b.inbox_cap + position] = request;
    return true;
}
/// A request that did not fit: parked for the next round, or — the parking lot full — dropped. Returns 1 when dropped.
UA_DEVICE std::uint32_t build_defer_request(const build_args_t& b, std::uint32_t target, cand_t request) {
    const std::uint32_t position = atomicAdd(b.deferred_count, 1u);
    if (position >= b.deferred_cap)
        return 1;
    b.deferred_targets[position] = target;
    b.deferred_requests[position] = request;
    return 0;
}

UA_DEVICE std::uint32_t* build_list(const build_args_t& b, const snapshot_view_t& ix, std::uint32_t slot) {
    return b.level ? b.upper + (std::uint64_t)(b.upper_ref[slot] + (b.level - 1)) * ix.m
                   : b.nbr0 + (std::uint64_t)slot * ix.m0;
}

/// Waves per SIMD the link kernels are cut for: what they had before the candidates' bitmap moved to LDS (one-chunk rows 6, others 4).
constexpr int build_waves(int lanes) { return lanes == 1 ? 6 : 4; }

/// LDS carve-up of the link kernels (after the staged query); `cap` = build_args_t::candidate_cap.
struct build_lds_t {
    std::uint32_t* cand_slots;   // [64] gather list of one measure_rows call
    float* cand_distances;       // [64]
    std::uint32_t* slots;        // [cap] candidates, ascending
    float* dists;                // [cap]
    std::uint32_t* sel;          // [192] accepted (≤ the list capacity); the wide reverse path also stages existing ∪ incoming here
    float* seld;                 // [192]
    std::uint64_t* alive;        // [cap / 64] one bit per candidate `refine_forward` has not struck yet
};
inline __host__ __device__ std::uint32_t build_lds_bytes(std::uint32_t cap) {
    return 64 * 4 * 2 + cap * 4 * 2 + build_selected_k * 4 * 2 + cap / 64 * 8;
}

UA_DEVICE build_lds_t build_lds(std::uint8_t* base, std::uint32_t cap) {
    build_lds_t l;
    l.cand_slots = reinterpret_cast<std::uint32_t*>(base);
    l.cand_distances = reinterpret_cast<float*>(base + 256);
    l.slots = reinterpret_cast<std::uint32_t*>(base + 512);
    l.dists = reinterpret_cast<float*>(base + 512 + cap * 4);
    l.sel = reinterpret_cast<std::uint32_t*>(base + 512 + cap * 8);
    l.seld = reinterpret_cast<float*>(base + 512 + cap * 8 + build_selected_k * 4);
    l.alive = reinterpret_cast<std::uint64_t*>(base + 512 + cap * 8 + build_selected_k * 8);
    return l;
}

/**
 *  `refine_` (index.hpp:4276-4318) over `count` candidates (LDS, ascending), at most `needed` (≤ 128) accepted into
 *  `sel/seld`. Returns how many. The caller handles the reference's shortcut for `count < needed`.
 */
template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak>
UA_DEVICE std::uint32_t refine_forward(const snapshot_view_t& ix, std::uint8_t* query_lds, const build_lds_t& l,
                                       std::uint32_t count, std::uint32_t needed, std::uint32_t& evaluated) {
    // the bitmap of live candidates sits in LDS, one 64-bit word per lane of the first `words` lanes: any expansion up to
    // build_max_candidates_k at the register cost of one
    const std::uint32_t lane = lane_id();
    const std::uint32_t words = (count + 63) / 64;
    auto word_at = [&](std::uint32_t w) -> std::uint64_t { // every lane reads the same cell; the value comes back wave-uniform
        const std::uint64_t v = l.alive[w];
        return ((std::uint64_t)uniform_u32((std::uint32_t)(v >> 32)) << 32) | uniform_u32((std::uint32_t)v);
    };
    if (lane < words) {
        const std::uint32_t first = 64u * lane;
        l.alive[lane] = count >= first + 64 ? ~0ull : (1ull << (count - first)) - 1ull;
    }
    wave_sync<false>();
    std::uint32_t accepted = 0;
    while (accepted < needed) {
        const std::uint64_t nonempty = ballot(lane < words && l.alive[lane] != 0); // bit w: word w still has candidates
        if (!nonempty)
            break;
        const std::uint32_t first_word = (std::uint32_t)__ffsll((long long)nonempty) - 1;
        const std::uint64_t first_live = word_at(first_word);
        const std::uint32_t chosen_index = 64u * first_word + (std::uint32_t)__ffsll((long long)first_live) - 1;
        const std::uint64_t first_left = first_live & (first_live - 1);
        const std::uint32_t chosen = uniform_u32(l.slots[chosen_index]);
        if (lane == 0) {
            l.alive[first_word] = first_left;
            l.sel[accepted] = chosen, l.seld[accepted] = l.dists[chosen_index];
        }
        wave_sync<false>();
        ++accepted;
        if (accepted == needed || !((nonempty & (nonempty - 1)) | first_left))
            break;
        const query_norm_t a2 = stage_row<metric_ak, scalar_ak, lanes_ak>(ix, chosen, query_lds);
        for (std::uint32_t w = first_word; w < words; ++w) {
            const std::uint64_t live = word_at(w);
            if (!live)
                continue;
            const bool mine = (live >> lane) & 1ull;
            const std::uint32_t position = rank_below(live, lane);
            if (mine)
                l.cand_slots[position] = l.slots[64u * w + lane];
            wave_sync<false>();
            const std::uint32_t batch = popcount64(live);
            measure_rows<metric_ak, scalar_ak, lanes_ak, unroll_ak, false>(ix, query_lds, a2, l.cand_slots,
                                                                           l.cand_distances, batch);
            evaluated += batch;
            const bool struck = mine && l.cand_distances[position] < l.dists[64u * w + lane]; // index.hpp:4300, strict
            const std::uint64_t left = live & ~ballot(struck);
            if (lane == 0)
                l.alive[w] = left;
            wave_sync<false>();
        }
    }
    wave_sync<false>();
    return accepted;
}

template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak>
__global__ __launch_bounds__(64, build_waves(lanes_ak)) void build_select_kernel(const snapshot_view_t ix, const build_args_t b) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    const std::uint32_t lane = lane_id();
    std::uint8_t* query_lds = lds;
    const build_lds_t l = build_lds(lds + query_lds_bytes<scalar_ak>(ix.chunks), b.candidate_cap);
    std::uint32_t evaluated = 0, dropped = 0;
    for (std::uint32_t t = blockIdx.x; t < b.count; t += gridDim.x) {
        const std::uint32_t node = uniform_u32(b.nodes[t]);
        std::uint32_t count = uniform_u32((std::uint32_t)b.cand_counts[t]);
        count = count < b.ef ? count : b.ef;
        count = count < build_max_candidates_k ? count : build_max_candidates_k;
        for (std::uint32_t i = lane; i < count; i += 64) {
            l.slots[i] = (std::uint32_t)b.cand_slots[(std::uint64_t)t * b.ef + i];
            l.dists[i] = b.cand_distances[(std::uint64_t)t * b.ef + i];
        }
        wave_sync<false>();
        std::uint32_t accepted;
        if (count < b.needed) { // index.hpp:4283-4286: a small candidate set is taken whole
            accepted = count;
            if (lane < count)
                l.sel[lane] = l.slots[lane], l.seld[lane] = l.dists[lane];
            wave_sync<false>();
        } else {
            accepted = refine_forward<metric_ak, scalar_ak, lanes_ak, unroll_ak>(ix, query_lds, l, count, b.needed, evaluated);
        }
        // outgoing links of the new node (index.hpp:3835-3842); nobody else touches this list
        std::uint32_t* list = build_list(b, ix, node);
        for (std::uint32_t i = lane; i < b.capacity; i += 64)
            list[i] = i < accepted ? l.sel[i] : none_slot_k;
        // reverse-link requests, applied by build_reverse_kernel once the whole batch has filed its own
        if (lane < accepted) {
            const std::uint32_t target = l.sel[lane];
            const cand_t request = make_cand(l.seld[lane], node);
            if (!build_file_request(b, target, request))
                dropped += build_defer_request(b, target, request);
        }
        wave_sync<false>();
    }
    if (b.counters) {
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1)
            dropped += __shfl_xor(dropped, offset, 64);
        if (lane == 0) {
            atomicAdd(b.counters + 0, (unsigned long long)evaluated);
            if (dropped)
                atomicAdd(b.counters + 3, (unsigned long long)dropped);
        }
    }
}

template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak>
__global__ __launch_bounds__(64, build_waves(lanes_ak)) void build_reverse_kernel(const snapshot_view_t ix, const build_args_t b) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    const std::uint32_t lane = lane_id();
    std::uint8_t* query_lds = lds;
    const build_lds_t l = build_lds(lds + query_lds_bytes<scalar_ak>(ix.chunks), b.candidate_cap);
    const std::uint32_t touched = uniform_u32(*b.touched_count);
    std::uint32_t evaluated = 0, repruned = 0;
    for (std::uint32_t t = blockIdx.x; t < touched; t += gridDim.x) {
        const std::uint32_t target = uniform_u32(b.touched[t]);
        std::uint32_t* list = build_list(b, ix, target);
        const std::uint32_t capacity = b.capacity;
        if (capacity + b.inbox_cap > 64) { // the list and its requests do not share a wave: through LDS
            std::uint32_t existing_count = 0;
            for (std::uint32_t tile = 0; tile < capacity; tile += 64) { // lists are prefix-compact
                const std::uint32_t cell = tile + lane;
                const std::uint32_t value = cell < capacity ? list[cell] : none_slot_k;
                if (value != none_slot_k)
                    l.sel[cell] = value;
                existing_count += popcount64(ballot(value != none_slot_k));
            }
            std::uint32_t incoming_count = uniform_u32(b.inbox_count[target]);
            incoming_count = incoming_count < b.inbox_cap ? incoming_count : b.inbox_cap;
            const cand_t incoming = lane < incoming_count ? b.inbox[(std::uint64_t)target * b.inbox_cap + lane] : 0;
            if (lane == 0)
                b.inbox_count[target] = 0; // ready for the next pass
            wave_sync<false>();
            // a requester the list already names (a member that is being re-linked in place keeps its inbound links) is not new
            const std::uint32_t mine = cand_slot(incoming);
            bool fresh = lane < incoming_count;
            for (std::uint32_t j = 0; j < existing_count; ++j)
                fresh = fresh && l.sel[j] != mine;
            const std::uint64_t fresh_mask = ballot(fresh);
            const std::uint32_t fresh_count = popcount64(fresh_mask);
            if (existing_count + fresh_count <= capacity) {
                // room left (index.hpp:3872-3875): append, in ascending requester order so that the build is reproducible
                std::uint32_t rank = 0;
                for (std::uint32_t j = 0; j < incoming_count; ++j)
                    rank += ((fresh_mask >> j) & 1ull) && read_lane_u32(mine, j) < mine ? 1u : 0u;
                if (fresh)
                    list[existing_count + rank] = mine;
            } else {
                // index.hpp:3877-3891: existing ∪ incoming, measured from `target`, refined down to the capacity
                const query_norm_t a2 = stage_row<metric_ak, scalar_ak, lanes_ak>(ix, target, query_lds);
                for (std::uint32_t tile = 0; tile < existing_count; tile += 64) {
                    const std::uint32_t batch = existing_count - tile < 64 ? existing_count - tile : 64;
                    if (lane < batch)
                        l.cand_slots[lane] = l.sel[tile + lane];
                    wave_sync<false>();
                    measure_rows<metric_ak, scalar_ak, lanes_ak, unroll_ak, false>(ix, query_lds, a2, l.cand_slots, l.cand_distances, batch);
                    if (lane < batch)
                        l.seld[tile + lane] = l.cand_distances[lane];
                    wave_sync<false>();
                }
                evaluated += existing_count;
                if (fresh) { // the requesters behind the old neighbours, with the distance they filed
                    const std::uint32_t position = existing_count + rank_below(fresh_mask, lane);
                    l.sel[position] = mine, l.seld[position] = cand_distance(incoming);
                }
                wave_sync<false>();
                const std::uint32_t total = existing_count + fresh_count; // ≤ 160: up to three candidates per lane
                for (std::uint32_t mine_index = lane; mine_index < total; mine_index += 64) { // ascending by (distance, slot)
                    const float my_distance = l.seld[mine_index];
                    const std::uint32_t my_slot = l.sel[mine_index];
                    std::uint32_t rank = 0;
                    for (std::uint32_t j = 0; j < total; ++j) {
                        const float other_distance = l.seld[j];
                        const std::uint32_t other_slot = l.sel[j];
                        rank += (other_distance < my_distance || (other_distance == my_distance && other_slot < my_slot)) ? 1u : 0u;
                    }
                    l.slots[rank] = my_slot, l.dists[rank] = my_distance;
                }
                wave_sync<false>();
                const std::uint32_t accepted =
                    refine_forward<metric_ak, scalar_ak, lanes_ak, unroll_ak>(ix, query_lds, l, total, capacity, evaluated);
                for (std::uint32_t cell = lane; cell < capacity; cell += 64)
                    list[cell] = cell < accepted ? l.sel[cell] : none_slot_k;
                ++repruned;
            }
            wave_sync<false>();
            continue;
        }
        const std::uint32_t existing = lane < capacity ? list[lane] : none_slot_k;
        const std::uint32_t existing_count = popcount64(ballot(existing != none_slot_k)); // lists are prefix-compact
        std::uint32_t incoming_count = uniform_u32(b.inbox_count[target]);
        incoming_count = incoming_count < b.inbox_cap ? incoming_count : b.inbox_cap;
        cand_t incoming = lane < incoming_count ? b.inbox[(std::uint64_t)target * b.inbox_cap + lane] : 0;
        if (lane == 0)
            b.inbox_count[target] = 0; // ready for the next pass
        {   // a requester the list already names (a member that is being re-linked in place keeps its inbound links) is not new:
            // the fresh ones move to the front lanes, in the order they were filed
            const std::uint32_t mine = cand_slot(incoming);
            bool fresh = lane < incoming_count;
            for (std::uint32_t j = 0; j < existing_count; ++j)
                fresh = fresh && read_lane_u32(existing, j) != mine;
            const std::uint64_t fresh_mask = ballot(fresh);
            if (popcount64(fresh_mask) != incoming_count) {
                const std::uint32_t position = rank_below(fresh_mask, lane);
                l.cand_slots[lane] = 0;
                wave_sync<false>();
                if (fresh)
                    l.cand_slots[position] = lane;
                wave_sync<false>();
                incoming_count = popcount64(fresh_mask);
                const int from = (int)l.cand_slots[lane];
                const std::uint32_t moved_slot = (std::uint32_t)__shfl((int)cand_slot(incoming), from, 64);
                const float moved_distance = __shfl(cand_distance(incoming), from, 64);
                incoming = lane < incoming_count ? make_cand(moved_distance, moved_slot) : 0;
                wave_sync<false>();
            }
        }
        if (existing_count + incoming_count <= capacity) {
            // room left (index.hpp:3872-3875): append, in ascending requester order so that the build is reproducible
            const std::uint32_t mine = cand_slot(incoming);
            std::uint32_t rank = 0;
            for (std::uint32_t j = 0; j < incoming_count; ++j)
                rank += read_lane_u32(mine, j) < mine ? 1u : 0u;
            if (lane < incoming_count)
                list[existing_count + rank] = mine;
        } else {
            // index.hpp:3877-3891: existing ∪ incoming, measured from `target`, refined down to the capacity
            const query_norm_t a2 = stage_row<metric_ak, scalar_ak, lanes_ak>(ix, target, query_lds);
            if (lane < existing_count)
                l.cand_slots[lane] = existing;
            wave_sync<false>();
            measure_rows<metric_ak, scalar_ak, lanes_ak, unroll_ak, false>(ix, query_lds, a2, l.cand_slots,
                                                                           l.cand_distances, existing_count);
            evaluated += existing_count;
            const std::uint32_t total = existing_count + incoming_count; // ≤ 64: one candidate per lane
            // lanes [0, existing) hold the old neighbours, lanes [existing, total) the requesters
            const int source = (int)((lane - existing_count) & 63u);
            const std::uint32_t shifted_slot = (std::uint32_t)__shfl((int)cand_slot(incoming), source, 64);
            const float shifted_distance = __shfl(cand_distance(incoming), source, 64);
            std::uint32_t my_slot = none_slot_k;
            float my_distance = __builtin_inff();
            if (lane < existing_count)
                my_slot = existing, my_distance = l.cand_distances[lane];
            else if (lane < total)
                my_slot = shifted_slot, my_distance = shifted_distance;
            std::uint32_t rank = 0; // ascending by (distance, slot)
            for (std::uint32_t j = 0; j < total; ++j) {
                const float other_distance = read_lane_f32(my_distance, j);
                const std::uint32_t other_slot = read_lane_u32(my_slot, j);
                rank += (other_distance < my_distance || (other_distance == my_distance && other_slot < my_slot)) ? 1u : 0u;
            }
            wave_sync<false>();
            if (lane < total)
                l.slots[rank] = my_slot, l.dists[rank] = my_distance;
            wave_sync<false>();
            const std::uint32_t accepted =
                refine_forward<metric_ak, scalar_ak, lanes_ak, unroll_ak>(ix, query_lds, l, total, capacity, evaluated);
            if (lane < capacity)
                list[lane] = lane < accepted ? l.sel[lane] : none_slot_k;
            ++repruned;
        }
        wave_sync<false>();
    }
    if (b.counters && lane == 0) {
        atomicAdd(b.counters + 1, (unsigned long long)evaluated);
        atomicAdd(b.counters + 2, (unsigned long long)repruned);
    }
}

} // namespace usearch_amd
