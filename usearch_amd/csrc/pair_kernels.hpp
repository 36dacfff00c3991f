/**
 *  usearch_amd/csrc/pair_kernels.hpp — the SHORT-ROW walk: two queries per wavefront, 32 lanes each (gfx950).
 *
 *  For rows of ≤ 128 bytes (b1 × 128 … i8 × 128) with level-0 lists of ≤ 32 neighbours a hop of the best-first beam
 *  (reference `search_to_find_in_base_`, /root/reference/include/usearch/index.hpp:4176-4246) moves a few hundred bytes and
 *  issues a few hundred instructions: the one-query-per-wave kernel of kernels.hpp leaves half of its lanes idle (M0 = 32
 *  neighbours on 64 lanes) and measured, on 20M × 128 b1, 278 vector + 293 scalar instructions per hop with both issue
 *  pipes at 50 % and the waves parked in s_waitcnt 70 % of their time (profiles/r03_short_rows.md) — issue-bound long before
 *  it is memory-bound. Here each HALF of a wave walks its own query:
 *
 *    * everything that is uniform per query (sizes, radius, the heap walk, the commit decisions) lives in vector registers,
 *      replicated over the 32 lanes of the half, so ONE instruction advances TWO queries;
 *    * a lane owns one neighbour of the hop from list cell to distance: it loads that neighbour's whole row itself (the row
 *      that travels with the list for 16-byte rows, else up to eight 16-byte chunks) — no staging of slots and distances
 *      through LDS, no cross-lane reduction, no lanes-per-row in the arithmetic (integer-valued pairs only: b1 and i8
 *      metrics, whose sums are exact in any order);
 *    * the halves are independent state machines (idle → entry point → greedy descent, index.hpp:3964-4003 → beam → dump):
 *      a half that finishes its query draws the next ticket on its own, so a long traversal never holds the other half;
 *      one step of the loop is one list expansion for either phase, so the halves share every instruction of it;
 *    * `top` (sorted_buffer_gt, index.hpp:845-956) is 2 or 4 register cells per lane (expansion ≤ 64 / 128), `next`
 *      (max_heap_gt, 664-835: the reference's sift rules, so equal distances pop in its order) is a binary heap in LDS with
 *      the root cached in registers, `visits` (growing_hash_set_gt, 1085-1211) an exact open-addressing hash in LDS, of any
 *      capacity (a multiple of 128 cells, not a power of two: LDS is what limits the queries in flight). A per-wave global
 *      slab is no alternative on this chip: every global compare-and-swap is executed at the memory side (one 64-byte
 *      request each, ≈ 2 µs per probe round under load — profiles/r03_short_rows.md), whatever its scope;
 *    * the list (and inline rows) of the heap's root is always already in registers when its hop begins: which member the
 *      next hop expands follows from the pop and from the distances of the hop's newcomers BEFORE any of them is committed,
 *      so its list is requested early and travels while the commits run;
 *    * no divergent branch around LDS traffic or wave-level operations: a lane with nothing to store stores into a private
 *      dummy cell, a lane with nothing to insert swaps "empty" for "empty" (each such branch cost an exec-mask save / restore
 *      and, nested in a wave-level loop, a serialising "waterfall" — 13 000 cycles per step in the first version).
 *
 *  Same results as the one-query kernel and as the reference — keys, distance bits, counts and both traversal counters
 *  (the parity tests do not know which kernel ran). Scratch overflow sets the query's status like there; the engine's retry
 *  ladder re-runs those queries with the one-query kernel.
 */
#pragma once
#include "kernels.hpp"

namespace usearch_amd {

/// Register cells of `top` per lane → capacity 32 × cells per half-wave.
constexpr int pair_cells_for(unsigned ef) { return ef <= 64 ? 2 : 4; }
constexpr unsigned pair_max_expansion_k = 128;
constexpr unsigned pair_max_list_k = 32;   ///< neighbours per list: one per lane of a half
constexpr unsigned pair_max_chunks_k = 8;  ///< 16-byte chunks per row
constexpr unsigned pair_query_bytes_k = pair_max_chunks_k * 16;
constexpr unsigned pair_dummy_bytes_k = 64 * 8; ///< one private 8-byte cell per lane: where predicated-off LDS stores land
constexpr unsigned pair_hash_granule_k = 128;   ///< visited-set capacities are multiples of this many cells (any multiple: no power of two)

/// LDS bytes of one half: the staged query | `next` | the visited hash.
inline __host__ __device__ std::uint32_t pair_half_lds_bytes(std::uint32_t next_cap, std::uint32_t hash_cells) {
    return pair_query_bytes_k + (std::uint32_t)align16((std::uint64_t)next_cap * 8) + hash_cells * 4;
}
inline __host__ __device__ std::uint32_t pair_wave_lds_bytes(std::uint32_t next_cap, std::uint32_t hash_cells) {
    return pair_dummy_bytes_k + 2 * pair_half_lds_bytes(next_cap, hash_cells);
}

/// Waves per SIMD the register budget is cut for. LDS, not registers, limits the residency of this kernel (≈ 30 KB per wave).
#ifndef USEARCH_AMD_PAIR_WAVES
#define USEARCH_AMD_PAIR_WAVES 2
#endif

enum pair_phase_t : std::uint32_t {
    pair_idle_k = 0,    ///< draws a ticket at the top of the loop
    pair_entry_k = 1,   ///< measures the entry point
    pair_descent_k = 2, ///< greedy descent, levels max_level … 1
    pair_beam_k = 3,    ///< best-first beam on level 0
    pair_exit_k = 4,    ///< the batch is drained
};

/// Value `v` of sub-lane `index` of the caller's own half. Every lane of the wave must execute it.
UA_DEVICE std::uint32_t half_read_u32(std::uint32_t v, std::uint32_t index) {
    return (std::uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 32u) + (index & 31u)) << 2), (int)v);
}
/// The caller's half of a wave-wide ballot.
UA_DEVICE std::uint32_t half_bits(std::uint64_t mask) {
    return (threadIdx.x & 32u) ? (std::uint32_t)(mask >> 32) : (std::uint32_t)mask;
}
/// `v` of lane `lane_a` for the lower half of the wave, of lane `lane_b` for the upper half; both wave-uniform. Two
/// `v_readlane` and a select: a few cycles, where a `ds_bpermute` is a trip through the LDS crossbar.
UA_DEVICE std::uint32_t pair_read_u32(std::uint32_t v, std::uint32_t lane_a, std::uint32_t lane_b) {
    const std::uint32_t a = (std::uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane_a);
    const std::uint32_t b = (std::uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane_b);
    return (threadIdx.x & 32u) ? b : a;
}
UA_DEVICE float pair_read_f32(float v, std::uint32_t lane_a, std::uint32_t lane_b) {
    return __builtin_bit_cast(float, pair_read_u32(__builtin_bit_cast(std::uint32_t, v), lane_a, lane_b));
}
/// Minimum of `v` over the 32 lanes of the caller's half, in every lane of the half: five DPP row steps (lane 15 of a row of
/// 16 collects the row), one `row_bcast:15` into the odd rows (lane 31 / 63 collect the half), two `v_readlane`.
template <int control_ak, int row_mask_ak, int bank_mask_ak> UA_DEVICE float dpp_min_step(float x) {
    const int moved = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), control_ak, row_mask_ak,
                                                  bank_mask_ak, false); // a lane without a source keeps its own value
    return fminf(x, __builtin_bit_cast(float, moved));
}
UA_DEVICE float half_min_f32(float v) {
    v = dpp_min_step<0x111 /* row_shr:1 */, 0xf, 0xf>(v);
    v = dpp_min_step<0x112 /* row_shr:2 */, 0xf, 0xf>(v);
    v = dpp_min_step<0x113 /* row_shr:3 */, 0xf, 0xf>(v);
    v = dpp_min_step<0x114 /* row_shr:4 */, 0xf, 0xe>(v);
    v = dpp_min_step<0x118 /* row_shr:8 */, 0xf, 0xc>(v);
    v = dpp_min_step<0x142 /* row_bcast:15 */, 0xa, 0xf>(v);
    return pair_read_f32(v, 31, 63);
}

/// Cell of slot `slot` in a visited set of `cap` cells (any capacity: the high half of hash × cap).
UA_DEVICE std::uint32_t pair_home_cell(std::uint32_t slot, std::uint32_t cap) { return __umulhi(hash_slot(slot), cap); }

template <int metric_ak, int scalar_ak, int cells_ak, bool inline_ak>
__global__ __launch_bounds__(64, USEARCH_AMD_PAIR_WAVES) void pair_search_kernel(const snapshot_view_t ix,
                                                                                 const search_args_t args) {
    static_assert(scalar_ak == scalar_b1x8_k || scalar_ak == scalar_i8_k, "integer-valued pairs only: no summation layout");
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    const std::uint32_t lane = threadIdx.x, sl = lane & 31u, half = lane >> 5;
    const bool upper = lane >= 32u;
    const std::uint32_t ef = args.ef, wanted = args.wanted, m0 = ix.m0, m = ix.m;
    const std::uint32_t used_chunks = (ix.bytes_per_vector + 15u) / 16u;
    const std::uint32_t cap = args.hash_cap;
    const std::uint32_t visits_limit = cap - cap / 8; // 87.5 % load
    const std::uint32_t next_cap = args.next_cap;

    // Divergent branches around LDS traffic cost more than the traffic (the compiler wraps each in an exec-mask save / restore
    // and serialises wave-level operations nested in them): every LDS access below is executed by ALL lanes — a lane that has
    // nothing to store stores into its private dummy cell, a lane that has nothing to insert swaps "empty" for "empty".
    cand_t* const dummy = reinterpret_cast<cand_t*>(lds) + lane;
    std::uint8_t* const mine_lds = lds + pair_dummy_bytes_k + half * pair_half_lds_bytes(next_cap, cap);
    std::uint8_t* const query_lds = mine_lds;
    cand_t* const heap = reinterpret_cast<cand_t*>(mine_lds + pair_query_bytes_k);
    std::uint32_t* const table = reinterpret_cast<std::uint32_t*>(mine_lds + pair_query_bytes_k + align16((std::uint64_t)next_cap * 8));

    // ---- per-half state, replicated over the half's lanes
    std::uint32_t phase = pair_idle_k, q = 0, level = 0, closest = 0, closest_ref = 0;
    std::uint32_t next_size = 0, visits_count = 0, computed = 0, cycles = 0, top_size = 0, peak_next = 0;
    float closest_distance = 0.f, radius = 0.f;
    query_norm_t a2;
    float td[cells_ak];
    std::uint32_t ts[cells_ak];
#pragma unroll
    for (int i = 0; i < cells_ak; ++i)
        td[i] = __builtin_inff(), ts[i] = none_slot_k;
    cand_t root = 0;                        // heap[0] while next_size > 0
    std::uint32_t ahead_slot = none_slot_k; // whose level-0 list sits in ahead_cell (/ ahead_row)
    std::uint32_t ahead_cell = none_slot_k;
    uint4 ahead_row = {0u, 0u, 0u, 0u};

#ifdef USEARCH_AMD_PHASES // diagnostic build only (`make PHASES=1`): shader-clock ticks per step of the loop, summed over all waves
    std::uint64_t phase_mark = args.phases ? __builtin_amdgcn_s_memtime() : 0;
    std::uint64_t phase_ticks[7] = {0, 0, 0, 0, 0, 0, 0};
    std::uint32_t diagnostic[5] = {0, 0, 0, 0, 0}; // loop steps, steps with both halves in the beam, commits, pop levels, probe rounds
    auto tick = [&](int step) {
        if (args.phases) {
            const std::uint64_t now = __builtin_amdgcn_s_memtime();
            phase_ticks[step] += now - phase_mark;
            phase_mark = now;
        }
    };
    auto count = [&](int what) { diagnostic[what] += 1; };
#else
    auto tick = [](int) {};
    auto count = [](int) {};
#endif
    auto request_ahead = [&](bool who, std::uint32_t node) { // the list (and the inline rows) of `node` into the ahead registers
        if (who) {
            ahead_slot = node;
            ahead_cell = sl < m0 ? ix.nbr0[(std::uint64_t)node * m0 + sl] : none_slot_k;
            if constexpr (inline_ak)
                if (sl < m0)
                    ahead_row = reinterpret_cast<const uint4*>(ix.nbr0_rows)[(std::uint64_t)node * m0 + sl];
        }
    };

    for (;;) {
        // ---- 1. a beam whose best frontier member is farther than a full `top`'s radius is over (index.hpp:4208-4211): dump
        {
            const bool over = phase == pair_beam_k && (next_size == 0 || (-cand_distance(root) > radius && top_size == ef));
            if (over) {
                const std::uint32_t found = top_size < wanted ? top_size : wanted;
                std::uint64_t* keys = args.keys + (std::uint64_t)q * wanted;
                std::uint32_t* bits = reinterpret_cast<std::uint32_t*>(args.distances) + (std::uint64_t)q * wanted;
#pragma unroll
                for (int i = 0; i < cells_ak; ++i) { // dump_to with the key 0 / signalling-NaN padding of index.hpp:2707-2722
                    const std::uint32_t g = sl * cells_ak + i;
                    if (g < wanted) {
                        keys[g] = g < found ? (args.emit_slots ? (std::uint64_t)ts[i] : ix.keys[ts[i]]) : 0;
                        bits[g] = g < found ? __builtin_bit_cast(std::uint32_t, td[i]) : signaling_nan_bits_k;
                    }
                }
                for (std::uint32_t g = 32u * cells_ak + sl; g < wanted; g += 32)
                    keys[g] = 0, bits[g] = signaling_nan_bits_k;
                if (sl == 0) {
                    args.counts[q] = found;
                    args.visited[q] = cycles;
                    args.computed[q] = computed;
                    args.status[q] = status_done_k;
                    if (args.peaks)
                        args.peaks[2 * (std::uint64_t)q] = peak_next, args.peaks[2 * (std::uint64_t)q + 1] = visits_count;
                }
                phase = pair_idle_k;
            }
        }

        // ---- 2. idle halves draw a ticket and stage their query
        if (ballot(phase == pair_idle_k)) {
            const bool idle = phase == pair_idle_k;
            std::uint32_t ticket = 0;
            if (idle && sl == 0)
                ticket = atomicAdd(args.queue, 1u);
            ticket = pair_read_u32(ticket, 0, 32);
            if (idle) {
                if (ticket >= args.count) {
                    phase = pair_exit_k;
                } else {
                    q = args.todo ? args.todo[ticket] : ticket;
                    const std::uint64_t query_row = args.query_ids ? args.query_ids[q] : q;
                    const std::uint8_t* query = args.queries + query_row * args.query_stride;
                    for (std::uint32_t b = sl; b < pair_query_bytes_k; b += 32)
                        query_lds[b] = b < ix.bytes_per_vector ? query[b] : (std::uint8_t)0;
                    // an empty visited set
                    uint4* cells = reinterpret_cast<uint4*>(table);
                    const uint4 empty = {none_slot_k, none_slot_k, none_slot_k, none_slot_k};
                    for (std::uint32_t i = sl; i < cap / 4; i += 32)
                        cells[i] = empty;
                    phase = pair_entry_k;
                    computed = 0, cycles = 0, next_size = 0, visits_count = 0, top_size = 0, peak_next = 1;
                    level = ix.max_level;
                    ahead_slot = none_slot_k;
                }
            }
            wave_sync<false>();
            if (idle && phase == pair_entry_k) {
                // query-side constants (Σa², Σa): exact integers, every lane of the half sums the whole staged query itself
                a2 = query_norm_t{};
                if constexpr (scalar_ak == scalar_i8_k && metric_ak != metric_ip_k) {
                    int sum = 0, plain = 0;
                    for (std::uint32_t c = 0; c < used_chunks; ++c) {
                        const uint4 a = *reinterpret_cast<const uint4*>(query_lds + c * 16);
                        sum = __builtin_amdgcn_sdot4((int)a.x, (int)a.x, sum, false);
                        sum = __builtin_amdgcn_sdot4((int)a.y, (int)a.y, sum, false);
                        sum = __builtin_amdgcn_sdot4((int)a.z, (int)a.z, sum, false);
                        sum = __builtin_amdgcn_sdot4((int)a.w, (int)a.w, sum, false);
                        if constexpr (metric_ak == metric_pearson_k) {
                            plain = __builtin_amdgcn_sdot4(0x01010101, (int)a.x, plain, false);
                            plain = __builtin_amdgcn_sdot4(0x01010101, (int)a.y, plain, false);
                            plain = __builtin_amdgcn_sdot4(0x01010101, (int)a.z, plain, false);
                            plain = __builtin_amdgcn_sdot4(0x01010101, (int)a.w, plain, false);
                        }
                    }
                    a2.i = sum, a2.j = plain;
                }
            }
        }
        if (!ballot(phase != pair_exit_k))
            break;
        tick(0);
        count(0);
#ifdef USEARCH_AMD_PHASES
        if (ballot(phase == pair_beam_k) == ~0ull)
            count(1);
#endif

        // ---- 3. whose list this step expands, one cell per lane. The beam's root had its list requested when it became the root.
        const bool entering = phase == pair_entry_k, descending = phase == pair_descent_k, beaming = phase == pair_beam_k;
        std::uint32_t slot = beaming ? ahead_cell : none_slot_k;
        uint4 row0 = ahead_row; // inline rows: the neighbour's row arrived with the list
        if (ballot(entering || descending)) {
            if (entering && sl == 0)
                slot = ix.entry_slot;
            if (descending && sl < m)
                slot = ix.upper[(std::uint64_t)(closest_ref + (level - 1)) * m + sl];
        }
        cycles += beaming ? 1u : 0u;
        const bool present = slot != none_slot_k;
        tick(1);

        // ---- 4. beam: visits.set(successor) for the whole list at once (index.hpp:4229); duplicates inside a list were removed
        //         on upload. The other phases measure every neighbour. Straight-line: a lane without a neighbour swaps "empty"
        //         for "empty" in its home cell, which changes nothing.
        const std::uint32_t present_count = (std::uint32_t)__popc(half_bits(ballot(present)));
        const bool overflow = beaming && (visits_count + present_count > visits_limit || next_size - 1 + present_count > next_cap);
        const bool probing = beaming && present && !overflow;
        bool fresh = false;
        if (ballot(beaming)) {
            std::uint32_t h = pair_home_cell(slot, cap);
            std::uint32_t old = atomicCAS(table + h, none_slot_k, probing ? slot : none_slot_k);
            for (;;) { // linear probing, index.hpp:1085-1211
                const bool again = probing && old != none_slot_k && old != slot;
                if (!ballot(again))
                    break;
                count(4);
                h = again ? (h + 1 < cap ? h + 1 : 0u) : h;
                const std::uint32_t seen = atomicCAS(table + h, none_slot_k, again ? slot : none_slot_k);
                old = again ? seen : old;
            }
            fresh = probing && old == none_slot_k;
            visits_count += (std::uint32_t)__popc(half_bits(ballot(fresh)));
        }
        const bool measured = fresh || (present && !beaming);
        tick(2);

        // ---- 5. rows: requested now, consumed after the pop. A lane that measures takes its neighbour's whole row itself.
        uint4 v[inline_ak ? 1 : pair_max_chunks_k];
#pragma unroll
        for (int c = 0; c < (inline_ak ? 1 : (int)pair_max_chunks_k); ++c)
            v[c] = uint4{0u, 0u, 0u, 0u};
        std::uint32_t ref = none_slot_k; // descent: where the neighbour's upper lists start, should it become the closest
        if constexpr (inline_ak) {
            v[0] = row0;
            if (ballot(measured && !beaming)) { // upper levels and the entry point: the row comes from the matrix
                if (measured && !beaming) {
                    v[0] = *reinterpret_cast<const uint4*>(ix.vectors + (std::uint64_t)slot * ix.row_stride);
                    ref = ix.upper_ref[slot];
                }
            }
        } else {
            if (measured) {
                const uint4* row = reinterpret_cast<const uint4*>(ix.vectors + (std::uint64_t)slot * ix.row_stride);
#pragma unroll
                for (int c = 0; c < (int)pair_max_chunks_k; ++c)
                    if ((std::uint32_t)c < used_chunks)
                        v[c] = row[c];
                if (!beaming)
                    ref = ix.upper_ref[slot];
            }
        }

        // ---- 6. beam: next.pop() (index.hpp:786-794, 819-834: the last element replaces the root and sinks; the left child wins
        //         unless the right one is strictly greater), one level per LDS round trip, both halves at once
        if (ballot(beaming)) {
            const bool popping = beaming && !overflow;
            const std::uint32_t n = next_size - 1;
            const cand_t last = heap[popping ? n : 0u];
            const float last_key = cand_distance(last);
            cand_t new_root = last;
            std::uint32_t i = 0;
            bool sinking = popping && n > 1; // with one element left it simply becomes the root
            while (ballot(sinking)) {
                count(3);
                const std::uint32_t left = 2 * i + 1, right = left + 1;
                const bool go = sinking && left < n;
                const bool has_right = go && right < n;
                const cand_t l = heap[go ? left : 0u];
                const cand_t r = heap[has_right ? right : (go ? left : 0u)];
                const float lk = cand_distance(l), rk = cand_distance(r);
                const bool take_left = go && last_key < lk;
                const float best_key = take_left ? lk : last_key;
                const bool take_right = has_right && best_key < rk;
                const bool moved = take_left || take_right;
                const cand_t best_cand = take_right ? r : l;
                cand_t* const where = (moved && sl == 0) ? heap + i : dummy;
                *where = best_cand;
                new_root = (moved && i == 0) ? best_cand : new_root;
                i = take_right ? right : (take_left ? left : i);
                sinking = moved;
            }
            cand_t* const where = (popping && n && sl == 0) ? heap + i : dummy;
            *where = last;
            next_size = popping ? n : next_size;
            root = popping ? new_root : root;
            wave_sync<false>();
            // the likeliest next hop, unless this hop finds something closer (step 8): its list is requested now
            request_ahead(popping && n > 0, cand_slot(new_root));
            ahead_slot = (popping && n == 0) ? none_slot_k : ahead_slot;
        }
        tick(3);

        // ---- 7. distances, computed by every lane (an idle lane's result is dropped: same instructions either way)
        float mine;
        {
            partial_t p;
            if constexpr (inline_ak) {
                accumulate_chunk<metric_ak, scalar_ak>(p, query_lds, 0, v[0]);
            } else {
#pragma unroll
                for (int c = 0; c < (int)pair_max_chunks_k; ++c)
                    if ((std::uint32_t)c < used_chunks)
                        accumulate_chunk<metric_ak, scalar_ak>(p, query_lds, (std::uint32_t)c, v[c]);
            }
            const float distance = finalize_distance<metric_ak, scalar_ak>(p, a2, ix.dimensions);
            mine = measured ? distance : __builtin_inff();
        }
        computed += (std::uint32_t)__popc(half_bits(ballot(measured)));
        // the closest measured neighbour of each half, the FIRST one in list order among equals
        const float best = half_min_f32(mine);
        const std::uint64_t winners = ballot(measured && mine == best);
        const std::uint32_t winners_a = (std::uint32_t)winners, winners_b = (std::uint32_t)(winners >> 32);
        const std::uint32_t first_a = winners_a ? (std::uint32_t)__builtin_ctz(winners_a) : 0u;
        const std::uint32_t first_b = 32u + (winners_b ? (std::uint32_t)__builtin_ctz(winners_b) : 0u);
        const bool any_winner = upper ? winners_b != 0 : winners_a != 0;
        const std::uint32_t best_slot = pair_read_u32(slot, first_a, first_b);
        tick(4);

        // ---- 8. beam: which member the NEXT hop expands is known before anything is committed. The measured newcomers enter
        //         `next` in list order and a key only overtakes strictly smaller ancestors (index.hpp:808-811), so the next root is
        //         the first of the closest newcomers if it is accepted and strictly closer than the root the pop left — else that
        //         root. Its list is requested now; the commits below run while it is in flight.
        {
            const bool accepted = beaming && any_winner && (top_size < ef || best < radius);
            const bool takes_root = accepted && (next_size == 0 || -best > cand_distance(root));
            request_ahead(takes_root && best_slot != ahead_slot, best_slot);
        }

        // ---- 9a. entry point and greedy descent (index.hpp:3964-4003): strict `<` while scanning in list order, so the FIRST
        //          occurrence of the minimum wins; the scan runs over the list of the node the step started from
        bool begin_beam = false;
        if (ballot(entering || descending)) {
            const std::uint32_t best_ref = pair_read_u32(ref, first_a, first_b);
            if (entering) {
                closest = ix.entry_slot, closest_ref = best_ref, closest_distance = best;
                phase = level ? pair_descent_k : pair_beam_k;
                begin_beam = level == 0;
            }
            if (descending) {
                ++cycles;
                if (any_winner && best < closest_distance) {
                    closest = best_slot, closest_ref = best_ref, closest_distance = best;
                } else {
                    level -= 1;
                    if (level == 0) {
                        phase = pair_beam_k;
                        begin_beam = true;
                    }
                }
            }
        }
        // ---- 9b. search_to_find_in_base_ begins (index.hpp:4185-4206): the start is measured once more, enters `next`,
        //          `visits` and `top`
        if (ballot(begin_beam)) {
            const cand_t start = make_cand(-closest_distance, closest);
            cand_t* const where = (begin_beam && sl == 0) ? heap : dummy;
            *where = start;
            (void)atomicCAS(table + pair_home_cell(closest, cap), none_slot_k, (begin_beam && sl == 0) ? closest : none_slot_k);
            if (begin_beam) {
                computed += 1;
                radius = closest_distance;
                root = start;
                next_size = 1, visits_count = 1, top_size = 1, peak_next = 1;
#pragma unroll
                for (int i = 0; i < cells_ak; ++i)
                    td[i] = __builtin_inff(), ts[i] = none_slot_k;
                td[0] = sl == 0 ? radius : td[0];
                ts[0] = sl == 0 ? closest : ts[0];
            }
            request_ahead(begin_beam, closest);
            wave_sync<false>();
        }
        tick(5);

        // ---- 9c. beam: commit the measured newcomers in list order with the reference's tests (index.hpp:4233-4240). Which
        //          candidate is next is a scalar question per half (two find-first-bit on the two halves of one ballot).
        if (ballot(beaming)) {
            if (overflow) { // scratch outgrown: the engine's retry ladder re-runs this query with more room
                if (sl == 0) {
                    args.status[q] = status_overflow_k;
                    atomicAdd(args.queue + 1, 1u);
                }
                phase = pair_idle_k;
            }
            const std::uint64_t pending = ballot(fresh && (top_size < ef || mine < radius)); // radius only shrinks
            std::uint32_t pending_a = (std::uint32_t)pending, pending_b = (std::uint32_t)(pending >> 32);
            const std::uint32_t last_lane = (ef - 1) / cells_ak, last_cell = (ef - 1) % cells_ak;
            const std::uint32_t drop_lane = ef / cells_ak, drop_cell = ef % cells_ak;
            while (pending_a | pending_b) {
                const std::uint32_t lane_a = pending_a ? (std::uint32_t)__builtin_ctz(pending_a) : 0u;
                const std::uint32_t lane_b = 32u + (pending_b ? (std::uint32_t)__builtin_ctz(pending_b) : 0u);
                const bool has = upper ? pending_b != 0 : pending_a != 0;
                pending_a &= pending_a - 1, pending_b &= pending_b - 1;
                const float d = pair_read_f32(mine, lane_a, lane_b);
                const std::uint32_t successor = pair_read_u32(slot, lane_a, lane_b);
                const bool ok = has && (top_size < ef || d < radius);
                if (!ballot(ok))
                    continue;
                count(2);
                // next.insert (index.hpp:765-770, 808-811): the ancestors the new key overtakes form a prefix of its path to
                // the root; every lane reads one ancestor, a ballot finds the prefix, the overtaken ones move one step down
                const float key = -d;
                const std::uint32_t leaf1 = next_size + 1;
                const std::uint32_t depth = 31u - (std::uint32_t)__clz((int)leaf1);
                const bool has_ancestor = sl >= 1 && sl <= depth;
                const cand_t ancestor = heap[has_ancestor ? (leaf1 >> sl) - 1 : 0u];
                // top.insert (index.hpp:928-939): lower_bound placement — the new element lands BEFORE equal ones — decided
                // cell by cell from local information (see top_gt::insert); a half that inserts nothing keeps every cell
                {
                    float below_d = lane_below_f32(td[cells_ak - 1]);
                    const std::uint32_t below_s = lane_below_u32(ts[cells_ak - 1]);
                    below_d = sl == 0 ? -__builtin_inff() : below_d;
                    bool keeps[cells_ak + 1];
                    keeps[0] = !ok || below_d < d;
#pragma unroll
                    for (int c = 0; c < cells_ak; ++c)
                        keeps[c + 1] = !ok || td[c] < d;
#pragma unroll
                    for (int c = cells_ak - 1; c >= 0; --c) {
                        const float under_d = c > 0 ? td[c > 0 ? c - 1 : 0] : below_d;
                        const std::uint32_t under_s = c > 0 ? ts[c > 0 ? c - 1 : 0] : below_s;
                        const float moved_d = keeps[c] ? d : under_d;
                        const std::uint32_t moved_s = keeps[c] ? successor : under_s;
                        td[c] = keeps[c + 1] ? td[c] : moved_d;
                        ts[c] = keeps[c + 1] ? ts[c] : moved_s;
                    }
                    const bool full = top_size == ef;
                    if (ef < 32u * cells_ak) { // what left cell `ef - 1` of a full buffer sits in cell `ef`: drop it
#pragma unroll
                        for (int c = 0; c < cells_ak; ++c)
                            if (drop_cell == (std::uint32_t)c) {
                                td[c] = (ok && full && sl == drop_lane) ? __builtin_inff() : td[c];
                                ts[c] = (ok && full && sl == drop_lane) ? none_slot_k : ts[c];
                            }
                    }
                    top_size += (ok && !full) ? 1u : 0u;
                    // radius = top.top().distance once full (index.hpp:891): cell ef - 1
                    float worst = 0.f;
#pragma unroll
                    for (int c = 0; c < cells_ak; ++c)
                        if (last_cell == (std::uint32_t)c)
                            worst = pair_read_f32(td[c], last_lane, 32u + last_lane);
                    radius = (ok && top_size == ef) ? worst : radius;
                }
                const std::uint32_t rises =
                    (std::uint32_t)__popc(half_bits(ballot(ok && has_ancestor && cand_distance(ancestor) < key)));
                const cand_t entry = make_cand(key, successor);
                const bool stores = ok && (sl == 0 || (has_ancestor && sl <= rises));
                const std::uint32_t cell = sl == 0 ? (leaf1 >> rises) - 1 : (leaf1 >> (sl - 1)) - 1;
                cand_t* const where = stores ? heap + cell : dummy;
                *where = sl == 0 ? entry : ancestor;
                next_size = ok ? leaf1 : next_size;
                root = (ok && rises == depth) ? entry : root;
                wave_sync<false>();
            }
            peak_next = next_size > peak_next ? next_size : peak_next;
            // step 8 named the root whatever happened above; should it ever not have, the hop still finds its list
            request_ahead(phase == pair_beam_k && beaming && next_size > 0 && cand_slot(root) != ahead_slot, cand_slot(root));
        }
        tick(6);
    }
#ifdef USEARCH_AMD_PHASES
    if (args.phases && lane == 0) {
#pragma unroll
        for (int step = 0; step < 7; ++step)
            atomicAdd(args.phases + step, (unsigned long long)phase_ticks[step]);
#pragma unroll
        for (int what = 0; what < 5; ++what)
            atomicAdd(args.phases + 7 + what, (unsigned long long)diagnostic[what]);
    }
#endif
}

} // namespace usearch_amd
