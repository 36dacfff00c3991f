/**
 *  usearch_amd/csrc/kernels.hpp — CDNA4 (gfx950) device code of the batched HNSW search.
 *
 *  One 64-lane wavefront (= one workgroup) walks one query: the greedy descent over the upper levels
 *  (reference `index_gt::search_for_one_`, /root/reference/include/usearch/index.hpp:3964-4003) and the best-first
 *  beam search on level 0 (`search_to_find_in_base_`, index.hpp:4176-4246). Its `top` (sorted_buffer_gt,
 *  index.hpp:845-956), `next` (max_heap_gt, index.hpp:664-835) and `visits` (growing_hash_set_gt, index.hpp:1085-1211)
 *  live in LDS; a second instantiation keeps them in a per-wave global scratch slab for queries that outgrow LDS.
 *
 *  Result parity with the reference is by construction, not by luck:
 *    * distances of all not-yet-visited neighbours of the popped node are evaluated wave-parallel (that is the
 *      HBM-bound part), but they are COMMITTED to `next`/`top` one by one in neighbour-list order with exactly the
 *      reference's comparisons (strict `<` / `>`), so `radius` evolves as on the CPU;
 *    * `top` keeps lower_bound placement (new before equal), `next` is a real binary heap with the reference's
 *      sift rules, so the pop order among equal distances — the norm for Hamming / i8 — is the reference's;
 *    * `visits` is an exact set.
 *  Floating-point sums use a fixed, documented layout (16-byte chunk `c` of a row belongs to lane `c % G` of the
 *  G-lane group that owns the row; one fused-multiply-add chain per lane in chunk order; XOR butterfly G/2…1) which
 *  the CPU oracle restates (`oracle/usearch_oracle.c`, `lanes = G`), so float results are bit-reproducible too.
 *
 *  No MFMA on purpose: every query gathers different rows (no operand reuse), ~3 FLOP per fetched byte.
 */
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace usearch_amd {

#define UA_DEVICE __device__ __forceinline__

// ---------------------------------------------------------------------------------------------------------------------
//  Wave-level helpers (wave = 64 lanes = the whole workgroup)
// ---------------------------------------------------------------------------------------------------------------------

/// Lane of the wave. Workgroups are one wave wide except the team kernel's (five waves share a query): the mask costs nothing
/// where the compiler knows the workgroup is 64 wide.
UA_DEVICE std::uint32_t lane_id() { return threadIdx.x & 63u; }

/**
 *  Orders this wave's scratch accesses across lanes: later reads see earlier writes of any lane. The LDS flavour
 *  fences the local address space only, so global loads that are still in flight are not waited for.
 */
template <bool global_ak> UA_DEVICE void wave_sync() {
    if constexpr (global_ak) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
}

UA_DEVICE std::uint32_t uniform_u32(std::uint32_t v) { return (std::uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
UA_DEVICE float uniform_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
UA_DEVICE std::uint32_t read_lane_u32(std::uint32_t v, std::uint32_t lane) {
    return (std::uint32_t)__builtin_amdgcn_readlane((int)v, (int)uniform_u32(lane));
}
UA_DEVICE float read_lane_f32(float v, std::uint32_t lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)uniform_u32(lane)));
}
UA_DEVICE std::uint64_t ballot(bool p) { return __ballot(p); }
UA_DEVICE std::uint32_t popcount64(std::uint64_t m) { return (std::uint32_t)__popcll(m); }
UA_DEVICE std::uint32_t rank_below(std::uint64_t m, std::uint32_t lane) {
    // `v_mbcnt_lo` / `v_mbcnt_hi` count a mask's bits below the executing lane: two instructions and no per-lane constant — the
    // shift-and-mask spelling kept (1 << lane) − 1 alive in two registers of every kernel for its whole life (round 6: the short-row
    // cuts reloaded such constants from scratch inside the hop loop, each reload behind a wait for every load in flight)
    (void)lane;
    return __builtin_amdgcn_mbcnt_hi((std::uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((std::uint32_t)m, 0u));
}
/// Bit `lane` of a wave-uniform mask as this lane's predicate: the mask IS the predicate register, no instruction and no per-lane
/// constant (the spelling (mask >> lane) & 1 compiles to an AND with a precomputed 1 << lane held in two registers).
UA_DEVICE bool lane_bit(std::uint64_t uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }

/// Lane `i` receives the value of lane `i ^ offset`. Inside a quad (offsets 1 and 2) that is one DPP `quad_perm` move on the
/// vector ALU; wider exchanges go through the LDS crossbar (`ds_bpermute`). Same value either way: summation order untouched.
UA_DEVICE std::uint32_t xor_lane_u32(std::uint32_t v, int offset) {
    if (offset == 1)
        return (std::uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
    if (offset == 2)
        return (std::uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
    return (std::uint32_t)__shfl_xor((int)v, offset, 64);
}
UA_DEVICE int xor_lane(int v, int offset) { return (int)xor_lane_u32((std::uint32_t)v, offset); }
UA_DEVICE float xor_lane(float v, int offset) {
    return __builtin_bit_cast(float, xor_lane_u32(__builtin_bit_cast(std::uint32_t, v), offset));
}
UA_DEVICE double xor_lane(double v, int offset) {
    const std::uint64_t bits = __builtin_bit_cast(std::uint64_t, v);
    const std::uint64_t low = xor_lane_u32((std::uint32_t)bits, offset), high = xor_lane_u32((std::uint32_t)(bits >> 32), offset);
    return __builtin_bit_cast(double, low | (high << 32));
}

/// The smallest value held by any lane, in every lane: quad exchanges and the two row mirrors on the vector ALU (DPP), then the
/// four rows through scalar registers — no trip through the LDS crossbar. All 64 lanes must be active.
UA_DEVICE float wave_min_f32(float v) {
    auto dpp = [](float x, int control) {
        const int bits = __builtin_bit_cast(int, x);
        if (control == 0)
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(bits, bits, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
        if (control == 1)
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(bits, bits, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
        if (control == 2)
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(bits, bits, 0x141 /* row_half_mirror */, 0xf, 0xf, false));
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(bits, bits, 0x140 /* row_mirror */, 0xf, 0xf, false));
    };
    v = fminf(v, dpp(v, 0));
    v = fminf(v, dpp(v, 1));
    v = fminf(v, dpp(v, 2));
    v = fminf(v, dpp(v, 3));
    return fminf(fminf(read_lane_f32(v, 0), read_lane_f32(v, 16)), fminf(read_lane_f32(v, 32), read_lane_f32(v, 48)));
}

/// {float distance; u32 slot} of index.hpp:2097-2101, packed so that one 8-byte LDS access moves it.
using cand_t = std::uint64_t;
UA_DEVICE cand_t make_cand(float d, std::uint32_t slot) {
    return (cand_t)__builtin_bit_cast(std::uint32_t, d) | ((cand_t)slot << 32);
}
UA_DEVICE float cand_distance(cand_t c) { return __builtin_bit_cast(float, (std::uint32_t)c); }
UA_DEVICE std::uint32_t cand_slot(cand_t c) { return (std::uint32_t)(c >> 32); }

/**
 *  Scratch accessors. LDS: plain accesses (one wave, in-order LDS pipe). Global slab: agent-scope relaxed atomics, so
 *  that lanes of the wave exchange data through L2 instead of a possibly stale per-CU L1 line.
 */
template <bool global_ak> struct scratch_gt {
    static UA_DEVICE std::uint32_t load(const std::uint32_t* p) {
        if constexpr (global_ak)
            return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            return *p;
    }
    static UA_DEVICE void store(std::uint32_t* p, std::uint32_t v) {
        if constexpr (global_ak)
            __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            *p = v;
    }
    static UA_DEVICE cand_t load(const cand_t* p) {
        if constexpr (global_ak)
            return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            return *p;
    }
    static UA_DEVICE void store(cand_t* p, cand_t v) {
        if constexpr (global_ak)
            __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            *p = v;
    }
    static UA_DEVICE float load(const float* p) {
        return __builtin_bit_cast(float, load(reinterpret_cast<const std::uint32_t*>(p)));
    }
    static UA_DEVICE void store(float* p, float v) {
        store(reinterpret_cast<std::uint32_t*>(p), __builtin_bit_cast(std::uint32_t, v));
    }
};

// ---------------------------------------------------------------------------------------------------------------------
//  `next`: binary max-heap on the NEGATED distance — index.hpp:664-835
// ---------------------------------------------------------------------------------------------------------------------

/**
 *  insert + shift_up (index.hpp:765-770, 808-811: swap while parent < child, strictly). The ancestors of the new leaf
 *  are non-decreasing towards the root, so the ones the new key overtakes form a prefix of the path: every lane reads
 *  one ancestor, a ballot finds the prefix length, the overtaken ancestors move one step down in parallel.
 *
 *  In two halves. The first only requests the ancestors and waits for nothing, so register-only work placed between the
 *  halves (the insert into `top`) runs while the reads are in flight.
 */
struct push_ticket_t {
    cand_t ancestor;
    std::uint32_t leaf1; ///< 1-based index of the new leaf
    bool has;            ///< this lane holds an ancestor
};
template <bool global_ak> UA_DEVICE push_ticket_t heap_push_begin(const cand_t* heap, std::uint32_t size) {
    using mem = scratch_gt<global_ak>;
    const std::uint32_t lane = lane_id();
    push_ticket_t ticket;
    ticket.leaf1 = size + 1;
    const std::uint32_t depth = 31u - (std::uint32_t)__clz((int)ticket.leaf1); // number of ancestors
    ticket.has = lane >= 1 && lane <= depth;
    ticket.ancestor = 0;
    if (ticket.has)
        ticket.ancestor = mem::load(heap + ((ticket.leaf1 >> lane) - 1));
    return ticket;
}
/// Second half of a push: how far the key rises, the overtaken ancestors move down, the key lands.
template <bool global_ak>
UA_DEVICE void heap_push_finish(cand_t* heap, std::uint32_t& size, const push_ticket_t& ticket, float key, std::uint32_t slot) {
    using mem = scratch_gt<global_ak>;
    const std::uint32_t lane = lane_id();
    const std::uint64_t overtaken = ballot(ticket.has && cand_distance(ticket.ancestor) < key);
    const std::uint32_t rises = popcount64(overtaken);
    // ONE store: lanes 1 … rises move their ancestor one level down, lane 0 puts the key where the last of them was (`rises` never
    // exceeds the number of ancestors, so every lane up to it holds one; the cells are distinct: levels 0 … rises of the path)
    if (lane <= rises) {
        const std::uint32_t up = lane == 0 ? rises : lane - 1;
        mem::store(heap + ((ticket.leaf1 >> up) - 1), lane == 0 ? make_cand(key, slot) : ticket.ancestor);
    }
    size = ticket.leaf1;
    wave_sync<global_ak>();
}
template <bool global_ak>
UA_DEVICE void heap_push(cand_t* heap, std::uint32_t& size, float key, std::uint32_t slot) {
    const push_ticket_t ticket = heap_push_begin<global_ak>(heap, size);
    heap_push_finish<global_ak>(heap, size, ticket, key, slot);
}

/**
 *  pop + shift_down (index.hpp:786-794, 819-834): the last element replaces the root and sinks; at every level the left
 *  child wins unless the right one is strictly greater. Inherently sequential: executed wave-uniformly, one scratch round
 *  trip per level. (Reference shape; the traversal uses `heap_pop` below.)
 */
template <bool global_ak> UA_DEVICE cand_t heap_pop_serial(cand_t* heap, std::uint32_t& size) {
    using mem = scratch_gt<global_ak>;
    const std::uint32_t lane = lane_id();
    const cand_t root = mem::load(heap);
    const std::uint32_t n = size - 1;
    const cand_t last = mem::load(heap + n);
    const float last_key = uniform_f32(cand_distance(last));
    std::uint32_t i = 0;
    for (;;) {
        const std::uint32_t left = 2 * i + 1, right = left + 1;
        if (left >= n)
            break;
        const cand_t l = mem::load(heap + left);
        const cand_t r = mem::load(heap + (right < n ? right : left));
        const float lk = uniform_f32(cand_distance(l)), rk = uniform_f32(cand_distance(r));
        std::uint32_t best = i;
        float best_key = last_key;
        cand_t best_cand = last;
        if (best_key < lk)
            best = left, best_key = lk, best_cand = l;
        if (right < n && best_key < rk)
            best = right, best_key = rk, best_cand = r;
        if (best == i)
            break;
        if (lane == 0)
            mem::store(heap + i, best_cand);
        i = best;
    }
    if (lane == 0 && n)
        mem::store(heap + i, last);
    size = n;
    wave_sync<global_ak>();
    return root;
}

/// The walk of one round of `heap_pop` through the five levels below the hole, on the two ballots re-indexed by the node's number
/// `m` inside the block (the hole is 1, its children 2 and 3, … — lane = m − 2): while the node says "goes on", step to the child it
/// names and mark it. Five scalar instructions per level — bit test, branch, bit test, add-with-carry (m ← 2m + turn), bit set —
/// where the compiler's rendering of the same loop took thirteen; the pop runs once per hop and goes two rounds deep, on walks whose
/// SIMDs are half issue-bound (short rows, profiles/r06_short_rows/README.md §3).
UA_DEVICE void heap_pop_walk(std::uint64_t on_by_node, std::uint64_t right_by_node, std::uint32_t& m, std::uint64_t& path_by_node,
                             std::uint32_t& settled) {
#define USEARCH_AMD_POP_STEP                                                                                                                  \
    "s_bitcmp1_b64 %[on], %[m]\n\ts_cbranch_scc0 9f\n\ts_bitcmp1_b64 %[right], %[m]\n\ts_addc_u32 %[m], %[m], %[m]\n\ts_bitset1_b64 %[path], %[m]\n\t"
    asm volatile("s_mov_b32 %[settled], 1\n\t" USEARCH_AMD_POP_STEP USEARCH_AMD_POP_STEP USEARCH_AMD_POP_STEP USEARCH_AMD_POP_STEP USEARCH_AMD_POP_STEP
                 "s_mov_b32 %[settled], 0\n"
                 "9:\n\t"
                 : [m] "+s"(m), [path] "+s"(path_by_node), [settled] "=&s"(settled)
                 : [on] "s"(on_by_node), [right] "s"(right_by_node)
                 : "scc");
#undef USEARCH_AMD_POP_STEP
}

/**
 *  The same pop — the same decisions, the same final layout — five levels per scratch round trip, with the decisions taken
 *  by all lanes at once instead of one after the other. The 62 descendants of the hole within five levels are fetched in one
 *  go, one per lane (lane 2^t - 2 + k holds descendant k of level t; lane 62 stands for the hole itself). What the sinking
 *  element would do at a node — stop, go left, go right (index.hpp:819-834: left if it is smaller than the left child, then
 *  right instead if that one is strictly greater still) — depends only on that node's two children and on the element, so
 *  every lane answers for its own node (two cross-lane reads), two ballots collect the answers, and the walk through the
 *  subtree is a handful of scalar bit tests. Children of the hole still hold their pre-pop values (only ancestors of the hole
 *  are rewritten), so deciding from the snapshot is exact. A frontier of 2 000 entries (depth 11) takes 3 rounds.
 */
template <bool global_ak> UA_DEVICE cand_t heap_pop(cand_t* heap, std::uint32_t& size) {
    using mem = scratch_gt<global_ak>;
    // a frontier in LDS has far fewer than 2^27 cells: 32-bit positions; the all-global fallback holds up to one cell per member
    using index_t = std::conditional_t<global_ak, std::uint64_t, std::uint32_t>;
    const std::uint32_t lane = lane_id();
    const cand_t root = mem::load(heap);
    const std::uint32_t n = size - 1;
    const cand_t last = mem::load(heap + n);
    const float last_key = uniform_f32(cand_distance(last));
    const bool stands_for_hole = lane == 62;
    const std::uint32_t my_level = stands_for_hole ? 0u : 31u - (std::uint32_t)__clz((int)(lane + 2)); // lane 63: 6, unused
    const std::uint32_t my_offset = stands_for_hole ? 0u : lane + 2 - (1u << my_level);
    const std::uint32_t left_lane = (2u << my_level) - 2 + 2 * my_offset; // lane of my node's left child (levels 0 … 4)
    index_t i = 0;
    while (2 * i + 1 < n) {
        const index_t my_index = ((i + 1) << my_level) - 1 + my_offset;
        cand_t mine = 0;
        if (lane < 62 && my_index < n)
            mine = mem::load(heap + my_index);
        const std::uint32_t key_bits = (std::uint32_t)mine;
        const float left_key = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(left_lane * 4), (int)key_bits));
        const float right_key =
            __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((left_lane + 1) * 4), (int)key_bits));
        const bool has_left = my_level <= 4 && 2 * my_index + 1 < n, has_right = my_level <= 4 && 2 * my_index + 2 < n;
        bool goes_on, goes_right;
        if (has_left && last_key < left_key) {
            goes_on = true;
            goes_right = has_right && left_key < right_key;
        } else {
            goes_right = has_right && last_key < right_key;
            goes_on = goes_right;
        }
        const std::uint64_t on_mask = ballot(goes_on), right_mask = ballot(goes_right);
        // the ballots by node number: node m ≥ 2 answered in lane m − 2, the hole (node 1) in lane 62
        const std::uint64_t on_by_node = (on_mask << 2) | ((on_mask >> 61) & 2ull);
        const std::uint64_t right_by_node = (right_mask << 2) | ((right_mask >> 61) & 2ull);
        std::uint64_t path_by_node = 0; // nodes that move up into their parent
        std::uint32_t m = 1, settled;
        heap_pop_walk(on_by_node, right_by_node, m, path_by_node, settled);
        if (lane_bit(path_by_node >> 2)) // node m answers in lane m − 2
            mem::store(heap + ((my_index - 1) >> 1), mine);
        // the hole went `steps` levels down to the node numbered m = 2^steps + its offset among that level's nodes
        const std::uint32_t steps = 31u - (std::uint32_t)__clz((int)m);
        i = ((i + 1) << steps) + (m - (1u << steps)) - 1;
        if (settled)
            break;
    }
    if (lane == 0 && n)
        mem::store(heap + i, last);
    size = n;
    wave_sync<global_ak>();
    return root;
}

// ---------------------------------------------------------------------------------------------------------------------
//  `top`: ascending sorted buffer — index.hpp:845-956
// ---------------------------------------------------------------------------------------------------------------------

/**
 *  insert(element, limit) (index.hpp:928-939): position = lower_bound by distance (so the new element lands BEFORE equal
 *  ones), the worst element falls off when full, an element not better than a full buffer's worst is refused.
 */
template <bool global_ak>
UA_DEVICE bool sorted_insert(cand_t* top, std::uint32_t& size, std::uint32_t limit, float d, std::uint32_t slot) {
    using mem = scratch_gt<global_ak>;
    const std::uint32_t lane = lane_id();
    std::uint32_t position = 0;
    for (std::uint32_t base = 0; base < size; base += 64) {
        const std::uint32_t j = base + lane;
        const bool less = j < size && cand_distance(mem::load(top + j)) < d;
        position += popcount64(ballot(less));
    }
    if (position == limit)
        return false;
    const bool full = size == limit;
    const std::uint32_t end = size - (full ? 1u : 0u); // entries [position, end) move one cell up
    if (end > position) {
        const std::uint32_t tiles = (end - position + 63) / 64;
        for (std::uint32_t t = tiles; t-- > 0;) { // highest tile first: its destination cells are already free
            const std::uint32_t j = position + t * 64 + lane;
            cand_t moved = 0;
            if (j < end)
                moved = mem::load(top + j);
            wave_sync<global_ak>();
            if (j < end)
                mem::store(top + j + 1, moved);
            wave_sync<global_ak>();
        }
    }
    if (lane == 0)
        mem::store(top + position, make_cand(d, slot));
    size += full ? 0u : 1u;
    wave_sync<global_ak>();
    return true;
}

/// Every lane receives the value of the lane below it (lane 0 keeps its own): one DPP `wave_shr:1` move, no LDS crossbar.
UA_DEVICE std::uint32_t lane_below_u32(std::uint32_t v) {
    return (std::uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
UA_DEVICE float lane_below_f32(float v) {
    return __builtin_bit_cast(float, lane_below_u32(__builtin_bit_cast(std::uint32_t, v)));
}

/**
 *  `top` as the search kernel holds it.
 *    epl_ak > 0   in REGISTERS, blocked layout: lane L owns entries [L·epl, (L+1)·epl) of the ascending array (capacity
 *                 64·epl ≥ expansion), unused cells hold +inf. An insert is decided cell by cell from purely local
 *                 information — a cell keeps its value while that is smaller than the new distance, takes the new element
 *                 if the cell below it is smaller, and its lower neighbour's value otherwise — so it is straight-line
 *                 vector code: no position to compute, no ballot, no scalar round trip (the measured cost of an insert was
 *                 the VALU→SALU→branch latency chain of those, not its instruction count).
 *    epl_ak == 0  in scratch memory (LDS, or the global slab of the fallback mode): any expansion.
 *  Same observable behaviour as sorted_buffer_gt (index.hpp:845-956) either way.
 */
template <int epl_ak, bool global_ak, bool flags_ak = false> struct top_gt {
    static constexpr int regs_k = epl_ak > 0 ? epl_ak : 1;
    static_assert(!flags_ak || epl_ak > 0, "the frontier flags ride in the register layout only");
    /// flags_ak: bit 31 of a cell's slot word says "closed" — the member was already expanded (or the cell is padding); the
    /// open cells ARE the traversal's frontier (see `search_one`, frontier_top_k). Slots are < 2^31 in that mode.
    static constexpr std::uint32_t closed_bit_k = 0x80000000u;
    // plain arrays that are only ever indexed by compile-time constants (every loop over them is fully unrolled): SROA turns
    // each cell into its own SSA value
    float d[regs_k];
    std::uint32_t s[regs_k];
    cand_t* cells = nullptr;
    std::uint32_t size = 0;

    UA_DEVICE void reset(cand_t* memory) {
        cells = memory;
        size = 0;
#pragma unroll
        for (int i = 0; i < regs_k; ++i)
            d[i] = __builtin_inff(), s[i] = none_slot_k;
    }

    /// insert(element, limit), index.hpp:928-939, under the traversal's precondition: the buffer is not full, or `nd` is
    /// smaller than its last element (so the element always lands; index.hpp:931-933 never refuses it). When the buffer is
    /// full afterwards, `radius` receives top().distance (index.hpp:891); before that the traversal never reads it
    /// (every use is guarded by `size == limit`).
    UA_DEVICE bool insert(float nd, std::uint32_t ns, std::uint32_t limit, float& radius) {
        if constexpr (epl_ak == 0) {
            const bool inserted = sorted_insert<global_ak>(cells, size, limit, nd, ns);
            if (inserted && size == limit)
                radius = worst();
            return inserted;
        } else {
            const std::uint32_t lane = lane_id();
            // the cell below this lane's first cell is the last cell of the lane below; below cell 0 lies -inf
            float below_d = lane_below_f32(d[epl_ak - 1]);
            const std::uint32_t below_s = lane_below_u32(s[epl_ak - 1]);
            below_d = lane == 0 ? -__builtin_inff() : below_d;
            // lower_bound placement: the landing cell is the first one that is not smaller than `nd` (new before equal)
            // (for lane 0 that holds whatever `nd` is: a NaN compares smaller than nothing, lands in the very first cell — where
            // the reference's lower_bound puts it, index.hpp:928-939 — and must not pull the cell "below" into the array)
            bool smaller[regs_k + 1];
            smaller[0] = lane == 0 || below_d < nd;
#pragma unroll
            for (int i = 0; i < epl_ak; ++i)
                smaller[i + 1] = d[i] < nd;
#pragma unroll
            for (int i = epl_ak - 1; i >= 0; --i) { // descending: cell i still reads the old cell i - 1
                const float under_d = i > 0 ? d[i > 0 ? i - 1 : 0] : below_d;
                const std::uint32_t under_s = i > 0 ? s[i > 0 ? i - 1 : 0] : below_s;
                const float moved_d = smaller[i] ? nd : under_d;
                const std::uint32_t moved_s = smaller[i] ? ns : under_s;
                d[i] = smaller[i + 1] ? d[i] : moved_d;
                s[i] = smaller[i + 1] ? s[i] : moved_s;
            }
            // wave-uniform bookkeeping; none of it waits for the vector work above
            const bool full = uniform_u32(size) == uniform_u32(limit);
            // (an expansion that is a multiple of the cells per lane — 64, 128, 256, 608 = 38 · 16 … — ends on a lane boundary: the cell
            // to drop is the first of its lane, the radius sits in the last of the lane below; the other expansions pick the cell by
            // number, which with 16 cells per lane compiles to a tree of branches and register moves, per insert)
            if (full && limit < 64u * epl_ak) { // what left cell `limit - 1` of a full buffer sits in cell `limit`: drop it
                const std::uint32_t drop_lane = limit / epl_ak, drop_cell = limit % epl_ak;
                if (drop_cell == 0) {
                    d[0] = lane == drop_lane ? __builtin_inff() : d[0];
                    s[0] = lane == drop_lane ? none_slot_k : s[0];
                } else {
#pragma unroll
                    for (int i = 1; i < epl_ak; ++i)
                        if (drop_cell == (std::uint32_t)i) {
                            d[i] = lane == drop_lane ? __builtin_inff() : d[i];
                            s[i] = lane == drop_lane ? none_slot_k : s[i];
                        }
                }
            }
            size += full ? 0u : 1u;
            if (size == limit) {
                const std::uint32_t last_lane = (limit - 1) / epl_ak, last_cell = (limit - 1) % epl_ak;
                if (last_cell == (std::uint32_t)(epl_ak - 1)) {
                    radius = read_lane_f32(d[epl_ak - 1], last_lane);
                } else {
#pragma unroll
                    for (int i = 0; i < epl_ak - 1; ++i)
                        if (last_cell == (std::uint32_t)i)
                            radius = read_lane_f32(d[i], last_lane);
                }
            }
            return true;
        }
    }

    /**
     *  The closest member that has not been expanded yet: the first open cell of the ascending array. Every lane finds its own
     *  first open cell with straight-line selects, one ballot names the lowest lane that has one.
     *  Returns false when every kept member has been expanded — the traversal is over.
     */
    UA_DEVICE bool first_open(float& distance, std::uint32_t& slot, std::uint32_t& owner_lane, std::uint32_t& owner_cell) const {
        static_assert(flags_ak, "frontier flags are not compiled into this layout");
        std::uint32_t my_cell = (std::uint32_t)regs_k, my_slot = 0;
        float my_distance = 0.f;
#pragma unroll
        for (int i = regs_k - 1; i >= 0; --i) {
            const bool open = (s[i] & closed_bit_k) == 0;
            my_cell = open ? (std::uint32_t)i : my_cell;
            my_slot = open ? s[i] : my_slot;
            my_distance = open ? d[i] : my_distance;
        }
        const std::uint64_t owners = ballot(my_cell < (std::uint32_t)regs_k);
        if (!owners)
            return false;
        owner_lane = (std::uint32_t)__ffsll((long long)owners) - 1;
        distance = read_lane_f32(my_distance, owner_lane);
        slot = read_lane_u32(my_slot, owner_lane);
        owner_cell = read_lane_u32(my_cell, owner_lane);
        return true;
    }

    /// Marks cell `owner_cell` of lane `owner_lane` as expanded.
    UA_DEVICE void close(std::uint32_t owner_lane, std::uint32_t owner_cell) {
        static_assert(flags_ak, "frontier flags are not compiled into this layout");
        const bool mine = lane_id() == owner_lane;
#pragma unroll
        for (int i = 0; i < regs_k; ++i)
            s[i] |= (mine && owner_cell == (std::uint32_t)i) ? closed_bit_k : 0u;
    }

    /// top.top() — the worst kept distance (index.hpp:891). Requires size > 0. (Scratch-memory layout only.)
    UA_DEVICE float worst() const {
        return uniform_f32(cand_distance(scratch_gt<global_ak>::load(cells + (size - 1))));
    }

    /// dump_to(keys, distances, capacity = wanted) with the key 0 / signalling-NaN padding of index.hpp:2707-2722.
    UA_DEVICE void dump(const snapshot_view_t& ix, const search_args_t& args, std::uint32_t q, std::uint32_t found,
                        std::uint32_t wanted) const {
        std::uint64_t* keys = args.keys + (std::uint64_t)q * wanted;
        std::uint32_t* bits = reinterpret_cast<std::uint32_t*>(args.distances) + (std::uint64_t)q * wanted;
        if constexpr (epl_ak == 0) {
            for (std::uint32_t i = lane_id(); i < wanted; i += 64) {
                std::uint64_t key = 0;
                std::uint32_t distance_bits = signaling_nan_bits_k;
                if (i < found) {
                    const cand_t c = scratch_gt<global_ak>::load(cells + i);
                    key = args.emit_slots ? (std::uint64_t)cand_slot(c) : ix.keys[cand_slot(c)];
                    distance_bits = (std::uint32_t)c;
                }
                keys[i] = key;
                bits[i] = distance_bits;
            }
        } else {
#pragma unroll
            for (int i = 0; i < epl_ak; ++i) {
                const std::uint32_t g = lane_id() * epl_ak + i;
                const float distance = d[i];
                const std::uint32_t slot = flags_ak ? (s[i] & ~closed_bit_k) : s[i];
                if (g < wanted) {
                    keys[g] = g < found ? (args.emit_slots ? (std::uint64_t)slot : ix.keys[slot]) : 0;
                    bits[g] = g < found ? __builtin_bit_cast(std::uint32_t, distance) : signaling_nan_bits_k;
                }
            }
            for (std::uint32_t g = 64 * epl_ak + lane_id(); g < wanted; g += 64) // wanted beyond capacity: padding only
                keys[g] = 0, bits[g] = signaling_nan_bits_k;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------
//  `visits`: exact set of slots — index.hpp:1085-1211. LDS: open addressing (CAS); global slab: one bit per slot.
// ---------------------------------------------------------------------------------------------------------------------

UA_DEVICE std::uint32_t hash_slot(std::uint32_t slot) {
    const std::uint32_t h = slot * 2654435761u;
    return h ^ (h >> 15);
}

/// Where the per-query scratch lives.
enum scratch_mode_t : int {
    scratch_lds_k = 0,    ///< top, next, visits (hash) all in LDS — small expansions
    scratch_hash_k = 1,   ///< top, next in LDS; visits = open-addressing hash in a per-wave global slab (L2/MALL resident)
    scratch_global_k = 2, ///< everything in a per-wave global slab, visits = one bit per slot: cannot overflow (fallback)
};

/**
 *  What holds the traversal's frontier (`next` of index.hpp:4176-4246).
 *
 *  frontier_heap_k  the reference's container: a binary max-heap on the negated distance (index.hpp:664-835) in scratch
 *                   memory, never pruned. Pop order among EQUAL distances is the reference's. At expansion ≈ 600 it holds
 *                   ≈ 1 100 entries (peaks at 3-4 × expansion) of which 25 % can still be popped.
 *  frontier_top_k   no container at all. A candidate enters `next` and `top` in the same breath (index.hpp:4233-4240), and an
 *                   entry that `top` has evicted is farther than the radius for good (the radius never grows), so the loop
 *                   `pop the closest of next; stop when it is farther than the radius` only ever expands members that are
 *                   still in `top`: the frontier IS the not-yet-expanded part of `top`. One flag bit per `top` cell replaces
 *                   the 16-KB heap, "pop" is a handful of register selects and a ballot. Same hops in the same order, same
 *                   counters, same results as the heap whenever the distances that meet in the frontier are distinct; two
 *                   members at exactly the same distance may be expanded in the other order than the reference's heap would
 *                   (and a member evicted from a full `top` at exactly the radius is not expanded, where the reference's strict
 *                   `>` of index.hpp:4210 still would). Requires every member to be a result candidate (no predicate, no
 *                   tombstones: a rejected member is traversed without entering `top`) and slots < 2^31. The engine uses it
 *                   for the float-valued pairs; the integer-valued ones (b1, i8), where ties are the norm, keep the heap.
 */
enum frontier_mode_t : int { frontier_heap_k = 0, frontier_top_k = 1 };

/// Returns true for lanes whose `slot` was NOT in the set before (and now is). Inactive lanes return false.
template <int mode_ak>
UA_DEVICE bool visits_set(std::uint32_t* cells, std::uint32_t mask, std::uint32_t slot, bool active) {
    bool fresh = false;
    if constexpr (mode_ak == scratch_global_k) {
        if (active) {
            const std::uint32_t bit = 1u << (slot & 31);
            const std::uint32_t old = atomicOr(cells + (slot >> 5), bit);
            fresh = (old & bit) == 0;
        }
    } else { // LDS or global hash: the same CAS probing, ds_cmpst vs global_atomic_cmpswap
        if (active) {
            std::uint32_t h = hash_slot(slot) & mask;
            for (;;) {
                const std::uint32_t old = atomicCAS(cells + h, none_slot_k, slot);
                if (old == none_slot_k) {
                    fresh = true;
                    break;
                }
                if (old == slot)
                    break;
                h = (h + 1) & mask;
            }
        }
    }
    return fresh;
}

// ---------------------------------------------------------------------------------------------------------------------
//  Distances — index_plugins.hpp:1309-1414 (ip, cos, l2sq, hamming) and 1583-1630 (cos_i8, l2sq_i8)
// ---------------------------------------------------------------------------------------------------------------------

UA_DEVICE float half_bits_to_float(std::uint32_t bits16) {
    return (float)__builtin_bit_cast(_Float16, (std::uint16_t)bits16);
}

/// Scalar kinds whose arithmetic runs in f32 (`result_t = f32_t` of index_plugins.hpp:1930-2001), and the two 16-bit ones
/// among them, which are widened to f32 once when the query is staged.
template <int scalar_ak> constexpr bool f32_math() {
    return scalar_ak == scalar_f32_k || scalar_ak == scalar_f16_k || scalar_ak == scalar_bf16_k;
}
template <int scalar_ak> constexpr bool narrow_float() { return scalar_ak == scalar_f16_k || scalar_ak == scalar_bf16_k; }

/// 16 stored bits → f32: IEEE binary16 (f16_to_f32, index_plugins.hpp:398-410) or bfloat16 = the upper half of an f32
/// (bf16_to_f32, 434-446).
template <int scalar_ak> UA_DEVICE float narrow_bits_to_float(std::uint32_t bits16) {
    if constexpr (scalar_ak == scalar_bf16_k)
        return __builtin_bit_cast(float, bits16 << 16);
    else
        return half_bits_to_float(bits16);
}

/// What one lane accumulates for one row; which members are live depends on (metric, scalar) — the others never leave
/// their initial constant and cost nothing.
///   f32 math  ip: fx = Σab · cos: fx = Σab, fy = Σb² · l2sq: fx = Σ(a-b)² · pearson: fx = Σab, fy = Σb², fz = Σb ·
///             divergence: fx, fy = the two Kullback-Leibler sums · haversine: fx = the haversine term of the pair
///   f64       the same in dx, dy, dz
///   i8        ix = Σab, iy = Σb², iz = Σb (pearson)
///   b1        hamming: ix = Σpopcount(a^b) · tanimoto: ix = Σpopcount(a&b), iy = Σpopcount(a|b) ·
///             sorensen: ix = Σpopcount(a&b), iy = Σ(popcount(a) + popcount(b))
struct partial_t {
    float fx = 0.f, fy = 0.f, fz = 0.f;
    double dx = 0.0, dy = 0.0, dz = 0.0;
    int ix = 0, iy = 0, iz = 0;
};

UA_DEVICE float fma_real(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
UA_DEVICE double fma_real(double a, double b, double c) { return __builtin_fma(a, b, c); }

/// One element pair of an equidimensional metric, `real_ak` = float or double (index_plugins.hpp:1309-1385, 1478-1551).
template <int metric_ak, typename real_ak>
UA_DEVICE void accumulate_real(real_ak& x, real_ak& y, real_ak& z, real_ak a, real_ak b) {
    if constexpr (metric_ak == metric_cos_k) {
        x = fma_real(a, b, x);
        y = fma_real(b, b, y);
    } else if constexpr (metric_ak == metric_ip_k) {
        x = fma_real(a, b, x);
    } else if constexpr (metric_ak == metric_l2sq_k) {
        const real_ak t = a - b;
        x = fma_real(t, t, x);
    } else if constexpr (metric_ak == metric_pearson_k) { // metric_pearson_gt 1478-1520; Σa, Σa² are query constants
        x = fma_real(a, b, x);
        y = fma_real(b, b, y);
        z = z + b;
    } else if constexpr (metric_ak == metric_divergence_k) { // metric_divergence_gt 1526-1551, p = query, q = stored
        real_ak epsilon, log_p, log_q;
        if constexpr (sizeof(real_ak) == 4)
            epsilon = 1.1920928955078125e-7f;
        else
            epsilon = 2.220446049250313e-16;
        const real_ak m = (a + b) / 2 + epsilon;
        if constexpr (sizeof(real_ak) == 4)
            log_p = logf((a + epsilon) / m), log_q = logf((b + epsilon) / m);
        else
            log_p = log((a + epsilon) / m), log_q = log((b + epsilon) / m);
        x = x + a * log_p;
        y = y + b * log_q;
    }
}

template <int metric_ak> UA_DEVICE void accumulate_float(partial_t& p, float a, float b) {
    accumulate_real<metric_ak, float>(p.fx, p.fy, p.fz, a, b);
}

/// metric_haversine_gt (index_plugins.hpp:1636-1657): the pair's (latitude, longitude) in degrees → the term under the
/// arcsine; `angle_to_radians` of line 203.
template <typename real_ak>
UA_DEVICE real_ak haversine_term(real_ak lat_a, real_ak lon_a, real_ak lat_b, real_ak lon_b) {
    const real_ak pi = (real_ak)3.14159265358979323846, straight = (real_ak)180;
    const real_ak lat_delta = ((lat_b - lat_a) * pi / straight) / 2, lon_delta = ((lon_b - lon_a) * pi / straight) / 2;
    const real_ak converted_a = lat_a * pi / straight, converted_b = lat_b * pi / straight;
    if constexpr (sizeof(real_ak) == 4) {
        const float s1 = sinf(lat_delta), s2 = sinf(lon_delta);
        return s1 * s1 + cosf(converted_a) * cosf(converted_b) * (s2 * s2);
    } else {
        const double s1 = sin(lat_delta), s2 = sin(lon_delta);
        return s1 * s1 + cos(converted_a) * cos(converted_b) * (s2 * s2);
    }
}

/// LDS bytes the query occupies per 16-byte row chunk: 16-bit float queries are widened to f32 once, at load time.
template <int scalar_ak> constexpr std::uint32_t query_chunk_bytes() { return narrow_float<scalar_ak>() ? 32u : 16u; }

/// Folds one 16-byte chunk `v` of a stored row against chunk `c` of the query (already in LDS).
template <int metric_ak, int scalar_ak>
UA_DEVICE void accumulate_chunk(partial_t& p, const std::uint8_t* query_lds, std::uint32_t c, uint4 v) {
    const std::uint8_t* q = query_lds + (std::size_t)c * query_chunk_bytes<scalar_ak>();
    if constexpr (metric_ak == metric_haversine_k) { // two scalars: the whole vector sits in chunk 0
        if (c == 0) {
            if constexpr (scalar_ak == scalar_f32_k) {
                const float2 a = *reinterpret_cast<const float2*>(q);
                p.fx = haversine_term<float>(a.x, a.y, __builtin_bit_cast(float, v.x), __builtin_bit_cast(float, v.y));
            } else {
                const double2 a = *reinterpret_cast<const double2*>(q);
                p.dx = haversine_term<double>(a.x, a.y, __builtin_bit_cast(double, ((std::uint64_t)v.y << 32) | v.x),
                                              __builtin_bit_cast(double, ((std::uint64_t)v.w << 32) | v.z));
            }
        }
    } else if constexpr (scalar_ak == scalar_f32_k) {
        const float4 a = *reinterpret_cast<const float4*>(q);
        accumulate_float<metric_ak>(p, a.x, __builtin_bit_cast(float, v.x));
        accumulate_float<metric_ak>(p, a.y, __builtin_bit_cast(float, v.y));
        accumulate_float<metric_ak>(p, a.z, __builtin_bit_cast(float, v.z));
        accumulate_float<metric_ak>(p, a.w, __builtin_bit_cast(float, v.w));
    } else if constexpr (narrow_float<scalar_ak>()) {
        const float4 a0 = *reinterpret_cast<const float4*>(q);
        const float4 a1 = *reinterpret_cast<const float4*>(q + 16);
        accumulate_float<metric_ak>(p, a0.x, narrow_bits_to_float<scalar_ak>(v.x & 0xFFFFu));
        accumulate_float<metric_ak>(p, a0.y, narrow_bits_to_float<scalar_ak>(v.x >> 16));
        accumulate_float<metric_ak>(p, a0.z, narrow_bits_to_float<scalar_ak>(v.y & 0xFFFFu));
        accumulate_float<metric_ak>(p, a0.w, narrow_bits_to_float<scalar_ak>(v.y >> 16));
        accumulate_float<metric_ak>(p, a1.x, narrow_bits_to_float<scalar_ak>(v.z & 0xFFFFu));
        accumulate_float<metric_ak>(p, a1.y, narrow_bits_to_float<scalar_ak>(v.z >> 16));
        accumulate_float<metric_ak>(p, a1.z, narrow_bits_to_float<scalar_ak>(v.w & 0xFFFFu));
        accumulate_float<metric_ak>(p, a1.w, narrow_bits_to_float<scalar_ak>(v.w >> 16));
    } else if constexpr (scalar_ak == scalar_f64_k) {
        const double2 a = *reinterpret_cast<const double2*>(q);
        accumulate_real<metric_ak, double>(p.dx, p.dy, p.dz, a.x,
                                           __builtin_bit_cast(double, ((std::uint64_t)v.y << 32) | v.x));
        accumulate_real<metric_ak, double>(p.dx, p.dy, p.dz, a.y,
                                           __builtin_bit_cast(double, ((std::uint64_t)v.w << 32) | v.z));
    } else if constexpr (scalar_ak == scalar_i8_k) {
        const uint4 a = *reinterpret_cast<const uint4*>(q);
        p.ix = __builtin_amdgcn_sdot4((int)a.x, (int)v.x, p.ix, false);
        p.ix = __builtin_amdgcn_sdot4((int)a.y, (int)v.y, p.ix, false);
        p.ix = __builtin_amdgcn_sdot4((int)a.z, (int)v.z, p.ix, false);
        p.ix = __builtin_amdgcn_sdot4((int)a.w, (int)v.w, p.ix, false);
        if constexpr (metric_ak != metric_ip_k) {
            p.iy = __builtin_amdgcn_sdot4((int)v.x, (int)v.x, p.iy, false);
            p.iy = __builtin_amdgcn_sdot4((int)v.y, (int)v.y, p.iy, false);
            p.iy = __builtin_amdgcn_sdot4((int)v.z, (int)v.z, p.iy, false);
            p.iy = __builtin_amdgcn_sdot4((int)v.w, (int)v.w, p.iy, false);
        }
        if constexpr (metric_ak == metric_pearson_k) { // Σb: a dot product with four ones
            p.iz = __builtin_amdgcn_sdot4(0x01010101, (int)v.x, p.iz, false);
            p.iz = __builtin_amdgcn_sdot4(0x01010101, (int)v.y, p.iz, false);
            p.iz = __builtin_amdgcn_sdot4(0x01010101, (int)v.z, p.iz, false);
            p.iz = __builtin_amdgcn_sdot4(0x01010101, (int)v.w, p.iz, false);
        }
    } else { // b1x8: hamming 1392-1414, tanimoto (= jaccard) 1420-1445, sorensen 1451-1476
        const uint4 a = *reinterpret_cast<const uint4*>(q);
        if constexpr (metric_ak == metric_hamming_k) {
            p.ix += __popc(a.x ^ v.x) + __popc(a.y ^ v.y) + __popc(a.z ^ v.z) + __popc(a.w ^ v.w);
        } else {
            p.ix += __popc(a.x & v.x) + __popc(a.y & v.y) + __popc(a.z & v.z) + __popc(a.w & v.w);
            if constexpr (metric_ak == metric_sorensen_k)
                p.iy += __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(v.x) + __popc(v.y) +
                        __popc(v.z) + __popc(v.w);
            else
                p.iy += __popc(a.x | v.x) + __popc(a.y | v.y) + __popc(a.z | v.z) + __popc(a.w | v.w);
        }
    }
}

/// XOR butterfly over the `lanes_ak` lanes that share a row (offsets lanes/2 … 1), over the members the pair uses.
template <int metric_ak, int scalar_ak, int lanes_ak> UA_DEVICE void reduce_partial(partial_t& p) {
    constexpr bool second = metric_ak == metric_cos_k || metric_ak == metric_pearson_k || metric_ak == metric_divergence_k;
    constexpr bool third = metric_ak == metric_pearson_k;
#pragma unroll
    for (int offset = lanes_ak / 2; offset >= 1; offset >>= 1) {
        if constexpr (f32_math<scalar_ak>()) {
            p.fx += xor_lane(p.fx, offset);
            if constexpr (second)
                p.fy += xor_lane(p.fy, offset);
            if constexpr (third)
                p.fz += xor_lane(p.fz, offset);
        } else if constexpr (scalar_ak == scalar_f64_k) {
            p.dx += xor_lane(p.dx, offset);
            if constexpr (second)
                p.dy += xor_lane(p.dy, offset);
            if constexpr (third)
                p.dz += xor_lane(p.dz, offset);
        } else {
            p.ix += xor_lane(p.ix, offset);
            p.iy += xor_lane(p.iy, offset);
            if constexpr (third)
                p.iz += xor_lane(p.iz, offset);
        }
    }
}

/// Query-side constants of a distance, in the same summation layout as the rows: Σa² (cos, pearson) and Σa (pearson);
/// exact integers for i8 (Σa² also serves l2sq there).
struct query_norm_t {
    float f = 0.f, g = 0.f;
    double d = 0.0, e = 0.0;
    int i = 0, j = 0;
};

/// metric_pearson_gt's closing arithmetic (index_plugins.hpp:1508-1519) from the five sums.
template <typename real_ak>
UA_DEVICE real_ak pearson_distance(std::uint32_t dimensions, real_ak ab, real_ak a2, real_ak b2, real_ak sa, real_ak sb) {
    if (dimensions <= 1)
        return 0;
    const real_ak n = (real_ak)dimensions;
    const real_ak denominator = (n * a2 - sa * sa) * (n * b2 - sb * sb);
    if (denominator == 0)
        return 0;
    const real_ak correlation = n * ab - sa * sb;
    if constexpr (sizeof(real_ak) == 4)
        return 1 - correlation / __builtin_sqrtf(denominator);
    else
        return 1 - correlation / __builtin_sqrt(denominator);
}

template <int metric_ak, int scalar_ak>
UA_DEVICE float finalize_distance(partial_t p, query_norm_t a2, std::uint32_t dimensions) {
    if constexpr (f32_math<scalar_ak>()) {
        if constexpr (metric_ak == metric_cos_k) { // metric_cos_gt, index_plugins.hpp:1334-1359
            if (a2.f == 0.f && p.fy == 0.f)
                return 0.f;
            if (a2.f == 0.f || p.fy == 0.f)
                return 1.f;
            return 1.f - p.fx / (__builtin_sqrtf(a2.f) * __builtin_sqrtf(p.fy));
        } else if constexpr (metric_ak == metric_ip_k) { // metric_ip_gt 1309-1326
            return 1.f - p.fx;
        } else if constexpr (metric_ak == metric_pearson_k) {
            return pearson_distance<float>(dimensions, p.fx, a2.f, p.fy, a2.g, p.fz);
        } else if constexpr (metric_ak == metric_divergence_k) {
            return (p.fx + p.fy) / 2;
        } else if constexpr (metric_ak == metric_haversine_k) {
            return 2 * asinf(__builtin_sqrtf(p.fx));
        } else { // metric_l2sq_gt 1365-1385
            return p.fx;
        }
    } else if constexpr (scalar_ak == scalar_f64_k) { // the same structs with result_t = f64_t, narrowed at the end (2010-2014)
        if constexpr (metric_ak == metric_cos_k) {
            if (a2.d == 0.0 && p.dy == 0.0)
                return 0.f;
            if (a2.d == 0.0 || p.dy == 0.0)
                return 1.f;
            return (float)(1.0 - p.dx / (__builtin_sqrt(a2.d) * __builtin_sqrt(p.dy)));
        } else if constexpr (metric_ak == metric_ip_k) {
            return (float)(1.0 - p.dx);
        } else if constexpr (metric_ak == metric_pearson_k) {
            return (float)pearson_distance<double>(dimensions, p.dx, a2.d, p.dy, a2.e, p.dz);
        } else if constexpr (metric_ak == metric_divergence_k) {
            return (float)((p.dx + p.dy) / 2);
        } else if constexpr (metric_ak == metric_haversine_k) {
            return (float)(2 * asin(__builtin_sqrt(p.dx)));
        } else {
            return (float)p.dx;
        }
    } else if constexpr (scalar_ak == scalar_i8_k) {
        if constexpr (metric_ak == metric_cos_k) { // metric_cos_i8_t 1583-1607, incl. `ab == 0 → 0`
            const float a2f = __builtin_sqrtf((float)a2.i), b2f = __builtin_sqrtf((float)p.iy);
            return p.ix != 0 ? 1.f - (float)p.ix / (a2f * b2f) : 0.f;
        } else if constexpr (metric_ak == metric_ip_k) { // metric_ip_gt<i8_t, f32_t>: exact while |Σ| < 2^24
            return 1.f - (float)p.ix;
        } else if constexpr (metric_ak == metric_pearson_k) { // metric_pearson_gt<i8_t, f32_t>: its f32 sums of small
            // integers are exact while they stay below 2^24 (dimensions ≤ 1040), which is where these equal them
            return pearson_distance<float>(dimensions, (float)p.ix, (float)a2.i, (float)p.iy, (float)a2.j, (float)p.iz);
        } else { // metric_l2sq_i8_t 1613-1630: Σ(a-b)² = Σa² + Σb² - 2Σab, exact in int32
            return (float)(a2.i + p.iy - 2 * p.ix);
        }
    } else {
        if constexpr (metric_ak == metric_hamming_k) // metric_hamming_gt<b1x8_t> 1392-1414
            return (float)p.ix;
        else if constexpr (metric_ak == metric_sorensen_k) // 1 - 2·|a∧b| / (|a| + |b|), f32 like the reference's result_t
            return 1 - 2 * (float)p.ix / (float)p.iy;
        else // tanimoto, jaccard: 1 - |a∧b| / |a∨b|
            return 1 - (float)p.ix / (float)p.iy;
    }
}

/**
 *  Distances from the query (in LDS) to `count` rows whose slots sit in `slots[0..count)`; results to `out[0..count)`.
 *  The wave is split into 64/G groups of G lanes; each lane streams 16-byte chunks `sub, sub+G, …` of its row with
 *  `unroll_ak` loads in flight before the first use. A group takes `rows_ak` rows per round (rows g, g + 64/G, …): their
 *  loads are all issued before the first one is consumed, so a round trip to HBM fetches rows_ak × 64/G rows per wave —
 *  more bytes in flight per wave instead of more waves (a batch's drain phase grows with the waves per query in flight,
 *  not with the bytes per wave). The summation layout of a row does not depend on `rows_ak`.
 */
template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak, bool global_ak, int rows_ak = 1>
UA_DEVICE void measure_rows(const snapshot_view_t& ix, const std::uint8_t* query_lds, query_norm_t a2,
                            const std::uint32_t* slots, float* out, std::uint32_t count) {
    using mem = scratch_gt<global_ak>;
    constexpr std::uint32_t rows_per_round = 64 / lanes_ak;
    const std::uint32_t lane = lane_id();
    const std::uint32_t group = lane / lanes_ak, sub = lane % lanes_ak;
    const std::uint32_t chunks_per_lane = ix.chunks / lanes_ak; // `chunks` is a multiple of G
    constexpr bool fence_ak = false; // see the multi-row path: single rows do not need the scheduling fence
    if constexpr (rows_ak == 1) {
        for (std::uint32_t base = 0; base < count; base += rows_per_round) {
            const std::uint32_t ci = base + group;
            if (ci < count) { // a group is active or idle as a whole, so the butterfly below stays inside active lanes
                const std::uint32_t slot = mem::load(slots + ci);
                const uint4* row = reinterpret_cast<const uint4*>(ix.vectors + (std::uint64_t)slot * ix.row_stride) + sub;
                partial_t p;
                std::uint32_t it = 0;
                for (; it + unroll_ak <= chunks_per_lane; it += unroll_ak) {
                    uint4 v[unroll_ak];
#pragma unroll
                    for (int u = 0; u < unroll_ak; ++u)
                        v[u] = row[(std::size_t)(it + u) * lanes_ak];
#pragma unroll
                    for (int u = 0; u < unroll_ak; ++u) {
                        accumulate_chunk<metric_ak, scalar_ak>(p, query_lds, sub + (it + u) * lanes_ak, v[u]);
                        if constexpr (fence_ak)
                            __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (it < chunks_per_lane) { // ragged tail: still issue every load before the first use
                    uint4 v[unroll_ak];
#pragma unroll
                    for (int u = 0; u < unroll_ak; ++u)
                        if (it + u < chunks_per_lane)
                            v[u] = row[(std::size_t)(it + u) * lanes_ak];
#pragma unroll
                    for (int u = 0; u < unroll_ak; ++u)
                        if (it + u < chunks_per_lane)
                            accumulate_chunk<metric_ak, scalar_ak>(p, query_lds, sub + (it + u) * lanes_ak, v[u]);
                }
                reduce_partial<metric_ak, scalar_ak, lanes_ak>(p);
                if (sub == 0)
                    mem::store(out + ci, finalize_distance<metric_ak, scalar_ak>(p, a2, ix.dimensions));
            }
        }
    } else {
        for (std::uint32_t base = 0; base < count; base += rows_per_round * rows_ak) {
            // row r of this group in this round; a row beyond `count` re-reads the group's first row (always present when the
            // round runs at all for this group) so that the loads below need no predicate — its result is dropped
            const std::uint32_t first = base + group;
            if (first >= count)
                continue;
            const uint4* row[rows_ak];
            bool present[rows_ak];
#pragma unroll
            for (int r = 0; r < rows_ak; ++r) {
                const std::uint32_t ci = first + (std::uint32_t)r * rows_per_round;
                present[r] = ci < count;
                const std::uint32_t slot = mem::load(slots + (present[r] ? ci : first));
                row[r] = reinterpret_cast<const uint4*>(ix.vectors + (std::uint64_t)slot * ix.row_stride) + sub;
            }
            partial_t p[rows_ak];
            std::uint32_t it = 0;
            for (; it + unroll_ak <= chunks_per_lane; it += unroll_ak) {
                uint4 v[rows_ak][unroll_ak];
#pragma unroll
                for (int r = 0; r < rows_ak; ++r)
#pragma unroll
                    for (int u = 0; u < unroll_ak; ++u)
                        v[r][u] = row[r][(std::size_t)(it + u) * lanes_ak];
                // chunk by chunk, every row against the same query chunk (read from LDS once); the scheduling fence keeps the
                // compiler from hoisting all the query reads to the top, which would cost a hundred registers
#pragma unroll
                for (int u = 0; u < unroll_ak; ++u) {
#pragma unroll
                    for (int r = 0; r < rows_ak; ++r)
                        accumulate_chunk<metric_ak, scalar_ak>(p[r], query_lds, sub + (it + u) * lanes_ak, v[r][u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (it < chunks_per_lane) { // ragged tail: a chunk beyond the row re-reads its last one and is not accumulated
                uint4 v[rows_ak][unroll_ak];
#pragma unroll
                for (int r = 0; r < rows_ak; ++r)
#pragma unroll
                    for (int u = 0; u < unroll_ak; ++u) {
                        const std::uint32_t chunk = it + u < chunks_per_lane ? it + u : chunks_per_lane - 1;
                        v[r][u] = row[r][(std::size_t)chunk * lanes_ak];
                    }
#pragma unroll
                for (int u = 0; u < unroll_ak; ++u) {
                    if (it + u < chunks_per_lane) {
#pragma unroll
                        for (int r = 0; r < rows_ak; ++r)
                            accumulate_chunk<metric_ak, scalar_ak>(p[r], query_lds, sub + (it + u) * lanes_ak, v[r][u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int r = 0; r < rows_ak; ++r) {
                reduce_partial<metric_ak, scalar_ak, lanes_ak>(p[r]);
                if (sub == 0 && present[r])
                    mem::store(out + first + (std::uint32_t)r * rows_per_round,
                               finalize_distance<metric_ak, scalar_ak>(p[r], a2, ix.dimensions));
            }
        }
    }
    wave_sync<global_ak>();
}

/// Query-side constants of the distance for the query already staged in LDS: Σa² (and Σa for pearson) in the row
/// summation layout — lane `sub` owns chunks sub, sub+G, …, one chain per lane, XOR butterfly; every group computes the
/// same value.
template <int metric_ak, int scalar_ak, int lanes_ak>
UA_DEVICE query_norm_t staged_norm(const snapshot_view_t& ix, const std::uint8_t* query_lds) {
    const std::uint32_t lane = lane_id();
    query_norm_t a2;
    constexpr bool squares = metric_ak == metric_cos_k || metric_ak == metric_pearson_k;
    if constexpr (f32_math<scalar_ak>() && squares) {
        const std::uint32_t sub = lane % lanes_ak;
        float sum = 0.f, plain = 0.f;
        for (std::uint32_t c = sub; c < ix.chunks; c += lanes_ak) {
            const float* a = reinterpret_cast<const float*>(query_lds + (std::size_t)c * query_chunk_bytes<scalar_ak>());
            constexpr int per_chunk = narrow_float<scalar_ak>() ? 8 : 4;
#pragma unroll
            for (int e = 0; e < per_chunk; ++e) {
                sum = __builtin_fmaf(a[e], a[e], sum);
                if constexpr (metric_ak == metric_pearson_k)
                    plain = plain + a[e];
            }
        }
#pragma unroll
        for (int offset = lanes_ak / 2; offset >= 1; offset >>= 1) {
            sum += xor_lane(sum, offset);
            if constexpr (metric_ak == metric_pearson_k)
                plain += xor_lane(plain, offset);
        }
        a2.f = sum, a2.g = plain;
    } else if constexpr (scalar_ak == scalar_f64_k && squares) {
        const std::uint32_t sub = lane % lanes_ak;
        double sum = 0.0, plain = 0.0;
        for (std::uint32_t c = sub; c < ix.chunks; c += lanes_ak) {
            const double* a = reinterpret_cast<const double*>(query_lds + (std::size_t)c * 16);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sum = __builtin_fma(a[e], a[e], sum);
                if constexpr (metric_ak == metric_pearson_k)
                    plain = plain + a[e];
            }
        }
#pragma unroll
        for (int offset = lanes_ak / 2; offset >= 1; offset >>= 1) {
            sum += xor_lane(sum, offset);
            if constexpr (metric_ak == metric_pearson_k)
                plain += xor_lane(plain, offset);
        }
        a2.d = sum, a2.e = plain;
    } else if constexpr (scalar_ak == scalar_i8_k && metric_ak != metric_ip_k) {
        int sum = 0, plain = 0;
        for (std::uint32_t c = lane; c < ix.chunks; c += 64) {
            const uint4 a = *reinterpret_cast<const uint4*>(query_lds + (std::size_t)c * 16);
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
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) {
            sum += xor_lane(sum, offset);
            plain += xor_lane(plain, offset);
        }
        a2.i = sum, a2.j = plain;
    }
    return a2;
}

/// Copies query `q` into LDS (zero padded to the row stride; f16 widened to f32) and derives its norm.
template <int metric_ak, int scalar_ak, int lanes_ak>
UA_DEVICE query_norm_t stage_query(const snapshot_view_t& ix, const std::uint8_t* query, std::uint8_t* query_lds) {
    const std::uint32_t lane = lane_id();
    if constexpr (narrow_float<scalar_ak>()) {
        float* dst = reinterpret_cast<float*>(query_lds);
        const std::uint32_t scalars = ix.chunks * 8;
        for (std::uint32_t e = lane; e < scalars; e += 64) {
            float value = 0.f;
            if (e < ix.dimensions)
                value = narrow_bits_to_float<scalar_ak>((std::uint32_t)query[2 * e] | ((std::uint32_t)query[2 * e + 1] << 8));
            dst[e] = value;
        }
    } else {
        const std::uint32_t bytes = ix.chunks * 16;
        for (std::uint32_t b = lane; b < bytes; b += 64)
            query_lds[b] = b < ix.bytes_per_vector ? query[b] : (std::uint8_t)0;
    }
    wave_sync<false>();
    return staged_norm<metric_ak, scalar_ak, lanes_ak>(ix, query_lds);
}

/// Same for a STORED row (16-byte aligned, already zero padded to the row stride) playing the query: 16-byte loads.
/// Produces the same LDS image as `stage_query` of that vector, so distances agree bit for bit.
template <int metric_ak, int scalar_ak, int lanes_ak>
UA_DEVICE query_norm_t stage_row(const snapshot_view_t& ix, std::uint32_t slot, std::uint8_t* query_lds) {
    const uint4* row = reinterpret_cast<const uint4*>(ix.vectors + (std::uint64_t)slot * ix.row_stride);
    for (std::uint32_t c = lane_id(); c < ix.chunks; c += 64) {
        const uint4 v = row[c];
        if constexpr (narrow_float<scalar_ak>()) {
            float4* dst = reinterpret_cast<float4*>(query_lds + (std::size_t)c * 32);
            dst[0] = float4{narrow_bits_to_float<scalar_ak>(v.x & 0xFFFFu), narrow_bits_to_float<scalar_ak>(v.x >> 16),
                            narrow_bits_to_float<scalar_ak>(v.y & 0xFFFFu), narrow_bits_to_float<scalar_ak>(v.y >> 16)};
            dst[1] = float4{narrow_bits_to_float<scalar_ak>(v.z & 0xFFFFu), narrow_bits_to_float<scalar_ak>(v.z >> 16),
                            narrow_bits_to_float<scalar_ak>(v.w & 0xFFFFu), narrow_bits_to_float<scalar_ak>(v.w >> 16)};
        } else {
            *reinterpret_cast<uint4*>(query_lds + (std::size_t)c * 16) = v;
        }
    }
    wave_sync<false>();
    return staged_norm<metric_ak, scalar_ak, lanes_ak>(ix, query_lds);
}

// ---------------------------------------------------------------------------------------------------------------------
//  The search kernel
// ---------------------------------------------------------------------------------------------------------------------

/// Byte offsets of the per-wave scratch areas; the same arithmetic runs on the host to size LDS / the global slab.
struct scratch_layout_t {
    std::uint64_t top, next, cand_slots, cand_distances, visits, total;
};

inline __host__ __device__ std::uint64_t align16(std::uint64_t v) { return (v + 15u) & ~(std::uint64_t)15u; }

/// top | next | candidates | visits. In `scratch_hash_k` mode the first three sit in LDS and `visits` alone in the slab;
/// `top_cells` is 0 when `top` lives in registers.
inline __host__ __device__ scratch_layout_t scratch_layout(std::uint64_t top_cells, std::uint64_t next_cap,
                                                           std::uint64_t visits_bytes) {
    scratch_layout_t l;
    l.top = 0;
    l.next = l.top + align16(top_cells * 8);
    l.cand_slots = l.next + align16(next_cap * 8);
    l.cand_distances = l.cand_slots + 256;
    l.visits = l.cand_distances + 256;
    l.total = l.visits + align16(visits_bytes);
    return l;
}

template <int scalar_ak> inline __host__ __device__ std::uint32_t query_lds_bytes(std::uint32_t chunks) {
    return chunks * ((scalar_ak == scalar_f16_k || scalar_ak == scalar_bf16_k) ? 32u : 16u); // a multiple of 16
}

/// What the waves of a TEAM share besides the leader's LDS areas (team_search_kernel): how many rows the hop in progress has
/// gathered, whether the leader measures a share of them itself, and the query's norms.
struct team_t {
    std::uint32_t count;
    std::uint32_t leader_in;
    std::uint32_t reserved[2];
    query_norm_t a2;
};
static_assert(sizeof(team_t) <= 64, "the engine reserves 64 bytes for the control block");
constexpr std::uint32_t team_exit_k = 0xFFFFFFFFu;
/// Behind the control block: the neighbour list of the member the walk is likeliest to expand NEXT (`team_ahead_rows_k` cells, absent
/// ones `none_slot_k`), published by a pipelining leader before a hop's second barrier. The helpers touch those members' rows between
/// that barrier and the next hop's first one — when they would wait for the leader to name and probe — so that the next gather finds
/// its rows in the caches and its pages translated: a lone query's hop is the latency of that gather (2.7 µs of a 4.3-µs hop at
/// expansion 608 over 10M × 768 f16, phase clock of round 6) and bytes are free on a chip that serves one query. Nothing is computed
/// from the touched rows: results, counters and their order cannot change. Measured: 2.56 → 2.44 ms per query at expansion 608
/// (profiles/r06_single_query/). Touching the LISTS of a hop's members as well — the next hop expands one of them a third of the time
/// and nobody could ask for that list ahead — was tried in front of the gather and cost more than it saved (2.47 ms): the gather's
/// loads return behind them.
constexpr std::uint32_t team_ahead_rows_k = 32;
constexpr std::uint32_t team_block_bytes_k = 64 + team_ahead_rows_k * 4;
UA_DEVICE std::uint32_t* team_ahead_list(team_t* team) {
    return reinterpret_cast<std::uint32_t*>(reinterpret_cast<std::uint8_t*>(team) + 64);
}

/// One helper's share of the warm-up: 8 lanes per row, the 64-byte halves `(lane & 7) + 8·k`, k = 0 … 2, of it — the memory side
/// answers a lone 4-byte load with 64 bytes, not with the 128-byte line (profiles/r03_short_rows: request sizes), so a row of 1 536
/// bytes takes 24 touches (rows of ≤ 1.5 KB are covered whole). The loads land in registers nobody reads; the caller keeps the
/// registers allocated until its next gather has waited for its own — younger — loads (loads return in order), so a late arrival
/// cannot land in a register that has a new owner.
struct team_landing_t {
    std::uint32_t cell[3] = {0, 0, 0};
};
UA_DEVICE void team_touch_rows(const snapshot_view_t& ix, const std::uint32_t* ahead, std::uint32_t helper, team_landing_t& landing) {
    const std::uint32_t lane = lane_id();
    const std::uint32_t row = helper * 8 + (lane >> 3);
    const std::uint32_t slot = row < team_ahead_rows_k ? ahead[row] : none_slot_k;
    const std::uint32_t halves = (ix.bytes_per_vector + 63u) / 64u;
    const std::uint8_t* half = ix.vectors + (std::uint64_t)slot * ix.row_stride + (lane & 7u) * 64u;
    if (slot != none_slot_k && (lane & 7u) < halves)
        asm volatile("global_load_dword %0, %1, off" : "=&v"(landing.cell[0]) : "v"(half) : "memory");
    if (slot != none_slot_k && 8u + (lane & 7u) < halves)
        asm volatile("global_load_dword %0, %1, off" : "=&v"(landing.cell[1]) : "v"(half + 512) : "memory");
    if (slot != none_slot_k && 16u + (lane & 7u) < halves)
        asm volatile("global_load_dword %0, %1, off" : "=&v"(landing.cell[2]) : "v"(half + 1024) : "memory");
}
/// Keeps the landing registers allocated up to this point of the program.
UA_DEVICE void team_landing_alive(const team_landing_t& landing) {
    asm volatile("" ::"v"(landing.cell[0]), "v"(landing.cell[1]), "v"(landing.cell[2]));
}

/// The leader and four helpers: 4 × 8 lane groups take the ≤ 32 rows of a hop in ONE round. (Two helpers per SIMD with four rows
/// each were measured: 7 % slower — a helper's round is instruction issue, which a second wave on the SIMD doubles.)
constexpr int team_waves_k = 5;

/// The workgroup barrier of a team: LDS traffic settled, every wave arrived — and NOT the wait for this wave's global loads that
/// `__syncthreads()` brings along: the leader keeps a neighbour list in flight across the barrier (requested a hop ahead), and
/// waiting for it there would put a memory round trip back on every hop.
UA_DEVICE void team_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/// Part `part` of `parts` measures its share of the `count` rows gathered in `slots` — whole rows, each by a lane group in the
/// very layout `measure_rows` uses everywhere, so every distance has the bits the one-wave kernel computes. `in_turn`: the parts
/// take one round of a wave (64 / lanes rows) each in turn and the last one whatever is left, instead of equal shares.
template <int metric_ak, int scalar_ak, int lanes_ak, int loads_ak>
UA_DEVICE void team_share(const snapshot_view_t& ix, const std::uint8_t* query_lds, query_norm_t a2, const std::uint32_t* slots,
                          float* out, std::uint32_t count, std::uint32_t part, std::uint32_t parts, bool in_turn = false) {
    const std::uint32_t per_part = in_turn ? 64u / lanes_ak : (count + parts - 1) / parts;
    const std::uint32_t begin = part * per_part;
    const std::uint32_t end = (in_turn && part + 1 == parts) || begin + per_part > count ? count : begin + per_part;
    if (begin < end)
        measure_rows<metric_ak, scalar_ak, lanes_ak, loads_ak, false, 1>(ix, query_lds, a2, slots + begin, out + begin, end - begin);
}

/**
 *  One query, start to finish. `heaps` = top/next/candidate arrays (LDS, or the slab in `scratch_global_k`), `visits` = the
 *  visited set (LDS hash, slab hash or slab bitmap). Returns false on scratch overflow (nothing written but `status`).
 */
template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak, int mode_ak, int epl_ak, int frontier_ak, int team_ak = 1,
          bool plain_ak = false>
UA_DEVICE bool search_one(const snapshot_view_t& ix, const search_args_t& args, std::uint32_t q,
                          std::uint8_t* query_lds, std::uint8_t* heaps, std::uint32_t* visits, team_t* team = nullptr) {
    constexpr bool global_ak = mode_ak == scratch_global_k;
    constexpr bool in_top_ak = frontier_ak == frontier_top_k;
    static_assert(!in_top_ak || (epl_ak > 0 && !global_ak), "the frontier rides in the register layout of `top`");
    // `plain_ak` (short rows, round 6): the walk of a plain `search` batch and nothing else — level 0, every member a result candidate
    // (no predicate, no tombstones, no member's own row to leave out), lists of one tile, the `seen` cells in place, rows inline
    // (G = 1) or gathered next to the probe (G = 2). The engine promises all of it (`launch_params_t::plain`) and the general build
    // stays the one everything else runs. What it buys: the hop loop of the general build keeps ≈ 45 uniform values of features a
    // plain batch never uses alive (the lazy predicate's six pointers, the beam level's lists, the modes' switches) — the b1 walk
    // restored 89 spilled scalars per pass through its hop loop, every restore a vector-pipe instruction on SIMDs that are half
    // issue-bound (profiles/r06_short_rows/README.md §3).
    static_assert(!plain_ak || (lanes_ak <= 2 && mode_ak == scratch_hash_k && team_ak == 1 && !in_top_ak), "short rows over the global hash");
    using mem = scratch_gt<global_ak>;
    const std::uint32_t lane = lane_id();
    const std::uint32_t ef = args.ef, wanted = args.wanted;
    const scratch_layout_t layout = scratch_layout(epl_ak ? 0 : ef, in_top_ak ? 0 : args.next_cap, 0);
    top_gt<epl_ak, global_ak, in_top_ak> top;
    top.reset(reinterpret_cast<cand_t*>(heaps + layout.top));
    cand_t* next = reinterpret_cast<cand_t*>(heaps + layout.next);
    std::uint32_t* cand_slots = reinterpret_cast<std::uint32_t*>(heaps + layout.cand_slots);
    float* cand_distances = reinterpret_cast<float*>(heaps + layout.cand_distances);
    const std::uint32_t visits_mask = global_ak ? 0u : args.hash_cap - 1;
    const std::uint32_t visits_limit = global_ak ? 0xFFFFFFFFu : args.hash_cap - args.hash_cap / 4; // 75 % load

    // diagnostic phase clock (USEARCH_AMD_PHASES=1): 0 setup+descent, 1 pop+list fetch, 2 visited set, 3 distances,
    // 4 commit, 5 result dump
#ifdef USEARCH_AMD_PHASES // diagnostic build only (`make PHASES=1`): the clock reads cost registers and waits
    std::uint64_t phase_mark = args.phases ? __builtin_amdgcn_s_memtime() : 0;
    std::uint64_t phase_ticks[6] = {0, 0, 0, 0, 0, 0};
    std::uint32_t diagnostic_pushes = 0, diagnostic_ready = 0, diagnostic_rechecks = 0;
    std::uint64_t diagnostic_push_ticks = 0, diagnostic_insert_ticks = 0;
    auto tick = [&](int phase) {
        if (args.phases) {
            const std::uint64_t now = __builtin_amdgcn_s_memtime();
            phase_ticks[phase] += now - phase_mark;
            phase_mark = now;
        }
    };
#else
    auto tick = [](int) {};
#endif
    const std::uint64_t query_row = !plain_ak && args.query_ids ? args.query_ids[q] : q;
    const query_norm_t a2 = stage_query<metric_ak, scalar_ak, lanes_ak>(
        ix, args.queries + query_row * args.query_stride, query_lds);

    if constexpr (!global_ak) { // the bitmap of `scratch_global_k` is zeroed by the host before the launch
        uint4* cells = reinterpret_cast<uint4*>(visits);
        const uint4 empty = {none_slot_k, none_slot_k, none_slot_k, none_slot_k};
        for (std::uint32_t i = lane; i < args.hash_cap / 4; i += 64)
            cells[i] = empty;
#ifdef USEARCH_AMD_EXPERIMENT_NO_SLAB_CLEAR_WAIT // profiles/r05_short_rows/: what a clear-free visited set could save at most in TIME — the
        wave_sync<false>();                        // clear's stores still go out, nobody waits for them (a probe may meet an uncleared cell)
#else
        if constexpr (mode_ak == scratch_hash_k) // the stores must have reached L2 before this wave's atomics probe the cells
            wave_sync<true>();
        else
            wave_sync<false>();
#endif
    }

    // ---- `seen`: an exact, lossy memory of this query's probes in LDS, in front of the visited set in its global slab (short rows).
    //      A probe of the slab is a compare-and-swap executed at the memory side, and the memory side turns over 27.5 G of them a
    //      second whatever the number of waves, their dependence or the slabs' footprint (scripts/probes/atomic_rate_probe.hip):
    //      the b1 walk of BASELINE's config 5 ran at 25 G. Four probes in ten ask about a slot that is already in the set — a
    //      neighbour shared with a member expanded a few hops ago — and every slot a probe has asked about is remembered here,
    //      direct-mapped: a lane whose slot is in its cell KNOWS the answer (visited) and skips the atomic; any other lane probes
    //      the slab as before, which stays the authority. Same answers, fewer atomics.
    //
    // ---- `aside` (the cut for plain batches over gathered rows, `plain_ak` with G = 2): a second table in LDS (`args.aside_offset`,
    //      `aside_cells`) holds the members whose HOME cell in the slab belongs to somebody else. The slab is then never probed past the home cell — ONE round
    //      trip to the memory side per hop instead of 1.8 (at a tenth of the cells taken, some lane of twenty meets a taken cell on
    //      nine hops in ten, and the whole wave waits for its second trip) — and what collides is settled in LDS, a compare-and-swap
    //      with linear probing at LDS latency. A member is in the set iff it sits in its home cell or in `aside`: a home cell never
    //      empties, so a member that found it taken finds it taken ever after. Exact; the room is checked before anything is set
    //      aside (a query that outgrows it is run again by the retry ladder with four times the slab: a quarter of the collisions).
    constexpr bool seen_ak = lanes_ak <= 2 && mode_ak == scratch_hash_k && team_ak == 1;
    // Measured on 20M-vector slices (profiles/r06_short_rows/README.md §4): i8 × 96 at expansion 80 +9.4 % over the general build
    // (the cut with the `seen` cells alone: +3 %), at 64 +8.4 % (+4.9 %); b1 × 128, whose rows travel with the lists and whose hop has
    // nothing to put in the shadow of a second probe round anyway, −1 … −3 %: that cut keeps the `seen` cells alone.
#ifdef USEARCH_AMD_EXPERIMENT_NO_ASIDE // scripts/ A/B runs: the cut for plain batches with the `seen` cells alone
    constexpr bool aside_ak = false;
#else
    constexpr bool aside_ak = plain_ak && lanes_ak == 2;
#endif
    std::uint32_t* seen = nullptr;
    std::uint32_t seen_mask = 0;
    std::uint32_t* aside = nullptr;
    std::uint32_t aside_mask = 0, aside_count = 0;
    if constexpr (aside_ak) {
        aside = reinterpret_cast<std::uint32_t*>(query_lds + args.aside_offset);
        aside_mask = args.aside_cells - 1;
        for (std::uint32_t i = lane; i < args.aside_cells; i += 64)
            aside[i] = none_slot_k;
        wave_sync<false>();
    }
    if constexpr (seen_ak) {
        if (args.seen_cells) {
            seen = reinterpret_cast<std::uint32_t*>(query_lds + args.seen_offset);
            seen_mask = args.seen_cells - 1;
            for (std::uint32_t i = lane; i < args.seen_cells; i += 64)
                seen[i] = none_slot_k;
            wave_sync<false>();
        }
        if (!plain_ak && args.probe_mode == probe_plain_k) { // the claim bits (zero between probe rounds; zeroed once more per query: free)
            std::uint32_t* claim = reinterpret_cast<std::uint32_t*>(query_lds + args.claim_offset);
            for (std::uint32_t i = lane; i < args.claim_bits / 32; i += 64)
                claim[i] = 0u;
            wave_sync<false>();
        }
    }

    std::uint32_t computed = 0, cycles = 0; // context_t counters, index.hpp:2208-2211
    // `unroll_ak` carries the rows a lane group takes per round in its hundreds (search_kernel packs it that way)
    constexpr int loads_ak = unroll_ak % 100, rows_ak = unroll_ak / 100 + 1;
    auto measure = [&](std::uint32_t count) { // the descent: a handful of rows per step, one row per lane group
        measure_rows<metric_ak, scalar_ak, lanes_ak, loads_ak, global_ak, 1>(ix, query_lds, a2, cand_slots, cand_distances,
                                                                            count);
        computed += count;
    };
    auto measure_hop = [&](std::uint32_t count) { // the beam: up to M0 fresh neighbours per hop
        if constexpr (team_ak > 1) {
            // the leader of a team: publish the gather list, take the first share, meet the helpers again when all of it is measured
            if (lane == 0)
                team->count = count, team->leader_in = 1u, team->a2 = a2;
            team_barrier();
            team_share<metric_ak, scalar_ak, lanes_ak, loads_ak>(ix, query_lds, a2, cand_slots, cand_distances, count, 0, 4);
            team_barrier();
        } else {
            measure_rows<metric_ak, scalar_ak, lanes_ak, loads_ak, global_ak, rows_ak>(ix, query_lds, a2, cand_slots,
                                                                                       cand_distances, count);
        }
        computed += count;
    };
    // index_dense.hpp:2071-2081: a member is a result candidate unless it is a tombstone or the caller's predicate
    // (evaluated on the host into one bit per slot) rejects it; it is traversed either way
    auto allowed = [&](std::uint32_t slot) -> bool {
        if constexpr (plain_ak)
            return true;
        if (!ix.has_tombstones && !args.allow_bits && !args.exclude_own)
            return true;
        bool ok = !(args.exclude_own && slot == (std::uint32_t)query_row); // index.hpp:4111, 4161: `updated_slot` never enters `top`
        if (ok && ix.has_tombstones)
            ok = ix.keys[slot] != free_key_k;
        if (ok && args.allow_bits) {
            const std::uint32_t word = slot >> 5, bit = 1u << (slot & 31);
            if (args.known_bits && !(args.known_bits[word] & bit)) {
                // the host has not been asked about this member yet (`search_args_t::known_bits`): ask, and go on with a GUESS — this
                // run's results are provisional, the host runs the query again once it knows. The guess admits a member with the
                // share of "yes" among the answers so far (a fixed pseudo-random draw per slot): a provisional `top` then fills about
                // as fast as the true one, the walk reaches about as far, and the next run has little left to ask.
                ok = slot * 0x9E3779B1u <= args.guess_threshold;
                if (lane == 0) {
                    const std::uint32_t at = atomicAdd(args.ask_cursor, 1u);
                    if (at < args.ask_cap)
                        args.ask_slots[at] = slot, args.ask_keys[at] = ix.keys[slot];
                }
            } else {
                ok = (args.allow_bits[word] & bit) != 0;
            }
        }
        return uniform_u32(ok ? 1u : 0u) != 0;
    };

    // ---- search_for_one_: greedy descent through levels max_level … 1 (index.hpp:3964-4003)
    std::uint32_t closest = ix.entry_slot;
    if (lane == 0)
        mem::store(cand_slots, closest);
    wave_sync<global_ak>();
    measure(1);
    float closest_distance = uniform_f32(mem::load(cand_distances));
    const std::uint32_t beam_level = plain_ak ? 0u : args.beam_level; // 0 for search; the level being linked during construction
    for (std::uint32_t level = ix.max_level; level > beam_level; --level) {
        bool changed;
        do {
            changed = false;
            const std::uint32_t* list = ix.upper + (std::uint64_t)(ix.upper_ref[closest] + (level - 1)) * ix.m;
            for (std::uint32_t tile = 0; tile < ix.m; tile += 64) {
                const std::uint32_t cell = tile + lane;
                const std::uint32_t neighbor = cell < ix.m ? list[cell] : none_slot_k;
                const std::uint32_t count = popcount64(ballot(neighbor != none_slot_k)); // lists are prefix-compact
                if (!count)
                    break;
                if (lane < count)
                    mem::store(cand_slots + lane, neighbor);
                wave_sync<global_ak>();
                measure(count);
                // strict `<` while scanning in list order ⇒ the FIRST occurrence of the minimum wins
                const float mine = lane < count ? mem::load(cand_distances + lane) : __builtin_inff();
                float best = mine;
#pragma unroll
                for (int offset = 32; offset >= 1; offset >>= 1)
                    best = fminf(best, __shfl_xor(best, offset, 64));
                if (best < closest_distance) {
                    const std::uint32_t winner =
                        (std::uint32_t)__ffsll((long long)ballot(lane < count && mine == best)) - 1;
                    closest_distance = best;
                    closest = read_lane_u32(neighbor, winner);
                    changed = true;
                }
                wave_sync<global_ak>();
            }
            ++cycles;
        } while (changed);
    }

    // ---- search_to_find_in_base_: best-first beam on level 0 (index.hpp:4176-4246)
    std::uint32_t next_size = 0, visits_count = 0, peak_next = 1;
    bool overflow = false;
    if (lane == 0)
        mem::store(cand_slots, closest);
    wave_sync<global_ak>();
    measure(1);
    float radius = uniform_f32(mem::load(cand_distances));
    // index_gt::cluster (index.hpp:3089-3125) is the descent alone, down to `beam_level`, plus one more evaluation of the
    // winner's distance (3115) — the very evaluation the beam starts with: one result per query (`wanted` = 1)
    if (!plain_ak && args.descent_only) {
        if (lane == 0) {
            args.keys[q] = args.emit_slots ? (std::uint64_t)closest : ix.keys[closest];
            args.distances[q] = radius;
            args.counts[q] = 1;
            args.visited[q] = cycles;
            args.computed[q] = computed;
            args.status[q] = status_done_k;
            if (args.peaks)
                args.peaks[2 * (std::uint64_t)q] = 0, args.peaks[2 * (std::uint64_t)q + 1] = 0;
        }
        return true;
    }
    if constexpr (!in_top_ak)
        heap_push<global_ak>(next, next_size, -radius, closest);
    visits_set<mode_ak>(visits, visits_mask, closest, lane == 0);
    visits_count = 1;
    if (in_top_ak || allowed(closest)) {
        float first_radius = radius;
        top.insert(radius, closest, ef, first_radius);
    }

    const std::uint32_t cells = plain_ak ? (ix.m0 < 64u ? ix.m0 : 64u) : beam_level ? ix.m : ix.m0;
    auto list_of = [&](std::uint32_t slot) -> const std::uint32_t* {
        if constexpr (plain_ak)
            return ix.nbr0 + (std::uint64_t)slot * ix.m0;
        return beam_level ? ix.upper + (std::uint64_t)(ix.upper_ref[slot] + (beam_level - 1)) * ix.m
                          : ix.nbr0 + (std::uint64_t)slot * ix.m0;
    };
    std::uint32_t ahead_slot = none_slot_k, ahead_cell = none_slot_k; // list tile requested ahead of its hop
    // rows stored next to the lists (`nbr0_rows`): one-chunk rows only, lists of one tile, level 0
    constexpr bool inline_ak = lanes_ak == 1 && !global_ak;
    const bool inline_rows = plain_ak ? inline_ak : inline_ak && ix.nbr0_rows != nullptr && !beam_level && cells <= 64 && ix.chunks == 1;
    uint4 ahead_row = {0u, 0u, 0u, 0u};
    // rows of ≤ 128 bytes over a visited set in a global slab: gathered next to the probe instead of behind it (see the hop loop)
    constexpr bool early_ak = lanes_ak == 2 && mode_ak == scratch_hash_k && team_ak == 1;
    const bool early_rows = plain_ak ? early_ak : early_ak && args.early_rows != 0 && !inline_rows && cells <= 64;
    tick(0);
    // ---- a team over the in-`top` frontier with a wide `top` (≥ 8 cells per lane: expansions above 256, where a commit costs about
    // what measuring the hop's rows does) walks the beam as a PIPELINE (lists of ≤ 64 cells): the leader names the next member to
    // expand BEFORE it commits the hop just measured, probes its list, hands the gather to the helpers and commits while they
    // measure. What makes the early naming exact: the next member is the first open cell of the array the commit WOULD leave — the
    // closest open member of `top` as it stands, or the closest newcomer if that one is strictly closer (it is then certain to
    // land: ahead of a kept member). Every tie — two newcomers at the smallest distance, a newcomer level with the open member, a
    // buffer that fills up mid-commit — is settled the long way: commit first, look again. Same commits in the same order, same
    // counters, same bits (tests/test_gpu_search_parity.py: test_team_and_one_wave_agree, test_team_settles_ties_like_one_wave).
    bool pipelined = false;
    if constexpr (team_ak > 1 && in_top_ak && epl_ak >= 8)
        pipelined = cells <= 64 && !inline_rows;
    if constexpr (team_ak > 1 && in_top_ak && epl_ak >= 8) {
        if (pipelined) {
            bool pending = false;        // the hop measured last has not been committed yet
            bool candidate = false;      // this lane holds one of its newcomers (lanes in list order)
            float mine = 0.f;
            std::uint32_t mine_slot = 0u;
            // commit in list order with the reference's tests (index.hpp:4233-4240); the newcomer that is being expanded lands closed
            auto commit = [&](std::uint32_t closed_lane) {
                std::uint64_t todo = ballot(candidate && (top.size < ef || mine < radius)); // radius only shrinks
                while (todo) {
                    const std::uint32_t i = (std::uint32_t)__ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const float d = read_lane_f32(mine, i);
                    if (!(top.size < ef || d < radius))
                        continue;
                    const std::uint32_t successor = read_lane_u32(mine_slot, i);
                    top.insert(d, i == closed_lane ? successor | top.closed_bit_k : successor, ef, radius);
                }
                pending = false;
            };
            for (;;) {
                float open_distance = 0.f;
                std::uint32_t open_slot = 0u, owner_lane = 0u, owner_cell = 0u;
                const bool has_open = top.first_open(open_distance, open_slot, owner_lane, owner_cell);
                std::uint32_t from_lane = 64u; // the newcomer expanded next, if it is one
                if (pending) {
                    const std::uint64_t newcomers = ballot(candidate);
                    const std::uint32_t room = ef - top.size;
                    bool exact = room == 0u || room >= popcount64(newcomers); // else the buffer fills up mid-commit
                    const bool lands = candidate && (room != 0u || mine < radius);
                    if (ballot(lands && mine != mine))
                        exact = false; // a NaN among the landing newcomers orders against nothing: the long way
                    if (exact && ballot(lands)) {
                        const float best = wave_min_f32(lands ? mine : __builtin_inff());
                        const std::uint64_t at_best = ballot(lands && mine == best);
                        if (!has_open || best < open_distance) {
                            if (at_best == 0 || (at_best & (at_best - 1)))
                                exact = false; // two newcomers at the smallest distance: their order in the array decides — or none
                                               // at it at all (every landing newcomer a NaN, which equals nothing): commit, look again
                            else
                                from_lane = (std::uint32_t)__ffsll((long long)at_best) - 1;
                        } else if (best == open_distance)
                            exact = false; // a newcomer lands in front of its equal
                    }
                    if (!exact) {
                        commit(64u);
                        continue;
                    }
                }
                if (from_lane == 64u && !has_open) {
                    if (pending)
                        commit(64u); // nothing of it lands (or it would have been named): kept for the shape of the loop
                    break;
                }
                std::uint32_t expanded;
                std::uint32_t neighbor = none_slot_k;
                std::uint32_t likely = none_slot_k; // where the walk goes after this hop unless it finds something closer
                if (from_lane != 64u) {
                    expanded = read_lane_u32(mine_slot, from_lane);
                    if (lane < cells) // nobody could ask for a newcomer's list ahead: a round trip of ≈ 0.4 µs, on a third of the hops
                        neighbor = list_of(expanded)[lane];
                    likely = has_open ? open_slot : none_slot_k; // the open member stays first in line
                } else {
                    expanded = open_slot;
                    top.close(owner_lane, owner_cell);
                    if (expanded == ahead_slot)
                        neighbor = ahead_cell;
                    else if (lane < cells)
                        neighbor = list_of(expanded)[lane];
#ifdef USEARCH_AMD_PHASES
                    diagnostic_ready += expanded == ahead_slot ? 1u : 0u;
#endif
                    float likely_distance;
                    std::uint32_t likely_lane, likely_cell;
                    if (!top.first_open(likely_distance, likely, likely_lane, likely_cell))
                        likely = none_slot_k;
                }
                ++cycles;
                const bool present = neighbor != none_slot_k;
                const std::uint64_t present_mask = ballot(present);
                const std::uint32_t present_count = popcount64(present_mask);
                std::uint32_t count = 0;
                bool fresh = false;
                tick(1);
                if (present_count) {
                    if (visits_count + present_count > visits_limit) {
                        overflow = true;
                        break;
                    }
                    // A team serves batches that leave the chip mostly idle (a `usearch_search` caller's single query above all):
                    // bytes are free, the hop's chain of round trips is what the caller waits for. So the helpers get EVERY
                    // neighbour of the list at once (round 6) — they gather and measure while the leader probes the visited set
                    // (two or three dependent round trips to the slab), instead of waiting for the probe to name the fresh ones.
                    // Four helpers of eight lane groups take 32 rows in one round either way; the distances of neighbours that
                    // turn out visited are dropped, `computed` counts the fresh ones as the reference does.
                    if (present)
                        cand_slots[rank_below(present_mask, lane)] = neighbor; // list order
                    if (lane == 0)
                        team->count = present_count, team->leader_in = 0u, team->a2 = a2;
                    team_barrier(); // the helpers fetch and measure the rows …
                    std::uint32_t h = hash_slot(neighbor) & visits_mask;
                    std::uint32_t old = neighbor; // an absent lane probes nothing
                    if (present)
                        old = atomicCAS(visits + h, none_slot_k, neighbor);
                    while (old != none_slot_k && old != neighbor) { // linear probing, index.hpp:1085-1211
                        h = (h + 1) & visits_mask;
                        old = atomicCAS(visits + h, none_slot_k, neighbor);
                    }
                    fresh = present && old == none_slot_k;
                    count = popcount64(ballot(fresh));
                    visits_count += count;
                }
                // its list is requested now, a hop ahead (the commit below may still put a newcomer of the hop before in front of it) —
                // and only now that this hop's own list has been consumed: loads return in order, and a wait for that list placed
                // after this request would wait for both
                if (likely != ahead_slot) {
                    ahead_slot = likely;
                    if (likely != none_slot_k)
                        ahead_cell = lane < cells ? list_of(likely)[lane] : none_slot_k;
                }
                tick(2);
                if (pending)
                    commit(from_lane); // … while the leader probes and the hop before lands in `top`
                // … and if that put a newcomer in front of the member whose list was just asked for, its list is asked for too
                {
                    float ahead_distance;
                    std::uint32_t first = none_slot_k, ahead_lane, ahead_index;
                    if (!top.first_open(ahead_distance, first, ahead_lane, ahead_index))
                        first = none_slot_k;
                    if (first != ahead_slot) {
                        ahead_slot = first;
                        if (first != none_slot_k)
                            ahead_cell = lane < cells ? list_of(first)[lane] : none_slot_k;
                    }
                }
                tick(4);
                if (present_count) {
                    // the list of the member likeliest to be expanded next, for the helpers to touch its members' rows (`team_ahead_list`);
                    // the wait for that list sits where the leader would wait for the helpers anyway
                    if (lane < team_ahead_rows_k)
                        team_ahead_list(team)[lane] = ahead_slot != none_slot_k ? ahead_cell : none_slot_k;
                    team_barrier(); // every share is in LDS
                    if (count) {
                        candidate = fresh; // lanes in list order, the visited ones among them skipped
                        mine = fresh ? cand_distances[rank_below(present_mask, lane)] : 0.f;
                        mine_slot = fresh ? neighbor : 0u;
                        computed += count;
                        pending = true;
                    }
                }
                tick(3);
            }
        }
    }
    for (; !pipelined;) {
        std::uint32_t expanded;
        if constexpr (in_top_ak) {
            // the closest member not expanded yet; every kept member is within the radius, so index.hpp:4210 never fires
            float open_distance;
            std::uint32_t owner_lane, owner_cell;
            if (!top.first_open(open_distance, expanded, owner_lane, owner_cell))
                break;
            top.close(owner_lane, owner_cell);
        } else {
            if (!next_size)
                break;
            const cand_t candidate = mem::load(next);
            const float candidate_distance = -uniform_f32(cand_distance(candidate));
            if (candidate_distance > radius && top.size == ef) // index.hpp:4210, strict `>`
                break;
            expanded = uniform_u32(cand_slot(candidate));
        }
        ++cycles;
        const std::uint32_t* list = list_of(expanded);
        const bool list_ready = expanded == ahead_slot; // its first tile was requested one hop ago
#ifdef USEARCH_AMD_PHASES
        diagnostic_ready += list_ready ? 1u : 0u;
#endif
        const std::uint32_t ready_cell = ahead_cell;
        std::uint32_t first_cell = none_slot_k;
        if (!list_ready && lane < cells)
            first_cell = list[lane];
        // rows of ≤ 16 bytes travel WITH the list: cell j of a node's list is followed, in `nbr0_rows`, by a copy of the row
        // it names — one contiguous block per hop instead of a list line plus up to M0 scattered sectors, and no dependent
        // round trip between "which neighbours" and "their vectors"
        uint4 inline_row = {0u, 0u, 0u, 0u};
        if constexpr (inline_ak) {
            if (inline_rows) {
                if (list_ready)
                    inline_row = ahead_row;
                else if (lane < cells)
                    inline_row = reinterpret_cast<const uint4*>(ix.nbr0_rows)[(std::uint64_t)expanded * cells + lane];
            }
        }
        // next.pop() only touches LDS: it runs in the shadow of a memory round trip — of the list load when the list was
        // not requested ahead, else of the first probe of the visited set. Right after it the frontier's new best is the
        // likeliest next hop (unless this hop finds something closer): its list is requested then, so that the row
        // arrives behind this hop's vector traffic instead of in front of the next hop's.
        bool popped = false;
        auto pop_now = [&]() {
            popped = true;
            ahead_slot = none_slot_k;
            if constexpr (in_top_ak) { // the member was closed above; the next open one is the likeliest next hop
                float ahead_distance;
                std::uint32_t ahead_lane, ahead_index;
                if (!top.first_open(ahead_distance, ahead_slot, ahead_lane, ahead_index))
                    ahead_slot = none_slot_k;
            } else {
                heap_pop<global_ak>(next, next_size);
                if (next_size)
                    ahead_slot = uniform_u32(cand_slot(mem::load(next)));
            }
            if (ahead_slot != none_slot_k) {
                ahead_cell = lane < cells ? list_of(ahead_slot)[lane] : none_slot_k;
                if constexpr (inline_ak)
                    if (inline_rows && lane < cells)
                        ahead_row = reinterpret_cast<const uint4*>(ix.nbr0_rows)[(std::uint64_t)ahead_slot * cells + lane];
            }
        };
        if (in_top_ak || !list_ready || cells > 64 || mode_ak == scratch_global_k)
            pop_now();
        for (std::uint32_t tile = 0; tile < cells; tile += 64) {
            const std::uint32_t cell = tile + lane;
            const std::uint32_t neighbor =
                tile == 0 ? (list_ready ? ready_cell : first_cell) : (cell < cells ? list[cell] : none_slot_k);
            const bool present = neighbor != none_slot_k;
            const std::uint32_t present_count = popcount64(ballot(present));
            if (!present_count)
                break;
            if (visits_count + present_count > visits_limit ||
                (!in_top_ak && next_size - (popped ? 0u : 1u) + present_count > args.next_cap)) {
                overflow = true;
                break;
            }
            // visits.set(successor) for the whole tile at once; duplicates inside a list were removed on upload
            tick(1);
            bool fresh;
            float early_mine = 0.f;          // inline rows: this lane's distance, computed in the shadow of the probe
            bool early_done = false;
            std::uint64_t asked_mask = 0;    // early rows: the lanes whose rows were gathered next to the probe, …
            bool measured_early = false;     // … their distances waiting in `cand_distances` in that order
            if constexpr (mode_ak == scratch_global_k) {
                fresh = visits_set<mode_ak>(visits, visits_mask, neighbor, present);
            } else {
                std::uint32_t h = hash_slot(neighbor) & visits_mask;
                std::uint32_t old = neighbor; // an absent lane probes nothing
                bool asks = present;
                std::uint32_t seen_cell = 0;
                if constexpr (seen_ak) {
                    if (seen) { // a slot this query has probed before is in the set: no atomic for it
                        seen_cell = ((neighbor * 0x9E3779B1u) >> 9) & seen_mask;
                        asks = present && seen[seen_cell] != neighbor;
                    }
                }
                // How the slab is probed (`probe_mode_t`, short rows only; the engine's choice rides in `args.probe_mode`):
                //  * `probe_swap_k` — a compare-and-swap per probe round, executed at the memory side.
                //  * `probe_load_first_k` (round 5's experiment) — a cell is LOADED and an atomic spent only to claim an empty one: a slot
                //    already in the set costs no atomic, a fresh one a load and the atomic — two round trips on the hop's chain.
                //  * `probe_plain_k` — NO atomic. The slab is private to this wave, so the only accesses that can race are those of the
                //    lanes of one instruction: a cell is loaded past the vector cache (`sc1`: nothing is allocated there per lane), an
                //    empty one is claimed with a plain store nobody waits for, and lanes of one round that want the SAME cell settle
                //    it in LDS — one bit per cell, `ds_or_rtn`: LDS serves the lanes of an instruction one after the other, the first
                //    to set the bit has the cell, the others look at it again next round and read the winner's slot (a wave's accesses
                //    to one address are served in issue order). The words touched go back to zero behind the ORs. A list holds every
                //    slot once (duplicates were removed on upload), so no two lanes of a round insert the same slot. One round trip
                //    per probe round, same members in the set, same answers (which CELL a member lands in is not observable).
                bool shadow_done = false;
                auto shadow_work = [&]() {
                    if (shadow_done)
                        return;
                    shadow_done = true;
                // ---- in the shadow of the probe's round trip (short rows, round 6). The walk of a short-row index is a chain of
                //      dependent round trips per hop — list, probe of the visited set, rows, commit — and not bytes; what does not
                //      NEED the probe's answer goes in front of the wait for it:
                //       * rows that arrived with the list (`nbr0_rows`): every lane measures its neighbour now, fresh or not;
                //       * rows of ≤ 128 bytes (G = 2; `args.early_rows`): the rows of every neighbour the probe asks about are
                //         gathered NOW, next to the probe instead of behind it — one dependent round trip less per hop for about
                //         two thirds more row traffic (four asked-about neighbours in ten turn out visited), on a walk that sits
                //         at a quarter of the memory's bandwidth. Distances of visited neighbours are dropped; `computed` counts
                //         the fresh ones as the reference does.
                if constexpr (inline_ak) {
                    if (inline_rows) {
                        partial_t p;
                        accumulate_chunk<metric_ak, scalar_ak>(p, query_lds, 0, inline_row);
                        early_mine = finalize_distance<metric_ak, scalar_ak>(p, a2, ix.dimensions);
                        early_done = true;
                    }
                }
                if constexpr (early_ak) {
                    if (early_rows && tile == 0) {
                        asked_mask = ballot(asks);
                        const std::uint32_t asked = popcount64(asked_mask);
                        if (asks)
                            cand_slots[rank_below(asked_mask, lane)] = neighbor;
                        wave_sync<false>();
                        if (asked)
                            measure_rows<metric_ak, scalar_ak, lanes_ak, loads_ak, false, 1>(ix, query_lds, a2, cand_slots, cand_distances, asked);
                        measured_early = true;
                    }
                }
                };
                std::uint32_t probe_mode = probe_swap_k;
#ifdef USEARCH_AMD_EXPERIMENT_PROBE_MODES // modes 1 and 2 measured slower (profiles/r06_short_rows/): product builds compile them away
                if constexpr (seen_ak)
                    probe_mode = args.probe_mode;
#endif
                const bool load_first = probe_mode == probe_load_first_k;
                if (probe_mode == probe_plain_k) {
                    std::uint32_t* claim = reinterpret_cast<std::uint32_t*>(query_lds + args.claim_offset);
                    const std::uint32_t claim_mask = args.claim_bits - 1;
                    bool looking = asks;
                    for (;;) {
                        std::uint32_t cell_value = none_slot_k;
                        if (looking)
                            cell_value = __hip_atomic_load(visits + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!popped)
                            pop_now(); // LDS and register work in the shadow of the first round trip
                        shadow_work();
                        const bool wants = looking && cell_value == none_slot_k;
                        if (looking && cell_value == neighbor)
                            looking = false; // in the set (`old` still names the slot itself: not fresh)
                        if (ballot(wants)) {
                            const std::uint32_t bit = h & claim_mask;
                            bool won = false;
                            if (wants) {
                                const std::uint32_t before = __hip_atomic_fetch_or(claim + (bit >> 5), 1u << (bit & 31u), __ATOMIC_RELAXED,
                                                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                                won = ((before >> (bit & 31u)) & 1u) == 0u;
                                claim[bit >> 5] = 0u;
                            }
                            if (won) {
                                __hip_atomic_store(visits + h, neighbor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                old = none_slot_k, looking = false;
                            }
                        }
                        if (looking && !wants)
                            h = (h + 1) & visits_mask; // linear probing, index.hpp:1085-1211
                        if (!ballot(looking))
                            break;
                    }
                } else {
                if (asks) {
                    if (load_first) {
                        old = __hip_atomic_load(visits + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old == none_slot_k)
                            old = atomicCAS(visits + h, none_slot_k, neighbor);
                    } else {
                        old = atomicCAS(visits + h, none_slot_k, neighbor);
                    }
                }
                if (!popped)
                    pop_now();
                shadow_work();
                if constexpr (aside_ak) {
                    // the home cell belongs to another member: this one lives in `aside` (see above) — or moves in now
                    bool taken = asks && old != none_slot_k && old != neighbor;
                    const std::uint64_t taken_mask = ballot(taken);
                    if (taken_mask) {
                        if (aside_count + popcount64(taken_mask) > aside_mask - aside_mask / 4) { // 75 % of the cells
                            overflow = true;
                            break;
                        }
                        std::uint32_t cell = ((neighbor * 0x9E3779B1u) >> 11) & aside_mask;
                        do {
                            std::uint32_t there = neighbor;
                            if (taken)
                                there = atomicCAS(aside + cell, none_slot_k, neighbor);
                            if (taken && (there == none_slot_k || there == neighbor))
                                old = there, taken = false; // moved in (fresh) / found (visited)
                            cell = (cell + 1) & aside_mask;
                        } while (ballot(taken));
                        aside_count += popcount64(ballot(lane_bit(taken_mask) && old == none_slot_k));
                    }
                }
                while (!aside_ak && old != none_slot_k && old != neighbor) { // linear probing, index.hpp:1085-1211
                    h = (h + 1) & visits_mask;
                    if (load_first) {
                        old = __hip_atomic_load(visits + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old == none_slot_k)
                            old = atomicCAS(visits + h, none_slot_k, neighbor);
                    } else {
                        old = atomicCAS(visits + h, none_slot_k, neighbor);
                    }
                }
                }
                fresh = present && old == none_slot_k;
                if constexpr (seen_ak) {
                    if (seen && asks)
                        seen[seen_cell] = neighbor; // probed (inserted or found): in the set from now on
                }
            }
            const std::uint64_t fresh_mask = ballot(fresh);
            tick(2);
            const std::uint32_t count = popcount64(fresh_mask);
            visits_count += count;
            if (!count)
                continue;
            float mine = 0.f;
            std::uint32_t mine_slot = 0u;
            bool candidate; // this lane holds a measured newcomer; lanes are in list order either way
            if (inline_rows) {
                // the row arrived with the list: every fresh lane measures its own neighbour, nothing is staged or gathered
                if constexpr (inline_ak) {
                    if (early_done) {
                        mine = early_mine; // measured while the probe was in flight
                    } else {
                        partial_t p;
                        accumulate_chunk<metric_ak, scalar_ak>(p, query_lds, 0, inline_row);
                        mine = finalize_distance<metric_ak, scalar_ak>(p, a2, ix.dimensions);
                    }
                }
                mine_slot = neighbor;
                candidate = fresh;
                computed += count;
            } else if (measured_early) {
                // the rows were gathered next to the probe: a fresh lane picks its distance up where the gather left it
                mine = fresh ? cand_distances[rank_below(asked_mask, lane)] : 0.f;
                mine_slot = neighbor;
                candidate = fresh;
                computed += count;
                wave_sync<false>();
            } else {
                if (fresh)
                    mem::store(cand_slots + rank_below(fresh_mask, lane), neighbor); // keeps list order
                wave_sync<global_ak>();
                measure_hop(count);
                mine = lane < count ? mem::load(cand_distances + lane) : 0.f;
                mine_slot = lane < count ? mem::load(cand_slots + lane) : 0u;
                candidate = lane < count;
            }
            tick(3);

            // commit in list order with the reference's tests (index.hpp:4233-4240)
            std::uint64_t pending = ballot(candidate && (top.size < ef || mine < radius)); // radius only shrinks
            while (pending) {
                const std::uint32_t i = (std::uint32_t)__ffsll((long long)pending) - 1;
                pending &= pending - 1;
                const float d = read_lane_f32(mine, i);
#ifdef USEARCH_AMD_PHASES
                ++diagnostic_rechecks;
#endif
                if (!(top.size < ef || d < radius))
                    continue;
                const std::uint32_t successor = read_lane_u32(mine_slot, i);
#ifdef USEARCH_AMD_PHASES
                ++diagnostic_pushes;
                const std::uint64_t t0 = __builtin_amdgcn_s_memtime();
#endif
                // the frontier lives in LDS, `top` in registers: the insert runs while the push's reads are in flight
                push_ticket_t ticket;
                if constexpr (!in_top_ak)
                    ticket = heap_push_begin<global_ak>(next, next_size);
#ifdef USEARCH_AMD_PHASES
                const std::uint64_t t1 = __builtin_amdgcn_s_memtime();
#endif
                if (in_top_ak || allowed(successor))
                    top.insert(d, successor, ef, radius); // radius = top.top() once full; the new cell is open
                if constexpr (!in_top_ak)
                    heap_push_finish<global_ak>(next, next_size, ticket, -d, successor);
#ifdef USEARCH_AMD_PHASES
                const std::uint64_t t2 = __builtin_amdgcn_s_memtime();
                diagnostic_push_ticks += t1 - t0, diagnostic_insert_ticks += t2 - t1;
#endif
            }
            peak_next = next_size > peak_next ? next_size : peak_next;
            wave_sync<global_ak>();
            tick(4);
        }
        if (overflow)
            break;
        if (!popped) // a node without neighbours: nothing was probed
            pop_now();
    }

    // ---- results: shrink to `wanted`, dump_to with padding (index.hpp:3067-3073, 2707-2722)
    if (overflow) {
        if (lane == 0) {
            args.status[q] = status_overflow_k;
            atomicAdd(args.queue + 1, 1u); // the host reads this one counter before it looks at any status
        }
        return false;
    }
    const std::uint32_t found = top.size < wanted ? top.size : wanted;
    top.dump(ix, args, q, found, wanted);
    tick(5);
#ifdef USEARCH_AMD_PHASES
    if (args.phases && lane == 0) {
#pragma unroll
        for (int phase = 0; phase < 6; ++phase)
            atomicAdd(args.phases + phase, (unsigned long long)phase_ticks[phase]);
        atomicAdd(args.phases + 6, (unsigned long long)diagnostic_pushes);
        atomicAdd(args.phases + 7, (unsigned long long)diagnostic_ready);
        atomicAdd(args.phases + 8, (unsigned long long)diagnostic_push_ticks);
        atomicAdd(args.phases + 9, (unsigned long long)diagnostic_insert_ticks);
        atomicAdd(args.phases + 10, (unsigned long long)diagnostic_rechecks);
    }
#endif
    if (lane == 0) {
        args.counts[q] = found;
        args.visited[q] = cycles;
        args.computed[q] = computed;
        args.status[q] = status_done_k;
        if (args.peaks) { // scratch-sizing telemetry: how big `next` and `visits` got
            args.peaks[2 * (std::uint64_t)q] = peak_next;
            args.peaks[2 * (std::uint64_t)q + 1] = visits_count;
        }
    }
    return true;
}

/**
 *  Persistent launch: every wave (= workgroup) pulls query indices from `args.queue` until the batch is drained, so a
 *  grid of (CUs × resident waves) covers any batch size with perfect dynamic balance and one scratch slab per wave.
 *  `scratch_global_k` is the exception: one wave per query, statically (its bitmaps are zeroed per launch by the host).
 */
/**
 *  Register/latency trade-off of one instantiation: how many 16-byte loads a lane keeps in flight inside a row
 *  (`unroll`) against how many waves per SIMD the register allocator must leave room for (`waves`).
 */
enum kernel_variant_t : int {
    variant_u4_w4_k = 0,    ///< 4 loads in flight per lane
    variant_u8_w3_k = 1,    ///< 8 loads in flight (8 loads under 128 VGPRs spills: measured 2× slower)
    variant_u12_w2_k = 2,   ///< 12 loads in flight (a whole 768-d f16 row per lane group)
    variant_u12x2_w2_k = 3, ///< two rows per lane group per round, 12 loads each: 24 in flight (frontier_top_k builds only)
    variant_count_k = 4,
};
/// Waves per SIMD the smallest builds are cut for — `top` of one cell per lane (expansion ≤ 64), 4 loads in flight, rows of one
/// 16-byte chunk (G = 1) or of ≤ 128 bytes (G = 2). Their hops are a chain of short dependent round trips (list, visited set, a
/// few rows, the commit), so what pays is hops in flight. 20 M × 128 b1: 9.9 → 10.9 → 11.7 M QPS at 4 → 5 → 6 waves per SIMD
/// (128 → 96 → 80 VGPRs); 20 M × 96 i8: 9.2 → 10.8 M QPS at 5, 9.6 M at 6 (the spills win) — profiles/r02_short_rows.log.
#ifndef USEARCH_AMD_TINY_ROW_WAVES
#define USEARCH_AMD_TINY_ROW_WAVES 6
#endif
#ifndef USEARCH_AMD_SHORT_ROW_WAVES
#define USEARCH_AMD_SHORT_ROW_WAVES 5
#endif
#ifndef USEARCH_AMD_PLAIN_TINY_ROW_WAVES // the cut for plain batches (`plain_ak`) needs fewer registers: its own residency
#define USEARCH_AMD_PLAIN_TINY_ROW_WAVES USEARCH_AMD_TINY_ROW_WAVES
#endif
constexpr int variant_unroll(int v) { return v == variant_u4_w4_k ? 4 : v == variant_u8_w3_k ? 8 : 12; }
constexpr int variant_rows(int v) { return v == variant_u12x2_w2_k ? 2 : 1; }
/// Waves per SIMD the register budget of an instantiation is cut for (512 VGPRs per SIMD lane: 128 → 4, 168 → 3, 256 → 2);
/// from the allocations the compiler reports for the widest rows (cos, G = 8) with `top` in `epl` register rows.
constexpr int kernel_waves(int variant, int epl, int frontier = 0, int lanes = 8, bool plain = false) {
    if (variant == variant_u12x2_w2_k)
        return 2;
    if (plain && variant == variant_u4_w4_k && epl == 1 && lanes == 1)
        return USEARCH_AMD_PLAIN_TINY_ROW_WAVES;
    if (variant == variant_u4_w4_k && epl == 1 && lanes <= 2)
        return lanes == 1 ? USEARCH_AMD_TINY_ROW_WAVES : USEARCH_AMD_SHORT_ROW_WAVES;
    if (variant == variant_u4_w4_k && epl == 2 && lanes <= 2) // expansion 65 … 128 on short rows (C4 needs 80)
        return USEARCH_AMD_SHORT_ROW_WAVES;
    if (frontier) // without the heap's bookkeeping the 4-deep build fits 128 registers with any `top`
        return variant == variant_u4_w4_k ? 4 : variant == variant_u8_w3_k ? 3 : 2;
    return variant == variant_u4_w4_k ? (epl >= 8 ? 3 : 4) : variant == variant_u8_w3_k ? (epl >= 16 ? 2 : 3) : 2;
}

template <int metric_ak, int scalar_ak, int lanes_ak, int variant_ak, int mode_ak, int epl_ak, int frontier_ak, bool plain_ak = false>
__global__ __launch_bounds__(64, kernel_waves(variant_ak, epl_ak, frontier_ak, lanes_ak, plain_ak)) void search_kernel(const snapshot_view_t ix,
                                                                                            const search_args_t args) {
    constexpr int unroll_ak = variant_unroll(variant_ak) + 100 * (variant_rows(variant_ak) - 1); // rows ride in the hundreds
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    std::uint8_t* query_lds = lds;
    const std::uint32_t query_bytes = query_lds_bytes<scalar_ak>(ix.chunks);
    std::uint8_t* slab = args.scratch + (std::uint64_t)blockIdx.x * args.scratch_stride;
    const scratch_layout_t layout = scratch_layout(epl_ak ? 0 : args.ef, frontier_ak == frontier_top_k ? 0 : args.next_cap, 0);
    // batch-tail telemetry: when every wave of the launch started and left (100 MHz wall clock, comparable across the chip)
    if (args.wave_clock && lane_id() == 0)
        args.wave_clock[2 * (std::uint64_t)blockIdx.x] = __builtin_amdgcn_s_memrealtime();

    if constexpr (mode_ak == scratch_global_k) {
        const std::uint32_t q = args.todo ? args.todo[blockIdx.x] : blockIdx.x;
        search_one<metric_ak, scalar_ak, lanes_ak, unroll_ak, mode_ak, epl_ak, frontier_ak>(
            ix, args, q, query_lds, slab, reinterpret_cast<std::uint32_t*>(slab + layout.visits));
    } else {
        std::uint8_t* heaps = lds + query_bytes;
        std::uint32_t* visits = mode_ak == scratch_lds_k ? reinterpret_cast<std::uint32_t*>(heaps + layout.visits)
                                                         : reinterpret_cast<std::uint32_t*>(slab);
        for (;;) {
            std::uint32_t ticket = 0;
            if (lane_id() == 0)
                ticket = atomicAdd(args.queue, 1u);
            ticket = uniform_u32(ticket);
            if (ticket >= args.count)
                break;
            const std::uint32_t q = args.todo ? args.todo[ticket] : ticket;
            search_one<metric_ak, scalar_ak, lanes_ak, unroll_ak, mode_ak, epl_ak, frontier_ak, 1, plain_ak>(ix, args, q, query_lds, heaps,
                                                                                                             visits);
            wave_sync<false>();
        }
    }
    if (args.wave_clock && lane_id() == 0)
        args.wave_clock[2 * (std::uint64_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

/**
 *  FIVE waves per query, for batches too small to fill the chip with one wave each (a `usearch_search` caller's single query
 *  above all) over long rows: wave 0 walks — pop, list, visited set, ordered commit, `top` in its registers — and the others take
 *  the one step a lone wave is slow at: the hop's ≤ M0 rows, whole rows per lane group (`team_share`), two workgroup barriers per
 *  hop. With a wide `top` over the in-`top` frontier the leader does not measure at all: it commits the hop before while the four
 *  helpers measure this one (`search_one`, the pipelined beam); otherwise waves 0 … 3 measure a quarter each and the fifth only
 *  keeps the barriers company. Same distances bit for bit, same order of commits, same counters. The helpers share the leader's
 *  LDS image of the query and its gather list.
 */
template <int metric_ak, int scalar_ak, int lanes_ak, int variant_ak, int mode_ak, int epl_ak, int frontier_ak>
__global__ __launch_bounds__(64 * team_waves_k) void team_search_kernel(const snapshot_view_t ix, const search_args_t args) {
    static_assert(mode_ak != scratch_global_k, "the team walks with its heaps in LDS");
    constexpr int team_ak = team_waves_k;
    constexpr int loads_ak = variant_unroll(variant_ak);
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    std::uint8_t* query_lds = lds;
    const std::uint32_t query_bytes = query_lds_bytes<scalar_ak>(ix.chunks);
    const scratch_layout_t layout = scratch_layout(epl_ak ? 0 : args.ef, frontier_ak == frontier_top_k ? 0 : args.next_cap, 0);
    std::uint8_t* heaps = lds + query_bytes;
    team_t* team = reinterpret_cast<team_t*>(lds + args.team_offset);
    const std::uint32_t wave = threadIdx.x / 64;
    if (wave == 0) {
        std::uint8_t* slab = args.scratch + (std::uint64_t)blockIdx.x * args.scratch_stride;
        std::uint32_t* visits = mode_ak == scratch_lds_k ? reinterpret_cast<std::uint32_t*>(heaps + layout.visits)
                                                         : reinterpret_cast<std::uint32_t*>(slab);
        for (;;) {
            std::uint32_t ticket = 0;
            if (lane_id() == 0)
                ticket = atomicAdd(args.queue, 1u);
            ticket = uniform_u32(ticket);
            if (ticket >= args.count)
                break;
            const std::uint32_t q = args.todo ? args.todo[ticket] : ticket;
            search_one<metric_ak, scalar_ak, lanes_ak, loads_ak, mode_ak, epl_ak, frontier_ak, team_ak>(ix, args, q, query_lds, heaps,
                                                                                                       visits, team);
            wave_sync<false>();
        }
        if (lane_id() == 0)
            team->count = team_exit_k;
        team_barrier();
    } else {
        const std::uint32_t* cand_slots = reinterpret_cast<const std::uint32_t*>(heaps + layout.cand_slots);
        float* cand_distances = reinterpret_cast<float*>(heaps + layout.cand_distances);
        // whose turn a helper has when the leader stays out: the waves of the other SIMDs first, the leader's SIMD-mate (waves go
        // to the four SIMDs round-robin) last — it gets rows only when a hop gathers more than the others take in one round each,
        // and that matters: the leader's commit is all vector ALU work
        const std::uint32_t turn = wave - 1;
        team_landing_t landing; // where the warm-up's loads land (team_touch_rows)
        std::uint32_t* ahead = team_ahead_list(team);
        if (lane_id() < team_ahead_rows_k / (team_waves_k - 1))
            ahead[turn * (team_ahead_rows_k / (team_waves_k - 1)) + lane_id()] = none_slot_k; // until a pipelining leader publishes a list
#ifdef USEARCH_AMD_PHASES // every helper's clock: what it spent between the two barriers of a hop
        std::uint64_t helper_mark = args.phases ? __builtin_amdgcn_s_memtime() : 0, helper_busy = 0, helper_idle = 0, helper_hops = 0;
#endif
        for (;;) {
            team_barrier(); // the leader has published a hop (or the end)
#ifdef USEARCH_AMD_PHASES
            if (args.phases) {
                const std::uint64_t now = __builtin_amdgcn_s_memtime();
                helper_idle += now - helper_mark, helper_mark = now;
            }
#endif
            const std::uint32_t count = uniform_u32(team->count);
            if (count == team_exit_k)
                break;
            const query_norm_t a2 = team->a2;
            if (uniform_u32(team->leader_in)) {
                if (wave < 4)
                    team_share<metric_ak, scalar_ak, lanes_ak, loads_ak>(ix, query_lds, a2, cand_slots, cand_distances, count, wave, 4);
            } else
                team_share<metric_ak, scalar_ak, lanes_ak, loads_ak>(ix, query_lds, a2, cand_slots, cand_distances, count, turn,
                                                                     team_ak - 1, true);
#ifdef USEARCH_AMD_PHASES
            if (args.phases) {
                const std::uint64_t now = __builtin_amdgcn_s_memtime();
                helper_busy += now - helper_mark, helper_mark = now, ++helper_hops;
            }
#endif
            team_landing_alive(landing); // past the gather's own waits
            team_barrier(); // every share is in LDS: the leader commits
            team_touch_rows(ix, ahead, turn, landing);
        }
#ifdef USEARCH_AMD_PHASES
        if (args.phases && lane_id() == 0) { // [11 … 14] what each helper spent measuring, [15] the first one's hops
            atomicAdd(args.phases + 10 + wave, (unsigned long long)helper_busy);
            if (wave == 1)
                atomicAdd(args.phases + 15, (unsigned long long)helper_hops);
        }
#endif
    }
}

/**
 *  Plain distance evaluation, one wave per query row: out[q][j] = metric(query q, row slots[q][j]). Serves `usearch_distance`-
 *  style checks of the arithmetic alone and is the building block of exact search.
 */
template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak>
__global__ __launch_bounds__(64) void distances_kernel(const snapshot_view_t ix, const std::uint8_t* queries,
                                                       std::uint64_t query_stride, const std::uint32_t* slots,
                                                       std::uint32_t slots_per_query, float* out) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    const std::uint32_t lane = lane_id();
    const std::uint32_t q = blockIdx.x;
    std::uint8_t* query_lds = lds;
    std::uint32_t* cand_slots = reinterpret_cast<std::uint32_t*>(lds + query_lds_bytes<scalar_ak>(ix.chunks));
    float* cand_distances = reinterpret_cast<float*>(cand_slots + 64);
    const query_norm_t a2 = stage_query<metric_ak, scalar_ak, lanes_ak>(ix, queries + (std::uint64_t)q * query_stride, query_lds);
    for (std::uint32_t base = 0; base < slots_per_query; base += 64) {
        const std::uint32_t count = slots_per_query - base < 64 ? slots_per_query - base : 64;
        if (lane < count)
            cand_slots[lane] = slots[(std::uint64_t)q * slots_per_query + base + lane];
        wave_sync<false>();
        measure_rows<metric_ak, scalar_ak, lanes_ak, unroll_ak, false>(ix, query_lds, a2, cand_slots, cand_distances, count);
        if (lane < count)
            out[(std::uint64_t)q * slots_per_query + base + lane] = cand_distances[lane];
        wave_sync<false>();
    }
}

/**
 *  Exact (brute-force) search — `index_gt::search_exact_` (index.hpp:4252-4268): every allowed slot in slot order goes
 *  through `top.insert(candidate, wanted)`. One wave owns one (query, row-partition) pair and streams its rows with the
 *  same `measure_rows` loop as the graph search (so distances are bit-identical to it); partitions are folded afterwards
 *  by `merge_kernel`. Because inserts use lower_bound, the result is the top-`wanted` under (distance ↑, slot ↓).
 *  out_* are laid out [partition][query][wanted]; `map_keys` = 0 returns slots (dataset offsets) instead of keys.
 */
template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak>
__global__ __launch_bounds__(64) void exact_kernel(const snapshot_view_t ix, const std::uint8_t* queries,
                                                   std::uint64_t query_stride, std::uint32_t query_count,
                                                   std::uint32_t wanted, std::uint64_t rows_per_partition,
                                                   std::uint32_t map_keys, const std::uint32_t* allow_bits,
                                                   float* out_distances, std::uint64_t* out_keys,
                                                   std::uint64_t* out_counts) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    const std::uint32_t lane = lane_id();
    const std::uint32_t q = blockIdx.x, partition = blockIdx.y;
    std::uint8_t* query_lds = lds;
    std::uint32_t* cand_slots = reinterpret_cast<std::uint32_t*>(lds + query_lds_bytes<scalar_ak>(ix.chunks));
    float* cand_distances = reinterpret_cast<float*>(cand_slots + 64);
    top_gt<0, false> top;
    top.reset(reinterpret_cast<cand_t*>(cand_distances + 64));
    const query_norm_t a2 = stage_query<metric_ak, scalar_ak, lanes_ak>(ix, queries + (std::uint64_t)q * query_stride, query_lds);

    const std::uint64_t first = (std::uint64_t)partition * rows_per_partition;
    const std::uint64_t last = first + rows_per_partition < ix.size ? first + rows_per_partition : ix.size;
    float worst = 0.f;
    for (std::uint64_t base = first; base < last; base += 64) {
        const std::uint32_t span = last - base < 64 ? (std::uint32_t)(last - base) : 64u;
        // the `allow` predicate of index_dense.hpp:2071-2081 (tombstones, then the caller's predicate as one bit per slot —
        // `search_exact_` skips `!predicate(member)`, index.hpp:4260-4263), then compaction in slot order
        const std::uint32_t slot = (std::uint32_t)base + lane;
        bool allowed = lane < span && (!ix.has_tombstones || ix.keys[slot] != free_key_k);
        if (allow_bits && allowed)
            allowed = ((allow_bits[slot >> 5] >> (slot & 31)) & 1u) != 0;
        const std::uint64_t allowed_mask = ballot(allowed);
        const std::uint32_t count = popcount64(allowed_mask);
        if (!count)
            continue;
        if (allowed)
            cand_slots[rank_below(allowed_mask, lane)] = slot;
        wave_sync<false>();
        measure_rows<metric_ak, scalar_ak, lanes_ak, unroll_ak, false>(ix, query_lds, a2, cand_slots, cand_distances, count);
        const float mine = lane < count ? cand_distances[lane] : 0.f;
        const std::uint32_t mine_slot = lane < count ? cand_slots[lane] : 0u;
        // an element equal to a full buffer's worst still gets in (lower_bound lands before it): only `>` is hopeless
        std::uint64_t pending = ballot(lane < count && (top.size < wanted || !(mine > worst)));
        while (pending) {
            const std::uint32_t i = (std::uint32_t)__ffsll((long long)pending) - 1;
            pending &= pending - 1;
            const float d = read_lane_f32(mine, i);
            if (top.size == wanted && d > worst)
                continue;
            top.insert(d, read_lane_u32(mine_slot, i), wanted, worst);
        }
        wave_sync<false>();
    }
    const std::uint64_t row = ((std::uint64_t)partition * query_count + q) * wanted;
    for (std::uint32_t i = lane; i < wanted; i += 64) {
        std::uint64_t key = 0;
        std::uint32_t bits = signaling_nan_bits_k;
        if (i < top.size) {
            const cand_t c = top.cells[i];
            key = map_keys ? ix.keys[cand_slot(c)] : (std::uint64_t)cand_slot(c);
            bits = (std::uint32_t)c;
        }
        out_keys[row + i] = key;
        reinterpret_cast<std::uint32_t*>(out_distances)[row + i] = bits;
    }
    if (lane == 0)
        out_counts[(std::uint64_t)partition * query_count + q] = top.size;
}

} // namespace usearch_amd
