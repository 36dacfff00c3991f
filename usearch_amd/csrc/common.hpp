/**
 *  usearch_amd/csrc/common.hpp — shared host/device types of the MI355X search engine.
 *
 *  Enumerator VALUES are the reference's on-disk ones (/root/reference/include/usearch/index_plugins.hpp:113-159),
 *  because the engine is fed from serialized `.usearch` v2 images (index_dense.hpp:42-79 stores them as raw bytes).
 */
#pragma once
#include <cstddef>
#include <cstdint>

namespace usearch_amd {

enum metric_kind_t : std::uint8_t {
    metric_unknown_k = 0,
    metric_ip_k = 'i',
    metric_cos_k = 'c',
    metric_l2sq_k = 'e',
    metric_hamming_k = 'b',
    metric_pearson_k = 'p',
    metric_haversine_k = 'h',
    metric_divergence_k = 'd',
    metric_jaccard_k = 'j',
    metric_tanimoto_k = 't',
    metric_sorensen_k = 's',
};

enum scalar_kind_t : std::uint8_t {
    scalar_unknown_k = 0,
    scalar_b1x8_k = 1,
    scalar_bf16_k = 4,
    scalar_f64_k = 10,
    scalar_f32_k = 11,
    scalar_f16_k = 12,
    scalar_i8_k = 23,
    scalar_u64_k = 14, // key kind of index_dense_t (index_dense.hpp:2229)
    scalar_u32_k = 15, // compressed slot kind of index_dense_t
};

/// Bytes one stored vector occupies (index_plugins.hpp:1853-1855: bits_per_scalar * dimensions, rounded up to bytes).
inline std::size_t bytes_per_vector(scalar_kind_t kind, std::size_t dimensions) {
    switch (kind) {
    case scalar_b1x8_k: return (dimensions + 7) / 8;
    case scalar_i8_k: return dimensions;
    case scalar_f16_k:
    case scalar_bf16_k: return dimensions * 2;
    case scalar_f32_k: return dimensions * 4;
    case scalar_f64_k: return dimensions * 8;
    default: return 0;
    }
}

/// LDS bytes a staged query takes per 16-byte chunk of a stored row: the 16-bit float kinds are widened to f32 once.
inline constexpr std::uint32_t query_chunk_bytes_of(scalar_kind_t kind) {
    return (kind == scalar_f16_k || kind == scalar_bf16_k) ? 32u : 16u;
}

/// Pairs that get every build of the search kernel (three loads-in-flight depths for long rows): the BASELINE configs'
/// and their neighbours. The other pairs of the reference's dispatch table (index_plugins.hpp:1930-2008) get the 4-deep
/// build only — same results, one third of the compile time.
inline constexpr bool all_kernel_builds(metric_kind_t metric, scalar_kind_t scalar) {
    const bool common_metric = metric == metric_ip_k || metric == metric_cos_k || metric == metric_l2sq_k;
    return (common_metric && (scalar == scalar_f32_k || scalar == scalar_f16_k || scalar == scalar_bf16_k ||
                              scalar == scalar_i8_k)) ||
           (metric == metric_hamming_k && scalar == scalar_b1x8_k);
}

constexpr std::uint32_t none_slot_k = 0xFFFFFFFFu;             ///< empty neighbour cell / empty hash cell
constexpr std::uint64_t free_key_k = 0xFFFFFFFFFFFFFFFFull;    ///< tombstone key, index_dense.hpp:513
constexpr std::uint32_t signaling_nan_bits_k = 0x7FA00000u;    ///< tail padding of distances, index.hpp:2717-2719
constexpr std::size_t default_expansion_search_k = 64;         ///< index.hpp:3029-3030

/**
 *  Immutable HBM snapshot of one index, as the kernels see it (passed by value).
 *
 *  Layout (all arrays `hipMalloc`ed, slot numbering identical to the reference's so that labels and tie-breaks are
 *  comparable — replaces the pointer-chased `nodes_[slot]` → tape / `vectors_lookup_[slot]` of index.hpp:2280 and
 *  index_dense.hpp:456-460):
 *    vectors   [size][row_stride] bytes   row = the stored vector, zero padded to a multiple of 16*G bytes
 *    nbr0      [size][m0] u32             level-0 neighbour list in the reference's order, later duplicates removed,
 *                                         unused cells = none_slot_k (the u32 count of index.hpp:2148-2195 is implied)
 *    upper_ref [size] u32                 index of the node's level-1 list inside `upper`, none_slot_k for level-0 nodes
 *    upper     [lists][m] u32             lists of levels 1..L of one node are consecutive; same cell convention
 *    keys      [size] u64                 node_t::key (index.hpp:2116-2137)
 *    nbr0_rows [size][m0][16] bytes       optional, rows of ≤ 16 bytes (b1 × 128 …): the stored rows of a node's level-0
 *                                         neighbours next to each other, in list order — the per-hop gather of up to M0
 *                                         scattered 16-byte rows becomes one contiguous 16·M0-byte read (docs/format.md:7-28
 *                                         keeps a node's neighbours together for the same reason)
 */
struct snapshot_view_t {
    const std::uint8_t* vectors;
    const std::uint32_t* nbr0;
    const std::uint32_t* upper_ref;
    const std::uint32_t* upper;
    const std::uint64_t* keys;
    const std::uint8_t* nbr0_rows; ///< optional [size][m0][16]: cell j = a copy of the stored row of nbr0[i][j] (rows of ≤ 16
                                   ///< bytes only: the neighbours' vectors travel with the list, one contiguous block per hop)
    std::uint64_t size;
    std::uint32_t row_stride; ///< bytes, multiple of 16*G
    std::uint32_t chunks;     ///< row_stride / 16
    std::uint32_t bytes_per_vector;
    std::uint32_t dimensions;
    std::uint32_t m, m0;
    std::uint32_t max_level;
    std::uint32_t entry_slot;
    std::uint32_t has_tombstones; ///< any key == free_key_k: the `allow` predicate of index_dense.hpp:2071-2081 must run
};

/** One batch of queries, everything device-resident. */
struct search_args_t {
    const std::uint8_t* queries; ///< storage scalar kind, row `i` at queries + i*query_stride, bytes_per_vector bytes used
    std::uint64_t query_stride;
    const std::uint32_t* todo; ///< optional: query indices this launch handles (retry passes); null = identity
    std::uint32_t count;       ///< launches' grid size (number of queries or of `todo` entries)
    std::uint32_t wanted;      ///< k
    std::uint32_t ef;          ///< max(expansion, wanted), index.hpp:3052
    std::uint64_t* keys;       ///< [Q][wanted]
    float* distances;          ///< [Q][wanted]
    std::uint64_t* counts;     ///< [Q]
    std::uint64_t* visited;    ///< [Q] visited_members  (index.hpp:3071)
    std::uint64_t* computed;   ///< [Q] computed_distances (index.hpp:3072)
    std::uint32_t* status;     ///< [Q] 0 = done, 1 = scratch overflow → rerun with bigger scratch
    std::uint32_t* queue;      ///< one zeroed counter: persistent waves draw query tickets from it
    std::uint32_t* peaks;      ///< optional [Q][2]: peak size of `next`, final size of `visits` (scratch-sizing telemetry)
    std::uint32_t hash_cap;    ///< LDS visited-set cells, power of two
    std::uint32_t next_cap;    ///< frontier heap capacity
    std::uint8_t* scratch;        ///< per-wave global slabs (visited hash, or everything in the fallback mode)
    std::uint64_t scratch_stride; ///< bytes per launched wave
    // index construction reuses the search (search_to_insert_, index.hpp:4011-4079, is the same beam on any level):
    const std::uint32_t* query_ids; ///< optional: query `q` is row query_ids[q] of `queries` (a stored vector); outputs stay at row q
    std::uint32_t beam_level;       ///< level the beam runs on (0 for `search`); the greedy descent stops above it
    std::uint32_t emit_slots;       ///< 1 = write slots instead of keys into `keys`
    std::uint32_t descent_only;     ///< 1 = `index_gt::cluster` (index.hpp:3089-3125): stop after the greedy descent to
                                    ///< `beam_level` and report the member it reached (one result per query)
    const std::uint32_t* allow_bits; ///< optional: one bit per slot, 0 = the caller's predicate rejects that member
                                     ///< (`usearch_filtered_search`, index_dense.hpp:2071-2081)
    /// A predicate the host evaluates LAZILY (`usearch_filtered_search`'s callback, dropin.hip): `known_bits` says which members the
    /// host has been asked about already (their answer is in `allow_bits`); a member the walk wants to admit to `top` that is not
    /// known yet is posted to `ask_slots` / `ask_keys` (cursor `ask_cursor`, room `ask_cap`) and guessed (`guess_threshold`) — the
    /// host evaluates what was asked and runs the query again until a run asks nothing: that run IS the reference's traversal.
    const std::uint32_t* known_bits;
    std::uint32_t* ask_slots;
    std::uint64_t* ask_keys;
    std::uint32_t* ask_cursor;
    std::uint32_t ask_cap;
    std::uint32_t guess_threshold;   ///< an unknown member counts as allowed when slot · 0x9E3779B1 (mod 2³²) ≤ this: the share of "yes" so far
    std::uint32_t exclude_own;       ///< 1 = query q's own stored row (`query_ids[q]`) routes but never becomes a result candidate:
                                     ///< `search_to_update_` (index.hpp:4087-4170), the insertion search of a member that is
                                     ///< being re-linked in place
    unsigned long long* phases;     ///< optional [8] diagnostic: shader-clock ticks per phase summed over all waves
    std::uint32_t team_offset;      ///< team_search_kernel: where in the workgroup's LDS the shared `team_t` sits
    unsigned long long* wave_clock; ///< optional [grid][2] telemetry: 100-MHz wall clock at the start and the exit of every
                                    ///< persistent wave (how long the drain phase of a batch leaves the chip part-idle)
    std::uint32_t seen_offset;      ///< short rows with the visited set in a global slab: where in the wave's LDS the `seen` cells
    std::uint32_t seen_cells;       ///< sit, and how many (a power of two; 0 = none) — see `search_one`
    std::uint32_t early_rows;       ///< rows of ≤ 128 bytes (G = 2): 1 = a hop's rows are gathered next to the probe of the visited set
                                    ///< instead of behind it (`search_one`, the hop loop)
    std::uint32_t probe_mode;       ///< how those walks probe the slab (`probe_mode_t`): compare-and-swap, a load first and the swap
                                    ///< only to claim, or no atomic at all (loads + plain stores, claims settled in LDS)
    std::uint32_t claim_offset;     ///< `probe_plain_k`: where in the wave's LDS the claim bits sit, and how many (a power of two,
    std::uint32_t claim_bits;       ///< ≤ hash_cap; one bit per cell when equal, else cell & (bits − 1))
    std::uint32_t aside_offset;     ///< the cut for plain batches (`plain_ak`): where in the wave's LDS the members sit whose home cell in the
    std::uint32_t aside_cells;      ///< slab was taken, and how many cells (a power of two) — the slab is probed at the home cell only
};

enum : std::uint32_t { status_done_k = 0, status_overflow_k = 1 };

/// How a short-row walk probes its wave-private visited-set slab in global memory (kernels.hpp `search_one`).
enum probe_mode_t : std::uint32_t {
    probe_swap_k = 0,       ///< one compare-and-swap per probe (executed at the memory side: 27.5 G a second chip-wide)
    probe_load_first_k = 1, ///< a load, and the swap only to claim an empty cell (round 5's experiment: two round trips per fresh slot)
    probe_plain_k = 2,      ///< no atomic: a load that bypasses the vector cache, a plain store nobody waits for, and the claims of
                            ///< one instruction's lanes on the same cell settled by an LDS bit per cell
};

} // namespace usearch_amd
