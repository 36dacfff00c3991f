/**
 *  include/usearch_amd.h — C ABI of the MI355X-native batched HNSW search engine (the additive, GPU-specific part).
 *
 *  This is the boundary a binding links against. Types and conventions are those of the reference C ABI
 *  (/root/reference/c/usearch.h:20-62): `usearch_key_t` = uint64, `usearch_distance_t` = float, errors are reported
 *  through `usearch_error_t* error` — the callee stores a pointer to a static NUL-terminated string on failure and leaves
 *  it untouched on success (reference convention, c/usearch.h:24-28, c/test.c:56-59). Scalar kinds use the reference's
 *  C enumerators (c/usearch.h:54-62), not the on-disk values.
 *
 *  A *snapshot* is an immutable HBM-resident copy of one serialized index (`.usearch` v2 image, as produced by the
 *  reference's `usearch_save` / `usearch_save_buffer`, c/usearch.h:162,195): dense aligned vector matrix + fixed-stride
 *  level-0 neighbour rows + compact upper-level lists + keys. Searching it reproduces `usearch_search`
 *  (c/usearch.h:371-374 → c/lib.cpp:398-411 → index_dense.hpp:2053-2085 → index.hpp:3016-3075) for a whole batch of
 *  queries per call — the loop the reference leaves to its callers (cpp/bench.cpp:352-377, python/lib.cpp:261-319).
 *
 *  The reference-compatible 38 `usearch_*` entry points are declared in `include/usearch_c_dropin.h`.
 */
#ifndef USEARCH_AMD_H
#define USEARCH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef USEARCH_AMD_EXPORT
#define USEARCH_AMD_EXPORT __attribute__((visibility("default")))
#endif

typedef void* usearch_amd_snapshot_t;
typedef uint64_t usearch_amd_key_t;       /* = usearch_key_t, c/usearch.h:21 */
typedef float usearch_amd_distance_t;     /* = usearch_distance_t, c/usearch.h:22 */
typedef char const* usearch_amd_error_t;  /* = usearch_error_t, c/usearch.h:28 */

/** Scalar kinds with the values of `usearch_scalar_kind_t` (c/usearch.h:54-62). */
enum {
    usearch_amd_scalar_f32_k = 1,
    usearch_amd_scalar_f64_k = 2,
    usearch_amd_scalar_f16_k = 3,
    usearch_amd_scalar_i8_k = 4,
    usearch_amd_scalar_b1_k = 5,
    usearch_amd_scalar_bf16_k = 6,
};

/** Optional knobs of one batched search; zero-initialise for the defaults. */
typedef struct usearch_amd_tuning_t {
    uint32_t hash_cap;     /**< visited-set cells per query (power of two); 0 = (30 × expansion + 1600) / 0.75, rounded up */
    uint32_t next_cap;     /**< frontier heap capacity per query; 0 = 3 × expansion + 256 */
    uint32_t variant;      /**< kernel build: 0 = auto, 1 = 4 row loads in flight per lane (≤128 VGPRs, 16 waves/CU), 2 = 8 loads
                                (≤168 VGPRs, 12 waves/CU), 3 = 12 loads (≤256 VGPRs, 8 waves/CU); for the in-`top` frontier also
                                4 = two rows per lane group per round, 2 × 12 loads in flight (8 waves/CU) */
    uint32_t mode;         /**< scratch placement: 0 = auto, 1 = visited set in LDS, 2 = visited set in a per-wave global
                                hash (heaps stay in LDS), 3 = everything in global memory with exact sizes (slow) */
    uint32_t waves_per_cu; /**< persistent waves per compute unit; 0 = as many as LDS and registers admit (≤ 16) */
    uint32_t frontier;     /**< who holds the traversal's frontier: 0 = auto, 1 = the reference's binary heap (index.hpp:664-835;
                                its pop order among EQUAL distances), 2 = the not-yet-expanded cells of `top` (no heap at all:
                                same hops, counters and results whenever the distances meeting in the frontier are distinct;
                                float-valued pairs, expansion ≤ 1024, no predicate / tombstones — refused otherwise).
                                Auto = 2 where it applies, else 1. DESIGN.md §3.1 */
    uint32_t wave_clock;   /**< 1 = record when every persistent wave started and left (fills stats.tail_idle / span_ms) */
    uint32_t reserved;
} usearch_amd_tuning_t;

/** What a batched search did, for profiling and tests. */
typedef struct usearch_amd_stats_t {
    uint32_t passes;         /**< kernel launches (1 = all queries fit the first LDS scratch) */
    uint32_t retried_lds;    /**< queries rerun with enlarged LDS scratch */
    uint32_t retried_global; /**< queries rerun with global-memory scratch */
    float kernel_ms;         /**< HIP-event duration of the search launches (device entry point with timing only) */
    uint32_t mode;           /**< scratch placement of the first launch (values of usearch_amd_tuning_t::mode) */
    uint32_t grid;           /**< persistent waves of the first launch */
    uint32_t lds_bytes;      /**< LDS bytes per wave of the first launch */
    uint32_t frontier;       /**< frontier of the first launch (values of usearch_amd_tuning_t::frontier) */
    uint32_t variant;        /**< kernel build of the first launch (values of usearch_amd_tuning_t::variant; 5 = five waves per query,
                                  the build small batches over rows of ≥ 128 bytes get on their own) */
    float tail_idle;         /**< with wave_clock: share of (waves × span) during which waves were not there — batch tail */
    float span_ms;           /**< with wave_clock: first wave start → last wave exit, device wall clock */
    uint32_t top_cells;      /**< cells of `top` per lane in registers (1, 4, 8, 16), 0 = `top` in scratch memory; with mode,
                                  variant and frontier this names the kernel instantiation of the first launch */
    uint32_t probe_mode;     /**< short rows over a global visited-set slab: 0 = one compare-and-swap per probe, 1 = a load first and the
                                  swap only to claim, 2 = no atomic at all (loads, plain stores, claims settled by bits in LDS) */
    uint32_t seen_cells;     /**< … `seen` cells in LDS in front of the slab */
    uint32_t claim_bits;     /**< … claim bits in LDS (probe_mode 2) */
    uint32_t early_rows;     /**< rows of ≤ 128 bytes: 1 = a hop's rows were gathered next to the probe of the visited set, not behind it */
    uint32_t plain;          /**< rows of ≤ 128 bytes: 1 = the launch ran the kernel build cut for plain batches (level 0, no predicate, no
                                  tombstones, lists of ≤ 64 cells); USEARCH_AMD_NO_PLAIN=1 keeps the general build. Same results */
    uint32_t aside_cells;    /**< … and its LDS cells for the members whose home cell in the slab was taken (the slab is then probed at the
                                  home cell only: one round trip per hop) */
} usearch_amd_stats_t;

/** Number of visible HIP devices; 0 (and an error) when the runtime finds none. */
USEARCH_AMD_EXPORT int usearch_amd_device_count(usearch_amd_error_t* error);

/**
 *  Uploads a serialized index image to `device`. Replaces `usearch_load_buffer` (c/usearch.h:204-205, c/lib.cpp:244-250)
 *  for the search path: the image is parsed on the host and flattened; it is not referenced after the call returns.
 */
USEARCH_AMD_EXPORT usearch_amd_snapshot_t usearch_amd_snapshot_from_buffer(void const* image, size_t length, int device,
                                                                           usearch_amd_error_t* error);
/**
 *  An image saved with `exclude_vectors` (`serialization_config_t`, index_dense.hpp:1004: the graph alone, the file starts with the
 *  64-byte head) together with the vectors the caller kept: `vectors` = one row per member in slot order, rows
 *  `vectors_stride` bytes apart (0 = dense), in the index's scalar kind. An image that carries its matrix loads here too (the
 *  external rows are ignored then).
 */
USEARCH_AMD_EXPORT usearch_amd_snapshot_t usearch_amd_snapshot_from_parts(void const* graph, size_t graph_length,
                                                                          void const* vectors, size_t vectors_stride,
                                                                          int device, usearch_amd_error_t* error);
/** Same from a `.usearch` file (memory-mapped during the call). Replaces `usearch_load` (c/usearch.h:170). */
USEARCH_AMD_EXPORT usearch_amd_snapshot_t usearch_amd_snapshot_from_file(char const* path, int device,
                                                                         usearch_amd_error_t* error);
USEARCH_AMD_EXPORT void usearch_amd_snapshot_free(usearch_amd_snapshot_t snapshot, usearch_amd_error_t* error);

/** Introspection, mirroring `usearch_size/dimensions/connectivity` (c/usearch.h:231-252). */
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_size(usearch_amd_snapshot_t snapshot);
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_dimensions(usearch_amd_snapshot_t snapshot);
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_connectivity(usearch_amd_snapshot_t snapshot);
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_max_level(usearch_amd_snapshot_t snapshot);
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_bytes_per_vector(usearch_amd_snapshot_t snapshot);
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_row_stride(usearch_amd_snapshot_t snapshot);
/** Bytes of HBM the snapshot occupies (cf. `usearch_memory_usage`, c/usearch.h:139). */
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_device_bytes(usearch_amd_snapshot_t snapshot);
/** How the matrix of stored rows was placed in HBM. Where a multi-gigabyte array lands decides how fast the walk runs over it
 *  (the headline batch: 44.8 … 51.9 ms for the same bytes; the state belongs to the physical frames, which the driver releases late
 *  and hands out by rules of its own), and no synthetic probe tells the placements apart. So the first launches that fill the chip
 *  each try ONE fresh device-to-device copy of the matrix against the incumbent — both timed on that launch's own first queries at
 *  the caller's expansion — and keep the faster (csrc/placement.hpp, `snapshot_t::try_matrix_placement`; at most
 *  USEARCH_AMD_PLACEMENT_DRAWS = 8 trials, 1 = off, ended early by three wins of the incumbent in a row, reopened for three more —
 *  twice at most — by a launch more than twice as wide as the last trial's: a host that tunes its expansion walks up; arrays under
 *  USEARCH_AMD_PLACEMENT_MIN_BYTES = 1 GiB stay where they are). `*draws` = trials made so far, `*kept` = how many moved the
 *  matrix, `judge_ms[i]` / `incumbent_ms[i]` = the candidate's / the incumbent's milliseconds in trial i (up to 8 each; either may
 *  be null), `*probe_ms` = what the trials have cost in all. Since round 6 the trials are OFF unless USEARCH_AMD_PLACEMENT_DRAWS
 *  = 2 … 8 asks for them: the matrix is placed once, after the settle window (below). */
USEARCH_AMD_EXPORT void usearch_amd_snapshot_placement(usearch_amd_snapshot_t snapshot, uint32_t* draws, uint32_t* kept,
                                                       float* judge_ms, float* incumbent_ms, float* probe_ms);
/**
 *  Settle, then allocate (csrc/placement.hpp; round 6). The driver hands a freed block's frames back to its allocator 0.3 … 1 s
 *  after the free, and a multi-gigabyte array allocated inside that window lands on other — slower — frames than the ones the driver
 *  prefers when everything is free (the headline batch: 44.8 ms on the preferred frames every time, 51 ms every other time without
 *  the wait). Every loader and builder of this library therefore waits, before it allocates the matrix of stored rows, until
 *  USEARCH_AMD_SETTLE_MS (default 1000; 0 = off) have passed since the last release of ≥ 64 MB it knows of: its own, and those a
 *  host announces here after freeing device memory through ANOTHER allocator (`torch.cuda.empty_cache()`, `hipFree` of its own).
 *  `usearch_amd_settle` waits out the window explicitly (before a host's own big allocation) and returns the milliseconds waited;
 *  `usearch_amd_snapshot_settle_ms` = what this snapshot's matrix waited when it was allocated.
 *
 *  Tuning. The settle window makes a placement reproducible inside a process (the same image loaded twice runs the headline batch in
 *  45.10 / 45.06 ms, or in 48.31 / 48.28 ms); WHICH level it lands on depends on the box and on what the process allocated before
 *  (44.4 … 51.4 ms over the same bytes: profiles/r06_settled/), and only the walk itself tells placements apart. A host about to
 *  serve one shape of batch hands a sample of it to `usearch_amd_snapshot_tune` (device-resident queries in the storage scalar kind,
 *  as for `usearch_amd_search_many_device`): up to `max_trials` (≤ 8) fresh device-to-device copies of the matrix are placed one
 *  after the other and timed against the incumbent on the sample's first queries at `expansion`; the faster stays, three wins of
 *  the incumbent in a row end it early. Explicit and synchronous — no trial ever runs inside a search call (USEARCH_AMD_PLACEMENT_
 *  DRAWS = 2 … 8 in the environment turns round 5's online trials back on) — and a second copy of the matrix exists in HBM only
 *  during the call. Returns the trials made (0: the matrix is under 1 GiB, rows travel inline with the lists, or the sample does not
 *  fill the chip twice over); `usearch_amd_snapshot_placement` reports every trial's two times.
 */
USEARCH_AMD_EXPORT uint32_t usearch_amd_snapshot_tune(usearch_amd_snapshot_t snapshot, void const* queries_device, size_t queries_count,
                                                      size_t queries_stride, size_t wanted, size_t expansion, uint32_t max_trials,
                                                      usearch_amd_error_t* error);
USEARCH_AMD_EXPORT void usearch_amd_note_device_free(void);
USEARCH_AMD_EXPORT float usearch_amd_settle(void);
USEARCH_AMD_EXPORT float usearch_amd_snapshot_settle_ms(usearch_amd_snapshot_t snapshot);
/** The HBM-resident arrays of a snapshot, read-only, for a host's own kernels over them (csrc/common.hpp `snapshot_view_t`): the
 *  padded matrix of stored rows `vectors[size][row_stride]`, the level-0 lists `level0[size][2 × connectivity]` (u32 slots,
 *  0xFFFFFFFF = empty cell) and the keys `keys[size]` (u64). Valid until the snapshot is freed or — for an index under
 *  construction — extended. */
typedef struct usearch_amd_arrays_t {
    void const* vectors;
    uint32_t const* level0;
    uint64_t const* keys;
    uint64_t size;
    uint32_t row_stride;
    uint32_t level0_cells;
    int device;
    uint32_t reserved;
} usearch_amd_arrays_t;
USEARCH_AMD_EXPORT void usearch_amd_snapshot_arrays(usearch_amd_snapshot_t snapshot, usearch_amd_arrays_t* arrays);
/** Storage scalar kind (C enumerator) and metric kind (`usearch_metric_kind_t` value, c/usearch.h:40-52). */
USEARCH_AMD_EXPORT int usearch_amd_snapshot_scalar_kind(usearch_amd_snapshot_t snapshot);
USEARCH_AMD_EXPORT int usearch_amd_snapshot_metric_kind(usearch_amd_snapshot_t snapshot);
/** Lanes that share one stored row (G): fixes the floating-point summation layout, see DESIGN.md. */
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_lanes_per_row(usearch_amd_snapshot_t snapshot);
/** 1 when the stored rows of every member's level-0 neighbours are kept next to its list (rows of ≤ 16 bytes: one contiguous
 *  read per hop instead of a list line plus scattered rows), 0 otherwise. */
USEARCH_AMD_EXPORT size_t usearch_amd_snapshot_inline_rows(usearch_amd_snapshot_t snapshot);

/**
 *  Batched search with HOST buffers — `usearch_search` (c/usearch.h:371-374) for `queries_count` queries at once.
 *
 *  @param queries        row `i` starts at `(char*)queries + i * queries_stride` and holds `dimensions` scalars of
 *                        `query_kind`; cast to the storage kind exactly like index_dense.hpp:2058-2064.
 *  @param wanted         k. Exactly `wanted` keys and distances are written per query; unused tail slots are key 0 and
 *                        a signalling NaN (index.hpp:2707-2722, c/lib.cpp:410).
 *  @param expansion      ef; 0 = 64 (index.hpp:3029-3030); the effective value is max(expansion, wanted) (index.hpp:3052).
 *  @param keys           [queries_count][wanted], may be NULL.
 *  @param distances      [queries_count][wanted], may be NULL.
 *  @param counts         [queries_count] found per query (`dump_to`'s return value), may be NULL.
 *  @param visited        [queries_count] `search_result_t::visited_members` (index.hpp:3071), may be NULL.
 *  @param computed       [queries_count] `search_result_t::computed_distances` (index.hpp:3072), may be NULL.
 */
USEARCH_AMD_EXPORT void usearch_amd_search_many(usearch_amd_snapshot_t snapshot, void const* queries, int query_kind,
                                                size_t queries_count, size_t queries_stride, size_t wanted,
                                                size_t expansion, usearch_amd_key_t* keys,
                                                usearch_amd_distance_t* distances, uint64_t* counts, uint64_t* visited,
                                                uint64_t* computed, usearch_amd_tuning_t const* tuning,
                                                usearch_amd_stats_t* stats, usearch_amd_error_t* error);

/**
 *  Batched search with DEVICE buffers (HBM-resident input and output, the timed path of bench.py). Queries must
 *  already be in the storage scalar kind; every output pointer is mandatory. `stream` is a `hipStream_t` (NULL = the
 *  snapshot's own stream); the call returns after the stream has drained. With `timed != 0` the search launches
 *  are bracketed by HIP events on that stream and `stats->kernel_ms` is filled.
 */
USEARCH_AMD_EXPORT void usearch_amd_search_many_device(usearch_amd_snapshot_t snapshot, void const* queries,
                                                       size_t queries_count, size_t queries_stride, size_t wanted,
                                                       size_t expansion, usearch_amd_key_t* keys,
                                                       usearch_amd_distance_t* distances, uint64_t* counts,
                                                       uint64_t* visited, uint64_t* computed, void* stream,
                                                       usearch_amd_tuning_t const* tuning, int timed,
                                                       usearch_amd_stats_t* stats, usearch_amd_error_t* error);

/**
 *  EXACT (brute-force) batched search over a snapshot — `usearch_search` on an index searched with `exact = true`
 *  (index_dense.hpp:767-772, index.hpp:3046-3049, 4252-4268): every stored vector is measured, the `count` best under
 *  (distance ascending, later slot first among equals) are returned, tombstones skipped. Host buffers; queries in any
 *  scalar kind. `kernel_ms` (may be NULL) receives the HIP-event time of the scan kernel. Also the recall ground truth.
 */
USEARCH_AMD_EXPORT void usearch_amd_exact_search_many(usearch_amd_snapshot_t snapshot, void const* queries,
                                                      int query_kind, size_t queries_count, size_t queries_stride,
                                                      size_t wanted, usearch_amd_key_t* keys,
                                                      usearch_amd_distance_t* distances, uint64_t* counts,
                                                      float* kernel_ms, usearch_amd_error_t* error);

/* ---- filtered search: the caller's predicate as an HBM-resident bitmap --------------------------------------------------
 *  The reference's predicate is a host callable evaluated inside the traversal (`usearch_filtered_search`, c/usearch.h:391-395 →
 *  c/lib.cpp:413-429 → index_dense.hpp:774-779, 2071-2084 → index.hpp:4200-4205, 4236-4240) and inside the brute-force scan
 *  (index.hpp:4260-4263). A host function cannot run on the device: the predicate travels as ONE BIT PER SLOT, tested by the
 *  kernels at exactly those places — members that fail it still route the walk, they never enter the result. A filter is made
 *  once (a kernel over the snapshot's keys; nothing per member happens on the host), lives in HBM, and serves any number of
 *  batches; it describes the snapshot it was made for, with the members that snapshot had at that moment.
 * -------------------------------------------------------------------------------------------------------------------------- */

typedef void* usearch_amd_filter_t;

/** Members whose key lies in [first_key, last_key], both ends included. */
USEARCH_AMD_EXPORT usearch_amd_filter_t usearch_amd_filter_from_key_range(usearch_amd_snapshot_t snapshot,
                                                                         usearch_amd_key_t first_key, usearch_amd_key_t last_key,
                                                                         usearch_amd_error_t* error);
/** Members whose key is among `keys[0 .. keys_count)` (`allow != 0`) or is NOT among them (`allow == 0`: a deny list). */
USEARCH_AMD_EXPORT usearch_amd_filter_t usearch_amd_filter_from_keys(usearch_amd_snapshot_t snapshot, usearch_amd_key_t const* keys,
                                                                    size_t keys_count, int allow, usearch_amd_error_t* error);
/** The caller's own bitmap (host memory): bit `s & 31` of `bits[s >> 5]` = the member in slot `s` passes; slots are the members in
 *  the order the image holds them (= the order they were added). `words` ≥ ⌈size / 32⌉. */
USEARCH_AMD_EXPORT usearch_amd_filter_t usearch_amd_filter_from_bits(usearch_amd_snapshot_t snapshot, uint32_t const* bits, size_t words,
                                                                    usearch_amd_error_t* error);
/** How many members pass (tombstones never do). */
USEARCH_AMD_EXPORT size_t usearch_amd_filter_allowed(usearch_amd_filter_t filter);
/** The bitmap in HBM (⌈size / 32⌉ words), e.g. to hand to `usearch_amd_search_many_device`-style callers of their own kernels. */
USEARCH_AMD_EXPORT void const* usearch_amd_filter_device_bits(usearch_amd_filter_t filter);
USEARCH_AMD_EXPORT void usearch_amd_filter_free(usearch_amd_filter_t filter, usearch_amd_error_t* error);

/** `usearch_amd_search_many` under a filter — `usearch_filtered_search` (c/usearch.h:391-395) for a batch. Keys, distances, counts
 *  and both traversal counters are the reference's for the predicate the bitmap stands for. `filter` = NULL: no predicate. */
USEARCH_AMD_EXPORT void usearch_amd_filtered_search_many(usearch_amd_snapshot_t snapshot, usearch_amd_filter_t filter,
                                                         void const* queries, int query_kind, size_t queries_count,
                                                         size_t queries_stride, size_t wanted, size_t expansion,
                                                         usearch_amd_key_t* keys, usearch_amd_distance_t* distances,
                                                         uint64_t* counts, uint64_t* visited, uint64_t* computed,
                                                         usearch_amd_tuning_t const* tuning, usearch_amd_stats_t* stats,
                                                         usearch_amd_error_t* error);
/** `usearch_amd_search_many_device` under a filter (HBM in / out, the caller's stream). */
USEARCH_AMD_EXPORT void usearch_amd_filtered_search_many_device(usearch_amd_snapshot_t snapshot, usearch_amd_filter_t filter,
                                                                void const* queries, size_t queries_count, size_t queries_stride,
                                                                size_t wanted, size_t expansion, usearch_amd_key_t* keys,
                                                                usearch_amd_distance_t* distances, uint64_t* counts,
                                                                uint64_t* visited, uint64_t* computed, void* stream,
                                                                usearch_amd_tuning_t const* tuning, int timed,
                                                                usearch_amd_stats_t* stats, usearch_amd_error_t* error);
/** `filtered_search(query, wanted, predicate, thread, exact = true)` of the class for a batch (index_dense.hpp:774-779 →
 *  `search_exact_` with the predicate, index.hpp:4252-4268): brute force over the members that pass. Host buffers, any query scalar
 *  kind; `tiled != 0` = the matrix-unit kernel (pairs and tolerances of `usearch_amd_exact_search_many_tiled`). */
USEARCH_AMD_EXPORT void usearch_amd_filtered_exact_search_many(usearch_amd_snapshot_t snapshot, usearch_amd_filter_t filter,
                                                               void const* queries, int query_kind, size_t queries_count,
                                                               size_t queries_stride, size_t wanted, usearch_amd_key_t* keys,
                                                               usearch_amd_distance_t* distances, uint64_t* counts, int tiled,
                                                               float* kernel_ms, usearch_amd_error_t* error);

/**
 *  Closest member on a given LEVEL of the hierarchy for a batch of queries — `index_dense_gt::cluster(query, level)`
 *  (index_dense.hpp:788-793 → index_gt::cluster, index.hpp:3089-3125): the greedy descent of `search_for_one_` from the top
 *  level down to `level`, without the level-0 beam. Level 0 and 1 both end on level 1's winner, a level above the top one
 *  returns the entry point. Host buffers, queries in any scalar kind; `keys`, `distances`, `visited`, `computed` hold one
 *  cell per query (the counters as `cluster_result_t` reports them; either may be NULL). An empty index yields key 0 and a
 *  signalling NaN (the reference fails that call with "No clusters to identify").
 */
USEARCH_AMD_EXPORT void usearch_amd_cluster_many(usearch_amd_snapshot_t snapshot, void const* queries, int query_kind,
                                                 size_t queries_count, size_t queries_stride, size_t level,
                                                 usearch_amd_key_t* keys, usearch_amd_distance_t* distances,
                                                 uint64_t* visited, uint64_t* computed, usearch_amd_error_t* error);

/**
 *  The same exact search as a TILED MATRIX PRODUCT on the matrix units (usearch_amd/csrc/exact_tiled.hip) — what the reference's
 *  `exact_search_t` does with its distance matrix (index_plugins.hpp:2071-2164): dataset rows are read once per 64 queries
 *  instead of once per query (once per 256 for batches above 512 queries). Pairs: cos / ip / l2sq over f16 and bf16 (f32 accumulation in the matrix unit: distances within the
 *  float tolerance of `usearch_amd_exact_search_many`, which remains the bit-exact path) and cos / ip / l2sq over i8
 *  (bit-identical to it, ties included); `wanted` ≤ 64. Other pairs: an error.
 */
USEARCH_AMD_EXPORT void usearch_amd_exact_search_many_tiled(usearch_amd_snapshot_t snapshot, void const* queries,
                                                            int query_kind, size_t queries_count, size_t queries_stride,
                                                            size_t wanted, usearch_amd_key_t* keys,
                                                            usearch_amd_distance_t* distances, uint64_t* counts,
                                                            float* kernel_ms, usearch_amd_error_t* error);

/**
 *  Either exact search over DEVICE buffers (queries in the storage scalar kind, `queries_stride` bytes apart; keys, distances and
 *  counts dense `[queries_count][wanted]` / `[queries_count]` in HBM), enqueued on `stream` (NULL: a stream of the snapshot);
 *  returns when the results are complete. `tiled != 0` selects the matrix-unit kernel. `kernel_ms` (may be NULL): HIP-event
 *  time of the kernels. What `bench.py --exact` times.
 */
USEARCH_AMD_EXPORT void usearch_amd_exact_search_many_device(usearch_amd_snapshot_t snapshot, void const* queries,
                                                             size_t queries_count, size_t queries_stride, size_t wanted,
                                                             usearch_amd_key_t* keys, usearch_amd_distance_t* distances,
                                                             uint64_t* counts, void* stream, int tiled, float* kernel_ms,
                                                             usearch_amd_error_t* error);

/**
 *  Exact search of a raw host dataset — `usearch_exact_search` (c/usearch.h:467-474, c/lib.cpp:468-501): keys are row
 *  offsets of `dataset`. `metric_kind` / `scalar_kind` use the C enumerators of c/usearch.h:40-62. Ties between equal
 *  distances (unspecified in the reference: `std::partial_sort` by distance) resolve to the later row first.
 */
USEARCH_AMD_EXPORT void usearch_amd_exact_search_dataset(void const* dataset, size_t dataset_count,
                                                         size_t dataset_stride, void const* queries,
                                                         size_t queries_count, size_t queries_stride, int scalar_kind,
                                                         size_t dimensions, int metric_kind, size_t wanted,
                                                         usearch_amd_key_t* keys, size_t keys_stride,
                                                         usearch_amd_distance_t* distances, size_t distances_stride,
                                                         usearch_amd_error_t* error);

/* ------------------------------------------------------------------------------------------------------------------
 *  Sharded search across the GPUs of one node — one process per GPU, one shard (its own HNSW) per process.
 *  Replaces the reference's `Indexes` (python/usearch/index.py:1473-1514 → python/lib.cpp:321-402: every sub-index searches
 *  every query, per-query results folded with `search_result_t::merge_into`, index.hpp:2650-2670).
 * ---------------------------------------------------------------------------------------------------------------- */

typedef void* usearch_amd_comm_t;

/** Collectives supplied by the caller (MPI, gloo, …) instead of RCCL. Every callback returns NULL or an error message. */
typedef struct usearch_amd_transport_t {
    void* context;
    /** every rank contributes `bytes` bytes at `send` and receives world × `bytes` at `receive`, in rank order */
    char const* (*all_gather)(void* context, void const* send, void* receive, size_t bytes, void* stream);
    /** optional: `buffer` of rank `root` to every rank */
    char const* (*broadcast)(void* context, void* buffer, size_t bytes, int root, void* stream);
    int buffers_on_host; /**< 0 = callbacks take device pointers and a hipStream_t; 1 = host pointers (stream is NULL) */
    /** optional stand-in for the device search: when set, the whole step (search, packing, exchange, merge) runs in host
     *  memory and no HIP call is made — the protocol on machines without a GPU */
    char const* (*local_search)(void* context, void const* queries, size_t queries_count, size_t queries_stride,
                                size_t wanted, size_t expansion, usearch_amd_key_t* keys, usearch_amd_distance_t* distances,
                                uint64_t* counts);
} usearch_amd_transport_t;

typedef struct usearch_amd_sharded_stats_t {
    uint64_t block_bytes;    /**< bytes this rank contributes to the all-gather: Q·k·12 + Q·8 + 8 */
    uint64_t gathered_bytes; /**< bytes it receives (world × block) */
    float exchange_ms;       /**< all-gather + merge kernel on the stream (HIP events, with `timed`) */
    uint32_t exchanges;      /**< 1; 2 when some rank's scratch retry ladder ran and the blocks were exchanged again */
} usearch_amd_sharded_stats_t;

/** 128 bytes that name a new RCCL communicator (`ncclGetUniqueId`): rank 0 makes them, the launcher carries them over. */
USEARCH_AMD_EXPORT void usearch_amd_comm_unique_id(void* out_128_bytes, usearch_amd_error_t* error);
/** RCCL communicator over xGMI for this rank's `device` (`ncclCommInitRank`; every rank must call it). */
USEARCH_AMD_EXPORT usearch_amd_comm_t usearch_amd_comm_init_rccl(void const* unique_id_128_bytes, int rank, int world,
                                                                 int device, usearch_amd_error_t* error);
/** Communicator over the caller's collectives. */
USEARCH_AMD_EXPORT usearch_amd_comm_t usearch_amd_comm_init_custom(usearch_amd_transport_t const* transport, int rank,
                                                                   int world, int device, usearch_amd_error_t* error);
USEARCH_AMD_EXPORT void usearch_amd_comm_free(usearch_amd_comm_t comm);
USEARCH_AMD_EXPORT int usearch_amd_comm_rank(usearch_amd_comm_t comm);
USEARCH_AMD_EXPORT int usearch_amd_comm_world(usearch_amd_comm_t comm);
/** `buffer` (device memory) of rank `root` to every rank, on `stream`. */
USEARCH_AMD_EXPORT void usearch_amd_comm_broadcast(usearch_amd_comm_t comm, void* buffer, size_t bytes, int root,
                                                   void* stream, usearch_amd_error_t* error);

/**
 *  ONE STEP of sharded search, on every rank: [broadcast the batch from `broadcast_root`, -1 = every rank already holds
 *  it] → search `snapshot` (this rank's shard; results go straight into the send block) → ONE all-gather of the packed
 *  block {distances | keys | counts} → merge kernel with the `merge_into` tie rule, shards taken in rank order. All of it
 *  on `stream` (NULL = the snapshot's own) with a single wait at the end. Device buffers; queries in the storage kind.
 *  keys / distances / counts receive the merged result (identical on every rank); visited / computed this rank's own
 *  traversal counters. With a transport that carries `local_search`, every buffer is host memory and `snapshot` may be
 *  NULL. Replaces `Indexes.search` (python/lib.cpp:321-402).
 *  Errors are collective: a rank whose broadcast, search or retry fails still enters the all-gather, with an abort bit in
 *  the flag word that closes its block, so EVERY rank returns an error for that step (the failing rank its own message,
 *  the others "aborted by rank r in …") and none is left waiting inside the collective.
 */
USEARCH_AMD_EXPORT void usearch_amd_sharded_search_many(usearch_amd_snapshot_t snapshot, usearch_amd_comm_t comm,
                                                        void* queries, size_t queries_count, size_t queries_stride,
                                                        size_t wanted, size_t expansion, int broadcast_root,
                                                        usearch_amd_key_t* keys, usearch_amd_distance_t* distances,
                                                        uint64_t* counts, uint64_t* visited, uint64_t* computed,
                                                        void* stream, usearch_amd_tuning_t const* tuning, int timed,
                                                        usearch_amd_stats_t* stats,
                                                        usearch_amd_sharded_stats_t* sharded_stats,
                                                        usearch_amd_error_t* error);

/**
 *  Exchange step of SHARDED search: merges per-shard results `distances/keys[shards][queries][wanted]`,
 *  `counts[shards][queries]` (device pointers, e.g. the output of an RCCL all-gather of every rank's
 *  `usearch_amd_search_many_device` results) into `[queries][wanted]`, reproducing `search_result_t::merge_into`
 *  (index.hpp:2650-2670) applied to shards 0…P-1 in order — what `Indexes.search` does per query (python/lib.cpp:321-402).
 *  `stream` is a `hipStream_t` (NULL = default); returns after it has drained.
 */
USEARCH_AMD_EXPORT void usearch_amd_merge_many_device(usearch_amd_distance_t const* distances,
                                                      usearch_amd_key_t const* keys, uint64_t const* counts,
                                                      size_t shards, size_t queries_count, size_t wanted,
                                                      usearch_amd_distance_t* out_distances,
                                                      usearch_amd_key_t* out_keys, uint64_t* out_counts, void* stream,
                                                      usearch_amd_error_t* error);
/** Same with host buffers (staged through the current device). */
USEARCH_AMD_EXPORT void usearch_amd_merge_many(usearch_amd_distance_t const* distances, usearch_amd_key_t const* keys,
                                               uint64_t const* counts, size_t shards, size_t queries_count,
                                               size_t wanted, usearch_amd_distance_t* out_distances,
                                               usearch_amd_key_t* out_keys, uint64_t* out_counts,
                                               usearch_amd_error_t* error);

/* ---- index construction on the device ------------------------------------------------------------------------------ */

typedef void* usearch_amd_builder_t;

/** Knobs of a build; zero-initialise for the reference's defaults (M = 16, M0 = 2·M, ef_construction = 128). */
typedef struct usearch_amd_build_config_t {
    uint32_t connectivity;      /**< M  (`usearch_init_options_t::connectivity`, c/usearch.h:90-95); 0 = 16 */
    uint32_t connectivity_base; /**< M0; 0 = 2·M */
    uint32_t expansion_add;     /**< ef_construction (`expansion_add`, c/usearch.h:97-101); 0 = 128 */
    uint32_t batch_divisor;     /**< a batch holds at most (nodes linked so far) / divisor new nodes; 0 = 16 */
    uint32_t max_batch;         /**< … and at most this many; 0 = 65536 */
    uint32_t reserved;
    uint64_t seed;              /**< level draw; 0 = a fixed default */
} usearch_amd_build_config_t;

typedef struct usearch_amd_build_stats_t {
    uint64_t batches, passes;
    uint64_t search_distances, search_hops;   /**< counters of the insertion searches (index.hpp:4011-4079) */
    uint64_t select_distances, reverse_distances, repruned_lists, dropped_requests;
    double seconds_total, seconds_search, seconds_link, seconds_upload;
    uint32_t max_level, reserved;
    uint64_t refiled_requests;                /**< reverse links that waited a round for room in a hub's inbox; none is lost */
} usearch_amd_build_stats_t;

/**
 *  Builds an HNSW index over `count` vectors on `device` — the whole `usearch_add` loop (c/usearch.h:338-339;
 *  cpp/bench.cpp:296-327) in one call: insertion search, `form_links_to_closest_` and `form_reverse_links_`
 *  (index.hpp:2855-2863) run as batched gfx950 kernels. `vectors` holds rows of the storage `scalar_kind`, `stride`
 *  bytes apart, in host memory or (`vectors_on_device != 0`) in HBM. `keys` (host) may be NULL: key = row number.
 *  The result owns a searchable snapshot and can be serialized in the reference's format.
 */
USEARCH_AMD_EXPORT usearch_amd_builder_t usearch_amd_build(void const* vectors, size_t count, size_t stride,
                                                           int scalar_kind, size_t dimensions, int metric_kind,
                                                           usearch_amd_key_t const* keys,
                                                           usearch_amd_build_config_t const* config, int device,
                                                           int vectors_on_device, usearch_amd_error_t* error);
USEARCH_AMD_EXPORT void usearch_amd_build_free(usearch_amd_builder_t builder, usearch_amd_error_t* error);
/** The snapshot the build produced; owned by the builder, valid until `usearch_amd_build_free`. */
USEARCH_AMD_EXPORT usearch_amd_snapshot_t usearch_amd_build_snapshot(usearch_amd_builder_t builder);
/** `usearch_serialized_length` / `usearch_save_buffer` (c/usearch.h:154, 195): a v2 image the reference loads. */
USEARCH_AMD_EXPORT size_t usearch_amd_build_serialized_length(usearch_amd_builder_t builder);
USEARCH_AMD_EXPORT void usearch_amd_build_save_buffer(usearch_amd_builder_t builder, void* buffer, size_t length,
                                                      usearch_amd_error_t* error);
USEARCH_AMD_EXPORT void usearch_amd_build_stats(usearch_amd_builder_t builder, usearch_amd_build_stats_t* stats);

/**
 *  Telemetry of the most recent search on this snapshot: out[q] = {peak frontier size, visited-set size} for the first
 *  `queries_count` queries — what DESIGN.md's scratch sizing is derived from.
 */
USEARCH_AMD_EXPORT void usearch_amd_last_peaks(usearch_amd_snapshot_t snapshot, uint32_t* out, size_t queries_count,
                                               usearch_amd_error_t* error);

/**
 *  out[q][j] = metric(query q, stored vector of slot slots[q][j]) — `usearch_distance` (c/usearch.h:441-445) against
 *  stored rows, host buffers, queries in the storage kind. Exposes the distance arithmetic alone.
 */
USEARCH_AMD_EXPORT void usearch_amd_distances(usearch_amd_snapshot_t snapshot, void const* queries,
                                              size_t queries_count, size_t queries_stride, uint32_t const* slots,
                                              size_t slots_per_query, usearch_amd_distance_t* out,
                                              usearch_amd_error_t* error);

/** HIP-event duration of the kernel of the most recent `usearch_amd_distances` call: the dependency-free row-gather rate,
 *  i.e. the ceiling the search kernel's distance phase is measured against. */
USEARCH_AMD_EXPORT float usearch_amd_last_distances_ms(usearch_amd_snapshot_t snapshot);

/**
 *  Host-side query cast used by `usearch_amd_search_many` (index_plugins.hpp:1105-1224). Returns 0 when the kinds are
 *  equal (nothing written), 1 after writing the cast vector.
 */
USEARCH_AMD_EXPORT int usearch_amd_cast(int from_kind, int to_kind, void const* input, size_t dimensions, void* output);

#ifdef __cplusplus
}
#endif
#endif /* USEARCH_AMD_H */
