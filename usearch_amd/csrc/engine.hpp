/**
 *  usearch_amd/csrc/engine.hpp — host side of the MI355X search engine: an immutable HBM snapshot of one index and the
 *  batched search over it.
 *
 *  Replaces, for the search path only, what `index_dense_gt` owns on the CPU (/root/reference/include/usearch/
 *  index_dense.hpp:419-460 `metric_proxy_t` + `vectors_lookup_`, index.hpp:2280 `nodes_`) and what its callers loop over
 *  (`cpp/bench.cpp:352-377`, `python/lib.cpp:261-319`): one call = one batch of independent queries.
 */
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "image.hpp"
#include "placement.hpp"

namespace usearch_amd {

/// Tunables of one search launch; zeros mean "choose for me".
struct search_tuning_t {
    std::uint32_t hash_cap = 0;     ///< visited-set cells per query (power of two)
    std::uint32_t next_cap = 0;     ///< frontier capacity per query
    std::uint32_t variant = 0;      ///< 0 = auto, else 1 + kernel_variant_t (loads in flight per lane vs waves per SIMD)
    std::uint32_t mode = 0;         ///< 0 = auto, 1 = visited set in LDS, 2 = visited set in a global hash, 3 = all-global fallback
    std::uint32_t waves_per_cu = 0; ///< persistent waves per compute unit (0 = as many as LDS / registers admit)
    std::uint32_t top_in_memory = 0; ///< 1 = keep `top` in scratch memory even when it would fit registers
    std::uint32_t frontier = 0;     ///< 0 = auto, 1 = the reference's heap (its pop order among equal distances), 2 = the open
                                    ///< cells of `top` (kernels.hpp frontier_top_k; refused where it does not apply)
    std::uint32_t wave_clock = 0;   ///< 1 = record every persistent wave's start / exit time (batch-tail telemetry)
};

struct search_stats_t {
    std::uint32_t passes = 0;            ///< launches needed (1 = every query fit its LDS scratch)
    std::uint32_t retried_lds = 0;       ///< queries rerun with the enlarged LDS scratch
    std::uint32_t retried_global = 0;    ///< queries rerun with the global-memory scratch
    float kernel_ms = 0.f;               ///< HIP-event time of the search launches of this call (when `timed`)
    std::uint32_t mode = 0;              ///< scratch mode of the last launch (1 = LDS, 2 = global hash, 3 = all global)
    std::uint32_t grid = 0;              ///< persistent waves of the last launch
    std::uint32_t lds_bytes = 0;         ///< LDS per wave of the last launch
    std::uint32_t frontier = 0;          ///< 1 = heap, 2 = open cells of `top` (first launch)
    std::uint32_t variant = 0;           ///< 1 + kernel_variant_t of the first launch
    float tail_idle = 0.f;               ///< with `wave_clock`: share of wave-time between the first start and the last exit
                                         ///< that waves spent gone (the drain phase of the batch), first launch
    float span_ms = 0.f;                 ///< with `wave_clock`: first start → last exit on the device's 100-MHz clock
    std::uint32_t top_cells = 0;         ///< `top` cells per lane in registers of the first launch (0 = scratch memory)
    std::uint32_t probe_mode = 0;        ///< short rows over a global slab: `probe_mode_t` of the last launch
    std::uint32_t seen_cells = 0;        ///< … its `seen` cells in LDS
    std::uint32_t claim_bits = 0;        ///< … its claim bits in LDS (`probe_plain_k`)
    std::uint32_t early_rows = 0;        ///< rows of ≤ 128 bytes: 1 = gathered next to the probe of the visited set, not behind it
    std::uint32_t plain = 0;             ///< short rows: 1 = the last launch ran the build cut for plain batches (kernels.hpp `plain_ak`)
    std::uint32_t aside_cells = 0;       ///< … and its LDS cells for the members whose home cell in the slab was taken
};

/// What index construction asks of the search on top of a plain query batch (see search_args_t).
struct search_extras_t {
    const std::uint32_t* query_ids = nullptr; ///< device: query q is row query_ids[q] of `queries`
    std::uint32_t beam_level = 0;             ///< level the beam runs on
    bool emit_slots = false;                  ///< slots instead of keys in the `keys` output
    bool descent_only = false;                ///< `cluster`: the greedy descent to `beam_level` alone, one result per query
    const std::uint32_t* allow_bits = nullptr; ///< device: one bit per slot, 0 = rejected by the caller's predicate
    // a predicate evaluated lazily by the host (search_args_t::known_bits): all device pointers
    const std::uint32_t* known_bits = nullptr;
    std::uint32_t* ask_slots = nullptr;
    std::uint64_t* ask_keys = nullptr;
    std::uint32_t* ask_cursor = nullptr;
    std::uint32_t ask_cap = 0;
    std::uint32_t guess_threshold = 0xFFFFFFFFu;
    bool reference_frontier = false;          ///< keep the reference's heap whatever the pair (index construction does)
    bool exclude_own = false;                 ///< `search_to_update_`: a query's own stored row routes, never becomes a candidate
};

/// Whether an instantiation of the search kernel also exists in the cut for plain batches (kernels.hpp `plain_ak`; launch_impl.hpp
/// instantiates exactly these): rows of ≤ 128 bytes of the common pairs, 4 loads in flight, the visited set in the global hash, `top` of
/// one or two cells per lane, the reference's heap.
inline constexpr bool plain_build_exists(metric_kind_t metric, scalar_kind_t scalar, int lanes, bool four_deep, bool global_hash,
                                         int top_cells, bool heap) {
    return all_kernel_builds(metric, scalar) && lanes <= 2 && four_deep && global_hash && (top_cells == 1 || top_cells == 2) && heap;
}

/// Per-scalar-kind launchers, one translation unit each (compile time): defined in search_<kind>.hip.
struct launch_params_t {
    metric_kind_t metric;
    std::uint32_t lanes;
    int variant; ///< kernel_variant_t of kernels.hpp
    int mode;    ///< scratch_mode_t of kernels.hpp
    std::uint32_t entries_per_lane; ///< `top` in registers: 1, 4, 8 or 16 entries per lane; 0 = in scratch memory
    int frontier;                   ///< frontier_mode_t of kernels.hpp
    std::uint32_t grid;
    std::uint32_t lds_bytes;
    hipStream_t stream;
    std::uint32_t team = 0;       ///< 1 = five waves per query (team_search_kernel): small batches over long rows
    std::uint32_t plain = 0;      ///< 1 = the short-row build cut for a plain `search` batch (kernels.hpp `plain_ak`): the engine vouches
                                  ///< for level 0, no predicate / tombstones / own row, lists of ≤ 64 cells, `seen` cells, rows inline or early
};

/**
 *  Everything ONE in-flight batch needs besides the immutable snapshot: a stream, the ticket counter, per-query status,
 *  scratch slabs, and the device + pinned-host staging of the host-buffer entry points. A snapshot keeps a small pool of
 *  these, so that concurrent callers (the reference leases one `context_t` per thread, index_dense.hpp:1984-2000) run
 *  side by side instead of queueing on one mutex.
 */
struct workspace_t {
    hipStream_t stream = nullptr;
    hipEvent_t event_begin = nullptr, event_end = nullptr;
    std::uint32_t* d_status = nullptr;
    std::uint32_t* d_todo = nullptr;
    std::uint32_t* d_queue = nullptr; ///< 256-byte block: [0] ticket counter, [1] overflowed queries, [16…] phase clock
    std::uint32_t* d_peaks = nullptr;
    std::uint32_t* h_status = nullptr; ///< pinned: [0..1] copy of the queue block's counters, then per-query status
    std::size_t queries = 0;
    std::uint8_t* d_scratch = nullptr;
    std::size_t scratch_bytes = 0;
    unsigned long long* d_wave_clock = nullptr;
    std::size_t wave_clock_waves = 0;
    std::uint8_t* d_stage = nullptr; ///< host-buffer API: queries | keys | distances | counts | visited | computed [| allow bits]
    std::uint8_t* h_stage = nullptr; ///< pinned mirror of the result part
    std::size_t stage_bytes = 0, host_stage_bytes = 0;
    std::size_t last_count = 0;

    const char* create();
    const char* reserve(std::size_t queries_wanted, std::size_t scratch_wanted);
    const char* reserve_stage(std::size_t device_bytes, std::size_t host_bytes);
    const char* reserve_wave_clock(std::size_t waves);
    void destroy();
};

class snapshot_t {
  public:
    snapshot_t() = default;
    ~snapshot_t();
    snapshot_t(const snapshot_t&) = delete;
    snapshot_t& operator=(const snapshot_t&) = delete;

    /// Flattens `image` into the HBM layout of `snapshot_view_t` on `device`. Returns nullptr or a static message.
    const char* build(const image_t& image, int device);

    const snapshot_view_t& view() const { return view_; }
    metric_kind_t metric() const { return metric_; }
    scalar_kind_t scalar() const { return scalar_; }
    std::uint32_t lanes_per_row() const { return lanes_; }
    int device() const { return device_; }
    std::size_t device_bytes() const { return device_bytes_; }
    std::uint64_t count_present() const { return count_present_; }
    std::uint64_t upper_lists() const { return upper_lists_; }
    float last_distances_ms() const { return last_distances_ms_; }
    /// How the matrix of stored rows was placed in HBM (placement.hpp): draws, their gather rates, which one was kept.
    const placement_t& placement() const { return placement_; }

    /**
     *  Batched search, all pointers device-resident, queries already in the storage scalar kind.
     *  Enqueues on `stream` (the leased workspace's own when null) and waits for it: the retry ladder needs to know
     *  whether any query outgrew its scratch (one 8-byte read-back).
     */
    const char* search_device(const void* queries, std::size_t count, std::size_t stride_bytes, std::size_t wanted,
                              std::size_t expansion, std::uint64_t* keys, float* distances, std::uint64_t* counts,
                              std::uint64_t* visited, std::uint64_t* computed, hipStream_t stream,
                              const search_tuning_t& tuning, search_stats_t* stats, bool timed,
                              const search_extras_t* extras = nullptr);

    /**
     *  TUNES WHERE THE MATRIX SITS, on the caller's own sample batch (round 6; `usearch_amd_snapshot_tune`). Which frames of HBM the
     *  matrix of stored rows received decides how fast the walk runs over it — 44.4 … 51.4 ms for the headline batch over the same
     *  bytes, per box and per what the process allocated before (profiles/r06_settled/) — and nothing but the walk itself tells the
     *  placements apart. So a host that is about to serve one shape of batch hands over a sample of it: up to `max_trials` times a
     *  fresh device-to-device copy of the matrix is placed and timed against the incumbent on the sample's first queries (one per
     *  resident wave, at the sample's expansion: `try_matrix_placement`), the faster stays, the other is freed — and because the
     *  driver hands freed frames back late, consecutive copies land on different frames. Three wins of the incumbent in a row end
     *  it early. EXPLICIT and synchronous: nothing of it ever happens inside a search call (round 5 ran these trials there; the
     *  advisor's finding), a second copy of the matrix exists in HBM only during this call, and the matrix does not move afterwards.
     *  Arrays under 1 GiB, inline-row snapshots and samples that do not fill the chip are left alone (returns 0 trials).
     */
    const char* tune(const void* queries, std::size_t count, std::size_t stride_bytes, std::size_t wanted, std::size_t expansion,
                     std::uint32_t max_trials, std::uint32_t* trials_made);

    /**
     *  The same search in two halves, for callers that append their own work to the stream before anybody waits (the
     *  sharded step: search → all-gather → merge, one stream, one wait): `search_begin` sizes the scratch, leases a
     *  workspace and launches; `search_finish` waits, and re-runs what outgrew its scratch (`reran` tells the caller that
     *  outputs changed after its appended work read them). A call object is used once.
     */
    struct search_call_t {
        workspace_t* workspace = nullptr;
        hipStream_t stream = nullptr;
        search_args_t args{};
        launch_params_t params{};
        std::size_t count = 0;
        std::uint32_t ef = 0, hash_cap = 0, next_cap = 0, query_lds = 0, entries_per_lane = 0, waves_cap = 0;
        int mode = 0;
        bool timed = false, want_phases = false, want_clock = false, done = false, reran = false;
        bool plain_possible = false; ///< short rows: nothing known at search_begin rules the build cut for plain batches out (`plain_ak`)
        bool keep_workspace = false; ///< search_finish leaves the workspace with the caller (who gives it back)
        float total_ms = 0.f;
        std::uint32_t passes = 0;
        search_stats_t stats{};
        std::vector<std::uint32_t> todo;
        bool have_todo = false;
    };
    const char* search_begin(search_call_t& call, const void* queries, std::size_t count, std::size_t stride_bytes,
                             std::size_t wanted, std::size_t expansion, std::uint64_t* keys, float* distances,
                             std::uint64_t* counts, std::uint64_t* visited, std::uint64_t* computed, hipStream_t stream,
                             const search_tuning_t& tuning, bool timed, const search_extras_t* extras = nullptr);
    const char* search_finish(search_call_t& call, search_stats_t* stats);

    /**
     *  Construction support (build.hip): allocates the HBM arrays of an index with room for `capacity` members and
     *  `lists_capacity` upper-level lists, every list empty; `append_for_build` uploads members, `grow_for_build` makes room
     *  for more. The graph arrays are filled in place by the link kernels while `set_frontier` tells the search how much of
     *  the graph exists.
     */
    const char* allocate_for_build(metric_kind_t metric, scalar_kind_t scalar, std::size_t dimensions,
                                   std::uint64_t capacity, std::uint64_t lists_capacity, std::uint32_t m, std::uint32_t m0,
                                   int device);
    /// Re-allocates the build arrays for `capacity` members / `lists_capacity` upper-level lists, keeping what is linked.
    const char* grow_for_build(std::uint64_t capacity, std::uint64_t lists_capacity);
    /// Uploads members [first, first + count): rows (re-pitched), keys (null = the slot number), upper-list references.
    const char* append_for_build(std::uint64_t first, std::uint64_t count, const std::uint32_t* upper_refs,
                                 const void* vectors, std::size_t stride, bool vectors_on_device, const std::uint64_t* keys);
    /// Overwrites one member's stored row (host memory, storage kind) and key in HBM: a recycled slot (builder_t::update).
    const char* overwrite_member(std::uint64_t slot, const void* vector, std::uint64_t key);
    /// Overwrites one member's key in HBM (rename; `free_key_k` = tombstone).
    const char* set_key(std::uint64_t slot, std::uint64_t key);
    /// Counts what changes WHO the members are without necessarily changing how many there are (`append_for_build`,
    /// `overwrite_member`, `set_key`): a `filter_t` made before is a bitmap of the keys as they were (filter.hpp: `check`).
    std::uint64_t mutations() const { return mutations_; }
    std::uint64_t build_capacity() const { return build_capacity_; }
    std::uint64_t build_lists_capacity() const { return build_lists_capacity_; }
    void set_upper_lists(std::uint64_t lists) { upper_lists_ = lists; }
    void set_frontier(std::uint64_t size, std::uint32_t entry_slot, std::uint32_t max_level) {
        view_.size = size, view_.entry_slot = entry_slot, view_.max_level = max_level;
    }
    void set_tombstones(bool any) { view_.has_tombstones = any ? 1u : 0u; }
    /// Once the graph no longer changes: lays the stored rows of every node's level-0 neighbours next to each other
    /// (`snapshot_view_t::nbr0_rows`) when rows are a single 16-byte chunk — b1 × 128, haversine … — so that a hop reads one
    /// contiguous block. Costs size × M0 × 16 bytes of HBM; USEARCH_AMD_INLINE_ROWS=0 turns it off.
    const char* finalize_layout();
    void suspend_layout() { view_.nbr0_rows = nullptr; } ///< while the lists are being rewritten (construction)
    std::uint32_t* mutable_nbr0() { return static_cast<std::uint32_t*>(d_nbr0_); }
    std::uint32_t* mutable_upper() { return static_cast<std::uint32_t*>(d_upper_); }
    hipStream_t stream() const { return stream_; }
    int compute_units() const { return compute_units_; }

    /// How many batches may be in flight at once (`usearch_change_threads_search`): the size of the workspace pool.
    void set_concurrency(std::size_t workspaces);
    std::size_t concurrency() const { return max_workspaces_; }

    /// Same with host buffers and a query scalar kind that may differ from the storage kind (cast first).
    const char* search_host(const void* queries, scalar_kind_t query_kind, std::size_t count, std::size_t stride_bytes,
                            std::size_t wanted, std::size_t expansion, std::uint64_t* keys, float* distances,
                            std::uint64_t* counts, std::uint64_t* visited, std::uint64_t* computed,
                            const search_tuning_t& tuning, search_stats_t* stats,
                            const std::uint32_t* allow_bits_host = nullptr, const search_extras_t* more = nullptr);

    /// `index_dense_gt::cluster(query, level)` for a batch (index_dense.hpp:788-793 → index.hpp:3089-3125): host buffers, any
    /// query scalar kind; keys / distances / visited / computed are [count] arrays.
    const char* cluster_host(const void* queries, scalar_kind_t query_kind, std::size_t count, std::size_t stride_bytes,
                             std::size_t level, std::uint64_t* keys, float* distances, std::uint64_t* visited,
                             std::uint64_t* computed);

    /// Telemetry of the last search_device call: per query {peak frontier size, visited-set size}; host copy.
    const char* last_peaks(std::uint32_t* out, std::size_t queries);

    /// `search(…, exact = true)` for a batch (index_dense.hpp:767-772 with exact, index.hpp:3046-3049): device buffers,
    /// queries in the storage kind. `allow_bits` (device, one bit per slot): the caller's predicate, which the brute-force scan
    /// applies too (index.hpp:4260-4263).
    const char* exact_device(const void* queries, std::size_t count, std::size_t stride_bytes, std::size_t wanted,
                             std::uint64_t* keys, float* distances, std::uint64_t* counts, hipStream_t stream,
                             float* kernel_ms, bool tiled = false, const std::uint32_t* allow_bits = nullptr);
    /// Same with host buffers and any query scalar kind. `tiled`: the matrix-unit kernel where one exists for the pair
    /// (exact_tiled.hip; an error otherwise) instead of the bit-exact wave-per-query one.
    const char* exact_host(const void* queries, scalar_kind_t query_kind, std::size_t count, std::size_t stride_bytes,
                           std::size_t wanted, std::uint64_t* keys, float* distances, std::uint64_t* counts,
                           float* kernel_ms, bool tiled = false, const std::uint32_t* allow_bits = nullptr);

    /// out[q][j] = metric(query q, stored row slots[q][j]); host buffers, queries in storage kind.
    const char* distances_host(const void* queries, std::size_t count, std::size_t stride_bytes,
                               const std::uint32_t* slots, std::size_t slots_per_query, float* out);

    /// Leases a workspace (waits while `concurrency()` of them are out); `give_back` returns it. RAII: `lease_t`.
    const char* take(workspace_t*& out);
    void give_back(workspace_t* workspace);
    struct lease_t {
        snapshot_t& owner;
        workspace_t* workspace = nullptr;
        explicit lease_t(snapshot_t& s) : owner(s) {}
        ~lease_t() {
            if (workspace)
                owner.give_back(workspace);
        }
        const char* take() { return owner.take(workspace); }
    };

  private:
    const char* run_ladder(search_call_t& call);
    /// One placement trial of the matrix of stored rows inside a launch that fills the chip (placement.hpp): a fresh device-to-device
    /// copy, incumbent and candidate timed alternately over the launch's first `grid` queries (`launch(view, ms)` runs them once and
    /// times them), the faster one stays. Needs the matrix to itself (no other batch in flight); does nothing otherwise.
    const char* try_matrix_placement(std::uint32_t expansion, const std::function<const char*(const snapshot_view_t&, float&)>& launch, hipStream_t stream);
    void release();

    snapshot_view_t view_{};
    metric_kind_t metric_ = metric_unknown_k;
    scalar_kind_t scalar_ = scalar_unknown_k;
    std::uint32_t lanes_ = 1;
    int device_ = 0;
    std::size_t device_bytes_ = 0;
    std::uint64_t count_present_ = 0, upper_lists_ = 0, mutations_ = 0;

    void* d_vectors_ = nullptr;
    void* d_nbr0_ = nullptr;
    void* d_upper_ref_ = nullptr;
    void* d_upper_ = nullptr;
    void* d_keys_ = nullptr;
    void* d_nbr0_rows_ = nullptr;
    std::uint64_t build_capacity_ = 0, build_lists_capacity_ = 0; ///< room in the arrays above while an index is under construction

    placement_t placement_{};
    std::uint32_t placement_trials_left_ = placement_max_draws_k; ///< trials `try_matrix_placement` may still make
    std::uint32_t placement_losses_ = 0;                          ///< trials in a row the incumbent has won
    std::uint32_t placement_last_ef_ = 0, placement_reopens_ = 0; ///< the expansion of the last trial; how often a wider regime reopened the search
    bool placing_ = false;            ///< a trial owns the matrix: `take` waits (guarded by pool_mutex_)
    std::uint32_t tuning_trials_ = 0; ///< > 0 only inside `tune`: the trials `run_ladder` may make there (0: the environment decides)
    std::size_t vectors_bytes_ = 0;   ///< bytes allocated behind d_vectors_
    int compute_units_ = 256;
    float last_distances_ms_ = 0.f;
    hipStream_t stream_ = nullptr; ///< the snapshot's own stream: construction and one-off kernels

    std::mutex pool_mutex_; ///< guards the pool below, never held while a batch runs
    std::condition_variable pool_ready_;
    std::vector<std::unique_ptr<workspace_t>> workspaces_;
    std::vector<workspace_t*> idle_;
    workspace_t* last_used_ = nullptr;
    std::size_t max_workspaces_ = 16;
};

#define USEARCH_AMD_DECLARE_LAUNCHERS(name)                                                                            \
    hipError_t launch_search_##name(const launch_params_t&, const snapshot_view_t&, const search_args_t&);            \
    hipError_t launch_distances_##name(const struct distances_params_t&, const snapshot_view_t&);                     \
    hipError_t launch_exact_##name(const struct exact_params_t&, const snapshot_view_t&);                             \
    hipError_t launch_build_##name(const struct build_params_t&, const snapshot_view_t&, const struct build_args_t&);
/// Every (metric, scalar) pair with a HIP kernel — the reference's own dispatch table, `configure_with_autovec`
/// (index_plugins.hpp:1930-2008); jaccard over bit sets is tanimoto there (2003-2004) and is mapped onto it at the boundary.
#define USEARCH_AMD_FOR_EACH_PAIR(X) \
    X(metric_ip_k, scalar_f32_k, ip_f32)\
    X(metric_cos_k, scalar_f32_k, cos_f32)\
    X(metric_l2sq_k, scalar_f32_k, l2sq_f32)\
    X(metric_pearson_k, scalar_f32_k, pearson_f32)\
    X(metric_ip_k, scalar_f16_k, ip_f16)\
    X(metric_cos_k, scalar_f16_k, cos_f16)\
    X(metric_l2sq_k, scalar_f16_k, l2sq_f16)\
    X(metric_pearson_k, scalar_f16_k, pearson_f16)\
    X(metric_ip_k, scalar_bf16_k, ip_bf16)\
    X(metric_cos_k, scalar_bf16_k, cos_bf16)\
    X(metric_l2sq_k, scalar_bf16_k, l2sq_bf16)\
    X(metric_pearson_k, scalar_bf16_k, pearson_bf16)\
    X(metric_ip_k, scalar_f64_k, ip_f64)\
    X(metric_cos_k, scalar_f64_k, cos_f64)\
    X(metric_l2sq_k, scalar_f64_k, l2sq_f64)\
    X(metric_pearson_k, scalar_f64_k, pearson_f64)\
    X(metric_ip_k, scalar_i8_k, ip_i8)\
    X(metric_cos_k, scalar_i8_k, cos_i8)\
    X(metric_l2sq_k, scalar_i8_k, l2sq_i8)\
    X(metric_pearson_k, scalar_i8_k, pearson_i8)\
    X(metric_divergence_k, scalar_f32_k, divergence_f32)\
    X(metric_divergence_k, scalar_f16_k, divergence_f16)\
    X(metric_divergence_k, scalar_bf16_k, divergence_bf16)\
    X(metric_divergence_k, scalar_f64_k, divergence_f64)\
    X(metric_haversine_k, scalar_f32_k, haversine_f32)\
    X(metric_haversine_k, scalar_f64_k, haversine_f64)\
    X(metric_hamming_k, scalar_b1x8_k, hamming_b1)\
    X(metric_tanimoto_k, scalar_b1x8_k, tanimoto_b1)\
    X(metric_sorensen_k, scalar_b1x8_k, sorensen_b1)
#define USEARCH_AMD_DECLARE_PAIR(metric_kind, scalar_kind, name) USEARCH_AMD_DECLARE_LAUNCHERS(name)
USEARCH_AMD_FOR_EACH_PAIR(USEARCH_AMD_DECLARE_PAIR)

/// Launch shape of one construction linking kernel (build_kernels.hpp): `reverse` = 0 select, 1 reverse links.
struct build_params_t {
    std::uint32_t lanes;
    std::uint32_t grid;
    int reverse;
    hipStream_t stream;
};

struct distances_params_t {
    metric_kind_t metric;
    std::uint32_t lanes;
    std::uint32_t lds_bytes;
    hipStream_t stream;
    const std::uint8_t* queries;
    std::uint64_t query_stride;
    const std::uint32_t* slots;
    std::uint32_t slots_per_query;
    std::uint32_t count;
    float* out;
};
struct exact_params_t {
    std::uint32_t lanes;
    std::uint32_t lds_bytes;
    hipStream_t stream;
    const std::uint8_t* queries;
    std::uint64_t query_stride;
    std::uint32_t query_count;
    std::uint32_t wanted;
    std::uint32_t partitions;
    std::uint64_t rows_per_partition;
    std::uint32_t map_keys;
    const std::uint32_t* allow_bits; ///< optional, device: one bit per slot, 0 = the caller's predicate rejects that member
    float* out_distances;      ///< [partitions][queries][wanted]
    std::uint64_t* out_keys;
    std::uint64_t* out_counts; ///< [partitions][queries]
};

/**
 *  Exact search over the rows of `view` (device pointers everywhere; `view` needs `vectors`, sizes and — when
 *  `map_keys` — `keys`): partition scan + fold. Results as `index_gt::search_exact_` gives them: top-`wanted` under
 *  (distance ↑, slot ↓), padded with key 0 / signalling NaN.
 */
const char* exact_search_device(metric_kind_t metric, scalar_kind_t scalar, std::uint32_t lanes,
                                const snapshot_view_t& view, const void* queries, std::size_t count,
                                std::size_t stride_bytes, std::size_t wanted, bool map_keys, std::uint64_t* keys,
                                float* distances, std::uint64_t* counts, hipStream_t stream, float* kernel_ms,
                                const std::uint32_t* allow_bits = nullptr);

/// Is there a tiled (matrix-unit) exact-search kernel for this pair and result count? (exact_tiled.hip: cos / ip over f16 and
/// bf16 — float tolerance — and cos / ip / l2sq over i8 — bit-identical to the wave-per-query kernel.)
bool exact_tiled_available(metric_kind_t metric, scalar_kind_t scalar, std::size_t wanted);

/// Many-to-many exact search as a tiled matrix product (index_plugins.hpp:2071-2164): rows are read once per 64 queries. Same
/// contract as `exact_search_device`.
const char* exact_search_tiled_device(metric_kind_t metric, scalar_kind_t scalar, const snapshot_view_t& view,
                                      const void* queries, std::size_t count, std::size_t stride_bytes, std::size_t wanted,
                                      bool map_keys, std::uint64_t* keys, float* distances, std::uint64_t* counts,
                                      hipStream_t stream, float* kernel_ms, const std::uint32_t* allow_bits = nullptr);

/// Host-buffer exact search of a raw dataset — `usearch_exact_search` (c/usearch.h:467-474): keys are dataset offsets.
const char* exact_search_dataset_host(metric_kind_t metric, scalar_kind_t scalar, std::size_t dimensions,
                                      const void* dataset, std::size_t dataset_count, std::size_t dataset_stride,
                                      const void* queries, std::size_t queries_count, std::size_t queries_stride,
                                      std::size_t wanted, std::uint64_t* keys, std::size_t keys_stride,
                                      float* distances, std::size_t distances_stride);

/// Lanes per stored row, the row pitch and the 16-byte chunks the kernels read of a row of `bytes` bytes (the summation layout
/// of DESIGN.md §3.3; the pitch may exceed chunks × 16 so that short rows never straddle a 128-byte line).
void row_geometry(std::size_t bytes, std::uint32_t& lanes, std::uint32_t& row_stride, std::uint32_t& chunks);

/// Exchange step of sharded search (merge.hip): [shards][queries][wanted] per-shard results → [queries][wanted], device
/// pointers, `merge_into` tie rule with shards merged in index order. Returns after the stream has drained.
const char* merge_shards_device(const float* distances, const std::uint64_t* keys, const std::uint64_t* counts,
                                std::size_t shards, std::size_t queries, std::size_t wanted, float* out_distances,
                                std::uint64_t* out_keys, std::uint64_t* out_counts, hipStream_t stream,
                                bool later_position_first = true);

/// The same merge without the wait, shards `*_stride` ELEMENTS apart (a packed all-gather block keeps the three arrays of one
/// shard together, so the strides differ from the dense [shards][queries][wanted] case).
const char* merge_shards_enqueue(const float* distances, const std::uint64_t* keys, const std::uint64_t* counts,
                                 std::uint64_t distances_stride, std::uint64_t keys_stride, std::uint64_t counts_stride,
                                 std::size_t shards, std::size_t queries, std::size_t wanted, float* out_distances,
                                 std::uint64_t* out_keys, std::uint64_t* out_counts, hipStream_t stream,
                                 bool later_position_first = true);

/// Is there a HIP kernel for this (metric, scalar) pair?
bool kernel_available(metric_kind_t metric, scalar_kind_t scalar);

} // namespace usearch_amd
