/**
 *  usearch_amd/csrc/build.hpp — batched HNSW construction on the MI355X: the host driver.
 *
 *  Replaces the `add` loop callers run over the reference (`usearch_add` per vector, c/usearch.h:338-339 →
 *  index_dense.hpp `add_` → `index_gt::add`, /root/reference/include/usearch/index.hpp:2759-2879; batch drivers
 *  cpp/bench.cpp:296-327, python/lib.cpp:194-252) by one call that links all vectors on the device and leaves a
 *  searchable snapshot plus a reference-compatible v2 image (`save_buffer`), so the reference can load what was built.
 *
 *  Nodes enter in slot order, in batches that grow with the graph; per batch and per level (bottom-up, so every read sees
 *  the pre-batch graph): insertion search (the search kernel on that level) → build_select_kernel → build_reverse_kernel.
 */
#pragma once
#include <cstdint>
#include <random>
#include <vector>

#include "engine.hpp"

namespace usearch_amd {

/// Largest `expansion_add` (and base connectivity) the link kernels take: a node's candidates are ranked by one wave
/// (build_kernels.hpp `build_max_candidates_k`; build.hip checks that the two agree).
constexpr std::uint32_t builder_max_expansion_k = 1024;
constexpr std::uint32_t builder_max_connectivity_base_k = 128;

struct build_config_t {
    std::uint32_t connectivity = 16;      ///< M,  index.hpp `default_connectivity()`
    std::uint32_t connectivity_base = 0;  ///< M0, 0 = 2·M (index.hpp:1867)
    std::uint32_t expansion_add = 128;    ///< ef_construction, index.hpp `default_expansion_add()`
    std::uint64_t seed = 0x5eed5eedull;   ///< level draw
    std::uint32_t batch_divisor = 16;     ///< a batch holds at most (nodes already linked) / divisor nodes …
    std::uint32_t max_batch = 65536;      ///< … and at most this many
    bool multi = false;                   ///< several members may share a key: written into the image's head (index_dense.hpp:1046)
};

struct build_stats_t {
    std::uint64_t batches = 0, passes = 0;
    std::uint64_t search_distances = 0; ///< `computed_distances` of the insertion searches
    std::uint64_t search_hops = 0;
    std::uint64_t select_distances = 0, reverse_distances = 0, repruned_lists = 0, dropped_requests = 0;
    std::uint64_t refiled_requests = 0; ///< reverse-link requests that waited a round for room in their target's inbox
    double seconds_total = 0, seconds_search = 0, seconds_link = 0, seconds_upload = 0;
    std::uint32_t max_level = 0;
};

class builder_t {
  public:
    builder_t() = default;
    ~builder_t();
    builder_t(const builder_t&) = delete;
    builder_t& operator=(const builder_t&) = delete;

    /**
     *  Builds the index of `count` vectors (`stride` bytes apart, storage scalar kind, host or device memory).
     *  `keys` (host) may be null: key = row number. Returns nullptr or a static message.
     */
    const char* build(metric_kind_t metric, scalar_kind_t scalar, std::size_t dimensions, const void* vectors,
                      std::uint64_t count, std::size_t stride, bool vectors_on_device, const std::uint64_t* keys,
                      const build_config_t& config, int device);

    /**
     *  Links `count` MORE vectors into the index built so far — what a run of `usearch_add` calls after the first build
     *  amounts to (index_gt::add, index.hpp:2780-2879, batch-deferred): the new members get the next slots and the next draws
     *  of the level generator, the device arrays grow geometrically, only the new members are searched for and linked.
     */
    const char* extend(const void* vectors, std::uint64_t count, std::size_t stride, bool vectors_on_device,
                       const std::uint64_t* keys, bool exact_capacity = false);

    /**
     *  Replaces members in place — what `usearch_add` does with a slot a `usearch_remove` freed (index_dense.hpp:1479-1511 pushes the
     *  slot into `free_keys_`, `add_` pops it and calls `index_gt::update`, index.hpp:2916-2999): member `slots[i]` gets vector i
     *  (host memory, `stride` bytes apart) and `keys[i]`, keeps its level, and is linked anew — insertion search that never names
     *  the member itself (`search_to_update_`), `form_links_to_closest_`, `form_reverse_links_`. Links other members still hold
     *  to the slot stay, as in the reference.
     */
    const char* update(const std::uint32_t* slots, std::uint64_t count, const void* vectors, std::size_t stride,
                       const std::uint64_t* keys);

    /// Renames a member in place; `free_key_k` makes it a tombstone (index_dense.hpp:1479-1511: it keeps routing, stops
    /// matching). No relinking.
    const char* set_key(std::uint64_t slot, std::uint64_t key);

    snapshot_t& snapshot() { return snapshot_; }
    const build_stats_t& stats() const { return stats_; }
    std::uint64_t size() const { return size_; }

    /// Size and content of the reference's serialized form (index_dense.hpp:995-1062, index.hpp:3277-3317).
    std::size_t serialized_length() const;
    const char* save_buffer(void* buffer, std::size_t length);

  private:
    const char* link_range(std::uint64_t begin, std::uint64_t end, const std::uint32_t* relinked = nullptr,
                           std::uint64_t relinked_count = 0);
    void release_workspace();

    snapshot_t snapshot_;
    build_config_t config_;
    build_stats_t stats_;
    std::vector<std::int16_t> levels_;
    std::vector<std::uint64_t> keys_; ///< unused while `identity_keys_`
    bool identity_keys_ = true;
    std::mt19937_64 generator_;
    std::uint64_t size_ = 0, upper_lists_ = 0;
    std::uint32_t entry_slot_ = 0, max_level_ = 0;
    metric_kind_t metric_ = metric_unknown_k;
    scalar_kind_t scalar_ = scalar_unknown_k;
    std::size_t dimensions_ = 0;
    // device workspace of the link passes, kept between `extend` calls
    std::vector<void*> workspace_;
    std::uint64_t workspace_nodes_ = 0, workspace_batch_ = 0;
    std::uint32_t *d_nodes_ = nullptr, *d_inbox_count_ = nullptr, *d_touched_ = nullptr, *d_touched_count_ = nullptr;
    std::uint64_t *d_cand_slots_ = nullptr, *d_cand_counts_ = nullptr, *d_visited_ = nullptr, *d_computed_ = nullptr;
    void* d_inbox_ = nullptr;
    float* d_cand_distances_ = nullptr;
    unsigned long long* d_counters_ = nullptr;
    // reverse-link requests that found an inbox full: two parking lots, filled and drained in turn (build_refile_kernel)
    std::uint32_t* d_deferred_targets_[2] = {nullptr, nullptr};
    void* d_deferred_requests_[2] = {nullptr, nullptr};
    std::uint32_t* d_deferred_count_ = nullptr; ///< [2]
    std::uint64_t unfiled_requests_ = 0;        ///< parked requests given up on (never, unless a re-filing round made no progress)
};

} // namespace usearch_amd
