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
#include <vector>

#include "engine.hpp"

namespace usearch_amd {

struct build_config_t {
    std::uint32_t connectivity = 16;      ///< M,  index.hpp `default_connectivity()`
    std::uint32_t connectivity_base = 0;  ///< M0, 0 = 2·M (index.hpp:1867)
    std::uint32_t expansion_add = 128;    ///< ef_construction, index.hpp `default_expansion_add()`
    std::uint64_t seed = 0x5eed5eedull;   ///< level draw
    std::uint32_t batch_divisor = 16;     ///< a batch holds at most (nodes already linked) / divisor nodes …
    std::uint32_t max_batch = 65536;      ///< … and at most this many
};

struct build_stats_t {
    std::uint64_t batches = 0, passes = 0;
    std::uint64_t search_distances = 0; ///< `computed_distances` of the insertion searches
    std::uint64_t search_hops = 0;
    std::uint64_t select_distances = 0, reverse_distances = 0, repruned_lists = 0, dropped_requests = 0;
    double seconds_total = 0, seconds_search = 0, seconds_link = 0, seconds_upload = 0;
    std::uint32_t max_level = 0;
};

class builder_t {
  public:
    /**
     *  Builds the index of `count` vectors (`stride` bytes apart, storage scalar kind, host or device memory).
     *  `keys` (host) may be null: key = row number. Returns nullptr or a static message.
     */
    const char* build(metric_kind_t metric, scalar_kind_t scalar, std::size_t dimensions, const void* vectors,
                      std::uint64_t count, std::size_t stride, bool vectors_on_device, const std::uint64_t* keys,
                      const build_config_t& config, int device);

    snapshot_t& snapshot() { return snapshot_; }
    const build_stats_t& stats() const { return stats_; }

    /// Size and content of the reference's serialized form (index_dense.hpp:995-1062, index.hpp:3277-3317).
    std::size_t serialized_length() const;
    const char* save_buffer(void* buffer, std::size_t length);

  private:
    snapshot_t snapshot_;
    build_config_t config_;
    build_stats_t stats_;
    std::vector<std::int16_t> levels_;
    std::vector<std::uint64_t> keys_; ///< empty = identity
    std::uint64_t size_ = 0, upper_lists_ = 0;
    std::uint32_t entry_slot_ = 0, max_level_ = 0;
    metric_kind_t metric_ = metric_unknown_k;
    scalar_kind_t scalar_ = scalar_unknown_k;
    std::size_t dimensions_ = 0;
};

} // namespace usearch_amd
