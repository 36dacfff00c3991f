/**
 *  include/usearch_amd.hpp — header-only C++17 face of the MI355X engine, shaped like the part of
 *  `unum::usearch::index_dense_gt` C++ callers use (/root/reference/include/usearch/index_dense.hpp:644-805:
 *  `make`, `add`, `search`, `filtered_search`, `get`, `contains`, `remove`, `rename`, `save`, `load`, `view`, `size`, …),
 *  so that code written against the reference's class reads the same:
 *
 *      auto made = usearch_amd::index_dense_t::make(768, usearch_metric_cos_k, usearch_scalar_f16_k);   // index_dense.hpp:644
 *      usearch_amd::index_dense_t index = std::move(made.index);
 *      index.add(42, vector);                                                                            // :764
 *      auto result = index.search(query, 10);                                                            // :771
 *      result.dump_to(keys, distances);                                                                  // index.hpp:2678
 *
 *  It is a veneer over the C ABI of include/usearch_c_dropin.h (link with `-lusearch_c` from usearch_amd/lib): no device
 *  code here, no state beyond the handle. `search_many` is the batched call the reference leaves to its callers
 *  (cpp/bench.cpp:352-377). Errors come back the reference's way: a result object whose `error` is a static C string, or
 *  empty; nothing throws.
 */
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "usearch_c_dropin.h"

namespace usearch_amd {

using vector_key_t = usearch_key_t;   ///< `default_key_t`, index_dense.hpp:2229
using distance_t = usearch_distance_t;
using f16_bits_t = std::uint16_t;     ///< half-precision scalars travel as their bit pattern (index_plugins.hpp:394-428)
struct bf16_bits_t {                  ///< brain floats travel as their bit pattern too (index_plugins.hpp:430-470): a distinct
    std::uint16_t bits;               ///< type, so that the overloads below can tell them from IEEE halves
};
struct b1x8_t {                       ///< eight bits per byte, MSB first (index_plugins.hpp:1139-1158)
    std::uint8_t byte;
};

/// `index_dense_config_t` (index_dense.hpp:95-126), the fields that matter on the device.
struct index_dense_config_t {
    std::size_t connectivity = 16;
    std::size_t expansion_add = 128;
    std::size_t expansion_search = 64;
    bool multi = false;
};

/// What `search` returns — `search_result_t` (index.hpp:2595-2742): results are copied out, not views onto a context.
struct search_result_t {
    struct match_t {
        vector_key_t key;
        distance_t distance;
    };
    std::vector<vector_key_t> keys;
    std::vector<distance_t> distances;
    std::size_t count = 0;
    std::size_t visited_members = 0, computed_distances = 0;
    char const* error = nullptr;

    explicit operator bool() const noexcept { return !error; }
    std::size_t size() const noexcept { return count; }
    bool empty() const noexcept { return !count; }
    match_t operator[](std::size_t i) const noexcept { return {keys[i], distances[i]}; }
    /// `dump_to` (index.hpp:2678-2741).
    std::size_t dump_to(vector_key_t* out_keys, distance_t* out_distances) const noexcept {
        for (std::size_t i = 0; i != count; ++i)
            out_keys[i] = keys[i], out_distances[i] = distances[i];
        return count;
    }
};

struct add_result_t {
    char const* error = nullptr;
    explicit operator bool() const noexcept { return !error; }
};
using serialization_result_t = add_result_t;

struct state_result_t;

class index_dense_t {
    usearch_index_t handle_ = nullptr;

  public:
    using state_result_t = usearch_amd::state_result_t; ///< `index_dense_gt::state_result_t`: {index, error}

    index_dense_t() = default;
    index_dense_t(index_dense_t&& other) noexcept : handle_(std::exchange(other.handle_, nullptr)) {}
    index_dense_t& operator=(index_dense_t&& other) noexcept {
        std::swap(handle_, other.handle_);
        return *this;
    }
    index_dense_t(index_dense_t const&) = delete;
    index_dense_t& operator=(index_dense_t const&) = delete;
    ~index_dense_t() {
        usearch_error_t error = nullptr;
        if (handle_)
            usearch_free(handle_, &error);
    }

    /// `index_dense_gt::make(metric, config)` (index_dense.hpp:644-679) and `make(path, view)` (681-700).
    static state_result_t make(std::size_t dimensions, usearch_metric_kind_t metric, usearch_scalar_kind_t scalar,
                               index_dense_config_t config = {});
    static state_result_t make(char const* path, bool view = false);

    explicit operator bool() const noexcept { return handle_ != nullptr; }
    std::size_t size() const noexcept { usearch_error_t e = nullptr; return usearch_size(handle_, &e); }
    std::size_t capacity() const noexcept { usearch_error_t e = nullptr; return usearch_capacity(handle_, &e); }
    std::size_t dimensions() const noexcept { usearch_error_t e = nullptr; return usearch_dimensions(handle_, &e); }
    std::size_t connectivity() const noexcept { usearch_error_t e = nullptr; return usearch_connectivity(handle_, &e); }
    std::size_t expansion_search() const noexcept { usearch_error_t e = nullptr; return usearch_expansion_search(handle_, &e); }
    void change_expansion_search(std::size_t n) noexcept { usearch_error_t e = nullptr; usearch_change_expansion_search(handle_, n, &e); }
    bool try_reserve(std::size_t n) noexcept { usearch_error_t e = nullptr; usearch_reserve(handle_, n, &e); return !e; }
    bool contains(vector_key_t key) const noexcept { usearch_error_t e = nullptr; return usearch_contains(handle_, key, &e); }
    std::size_t count(vector_key_t key) const noexcept { usearch_error_t e = nullptr; return usearch_count(handle_, key, &e); }
    std::size_t remove(vector_key_t key) noexcept { usearch_error_t e = nullptr; return usearch_remove(handle_, key, &e); }
    std::size_t rename(vector_key_t from, vector_key_t to) noexcept { usearch_error_t e = nullptr; return usearch_rename(handle_, from, to, &e); }

    // ---- add: one overload per scalar type, index_dense.hpp:760-765
    add_result_t add(vector_key_t key, float const* vector) { return add_(key, vector, usearch_scalar_f32_k); }
    add_result_t add(vector_key_t key, f16_bits_t const* vector) { return add_(key, vector, usearch_scalar_f16_k); }
    add_result_t add(vector_key_t key, std::int8_t const* vector) { return add_(key, vector, usearch_scalar_i8_k); }
    add_result_t add(vector_key_t key, b1x8_t const* vector) { return add_(key, vector, usearch_scalar_b1_k); }
    add_result_t add(vector_key_t key, double const* vector) { return add_(key, vector, usearch_scalar_f64_k); }
    add_result_t add(vector_key_t key, bf16_bits_t const* vector) { return add_(key, vector, usearch_scalar_bf16_k); }

    // ---- search: index_dense.hpp:767-772 (`thread` is accepted and ignored: the batch is the parallelism)
    search_result_t search(float const* q, std::size_t wanted, std::size_t = 0, bool exact = false) const { return search_(q, usearch_scalar_f32_k, 1, 0, wanted, exact); }
    search_result_t search(f16_bits_t const* q, std::size_t wanted, std::size_t = 0, bool exact = false) const { return search_(q, usearch_scalar_f16_k, 1, 0, wanted, exact); }
    search_result_t search(std::int8_t const* q, std::size_t wanted, std::size_t = 0, bool exact = false) const { return search_(q, usearch_scalar_i8_k, 1, 0, wanted, exact); }
    search_result_t search(b1x8_t const* q, std::size_t wanted, std::size_t = 0, bool exact = false) const { return search_(q, usearch_scalar_b1_k, 1, 0, wanted, exact); }
    search_result_t search(double const* q, std::size_t wanted, std::size_t = 0, bool exact = false) const { return search_(q, usearch_scalar_f64_k, 1, 0, wanted, exact); }
    search_result_t search(bf16_bits_t const* q, std::size_t wanted, std::size_t = 0, bool exact = false) const { return search_(q, usearch_scalar_bf16_k, 1, 0, wanted, exact); }

    /// The whole batch in one call: row `i` of the results holds query `i`'s `wanted` cells (`counts[i]` of them filled).
    struct batch_result_t {
        std::vector<vector_key_t> keys;
        std::vector<distance_t> distances;
        std::vector<std::size_t> counts;
        std::size_t visited_members = 0, computed_distances = 0;
        char const* error = nullptr;
        explicit operator bool() const noexcept { return !error; }
    };
    batch_result_t search_many(void const* queries, usearch_scalar_kind_t kind, std::size_t queries_count,
                               std::size_t queries_stride_bytes, std::size_t wanted) const {
        batch_result_t result;
        result.keys.resize(queries_count * wanted), result.distances.resize(queries_count * wanted);
        result.counts.resize(queries_count);
        usearch_search_many(handle_, queries, kind, queries_count, queries_stride_bytes, wanted, result.keys.data(),
                            wanted * sizeof(vector_key_t), result.distances.data(), wanted * sizeof(distance_t),
                            result.counts.data(), &result.visited_members, &result.computed_distances, &result.error);
        return result;
    }

    /// `cluster(vector, level)` (index_dense.hpp:788-793): the member the greedy descent reaches on `level` of the hierarchy.
    struct cluster_result_t {
        vector_key_t key = 0;
        distance_t distance = 0;
        char const* error = nullptr;
        explicit operator bool() const noexcept { return !error; }
    };
    cluster_result_t cluster(float const* q, std::size_t level, std::size_t = 0) const { return cluster_(q, usearch_scalar_f32_k, level); }
    cluster_result_t cluster(f16_bits_t const* q, std::size_t level, std::size_t = 0) const { return cluster_(q, usearch_scalar_f16_k, level); }
    cluster_result_t cluster(std::int8_t const* q, std::size_t level, std::size_t = 0) const { return cluster_(q, usearch_scalar_i8_k, level); }
    cluster_result_t cluster(b1x8_t const* q, std::size_t level, std::size_t = 0) const { return cluster_(q, usearch_scalar_b1_k, level); }
    cluster_result_t cluster(double const* q, std::size_t level, std::size_t = 0) const { return cluster_(q, usearch_scalar_f64_k, level); }
    cluster_result_t cluster(bf16_bits_t const* q, std::size_t level, std::size_t = 0) const { return cluster_(q, usearch_scalar_bf16_k, level); }

    /// `filtered_search` (index_dense.hpp:774-779): `predicate(key) -> bool`, evaluated on the host once per member.
    template <typename predicate_at>
    search_result_t filtered_search(float const* query, std::size_t wanted, predicate_at&& predicate) const {
        search_result_t result;
        result.keys.resize(wanted), result.distances.resize(wanted);
        auto trampoline = [](usearch_key_t key, void* state) -> int { return (*static_cast<predicate_at*>(state))(key) ? 1 : 0; };
        result.count = usearch_filtered_search(handle_, query, usearch_scalar_f32_k, wanted, +trampoline, (void*)&predicate,
                                               result.keys.data(), result.distances.data(), &result.error);
        return result;
    }

    std::size_t get(vector_key_t key, float* vector, std::size_t vectors_count = 1) const {
        usearch_error_t e = nullptr;
        return usearch_get(handle_, key, vectors_count, vector, usearch_scalar_f32_k, &e);
    }

    serialization_result_t save(char const* path) const { serialization_result_t r; usearch_save(handle_, path, &r.error); return r; }
    serialization_result_t load(char const* path) { serialization_result_t r; usearch_load(handle_, path, &r.error); return r; }
    serialization_result_t view(char const* path) { serialization_result_t r; usearch_view(handle_, path, &r.error); return r; }
    std::size_t serialized_length() const { usearch_error_t e = nullptr; return usearch_serialized_length(handle_, &e); }
    serialization_result_t save_to_buffer(void* buffer, std::size_t length) const { serialization_result_t r; usearch_save_buffer(handle_, buffer, length, &r.error); return r; }
    serialization_result_t load_from_buffer(void const* buffer, std::size_t length) { serialization_result_t r; usearch_load_buffer(handle_, buffer, length, &r.error); return r; }

  private:
    add_result_t add_(vector_key_t key, void const* vector, usearch_scalar_kind_t kind) {
        add_result_t result;
        usearch_add(handle_, key, vector, kind, &result.error);
        return result;
    }
    cluster_result_t cluster_(void const* query, usearch_scalar_kind_t kind, std::size_t level) const {
        cluster_result_t result;
        usearch_cluster_many(handle_, query, kind, 1, 0, level, &result.key, &result.distance, &result.error);
        return result;
    }

    search_result_t search_(void const* query, usearch_scalar_kind_t kind, std::size_t, std::size_t, std::size_t wanted,
                            bool exact) const {
        search_result_t result;
        result.keys.resize(wanted), result.distances.resize(wanted);
        if (exact) { // `search(…, exact = true)`: brute force over every member (index.hpp:3046-3049)
            result.error = "exact search of an index goes through usearch_amd_exact_search_many (include/usearch_amd.h)";
            return result;
        }
        result.count = usearch_search(handle_, query, kind, wanted, result.keys.data(), result.distances.data(), &result.error);
        return result;
    }
};

struct state_result_t {
    index_dense_t index;
    char const* error = nullptr;
    explicit operator bool() const noexcept { return !error; }
};

inline state_result_t index_dense_t::make(std::size_t dimensions, usearch_metric_kind_t metric, usearch_scalar_kind_t scalar,
                                          index_dense_config_t config) {
    state_result_t result;
    usearch_init_options_t options{};
    options.metric_kind = metric, options.quantization = scalar, options.dimensions = dimensions;
    options.connectivity = config.connectivity, options.expansion_add = config.expansion_add;
    options.expansion_search = config.expansion_search, options.multi = config.multi;
    result.index.handle_ = usearch_init(&options, &result.error);
    return result;
}

inline state_result_t index_dense_t::make(char const* path, bool view) {
    state_result_t result;
    result.index.handle_ = usearch_init(nullptr, &result.error);
    if (!result.error)
        (view ? usearch_view : usearch_load)(result.index.handle_, path, &result.error);
    return result;
}

} // namespace usearch_amd
