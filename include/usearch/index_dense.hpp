/**
 *  include/usearch/index_dense.hpp — the `unum::usearch::index_dense_gt` C++ SURFACE of USearch v2.21 over the MI355X engine.
 *
 *  Python, Rust, Java, JavaScript, ObjC … bind the reference through this class, not through its C ABI; so does the
 *  reference's own `c/lib.cpp`. This header carries the same names in the same namespace with the same call shapes —
 *  `metric_punned_t(dimensions, kind, scalar)`, `index_dense_config_t`, `index_limits_t`, `index_dense_gt<key, slot>::make /
 *  try_reserve / add / search(query, wanted, thread, exact) / filtered_search / get / remove / rename / save / load / view / …`,
 *  `search_result_t::dump_to(keys, distances[, capacity])` (/root/reference/include/usearch/index_dense.hpp:644-805, 2229;
 *  index.hpp:1401-1415, 2595-2742; index_plugins.hpp:1659-2014) — so that a translation unit written against the reference compiles
 *  with nothing but the include path swapped. The proof is in the tests: the reference's own `c/lib.cpp`, compiled from where it
 *  lies against THIS header (`oracle/Makefile class_lib`), yields a `libusearch_c` whose `c/test.c` passes on the MI355X
 *  (tests/test_gpu_class.py), and `tests/cpp/bench_loop.cpp` runs cpp/bench.cpp:352-377's search loop verbatim.
 *
 *  What it is NOT: the reference's implementation. No graph, no metric, no allocator lives here — every call lands in the
 *  drop-in library (`usearch_amd/lib/libusearch_c.so`, include/usearch_c_dropin.h), reached through its function table
 *  (`usearch_amd_c_api`), because a file like `c/lib.cpp` defines functions named `usearch_*` itself. Differences a caller can
 *  observe are the drop-in's: `add` stages and the next search links, the `thread` argument is accepted and
 *  ignored (a batch is the parallelism: see `search_many`), user-defined metric functions are refused, results are copied out of
 *  the device instead of being views onto a thread's context.
 */
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../usearch_c_dropin.h"

#define USEARCH_VERSION_MAJOR 2
#define USEARCH_VERSION_MINOR 21
#define USEARCH_VERSION_PATCH 0

namespace unum {
namespace usearch {

using byte_t = char;
using f32_t = float;
using f64_t = double;
using i8_t = std::int8_t;
using default_key_t = std::uint64_t;
using default_slot_t = std::uint32_t;
using default_distance_t = float;

/// Half-precision scalars travel as their bit pattern (index_plugins.hpp:394-470); conversions on the host for convenience.
class f16_bits_t {
    std::uint16_t uint16_{};

  public:
    f16_bits_t() noexcept = default;
    f16_bits_t(float v) noexcept {
        std::uint32_t bits;
        std::memcpy(&bits, &v, 4);
        const std::uint32_t sign = (bits >> 16) & 0x8000u;
        const std::int32_t exponent = (std::int32_t)((bits >> 23) & 0xFF) - 127 + 15;
        std::uint32_t mantissa = bits & 0x7FFFFFu;
        if (((bits >> 23) & 0xFF) == 0xFF)
            uint16_ = (std::uint16_t)(sign | 0x7C00u | (mantissa ? 0x200u : 0u));
        else if (exponent >= 31)
            uint16_ = (std::uint16_t)(sign | 0x7C00u);
        else if (exponent <= 0) {
            if (exponent < -10)
                uint16_ = (std::uint16_t)sign;
            else {
                mantissa |= 0x800000u;
                const int shift = 14 - exponent;
                std::uint32_t half = mantissa >> shift;
                const std::uint32_t rest = mantissa & ((1u << shift) - 1), midpoint = 1u << (shift - 1);
                half += rest > midpoint || (rest == midpoint && (half & 1u));
                uint16_ = (std::uint16_t)(sign | half);
            }
        } else {
            std::uint32_t half = ((std::uint32_t)exponent << 10) | (mantissa >> 13);
            const std::uint32_t rest = mantissa & 0x1FFFu;
            half += rest > 0x1000u || (rest == 0x1000u && (half & 1u));
            uint16_ = (std::uint16_t)(sign | half);
        }
    }
    operator float() const noexcept {
        const std::uint32_t sign = (std::uint32_t)(uint16_ & 0x8000u) << 16, exponent = (uint16_ >> 10) & 0x1Fu, mantissa = uint16_ & 0x3FFu;
        std::uint32_t bits;
        if (exponent == 0x1F)
            bits = sign | 0x7F800000u | (mantissa << 13);
        else if (exponent)
            bits = sign | ((exponent + 112) << 23) | (mantissa << 13);
        else if (!mantissa)
            bits = sign;
        else {
            int shift = 0;
            std::uint32_t m = mantissa;
            while (!(m & 0x400u))
                m <<= 1, ++shift;
            bits = sign | ((std::uint32_t)(113 - shift) << 23) | ((m & 0x3FFu) << 13);
        }
        float v;
        std::memcpy(&v, &bits, 4);
        return v;
    }
    std::uint16_t bits() const noexcept { return uint16_; }
};
class bf16_bits_t {
    std::uint16_t uint16_{};

  public:
    bf16_bits_t() noexcept = default;
    bf16_bits_t(float v) noexcept { // truncation, as the reference narrows (index_plugins.hpp:453-469)
        std::uint32_t bits;
        std::memcpy(&bits, &v, 4);
        uint16_ = (std::uint16_t)(bits >> 16);
    }
    operator float() const noexcept {
        const std::uint32_t bits = (std::uint32_t)uint16_ << 16;
        float v;
        std::memcpy(&v, &bits, 4);
        return v;
    }
    std::uint16_t bits() const noexcept { return uint16_; }
};
using f16_t = f16_bits_t;
using bf16_t = bf16_bits_t;
/// Eight dimensions per byte, most significant bit first (index_plugins.hpp:1139-1158).
class b1x8_t {
    std::uint8_t byte_{};

  public:
    b1x8_t() noexcept = default;
    b1x8_t(std::uint8_t byte) noexcept : byte_(byte) {}
    operator std::uint8_t() const noexcept { return byte_; }
};

enum class metric_kind_t : std::uint8_t { // index_plugins.hpp:113-129
    unknown_k = 0,
    ip_k = 'i',
    cos_k = 'c',
    l2sq_k = 'e',
    pearson_k = 'p',
    haversine_k = 'h',
    divergence_k = 'd',
    hamming_k = 'b',
    tanimoto_k = 't',
    sorensen_k = 's',
    jaccard_k = 'j',
};
enum class scalar_kind_t : std::uint8_t { // index_plugins.hpp:131-159, the kinds vectors are stored in
    unknown_k = 0,
    b1x8_k = 1,
    bf16_k = 4,
    f64_k = 10,
    f32_k = 11,
    f16_k = 12,
    i8_k = 23,
};
enum class metric_punned_signature_t { array_array_k = 0, array_array_size_k, array_array_state_k }; // index_plugins.hpp:1659-1666

namespace amd_detail {
inline usearch_amd_c_api_t const& api() noexcept {
    static usearch_amd_c_api_t const* const table = usearch_amd_c_api();
    return *table;
}
inline usearch_metric_kind_t to_c(metric_kind_t kind) noexcept {
    switch (kind) {
    case metric_kind_t::ip_k: return usearch_metric_ip_k;
    case metric_kind_t::cos_k: return usearch_metric_cos_k;
    case metric_kind_t::l2sq_k: return usearch_metric_l2sq_k;
    case metric_kind_t::pearson_k: return usearch_metric_pearson_k;
    case metric_kind_t::haversine_k: return usearch_metric_haversine_k;
    case metric_kind_t::divergence_k: return usearch_metric_divergence_k;
    case metric_kind_t::hamming_k: return usearch_metric_hamming_k;
    case metric_kind_t::tanimoto_k: return usearch_metric_tanimoto_k;
    case metric_kind_t::sorensen_k: return usearch_metric_sorensen_k;
    case metric_kind_t::jaccard_k: return usearch_metric_jaccard_k;
    default: return usearch_metric_unknown_k;
    }
}
inline metric_kind_t from_c(usearch_metric_kind_t kind) noexcept {
    switch (kind) {
    case usearch_metric_ip_k: return metric_kind_t::ip_k;
    case usearch_metric_cos_k: return metric_kind_t::cos_k;
    case usearch_metric_l2sq_k: return metric_kind_t::l2sq_k;
    case usearch_metric_pearson_k: return metric_kind_t::pearson_k;
    case usearch_metric_haversine_k: return metric_kind_t::haversine_k;
    case usearch_metric_divergence_k: return metric_kind_t::divergence_k;
    case usearch_metric_hamming_k: return metric_kind_t::hamming_k;
    case usearch_metric_tanimoto_k: return metric_kind_t::tanimoto_k;
    case usearch_metric_sorensen_k: return metric_kind_t::sorensen_k;
    case usearch_metric_jaccard_k: return metric_kind_t::jaccard_k;
    default: return metric_kind_t::unknown_k;
    }
}
inline usearch_scalar_kind_t to_c(scalar_kind_t kind) noexcept {
    switch (kind) {
    case scalar_kind_t::f32_k: return usearch_scalar_f32_k;
    case scalar_kind_t::f64_k: return usearch_scalar_f64_k;
    case scalar_kind_t::f16_k: return usearch_scalar_f16_k;
    case scalar_kind_t::bf16_k: return usearch_scalar_bf16_k;
    case scalar_kind_t::i8_k: return usearch_scalar_i8_k;
    case scalar_kind_t::b1x8_k: return usearch_scalar_b1_k;
    default: return usearch_scalar_unknown_k;
    }
}
inline scalar_kind_t from_c(usearch_scalar_kind_t kind) noexcept {
    switch (kind) {
    case usearch_scalar_f32_k: return scalar_kind_t::f32_k;
    case usearch_scalar_f64_k: return scalar_kind_t::f64_k;
    case usearch_scalar_f16_k: return scalar_kind_t::f16_k;
    case usearch_scalar_bf16_k: return scalar_kind_t::bf16_k;
    case usearch_scalar_i8_k: return scalar_kind_t::i8_k;
    case usearch_scalar_b1_k: return scalar_kind_t::b1x8_k;
    default: return scalar_kind_t::unknown_k;
    }
}
template <typename scalar_at> struct kind_of;
template <> struct kind_of<f32_t> { static constexpr scalar_kind_t value = scalar_kind_t::f32_k; };
template <> struct kind_of<f64_t> { static constexpr scalar_kind_t value = scalar_kind_t::f64_k; };
template <> struct kind_of<f16_bits_t> { static constexpr scalar_kind_t value = scalar_kind_t::f16_k; };
template <> struct kind_of<bf16_bits_t> { static constexpr scalar_kind_t value = scalar_kind_t::bf16_k; };
template <> struct kind_of<i8_t> { static constexpr scalar_kind_t value = scalar_kind_t::i8_k; };
template <> struct kind_of<b1x8_t> { static constexpr scalar_kind_t value = scalar_kind_t::b1x8_k; };
} // namespace amd_detail

/// `error_t` (index.hpp:315-372): owns nothing — the message is a static string of the library — and is released, not thrown.
class error_t {
    char const* message_{};

  public:
    error_t(char const* message = nullptr) noexcept : message_(message) {}
    error_t(error_t&& other) noexcept : message_(std::exchange(other.message_, nullptr)) {}
    error_t& operator=(error_t&& other) noexcept {
        std::swap(message_, other.message_);
        return *this;
    }
    error_t& operator=(char const* message) noexcept {
        message_ = message;
        return *this;
    }
    explicit operator bool() const noexcept { return message_ != nullptr; }
    char const* what() const noexcept { return message_; }
    char const* release() noexcept { return std::exchange(message_, nullptr); }
};

/// A view of `count` scalars; converts to the pointer the overloads take (cpp/bench.cpp passes these).
template <typename scalar_at> class span_gt {
    scalar_at* data_{};
    std::size_t size_{};

  public:
    span_gt() noexcept = default;
    span_gt(scalar_at* begin, std::size_t count) noexcept : data_(begin), size_(count) {}
    span_gt(scalar_at* begin, scalar_at* end) noexcept : data_(begin), size_((std::size_t)(end - begin)) {}
    scalar_at* data() const noexcept { return data_; }
    std::size_t size() const noexcept { return size_; }
    scalar_at* begin() const noexcept { return data_; }
    scalar_at* end() const noexcept { return data_ + size_; }
    scalar_at& operator[](std::size_t i) const noexcept { return data_[i]; }
    operator scalar_at*() const noexcept { return data_; }
};

struct dummy_predicate_t { // index.hpp:1480-1482
    template <typename member_at> constexpr bool operator()(member_at&&) const noexcept { return true; }
};

/// `memory_mapped_file_t` (index_plugins.hpp:925-1087) as far as callers hand buffers around with it: a span of bytes they own.
class memory_mapped_file_t {
    byte_t* data_{};
    std::size_t length_{};

  public:
    memory_mapped_file_t() noexcept = default;
    memory_mapped_file_t(byte_t* data, std::size_t length) noexcept : data_(data), length_(length) {}
    explicit operator bool() const noexcept { return data_ != nullptr; }
    byte_t* data() const noexcept { return data_; }
    std::size_t size() const noexcept { return length_; }
};

struct index_config_t { // index.hpp:1359-1399
    std::size_t connectivity = 16;
    std::size_t connectivity_base = 32;
};
struct index_dense_config_t : public index_config_t { // index_dense.hpp:102-150
    std::size_t expansion_add = 128;
    std::size_t expansion_search = 64;
    bool exclude_vectors = false;
    bool multi = false;
    bool enable_key_lookups = true;
    index_dense_config_t(index_config_t base) noexcept : index_config_t(base) {}
    index_dense_config_t(std::size_t c = 0, std::size_t ea = 0, std::size_t es = 0) noexcept {
        connectivity = c ? c : 16, connectivity_base = 2 * connectivity;
        expansion_add = ea ? ea : 128, expansion_search = es ? es : 64;
    }
};
struct index_limits_t { // index.hpp:1401-1415
    std::size_t members = 0;
    std::size_t threads_add = std::thread::hardware_concurrency();
    std::size_t threads_search = std::thread::hardware_concurrency();
    index_limits_t(std::size_t n, std::size_t t) noexcept : members(n), threads_add(t), threads_search(t) {}
    index_limits_t(std::size_t n = 0) noexcept : index_limits_t(n, std::thread::hardware_concurrency()) {}
    std::size_t threads() const noexcept { return (std::max)(threads_add, threads_search); }
    std::size_t concurrency() const noexcept { return (std::min)(threads_add, threads_search); }
};
struct index_update_config_t {
    std::size_t expansion = 128;
    std::size_t thread = 0;
};
struct index_search_config_t {
    std::size_t expansion = 64;
    std::size_t thread = 0;
    bool exact = false;
};
struct serialization_config_t { // index_dense.hpp:1000-1012
    bool exclude_vectors = false;
    bool use_64_bit_dimensions = false;
};

/**
 *  `metric_punned_t` (index_plugins.hpp:1659-2014): which distance, over which scalars, in how many dimensions. The arithmetic
 *  itself runs on the device; `operator()` measures one pair there.
 */
class metric_punned_t {
    std::size_t dimensions_ = 0;
    metric_kind_t metric_kind_ = metric_kind_t::unknown_k;
    scalar_kind_t scalar_kind_ = scalar_kind_t::unknown_k;
    std::uintptr_t function_ = 0, state_ = 0; ///< a caller's own metric: kept so that the refusal comes from the library, by name

  public:
    using scalar_t = byte_t;
    using result_t = default_distance_t;
    metric_punned_t() noexcept = default;
    metric_punned_t(std::size_t dimensions, metric_kind_t metric_kind = metric_kind_t::l2sq_k,
                    scalar_kind_t scalar_kind = scalar_kind_t::f32_k) noexcept
        : dimensions_(dimensions), metric_kind_(metric_kind), scalar_kind_(scalar_kind) {}
    static metric_punned_t builtin(std::size_t dimensions, metric_kind_t metric_kind = metric_kind_t::l2sq_k,
                                   scalar_kind_t scalar_kind = scalar_kind_t::f32_k) noexcept {
        return metric_punned_t(dimensions, metric_kind, scalar_kind);
    }
    static metric_punned_t stateless(std::size_t dimensions, std::uintptr_t function, metric_punned_signature_t,
                                     metric_kind_t metric_kind = metric_kind_t::unknown_k,
                                     scalar_kind_t scalar_kind = scalar_kind_t::unknown_k) noexcept {
        metric_punned_t metric(dimensions, metric_kind, scalar_kind);
        metric.function_ = function;
        return metric;
    }
    static metric_punned_t stateful(std::size_t dimensions, std::uintptr_t function, std::uintptr_t state,
                                    metric_kind_t metric_kind = metric_kind_t::unknown_k,
                                    scalar_kind_t scalar_kind = scalar_kind_t::unknown_k) noexcept {
        metric_punned_t metric(dimensions, metric_kind, scalar_kind);
        metric.function_ = function, metric.state_ = state;
        return metric;
    }
    std::size_t dimensions() const noexcept { return dimensions_; }
    metric_kind_t metric_kind() const noexcept { return metric_kind_; }
    scalar_kind_t scalar_kind() const noexcept { return scalar_kind_; }
    std::uintptr_t user_function() const noexcept { return function_; }
    std::uintptr_t user_state() const noexcept { return state_; }
    explicit operator bool() const noexcept { return !missing(); }
    bool missing() const noexcept {
        return !function_ && (amd_detail::to_c(metric_kind_) == usearch_metric_unknown_k ||
                              amd_detail::to_c(scalar_kind_) == usearch_scalar_unknown_k);
    }
    char const* isa_name() const noexcept { return "gfx950"; }
    std::size_t bytes_per_vector() const noexcept {
        switch (scalar_kind_) {
        case scalar_kind_t::b1x8_k: return (dimensions_ + 7) / 8;
        case scalar_kind_t::i8_k: return dimensions_;
        case scalar_kind_t::f16_k:
        case scalar_kind_t::bf16_k: return dimensions_ * 2;
        case scalar_kind_t::f32_k: return dimensions_ * 4;
        case scalar_kind_t::f64_k: return dimensions_ * 8;
        default: return 0;
        }
    }
    result_t operator()(byte_t const* a, byte_t const* b) const noexcept {
        usearch_error_t error = nullptr;
        return amd_detail::api().distance(a, b, amd_detail::to_c(scalar_kind_), dimensions_, amd_detail::to_c(metric_kind_), &error);
    }
};

struct serialization_result_t { // index.hpp:1427-1440
    error_t error;
    explicit operator bool() const noexcept { return !error; }
    serialization_result_t failed(error_t message) noexcept {
        error = std::move(message);
        return std::move(*this);
    }
};

/// What `index_dense_metadata_from_path / _from_buffer` report of a file (index_dense.hpp:42-79, 182-262).
struct index_dense_head_t {
    metric_kind_t kind_metric = metric_kind_t::unknown_k;
    scalar_kind_t kind_scalar = scalar_kind_t::unknown_k;
    std::uint64_t dimensions = 0;
    bool multi = false;
};
struct index_dense_metadata_result_t {
    index_dense_head_t head;
    error_t error;
    explicit operator bool() const noexcept { return !error; }
    index_dense_metadata_result_t failed(error_t message) noexcept {
        error = std::move(message);
        return std::move(*this);
    }
};
inline index_dense_metadata_result_t index_dense_metadata_from_path(char const* path) noexcept {
    index_dense_metadata_result_t result;
    usearch_init_options_t options{};
    usearch_error_t error = nullptr;
    amd_detail::api().metadata(path, &options, &error);
    if (error)
        return result.failed(error);
    result.head = {amd_detail::from_c(options.metric_kind), amd_detail::from_c(options.quantization), options.dimensions, options.multi};
    return result;
}
inline index_dense_metadata_result_t index_dense_metadata_from_buffer(memory_mapped_file_t const& file, std::size_t offset = 0) noexcept {
    index_dense_metadata_result_t result;
    usearch_init_options_t options{};
    usearch_error_t error = nullptr;
    amd_detail::api().metadata_buffer(file.data() + offset, file.size() - offset, &options, &error);
    if (error)
        return result.failed(error);
    result.head = {amd_detail::from_c(options.metric_kind), amd_detail::from_c(options.quantization), options.dimensions, options.multi};
    return result;
}

/// Executors exist for source compatibility: the device is the executor (index_plugins.hpp:605-790).
class executor_stl_t {
    std::size_t threads_{};

  public:
    executor_stl_t(std::size_t threads = 0) noexcept : threads_(threads ? threads : std::thread::hardware_concurrency()) {}
    std::size_t size() const noexcept { return threads_; }
};
using executor_default_t = executor_stl_t;
class dummy_progress_t {
  public:
    bool operator()(std::size_t, std::size_t) const noexcept { return true; }
};

/// `exact_search_t` (index_plugins.hpp:2071-2164): many queries against a raw dataset, keys are dataset offsets.
class exact_search_results_t {
  public:
    struct result_t {
        std::size_t offset;
        default_distance_t distance;
    };
    class query_results_t {
        exact_search_results_t const* owner_;
        std::size_t query_;

      public:
        query_results_t(exact_search_results_t const* owner, std::size_t query) noexcept : owner_(owner), query_(query) {}
        result_t operator[](std::size_t i) const noexcept {
            return {(std::size_t)owner_->keys_[query_ * owner_->wanted_ + i], owner_->distances_[query_ * owner_->wanted_ + i]};
        }
        std::size_t size() const noexcept { return owner_->wanted_; }
    };
    exact_search_results_t() noexcept = default;
    explicit operator bool() const noexcept { return ok_; }
    std::size_t size() const noexcept { return queries_; }
    query_results_t at(std::size_t query) const noexcept { return {this, query}; }

  private:
    friend class exact_search_t;
    std::vector<usearch_key_t> keys_;
    std::vector<usearch_distance_t> distances_;
    std::size_t queries_ = 0, wanted_ = 0;
    bool ok_ = false;
};
class exact_search_t {
  public:
    template <typename executor_at = executor_default_t, typename progress_at = dummy_progress_t>
    exact_search_results_t operator()(byte_t const* dataset, std::size_t dataset_count, std::size_t dataset_stride,
                                      byte_t const* queries, std::size_t queries_count, std::size_t queries_stride,
                                      std::size_t wanted, metric_punned_t const& metric, executor_at&& executor = executor_at{},
                                      progress_at&& = progress_at{}) {
        exact_search_results_t result;
        result.queries_ = queries_count, result.wanted_ = wanted;
        result.keys_.resize(queries_count * wanted), result.distances_.resize(queries_count * wanted);
        usearch_error_t error = nullptr;
        amd_detail::api().exact_search(dataset, dataset_count, dataset_stride, queries, queries_count, queries_stride,
                                       amd_detail::to_c(metric.scalar_kind()), metric.dimensions(),
                                       amd_detail::to_c(metric.metric_kind()), wanted, executor.size(), result.keys_.data(),
                                       wanted * sizeof(usearch_key_t), result.distances_.data(), wanted * sizeof(usearch_distance_t),
                                       &error);
        result.ok_ = error == nullptr;
        return result;
    }
};

/**
 *  `index_dense_gt<key, slot>` (index_dense.hpp:376-2227). One instantiation exists on the device: 64-bit keys, 32-bit slots —
 *  `index_dense_t` (index_dense.hpp:2229); the template parameters are there so that code naming them compiles.
 */
template <typename key_at = default_key_t, typename compressed_slot_at = default_slot_t> class index_dense_gt {
    static_assert(sizeof(key_at) == 8, "the device index keys its members with 64 bits (index_dense_t)");
    usearch_index_t handle_ = nullptr;
    metric_punned_t metric_;
    index_dense_config_t config_;
    index_limits_t limits_{0, 0};

  public:
    using vector_key_t = key_at;
    using key_t = vector_key_t;
    using compressed_slot_t = compressed_slot_at;
    using distance_t = default_distance_t;
    using metric_t = metric_punned_t;

    struct member_cref_t { // index.hpp:2038-2046
        vector_key_t key;
        std::size_t slot;
    };
    struct match_t { // index.hpp:2576-2593
        member_cref_t member;
        distance_t distance;
    };

    /// `search_result_t` (index.hpp:2595-2742). Results are copied out of the device, so a result outlives the next search.
    class search_result_t {
        friend class index_dense_gt;
        std::vector<vector_key_t> keys_;
        std::vector<distance_t> distances_;

      public:
        std::size_t count{};
        std::size_t visited_members{};
        std::size_t computed_distances{};
        error_t error{};

        search_result_t() noexcept {}
        explicit search_result_t(index_dense_gt const&) noexcept {}
        search_result_t(search_result_t&&) = default;
        search_result_t& operator=(search_result_t&&) = default;
        explicit operator bool() const noexcept { return !error; }
        search_result_t failed(error_t message) noexcept {
            error = std::move(message);
            return std::move(*this);
        }
        operator std::size_t() const noexcept { return count; }
        std::size_t size() const noexcept { return count; }
        bool empty() const noexcept { return !count; }
        match_t at(std::size_t i) const noexcept { return {member_cref_t{keys_[i], 0}, distances_[i]}; }
        match_t operator[](std::size_t i) const noexcept { return at(i); }
        match_t front() const noexcept { return at(0); }
        match_t back() const noexcept { return at(count - 1); }
        bool contains(vector_key_t key) const noexcept {
            for (std::size_t i = 0; i != count; ++i)
                if (keys_[i] == key)
                    return true;
            return false;
        }
        /// Folds these results into a caller's sorted buffer that may already hold some (index.hpp:2650-2670): a new result
        /// lands BEFORE equal distances, the worst falls off a full buffer.
        std::size_t merge_into(vector_key_t* keys, distance_t* distances, std::size_t old_count, std::size_t max_count) const noexcept {
            std::size_t merged = old_count;
            for (std::size_t i = 0; i != count; ++i) {
                const std::size_t offset = (std::size_t)(std::lower_bound(distances, distances + merged, distances_[i]) - distances);
                if (offset == max_count)
                    continue;
                const std::size_t worse = merged - offset - (max_count == merged);
                std::memmove(keys + offset + 1, keys + offset, worse * sizeof(vector_key_t));
                std::memmove(distances + offset + 1, distances + offset, worse * sizeof(distance_t));
                keys[offset] = keys_[i], distances[offset] = distances_[i];
                merged += merged != max_count;
            }
            return merged;
        }
        std::size_t dump_to(vector_key_t* keys, distance_t* distances) const noexcept {
            for (std::size_t i = 0; i != count; ++i)
                keys[i] = keys_[i], distances[i] = distances_[i];
            return count;
        }
        std::size_t dump_to(vector_key_t* keys) const noexcept {
            for (std::size_t i = 0; i != count; ++i)
                keys[i] = keys_[i];
            return count;
        }
        /// With a capacity the tail is padded: key 0 and a signalling NaN (index.hpp:2707-2722).
        std::size_t dump_to(vector_key_t* keys, distance_t* distances, std::size_t capacity) const noexcept {
            const std::size_t initialized = (std::min)(count, capacity);
            std::size_t i = 0;
            for (; i != initialized; ++i)
                keys[i] = keys_[i], distances[i] = distances_[i];
            for (; i != capacity; ++i)
                keys[i] = vector_key_t{}, distances[i] = std::numeric_limits<distance_t>::signaling_NaN();
            return initialized;
        }
        std::size_t dump_to(vector_key_t* keys, std::size_t capacity) const noexcept {
            const std::size_t initialized = (std::min)(count, capacity);
            std::size_t i = 0;
            for (; i != initialized; ++i)
                keys[i] = keys_[i];
            for (; i != capacity; ++i)
                keys[i] = vector_key_t{};
            return initialized;
        }
    };
    struct add_result_t { // index.hpp:2744-2757
        error_t error{};
        std::size_t new_size{};
        std::size_t visited_members{};
        std::size_t computed_distances{};
        std::size_t slot{};
        explicit operator bool() const noexcept { return !error; }
        add_result_t failed(error_t message) noexcept {
            error = std::move(message);
            return std::move(*this);
        }
    };
    struct labeling_result_t { // index_dense.hpp:1370-1380
        error_t error{};
        std::size_t completed{};
        explicit operator bool() const noexcept { return !error; }
        labeling_result_t failed(error_t message) noexcept {
            error = std::move(message);
            return std::move(*this);
        }
    };
    struct cluster_result_t { // index.hpp:2744 ff.
        error_t error{};
        std::size_t visited_members{};
        std::size_t computed_distances{};
        match_t cluster{};
        explicit operator bool() const noexcept { return !error; }
        cluster_result_t failed(error_t message) noexcept {
            error = std::move(message);
            return std::move(*this);
        }
    };
    struct state_result_t { // index_dense.hpp:609-623
        index_dense_gt index;
        error_t error;
        explicit operator bool() const noexcept { return !error; }
        state_result_t failed(error_t message) noexcept {
            error = std::move(message);
            return std::move(*this);
        }
        operator index_dense_gt&&() && noexcept { return std::move(index); }
    };

    index_dense_gt() noexcept { open_empty_(); }
    index_dense_gt(index_dense_gt&& other) noexcept
        : handle_(std::exchange(other.handle_, nullptr)), metric_(other.metric_), config_(other.config_), limits_(other.limits_) {}
    index_dense_gt& operator=(index_dense_gt&& other) noexcept {
        std::swap(handle_, other.handle_);
        std::swap(metric_, other.metric_);
        std::swap(config_, other.config_);
        std::swap(limits_, other.limits_);
        return *this;
    }
    index_dense_gt(index_dense_gt const&) = delete;
    index_dense_gt& operator=(index_dense_gt const&) = delete;
    ~index_dense_gt() noexcept {
        usearch_error_t error = nullptr;
        if (handle_)
            amd_detail::api().free(handle_, &error);
    }

    /// `make(metric, config, free_key)` (index_dense.hpp:644-679).
    static state_result_t make(metric_t metric, index_dense_config_t config = {}, vector_key_t = (vector_key_t)~0ull) noexcept {
        state_result_t result;
        usearch_init_options_t options{};
        options.metric_kind = amd_detail::to_c(metric.metric_kind());
        options.metric = reinterpret_cast<usearch_metric_t>(metric.user_function());
        options.quantization = amd_detail::to_c(metric.scalar_kind());
        options.dimensions = metric.dimensions();
        options.connectivity = config.connectivity, options.expansion_add = config.expansion_add;
        options.expansion_search = config.expansion_search, options.multi = config.multi;
        usearch_error_t error = nullptr;
        usearch_index_t handle = amd_detail::api().init(&options, &error);
        if (error || !handle)
            return result.failed(error ? error : "Out of memory!");
        result.index.close_();
        result.index.handle_ = handle;
        result.index.metric_ = metric, result.index.config_ = config;
        return result;
    }
    /// `make(path, view)` (index_dense.hpp:681-700).
    static state_result_t make(char const* path, bool view = false) noexcept {
        state_result_t result;
        serialization_result_t loaded = view ? result.index.view(path) : result.index.load(path);
        if (!loaded)
            return result.failed(loaded.error.release());
        return result;
    }

    explicit operator bool() const noexcept { return handle_ != nullptr && metric_.dimensions() != 0; }
    std::size_t size() const noexcept { return get_(amd_detail::api().size); }
    std::size_t capacity() const noexcept { return get_(amd_detail::api().capacity); }
    std::size_t dimensions() const noexcept { return get_(amd_detail::api().dimensions); }
    std::size_t connectivity() const noexcept { return get_(amd_detail::api().connectivity); }
    std::size_t expansion_add() const noexcept { return get_(amd_detail::api().expansion_add); }
    std::size_t expansion_search() const noexcept { return get_(amd_detail::api().expansion_search); }
    std::size_t memory_usage() const noexcept { return get_(amd_detail::api().memory_usage); }
    std::size_t serialized_length(serialization_config_t = {}) const noexcept { return get_(amd_detail::api().serialized_length); }
    std::size_t scalar_words() const noexcept { return metric_.dimensions(); }
    std::size_t bytes_per_vector() const noexcept { return metric_.bytes_per_vector(); }
    std::size_t max_level() const noexcept { return 0; }
    bool multi() const noexcept { return config_.multi; }
    metric_t const& metric() const noexcept { return metric_; }
    scalar_kind_t scalar_kind() const noexcept { return metric_.scalar_kind(); }
    index_dense_config_t const& config() const noexcept { return config_; }
    index_limits_t limits() const noexcept {
        index_limits_t result(capacity(), 0);
        result.threads_add = limits_.threads_add, result.threads_search = limits_.threads_search;
        return result;
    }
    static constexpr std::size_t any_thread() noexcept { return (std::numeric_limits<std::size_t>::max)(); }

    void change_expansion_add(std::size_t n) noexcept { set_(amd_detail::api().change_expansion_add, n); }
    void change_expansion_search(std::size_t n) noexcept { set_(amd_detail::api().change_expansion_search, n); }
    void change_metric(metric_t metric) noexcept {
        usearch_error_t error = nullptr;
        if (metric.user_function())
            amd_detail::api().change_metric(handle_, reinterpret_cast<usearch_metric_t>(metric.user_function()),
                                            reinterpret_cast<void*>(metric.user_state()), amd_detail::to_c(metric.metric_kind()), &error);
        else
            amd_detail::api().change_metric_kind(handle_, amd_detail::to_c(metric.metric_kind()), &error);
        if (!error)
            metric_ = metric;
    }

    /// `try_reserve(limits)` (index_dense.hpp:925-948): room for `limits.members`, and how many batches may be in flight.
    bool try_reserve(index_limits_t limits) noexcept {
        usearch_error_t error = nullptr;
        amd_detail::api().reserve(handle_, limits.members, &error);
        if (error)
            return false;
        amd_detail::api().change_threads_add(handle_, limits.threads_add, &error);
        amd_detail::api().change_threads_search(handle_, limits.threads_search, &error);
        limits_.threads_add = limits.threads_add, limits_.threads_search = limits.threads_search;
        return !error;
    }
    void reserve(index_limits_t limits) noexcept { (void)try_reserve(limits); }

    // ---- add (index_dense.hpp:760-765): one overload per scalar type
    add_result_t add(vector_key_t key, b1x8_t const* vector, std::size_t thread = any_thread(), bool copy = true) { return add_(key, vector, scalar_kind_t::b1x8_k, thread, copy); }
    add_result_t add(vector_key_t key, i8_t const* vector, std::size_t thread = any_thread(), bool copy = true) { return add_(key, vector, scalar_kind_t::i8_k, thread, copy); }
    add_result_t add(vector_key_t key, f16_t const* vector, std::size_t thread = any_thread(), bool copy = true) { return add_(key, vector, scalar_kind_t::f16_k, thread, copy); }
    add_result_t add(vector_key_t key, bf16_t const* vector, std::size_t thread = any_thread(), bool copy = true) { return add_(key, vector, scalar_kind_t::bf16_k, thread, copy); }
    add_result_t add(vector_key_t key, f32_t const* vector, std::size_t thread = any_thread(), bool copy = true) { return add_(key, vector, scalar_kind_t::f32_k, thread, copy); }
    add_result_t add(vector_key_t key, f64_t const* vector, std::size_t thread = any_thread(), bool copy = true) { return add_(key, vector, scalar_kind_t::f64_k, thread, copy); }

    // ---- search (index_dense.hpp:767-772)
    search_result_t search(b1x8_t const* q, std::size_t wanted, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::b1x8_k, wanted, dummy_predicate_t{}, thread, exact); }
    search_result_t search(i8_t const* q, std::size_t wanted, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::i8_k, wanted, dummy_predicate_t{}, thread, exact); }
    search_result_t search(f16_t const* q, std::size_t wanted, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::f16_k, wanted, dummy_predicate_t{}, thread, exact); }
    search_result_t search(bf16_t const* q, std::size_t wanted, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::bf16_k, wanted, dummy_predicate_t{}, thread, exact); }
    search_result_t search(f32_t const* q, std::size_t wanted, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::f32_k, wanted, dummy_predicate_t{}, thread, exact); }
    search_result_t search(f64_t const* q, std::size_t wanted, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::f64_k, wanted, dummy_predicate_t{}, thread, exact); }

    // ---- filtered_search (index_dense.hpp:774-779): `predicate(key) -> bool`
    template <typename predicate_at> search_result_t filtered_search(b1x8_t const* q, std::size_t wanted, predicate_at&& predicate, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::b1x8_k, wanted, std::forward<predicate_at>(predicate), thread, exact); }
    template <typename predicate_at> search_result_t filtered_search(i8_t const* q, std::size_t wanted, predicate_at&& predicate, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::i8_k, wanted, std::forward<predicate_at>(predicate), thread, exact); }
    template <typename predicate_at> search_result_t filtered_search(f16_t const* q, std::size_t wanted, predicate_at&& predicate, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::f16_k, wanted, std::forward<predicate_at>(predicate), thread, exact); }
    template <typename predicate_at> search_result_t filtered_search(bf16_t const* q, std::size_t wanted, predicate_at&& predicate, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::bf16_k, wanted, std::forward<predicate_at>(predicate), thread, exact); }
    template <typename predicate_at> search_result_t filtered_search(f32_t const* q, std::size_t wanted, predicate_at&& predicate, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::f32_k, wanted, std::forward<predicate_at>(predicate), thread, exact); }
    template <typename predicate_at> search_result_t filtered_search(f64_t const* q, std::size_t wanted, predicate_at&& predicate, std::size_t thread = any_thread(), bool exact = false) const { return search_(q, scalar_kind_t::f64_k, wanted, std::forward<predicate_at>(predicate), thread, exact); }

    /// The batch the reference leaves to its callers' loops (cpp/bench.cpp:352-377): `queries_count` rows `stride_bytes` apart, one
    /// device launch; row i of `keys` / `distances` holds exactly `wanted` cells (padded like `dump_to` with a capacity).
    template <typename scalar_at>
    search_result_t search_many(scalar_at const* queries, std::size_t queries_count, std::size_t stride_bytes, std::size_t wanted,
                                vector_key_t* keys, distance_t* distances, std::size_t* counts = nullptr) const {
        search_result_t result;
        usearch_error_t error = nullptr;
        amd_detail::api().search_many(handle_, queries, amd_detail::to_c(amd_detail::kind_of<scalar_at>::value), queries_count,
                                      stride_bytes, wanted, keys, wanted * sizeof(vector_key_t), distances,
                                      wanted * sizeof(distance_t), counts, &result.visited_members, &result.computed_distances, &error);
        if (error)
            return result.failed(error);
        result.count = queries_count;
        return result;
    }

    // ---- get (index_dense.hpp:781-786)
    std::size_t get(vector_key_t key, b1x8_t* vector, std::size_t count = 1) const { return get_vector_(key, vector, count, scalar_kind_t::b1x8_k); }
    std::size_t get(vector_key_t key, i8_t* vector, std::size_t count = 1) const { return get_vector_(key, vector, count, scalar_kind_t::i8_k); }
    std::size_t get(vector_key_t key, f16_t* vector, std::size_t count = 1) const { return get_vector_(key, vector, count, scalar_kind_t::f16_k); }
    std::size_t get(vector_key_t key, bf16_t* vector, std::size_t count = 1) const { return get_vector_(key, vector, count, scalar_kind_t::bf16_k); }
    std::size_t get(vector_key_t key, f32_t* vector, std::size_t count = 1) const { return get_vector_(key, vector, count, scalar_kind_t::f32_k); }
    std::size_t get(vector_key_t key, f64_t* vector, std::size_t count = 1) const { return get_vector_(key, vector, count, scalar_kind_t::f64_k); }

    // ---- cluster (index_dense.hpp:788-793): the member the greedy descent reaches on `level`
    template <typename scalar_at> cluster_result_t cluster(scalar_at const* query, std::size_t level, std::size_t = any_thread()) const {
        cluster_result_t result;
        usearch_error_t error = nullptr;
        usearch_key_t key = 0;
        distance_t distance = 0;
        amd_detail::api().cluster_many(handle_, query, amd_detail::to_c(amd_detail::kind_of<scalar_at>::value), 1, 0, level, &key,
                                       &distance, &error);
        if (error)
            return result.failed(error);
        result.cluster = {member_cref_t{key, 0}, distance};
        return result;
    }

    bool contains(vector_key_t key) const noexcept {
        usearch_error_t error = nullptr;
        return amd_detail::api().contains(handle_, key, &error);
    }
    std::size_t count(vector_key_t key) const noexcept {
        usearch_error_t error = nullptr;
        return amd_detail::api().count(handle_, key, &error);
    }
    labeling_result_t remove(vector_key_t key) noexcept {
        labeling_result_t result;
        usearch_error_t error = nullptr;
        result.completed = amd_detail::api().remove(handle_, key, &error);
        if (error)
            result.error = error;
        return result;
    }
    labeling_result_t rename(vector_key_t from, vector_key_t to) noexcept {
        labeling_result_t result;
        usearch_error_t error = nullptr;
        result.completed = amd_detail::api().rename(handle_, from, to, &error);
        if (error)
            result.error = error;
        return result;
    }
    void clear() noexcept {
        usearch_error_t error = nullptr;
        amd_detail::api().clear(handle_, &error);
    }
    void reset() noexcept { clear(); }

    // ---- serialization (index_dense.hpp:950-1313)
    template <typename progress_at = dummy_progress_t>
    serialization_result_t save(char const* path, serialization_config_t = {}, progress_at&& = progress_at{}) const noexcept {
        return io_([&](usearch_error_t* e) { amd_detail::api().save(handle_, path, e); });
    }
    template <typename progress_at = dummy_progress_t>
    serialization_result_t load(char const* path, serialization_config_t = {}, progress_at&& = progress_at{}) noexcept {
        serialization_result_t result = io_([&](usearch_error_t* e) { amd_detail::api().load(handle_, path, e); });
        if (result)
            adopt_(index_dense_metadata_from_path(path));
        return result;
    }
    template <typename progress_at = dummy_progress_t>
    serialization_result_t view(char const* path, std::size_t = 0, serialization_config_t = {}, progress_at&& = progress_at{}) noexcept {
        serialization_result_t result = io_([&](usearch_error_t* e) { amd_detail::api().view(handle_, path, e); });
        if (result)
            adopt_(index_dense_metadata_from_path(path));
        return result;
    }
    template <typename progress_at = dummy_progress_t>
    serialization_result_t save(memory_mapped_file_t file, std::size_t offset = 0, serialization_config_t = {},
                                progress_at&& = progress_at{}) const noexcept {
        return io_([&](usearch_error_t* e) { amd_detail::api().save_buffer(handle_, file.data() + offset, file.size() - offset, e); });
    }
    template <typename progress_at = dummy_progress_t>
    serialization_result_t load(memory_mapped_file_t file, std::size_t offset = 0, serialization_config_t = {},
                                progress_at&& = progress_at{}) noexcept {
        serialization_result_t result =
            io_([&](usearch_error_t* e) { amd_detail::api().load_buffer(handle_, file.data() + offset, file.size() - offset, e); });
        if (result)
            adopt_(index_dense_metadata_from_buffer(file, offset));
        return result;
    }
    template <typename progress_at = dummy_progress_t>
    serialization_result_t view(memory_mapped_file_t file, std::size_t offset = 0, serialization_config_t = {},
                                progress_at&& = progress_at{}) noexcept {
        serialization_result_t result =
            io_([&](usearch_error_t* e) { amd_detail::api().view_buffer(handle_, file.data() + offset, file.size() - offset, e); });
        if (result)
            adopt_(index_dense_metadata_from_buffer(file, offset));
        return result;
    }

    /// Not in the reference: brings the device index up to date now (members added since the last search are linked) instead
    /// of at the next search.
    serialization_result_t sync() noexcept {
        return io_([&](usearch_error_t* e) { amd_detail::api().gpu_sync(handle_, e); });
    }

  private:
    void close_() noexcept {
        usearch_error_t error = nullptr;
        if (handle_)
            amd_detail::api().free(handle_, &error);
        handle_ = nullptr;
    }
    void open_empty_() noexcept { // the shell `usearch_init(NULL)` makes: filled by `load` / `view` (c/lib.cpp:142-147)
        usearch_error_t error = nullptr;
        handle_ = amd_detail::api().init(nullptr, &error);
    }
    template <typename getter_at> std::size_t get_(getter_at getter) const noexcept {
        usearch_error_t error = nullptr;
        return handle_ ? getter(handle_, &error) : 0;
    }
    template <typename setter_at> void set_(setter_at setter, std::size_t value) noexcept {
        usearch_error_t error = nullptr;
        if (handle_)
            setter(handle_, value, &error);
    }
    template <typename body_at> serialization_result_t io_(body_at&& body) const noexcept {
        serialization_result_t result;
        usearch_error_t error = nullptr;
        body(&error);
        if (error)
            result.error = error;
        return result;
    }
    void adopt_(index_dense_metadata_result_t meta) noexcept {
        if (meta)
            metric_ = metric_punned_t(meta.head.dimensions, meta.head.kind_metric, meta.head.kind_scalar), config_.multi = meta.head.multi;
        config_.connectivity = connectivity(), config_.connectivity_base = 2 * config_.connectivity;
    }
    add_result_t add_(vector_key_t key, void const* vector, scalar_kind_t kind, std::size_t, bool) {
        add_result_t result;
        usearch_error_t error = nullptr;
        amd_detail::api().add(handle_, key, vector, amd_detail::to_c(kind), &error);
        if (error)
            return result.failed(error);
        result.new_size = size();
        result.slot = result.new_size ? result.new_size - 1 : 0;
        return result;
    }
    std::size_t get_vector_(vector_key_t key, void* vector, std::size_t count, scalar_kind_t kind) const {
        usearch_error_t error = nullptr;
        return amd_detail::api().get(handle_, key, count, vector, amd_detail::to_c(kind), &error);
    }
    template <typename predicate_at>
    search_result_t search_(void const* query, scalar_kind_t kind, std::size_t wanted, predicate_at&& predicate, std::size_t,
                            bool exact) const {
        using predicate_t = typename std::decay<predicate_at>::type;
        search_result_t result;
        result.keys_.resize(wanted), result.distances_.resize(wanted);
        usearch_error_t error = nullptr;
        std::size_t found = 0;
        if (exact && std::is_same<predicate_t, dummy_predicate_t>::value) { // brute force over every member (index.hpp:3046-3049)
            amd_detail::api().search_exact_many(handle_, query, amd_detail::to_c(kind), 1, 0, wanted, result.keys_.data(),
                                                wanted * sizeof(vector_key_t), result.distances_.data(), wanted * sizeof(distance_t),
                                                &found, &error);
        } else if (exact) { // … over the members the predicate lets through (`search_exact_` skips the others, index.hpp:4260-4263)
            using callable_t = typename std::remove_reference<predicate_at>::type;
            auto trampoline = [](usearch_key_t key, void* opaque) -> int {
                return (*static_cast<callable_t*>(opaque))((vector_key_t)key) ? 1 : 0;
            };
            usearch_filter_t made = amd_detail::api().filter_from_callback(
                handle_, +trampoline, const_cast<void*>(static_cast<void const*>(std::addressof(predicate))), &error);
            if (!error) {
                amd_detail::api().filtered_search_exact_many(handle_, made, query, amd_detail::to_c(kind), 1, 0, wanted,
                                                             result.keys_.data(), wanted * sizeof(vector_key_t),
                                                             result.distances_.data(), wanted * sizeof(distance_t), &found, &error);
                usearch_error_t ignored = nullptr;
                amd_detail::api().filter_free(made, &ignored);
            }
        } else if (std::is_same<predicate_t, dummy_predicate_t>::value) {
            amd_detail::api().search_many(handle_, query, amd_detail::to_c(kind), 1, 0, wanted, result.keys_.data(),
                                          wanted * sizeof(vector_key_t), result.distances_.data(), wanted * sizeof(distance_t), &found,
                                          &result.visited_members, &result.computed_distances, &error);
        } else {
            using callable_t = typename std::remove_reference<predicate_at>::type; // keeps a const predicate const
            auto trampoline = [](usearch_key_t key, void* opaque) -> int {
                return (*static_cast<callable_t*>(opaque))((vector_key_t)key) ? 1 : 0;
            };
            found = amd_detail::api().filtered_search(handle_, query, amd_detail::to_c(kind), wanted, +trampoline,
                                                      const_cast<void*>(static_cast<void const*>(std::addressof(predicate))),
                                                      result.keys_.data(), result.distances_.data(), &error);
        }
        if (error)
            return result.failed(error);
        result.count = found;
        return result;
    }
};

using index_dense_t = index_dense_gt<>;

} // namespace usearch
} // namespace unum
