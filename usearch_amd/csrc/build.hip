/**
 *  usearch_amd/csrc/build.hip — host driver of batched HNSW construction (see build.hpp / build_kernels.hpp) and the
 *  writer of the reference's serialized form.
 */
#include "build.hpp"

#include <chrono>
#include <cmath>
#include <cstring>
#include <random>

#include "build_kernels.hpp"
#include "host_util.hpp"

namespace usearch_amd {

namespace {

double seconds_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

hipError_t launch_build(metric_kind_t metric, scalar_kind_t scalar, const build_params_t& p, const snapshot_view_t& view,
                        const build_args_t& args) {
    if (metric == metric_jaccard_k) // the same kernel, index_plugins.hpp:2003-2004
        metric = metric_tanimoto_k;
#define UA_PAIR(m, sc, name)                                                                                           \
    if (metric == m && scalar == sc)                                                                                   \
        return launch_build_##name(p, view, args);
    USEARCH_AMD_FOR_EACH_PAIR(UA_PAIR)
#undef UA_PAIR
    return hipErrorInvalidValue;
}

/// Frees a set of device allocations when the build function returns, whatever the path.
struct device_block_t {
    std::vector<void*> pointers;
    ~device_block_t() {
        for (void* p : pointers)
            if (p)
                (void)hipFree(p);
    }
    template <typename pointer_at> hipError_t allocate(pointer_at** out, std::size_t bytes) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<std::size_t>(bytes, 16));
        if (e == hipSuccess)
            pointers.push_back(p);
        *out = static_cast<pointer_at*>(p);
        return e;
    }
};

} // namespace

const char* snapshot_t::allocate_for_build(metric_kind_t metric, scalar_kind_t scalar, std::size_t dimensions,
                                           std::uint64_t capacity, std::uint32_t m, std::uint32_t m0,
                                           const std::int16_t* levels, const void* vectors, std::size_t stride,
                                           bool vectors_on_device, const std::uint64_t* keys, int device) {
    if (!kernel_available(metric, scalar))
        return "No MI355X kernel for this metric / scalar kind combination";
    release();
    device_ = device;
    UA_HIP(hipSetDevice(device));
    metric_ = metric;
    scalar_ = scalar;
    count_present_ = capacity;
    const std::uint64_t n = capacity;
    const std::uint32_t bpv = (std::uint32_t)bytes_per_vector(scalar, dimensions);
    std::uint32_t row_stride = 0, row_chunks = 0;
    row_geometry(bpv, lanes_, row_stride, row_chunks);

    std::vector<std::uint32_t> upper_ref(n);
    std::uint64_t lists = 0;
    for (std::uint64_t i = 0; i < n; ++i) {
        upper_ref[i] = levels[i] ? (std::uint32_t)lists : none_slot_k;
        lists += (std::uint64_t)levels[i];
    }
    if (lists >= none_slot_k)
        return "Too many upper-level lists for 32-bit references";
    upper_lists_ = lists;

    device_bytes_ = 0;
    auto allocate = [&](void** p, std::size_t bytes) -> hipError_t {
        device_bytes_ += std::max<std::size_t>(bytes, 16);
        return hipMalloc(p, std::max<std::size_t>(bytes, 16));
    };
    const std::size_t nbr0_bytes = (std::size_t)n * m0 * 4, upper_bytes = (std::size_t)std::max<std::uint64_t>(lists, 1) * m * 4;
    UA_HIP(allocate(&d_vectors_, (std::size_t)n * row_stride));
    UA_HIP(allocate(&d_nbr0_, nbr0_bytes));
    UA_HIP(allocate(&d_upper_ref_, n * 4));
    UA_HIP(allocate(&d_upper_, upper_bytes));
    UA_HIP(allocate(&d_keys_, n * 8));
    UA_HIP(hipMemset(d_nbr0_, 0xFF, std::max<std::size_t>(nbr0_bytes, 16))); // every cell = none_slot_k
    UA_HIP(hipMemset(d_upper_, 0xFF, upper_bytes));
    UA_HIP(hipMemcpy(d_upper_ref_, upper_ref.data(), n * 4, hipMemcpyHostToDevice));
    if (vectors_on_device) {
        if (row_stride == bpv && stride == bpv) {
            UA_HIP(hipMemcpy(d_vectors_, vectors, (std::size_t)n * bpv, hipMemcpyDeviceToDevice));
        } else {
            UA_HIP(hipMemset(d_vectors_, 0, (std::size_t)n * row_stride));
            UA_HIP(hipMemcpy2D(d_vectors_, row_stride, vectors, stride, bpv, n, hipMemcpyDeviceToDevice));
        }
    } else if (const char* e = upload_rows(static_cast<std::uint8_t*>(d_vectors_), row_stride,
                                           static_cast<const std::uint8_t*>(vectors), stride, bpv, n)) {
        return e;
    }
    if (keys) {
        UA_HIP(hipMemcpy(d_keys_, keys, n * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<std::uint64_t> identity(n);
        for (std::uint64_t i = 0; i < n; ++i)
            identity[i] = i;
        UA_HIP(hipMemcpy(d_keys_, identity.data(), n * 8, hipMemcpyHostToDevice));
    }

    view_ = snapshot_view_t{};
    view_.vectors = static_cast<const std::uint8_t*>(d_vectors_);
    view_.nbr0 = static_cast<const std::uint32_t*>(d_nbr0_);
    view_.upper_ref = static_cast<const std::uint32_t*>(d_upper_ref_);
    view_.upper = static_cast<const std::uint32_t*>(d_upper_);
    view_.keys = static_cast<const std::uint64_t*>(d_keys_);
    view_.size = 0;
    view_.row_stride = row_stride;
    view_.chunks = row_chunks;
    view_.bytes_per_vector = bpv;
    view_.dimensions = (std::uint32_t)dimensions;
    view_.m = m;
    view_.m0 = m0;

    hipDeviceProp_t properties;
    UA_HIP(hipGetDeviceProperties(&properties, device));
    compute_units_ = properties.multiProcessorCount > 0 ? properties.multiProcessorCount : 256;
    UA_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    return nullptr;
}

const char* builder_t::build(metric_kind_t metric, scalar_kind_t scalar, std::size_t dimensions, const void* vectors,
                             std::uint64_t count, std::size_t stride, bool vectors_on_device,
                             const std::uint64_t* keys, const build_config_t& config, int device) {
    const double t_begin = seconds_now();
    config_ = config;
    stats_ = build_stats_t{};
    if (!kernel_available(metric, scalar))
        return "No MI355X kernel for this metric / scalar kind combination";
    if (!count || !vectors)
        return "Nothing to build";
    if (count >= none_slot_k)
        return "Index is too large for 32-bit slots";
    const std::uint32_t m = config_.connectivity;
    if (m < 2)
        return "Connectivity must be at least 2";
    if (!config_.connectivity_base)
        config_.connectivity_base = 2 * m;
    const std::uint32_t m0 = config_.connectivity_base;
    const std::uint32_t widest = std::max(m, m0);
    if (widest > 56)
        return "Connectivity is too large for the device builder (base connectivity must not exceed 56)";
    const std::uint32_t inbox_cap = std::min<std::uint32_t>(32, 64 - widest); // existing + incoming fit one wave
    const std::uint32_t ef = std::max<std::uint32_t>(widest + 1, config_.expansion_add); // index.hpp:2799-2800
    if (ef > build_max_candidates_k)
        return "Expansion is too large for the device builder";
    if (bytes_per_vector(scalar, dimensions) == 0 || stride < bytes_per_vector(scalar, dimensions))
        return "Stride is smaller than one vector";
    config_.batch_divisor = std::max<std::uint32_t>(1, config_.batch_divisor);
    config_.max_batch = std::max<std::uint32_t>(1, config_.max_batch);
    metric_ = metric, scalar_ = scalar, dimensions_ = dimensions, size_ = count;

    // ---- levels: the reference's distribution (index.hpp:3895-3899), one seeded generator for the whole index
    levels_.resize(count);
    {
        std::mt19937_64 generator(config_.seed);
        std::uniform_real_distribution<double> uniform(0.0, 1.0);
        const double inverse_log_connectivity = 1.0 / std::log((double)m);
        for (std::uint64_t i = 0; i < count; ++i) {
            double u = uniform(generator);
            if (u <= 0.0)
                u = 1e-300;
            const double r = -std::log(u) * inverse_log_connectivity;
            levels_[i] = (std::int16_t)std::min(r, 30.0);
        }
    }
    keys_.clear();
    if (keys)
        keys_.assign(keys, keys + count);

    const double t_upload = seconds_now();
    if (const char* e = snapshot_.allocate_for_build(metric, scalar, dimensions, count, m, m0, levels_.data(), vectors,
                                                     stride, vectors_on_device, keys, device))
        return e;
    stats_.seconds_upload = seconds_now() - t_upload;
    upper_lists_ = snapshot_.upper_lists();
    const snapshot_view_t& view = snapshot_.view();
    hipStream_t stream = snapshot_.stream();

    const std::uint64_t max_batch = std::min<std::uint64_t>(config_.max_batch, count);
    device_block_t block;
    std::uint32_t *d_nodes = nullptr, *d_inbox_count = nullptr, *d_touched = nullptr, *d_touched_count = nullptr;
    std::uint64_t *d_cand_slots = nullptr, *d_cand_counts = nullptr, *d_visited = nullptr, *d_computed = nullptr;
    cand_t* d_inbox = nullptr;
    float* d_cand_distances = nullptr;
    unsigned long long* d_counters = nullptr;
    UA_HIP(block.allocate(&d_nodes, max_batch * 4));
    UA_HIP(block.allocate(&d_cand_slots, max_batch * ef * 8));
    UA_HIP(block.allocate(&d_cand_distances, max_batch * ef * 4));
    UA_HIP(block.allocate(&d_cand_counts, max_batch * 8));
    UA_HIP(block.allocate(&d_visited, max_batch * 8));
    UA_HIP(block.allocate(&d_computed, max_batch * 8));
    UA_HIP(block.allocate(&d_inbox_count, count * 4));
    UA_HIP(block.allocate(&d_inbox, count * inbox_cap * 8));
    UA_HIP(block.allocate(&d_touched, max_batch * m * 4));
    UA_HIP(block.allocate(&d_touched_count, 16));
    UA_HIP(block.allocate(&d_counters, 64));
    UA_HIP(hipMemset(d_inbox_count, 0, count * 4));
    UA_HIP(hipMemset(d_counters, 0, 64));

    build_args_t args{};
    args.nbr0 = snapshot_.mutable_nbr0();
    args.upper = snapshot_.mutable_upper();
    args.upper_ref = view.upper_ref;
    args.needed = m;
    args.nodes = d_nodes;
    args.cand_slots = d_cand_slots;
    args.cand_distances = d_cand_distances;
    args.cand_counts = d_cand_counts;
    args.ef = ef;
    args.inbox_count = d_inbox_count;
    args.inbox = d_inbox;
    args.inbox_cap = inbox_cap;
    args.touched = d_touched;
    args.touched_count = d_touched_count;
    args.counters = d_counters;

    entry_slot_ = 0;
    max_level_ = (std::uint32_t)levels_[0]; // the first node only becomes the entry point, index.hpp:2835-2840
    std::vector<std::uint32_t> nodes;
    std::vector<std::uint64_t> host_counters(max_batch);
    const std::uint32_t resident = (std::uint32_t)snapshot_.compute_units() * 8;

    for (std::uint64_t begin = 1; begin < count;) {
        const std::uint64_t limit = std::max<std::uint64_t>(1, std::min<std::uint64_t>(max_batch, begin / config_.batch_divisor));
        const std::uint64_t end = std::min<std::uint64_t>(count, begin + limit);
        std::uint32_t batch_top = 0;
        for (std::uint64_t i = begin; i < end; ++i)
            batch_top = std::max<std::uint32_t>(batch_top, (std::uint32_t)levels_[i]);
        const std::uint32_t linked_top = std::min(batch_top, max_level_);
        snapshot_.set_frontier(begin, entry_slot_, max_level_);

        // bottom-up: the search on level l reads level l and the levels above it, none of which this batch has touched yet
        for (std::uint32_t level = 0; level <= linked_top; ++level) {
            nodes.clear();
            for (std::uint64_t i = begin; i < end; ++i)
                if ((std::uint32_t)levels_[i] >= level)
                    nodes.push_back((std::uint32_t)i);
            if (nodes.empty())
                continue;
            const std::uint32_t pass_count = (std::uint32_t)nodes.size();
            UA_HIP(hipMemcpyAsync(d_nodes, nodes.data(), (std::size_t)pass_count * 4, hipMemcpyHostToDevice, stream));
            UA_HIP(hipStreamSynchronize(stream)); // `nodes` is pageable and reused

            const double t_search = seconds_now();
            search_extras_t extras;
            extras.query_ids = d_nodes;
            extras.beam_level = level;
            extras.emit_slots = true;
            extras.reference_frontier = true; // builds stay byte-for-byte reproducible against the reference-shaped oracle
            search_stats_t search_stats;
            if (const char* e = snapshot_.search_device(view.vectors, pass_count, view.row_stride, ef, ef, d_cand_slots,
                                                        d_cand_distances, d_cand_counts, d_visited, d_computed, stream,
                                                        search_tuning_t{}, &search_stats, false, &extras))
                return e;
            stats_.seconds_search += seconds_now() - t_search;

            const double t_link = seconds_now();
            UA_HIP(hipMemsetAsync(d_touched_count, 0, 4, stream));
            args.level = level;
            args.capacity = level ? m : m0;
            args.count = pass_count;
            build_params_t params{};
            params.lanes = snapshot_.lanes_per_row();
            params.stream = stream;
            params.reverse = 0;
            params.grid = std::min<std::uint32_t>(pass_count, resident);
            UA_HIP(launch_build(metric, scalar, params, view, args));
            params.reverse = 1;
            params.grid = (std::uint32_t)std::min<std::uint64_t>((std::uint64_t)pass_count * m, resident);
            UA_HIP(launch_build(metric, scalar, params, view, args));
            // traversal counters of this pass (tiny) while the link kernels run
            UA_HIP(hipMemcpyAsync(host_counters.data(), d_computed, (std::size_t)pass_count * 8, hipMemcpyDeviceToHost, stream));
            UA_HIP(hipStreamSynchronize(stream));
            for (std::uint32_t i = 0; i < pass_count; ++i)
                stats_.search_distances += host_counters[i];
            UA_HIP(hipMemcpyAsync(host_counters.data(), d_visited, (std::size_t)pass_count * 8, hipMemcpyDeviceToHost, stream));
            UA_HIP(hipStreamSynchronize(stream));
            for (std::uint32_t i = 0; i < pass_count; ++i)
                stats_.search_hops += host_counters[i];
            stats_.seconds_link += seconds_now() - t_link;
            ++stats_.passes;
        }
        if (batch_top > max_level_) { // index.hpp:2874-2877: a taller node becomes the entry point
            for (std::uint64_t i = begin; i < end; ++i)
                if ((std::uint32_t)levels_[i] == batch_top) {
                    entry_slot_ = (std::uint32_t)i;
                    break;
                }
            max_level_ = batch_top;
        }
        ++stats_.batches;
        begin = end;
    }
    snapshot_.set_frontier(count, entry_slot_, max_level_);
    // removed members (key == free_key_, index_dense.hpp:513) were linked like any other node — the reference keeps them in
    // the graph too — and stop matching from here on
    bool tombstones = false;
    for (std::uint64_t key : keys_)
        tombstones |= key == free_key_k;
    snapshot_.set_tombstones(tombstones);
    if (const char* e = snapshot_.finalize_layout()) // the graph is final: rows of ≤ 16 bytes move next to the lists
        return e;
    unsigned long long counters[4] = {0, 0, 0, 0};
    UA_HIP(hipMemcpy(counters, d_counters, sizeof(counters), hipMemcpyDeviceToHost));
    stats_.select_distances = counters[0];
    stats_.reverse_distances = counters[1];
    stats_.repruned_lists = counters[2];
    stats_.dropped_requests = counters[3];
    stats_.max_level = max_level_;
    stats_.seconds_total = seconds_now() - t_begin;
    return nullptr;
}

std::size_t builder_t::serialized_length() const {
    const std::size_t bpv = bytes_per_vector(scalar_, dimensions_);
    const std::size_t m = config_.connectivity, m0 = config_.connectivity_base;
    return 8 + (std::size_t)size_ * bpv + 64 + 40 + (std::size_t)size_ * 2 +
           (std::size_t)size_ * (10 + 4 + 4 * m0) + (std::size_t)upper_lists_ * (4 + 4 * m);
}

const char* builder_t::save_buffer(void* buffer, std::size_t length) {
    if (!size_)
        return "Nothing was built";
    if (length < serialized_length())
        return "Buffer is too small";
    const snapshot_view_t& view = snapshot_.view();
    const std::size_t bpv = view.bytes_per_vector;
    const std::uint64_t n = size_;
    const std::uint32_t m = view.m, m0 = view.m0;
    UA_HIP(hipSetDevice(snapshot_.device()));
    std::uint8_t* p = static_cast<std::uint8_t*>(buffer);

    // matrix block, index_dense.hpp:1004-1030
    const std::uint32_t rows = (std::uint32_t)n, cols = (std::uint32_t)bpv;
    std::memcpy(p, &rows, 4);
    std::memcpy(p + 4, &cols, 4);
    p += 8;
    if (view.row_stride == bpv)
        UA_HIP(hipMemcpy(p, view.vectors, (std::size_t)n * bpv, hipMemcpyDeviceToHost));
    else
        UA_HIP(hipMemcpy2D(p, bpv, view.vectors, view.row_stride, bpv, n, hipMemcpyDeviceToHost));
    p += (std::size_t)n * bpv;

    // 64-byte head, index_dense.hpp:42-79 / 1036-1053
    std::memset(p, 0, 64);
    std::memcpy(p, "usearch", 7);
    const std::uint16_t version[3] = {2, 21, 0};
    std::memcpy(p + 7, version, 6);
    p[13] = (std::uint8_t)metric_;
    p[14] = (std::uint8_t)scalar_;
    p[15] = (std::uint8_t)scalar_u64_k;
    p[16] = (std::uint8_t)scalar_u32_k;
    std::uint64_t deleted = 0; // members whose key is the tombstone value (index_dense.hpp:1051-1052)
    for (std::uint64_t key : keys_)
        deleted += key == free_key_k;
    const std::uint64_t present = n - deleted, dimensions = dimensions_;
    std::memcpy(p + 17, &present, 8);
    std::memcpy(p + 25, &deleted, 8);
    std::memcpy(p + 33, &dimensions, 8);
    p[41] = 0; // multi
    p += 64;

    // graph header, index.hpp:1863-1869
    const std::uint64_t header[5] = {n, m, m0, max_level_, entry_slot_};
    std::memcpy(p, header, 40);
    p += 40;
    std::memcpy(p, levels_.data(), (std::size_t)n * 2); // index.hpp:3298-3305
    p += (std::size_t)n * 2;

    // node tapes, index.hpp:3308-3314: u64 key | i16 level | {u32 count, u32 × M0} | level × {u32 count, u32 × M}
    std::vector<std::uint32_t> nbr0((std::size_t)n * m0), upper((std::size_t)std::max<std::uint64_t>(upper_lists_, 1) * m);
    UA_HIP(hipMemcpy(nbr0.data(), view.nbr0, nbr0.size() * 4, hipMemcpyDeviceToHost));
    UA_HIP(hipMemcpy(upper.data(), view.upper, upper.size() * 4, hipMemcpyDeviceToHost));
    std::vector<std::uint64_t> offsets(n + 1);
    std::uint64_t offset = 0, lists = 0;
    std::vector<std::uint64_t> first_list(n);
    for (std::uint64_t i = 0; i < n; ++i) {
        offsets[i] = offset;
        first_list[i] = lists;
        lists += (std::uint64_t)levels_[i];
        offset += 10 + (4 + 4 * (std::size_t)m0) + (std::size_t)levels_[i] * (4 + 4 * (std::size_t)m);
    }
    offsets[n] = offset;
    std::uint8_t* tapes = p;
    const bool identity = keys_.empty();
    parallel_ranges(n, [&](std::uint64_t begin, std::uint64_t end) {
        for (std::uint64_t i = begin; i < end; ++i) {
            std::uint8_t* tape = tapes + offsets[i];
            const std::uint64_t key = identity ? i : keys_[i];
            std::memcpy(tape, &key, 8);
            std::memcpy(tape + 8, &levels_[i], 2);
            std::uint8_t* list = tape + 10;
            auto write_list = [&](const std::uint32_t* cells, std::uint32_t capacity) {
                std::uint32_t used = 0;
                while (used < capacity && cells[used] != none_slot_k)
                    ++used;
                std::memcpy(list, &used, 4);
                std::memcpy(list + 4, cells, (std::size_t)used * 4);
                std::memset(list + 4 + (std::size_t)used * 4, 0, (std::size_t)(capacity - used) * 4);
                list += 4 + 4 * (std::size_t)capacity;
            };
            write_list(nbr0.data() + (std::size_t)i * m0, m0);
            for (std::int16_t l = 1; l <= levels_[i]; ++l)
                write_list(upper.data() + (std::size_t)(first_list[i] + (l - 1)) * m, m);
        }
    });
    return nullptr;
}

} // namespace usearch_amd
