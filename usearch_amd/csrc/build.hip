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

static_assert(builder_max_expansion_k == build_max_candidates_k, "build.hpp and build_kernels.hpp disagree");

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

/// One thread per parked request of the previous round (`from_*`, `from_count` of them): filed again now that the reverse kernel
/// has applied and emptied the inboxes; what still does not fit is parked for the round after (`b.deferred_*`).
__global__ void build_refile_kernel(const build_args_t b, const std::uint32_t* from_targets, const cand_t* from_requests,
                                    std::uint32_t from_count) {
    const std::uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= from_count)
        return;
    const std::uint32_t target = from_targets[i];
    const cand_t request = from_requests[i];
    if (!build_file_request(b, target, request))
        if (build_defer_request(b, target, request))
            atomicAdd(b.counters + 3, 1ull);
}

/// Frees a set of device allocations when the build function returns, whatever the path.
struct device_block_t {
    std::vector<void*> pointers;
    ~device_block_t() {
        for (void* p : pointers)
            placed_free(p); // time-stamps the release of large blocks (placement.hpp: settle, then allocate)
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
                                           std::uint64_t capacity, std::uint64_t lists_capacity, std::uint32_t m,
                                           std::uint32_t m0, int device) {
    if (!kernel_available(metric, scalar))
        return "No MI355X kernel for this metric / scalar kind combination";
    release();
    device_ = device;
    UA_HIP(hipSetDevice(device));
    metric_ = metric;
    scalar_ = scalar;
    count_present_ = 0;
    upper_lists_ = 0;
    const std::uint32_t bpv = (std::uint32_t)bytes_per_vector(scalar, dimensions);
    std::uint32_t row_stride = 0, row_chunks = 0;
    row_geometry(bpv, lanes_, row_stride, row_chunks);

    view_ = snapshot_view_t{};
    view_.row_stride = row_stride;
    view_.chunks = row_chunks;
    view_.bytes_per_vector = bpv;
    view_.dimensions = (std::uint32_t)dimensions;
    view_.m = m;
    view_.m0 = m0;
    device_bytes_ = 0;
    build_capacity_ = build_lists_capacity_ = 0;
    if (const char* e = grow_for_build(capacity, lists_capacity))
        return e;

    hipDeviceProp_t properties;
    UA_HIP(hipGetDeviceProperties(&properties, device));
    compute_units_ = properties.multiProcessorCount > 0 ? properties.multiProcessorCount : 256;
    UA_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    return nullptr;
}

const char* snapshot_t::grow_for_build(std::uint64_t capacity, std::uint64_t lists_capacity) {
    capacity = std::max<std::uint64_t>(capacity, build_capacity_);
    lists_capacity = std::max<std::uint64_t>(std::max<std::uint64_t>(lists_capacity, 1), build_lists_capacity_);
    if (capacity == build_capacity_ && lists_capacity == build_lists_capacity_)
        return nullptr;
    if (capacity >= none_slot_k || lists_capacity >= none_slot_k)
        return "Index is too large for 32-bit slots";
    UA_HIP(hipSetDevice(device_));
    const std::uint32_t m = view_.m, m0 = view_.m0, row_stride = view_.row_stride;
    // every array is re-allocated at the new size and the linked part copied over, device to device; new cells are empty
    struct array_t {
        void** pointer;
        std::size_t old_bytes, new_bytes;
        int fill; // byte value of the fresh part
    };
    array_t arrays[] = {
        {&d_vectors_, (std::size_t)build_capacity_ * row_stride, (std::size_t)capacity * row_stride, 0},
        {&d_nbr0_, (std::size_t)build_capacity_ * m0 * 4, (std::size_t)capacity * m0 * 4, 0xFF},
        {&d_upper_ref_, (std::size_t)build_capacity_ * 4, (std::size_t)capacity * 4, 0xFF},
        {&d_upper_, (std::size_t)build_lists_capacity_ * m * 4, (std::size_t)lists_capacity * m * 4, 0xFF},
        {&d_keys_, (std::size_t)build_capacity_ * 8, (std::size_t)capacity * 8, 0},
    };
    for (array_t& array : arrays) {
        if (array.new_bytes == array.old_bytes && *array.pointer)
            continue;
        void* fresh = nullptr;
        if (array.pointer == &d_vectors_) { // the matrix the walk gathers rows from (placement.hpp)
            UA_HIP(placed_malloc(&fresh, array.new_bytes, row_stride, &placement_));
            placement_trials_left_ = placement_max_draws_k, placement_losses_ = 0, placement_last_ef_ = 0, placement_reopens_ = 0; // a new matrix: its placement is judged anew
            vectors_bytes_ = array.new_bytes;
        } else if (array.pointer == &d_nbr0_) {
            UA_HIP(placed_malloc(&fresh, array.new_bytes, (std::size_t)m0 * 4, nullptr));
        } else {
            UA_HIP(hipMalloc(&fresh, std::max<std::size_t>(array.new_bytes, 16)));
        }
        if (array.old_bytes && *array.pointer)
            UA_HIP(hipMemcpy(fresh, *array.pointer, array.old_bytes, hipMemcpyDeviceToDevice));
        if (array.new_bytes > array.old_bytes)
            UA_HIP(hipMemset(static_cast<std::uint8_t*>(fresh) + array.old_bytes, array.fill, array.new_bytes - array.old_bytes));
        if (*array.pointer) {
            placed_free(*array.pointer);
        }
        *array.pointer = fresh;
        device_bytes_ += std::max<std::size_t>(array.new_bytes, 16) - (array.old_bytes ? std::max<std::size_t>(array.old_bytes, 16) : 0);
    }
    build_capacity_ = capacity;
    build_lists_capacity_ = lists_capacity;
    view_.vectors = static_cast<const std::uint8_t*>(d_vectors_);
    view_.nbr0 = static_cast<const std::uint32_t*>(d_nbr0_);
    view_.upper_ref = static_cast<const std::uint32_t*>(d_upper_ref_);
    view_.upper = static_cast<const std::uint32_t*>(d_upper_);
    view_.keys = static_cast<const std::uint64_t*>(d_keys_);
    if (d_nbr0_rows_) { // laid out for the old arrays: rebuilt by the next finalize_layout
        placed_free(d_nbr0_rows_); // finalize_layout allocates it with placed_malloc (possibly a mapped range)
        d_nbr0_rows_ = nullptr;
        view_.nbr0_rows = nullptr;
    }
    return nullptr;
}

const char* snapshot_t::append_for_build(std::uint64_t first, std::uint64_t count, const std::uint32_t* upper_refs,
                                         const void* vectors, std::size_t stride, bool vectors_on_device,
                                         const std::uint64_t* keys) {
    if (!count)
        return nullptr;
    if (first + count > build_capacity_)
        return "The build arrays are too small";
    UA_HIP(hipSetDevice(device_));
    const std::uint32_t bpv = view_.bytes_per_vector, row_stride = view_.row_stride;
    std::uint8_t* rows = static_cast<std::uint8_t*>(d_vectors_) + first * row_stride;
    if (vectors_on_device) {
        if (row_stride == bpv && stride == bpv)
            UA_HIP(hipMemcpy(rows, vectors, (std::size_t)count * bpv, hipMemcpyDeviceToDevice));
        else // the padding was zeroed when the array was allocated
            UA_HIP(hipMemcpy2D(rows, row_stride, vectors, stride, bpv, count, hipMemcpyDeviceToDevice));
    } else if (const char* e = upload_rows(rows, row_stride, static_cast<const std::uint8_t*>(vectors), stride, bpv, count)) {
        return e;
    }
    UA_HIP(hipMemcpy(static_cast<std::uint32_t*>(d_upper_ref_) + first, upper_refs, count * 4, hipMemcpyHostToDevice));
    if (keys) {
        UA_HIP(hipMemcpy(static_cast<std::uint64_t*>(d_keys_) + first, keys, count * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<std::uint64_t> identity(count);
        for (std::uint64_t i = 0; i < count; ++i)
            identity[i] = first + i;
        UA_HIP(hipMemcpy(static_cast<std::uint64_t*>(d_keys_) + first, identity.data(), count * 8, hipMemcpyHostToDevice));
    }
    count_present_ = first + count;
    ++mutations_;
    return nullptr;
}

const char* snapshot_t::overwrite_member(std::uint64_t slot, const void* vector, std::uint64_t key) {
    if (slot >= view_.size)
        return "No such member";
    UA_HIP(hipSetDevice(device_));
    std::uint8_t* row = static_cast<std::uint8_t*>(d_vectors_) + slot * view_.row_stride; // the padding behind the row stays zero
    if (const char* e = upload_rows(row, view_.row_stride, static_cast<const std::uint8_t*>(vector), view_.bytes_per_vector,
                                    view_.bytes_per_vector, 1))
        return e;
    UA_HIP(hipMemcpy(static_cast<std::uint64_t*>(d_keys_) + slot, &key, 8, hipMemcpyHostToDevice));
    ++mutations_;
    return nullptr;
}

const char* snapshot_t::set_key(std::uint64_t slot, std::uint64_t key) {
    if (slot >= view_.size && slot >= build_capacity_)
        return "No such member";
    UA_HIP(hipSetDevice(device_));
    UA_HIP(hipMemcpy(static_cast<std::uint64_t*>(d_keys_) + slot, &key, 8, hipMemcpyHostToDevice));
    if (key == free_key_k)
        view_.has_tombstones = 1;
    ++mutations_;
    return nullptr;
}

builder_t::~builder_t() { release_workspace(); }

void builder_t::release_workspace() {
    for (void* p : workspace_)
        placed_free(p); // time-stamps the release of large blocks (placement.hpp: settle, then allocate)
    workspace_.clear();
    workspace_nodes_ = 0;
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
    const std::uint32_t widest = std::max(m, config_.connectivity_base);
    // a list and the requests filed against it are re-pruned by one wave (build_reverse_kernel): 64 - widest requests per round;
    // what does not fit waits for the next round (build_refile_kernel), so even a one-request inbox builds, in more rounds
    if (widest > build_max_capacity_k || m > 64)
        return "Connectivity is too large for the device builder (connectivity ≤ 64, base connectivity ≤ 128)";
    const std::uint32_t ef = std::max<std::uint32_t>(widest + 1, config_.expansion_add); // index.hpp:2799-2800
    if (ef > build_max_candidates_k)
        return "Expansion is too large for the device builder";
    if (bytes_per_vector(scalar, dimensions) == 0 || stride < bytes_per_vector(scalar, dimensions))
        return "Stride is smaller than one vector";
    config_.batch_divisor = std::max<std::uint32_t>(1, config_.batch_divisor);
    config_.max_batch = std::max<std::uint32_t>(1, config_.max_batch);
    metric_ = metric, scalar_ = scalar, dimensions_ = dimensions;
    size_ = 0, upper_lists_ = 0, entry_slot_ = 0, max_level_ = 0;
    levels_.clear();
    keys_.clear();
    identity_keys_ = keys == nullptr;
    generator_.seed(config_.seed); // one seeded generator for the whole life of the index: levels do not depend on how the
                                   // vectors arrive (one call or many `extend`s)
    release_workspace();
    // the first call knows its size: exact arrays; later `extend`s grow them geometrically
    if (const char* e = snapshot_.allocate_for_build(metric, scalar, dimensions, 0, 0, m, config_.connectivity_base, device))
        return e;
    const char* error = extend(vectors, count, stride, vectors_on_device, keys, true);
    release_workspace(); // a one-shot build gives its link workspace back (an inbox per member: 32 GB for 125M members);
                         // `extend` makes a new one — and keeps it — when members are added later
    stats_.seconds_total = seconds_now() - t_begin;
    return error;
}

const char* builder_t::extend(const void* vectors, std::uint64_t count, std::size_t stride, bool vectors_on_device,
                              const std::uint64_t* keys, bool exact_capacity) {
    const double t_begin = seconds_now();
    if (!count)
        return nullptr;
    if (!vectors)
        return "Nothing to add";
    const std::uint64_t first = size_, total = size_ + count;
    if (total >= none_slot_k)
        return "Index is too large for 32-bit slots";
    if (stride < bytes_per_vector(scalar_, dimensions_))
        return "Stride is smaller than one vector";
    if (!keys && !identity_keys_)
        return "Keys are needed: earlier members have them";
    if (keys && identity_keys_ && first) { // earlier members were keyed by their row number
        keys_.resize(first);
        for (std::uint64_t i = 0; i < first; ++i)
            keys_[i] = i;
    }
    const std::uint32_t m = config_.connectivity;

    // ---- levels: the reference's distribution (index.hpp:3895-3899)
    levels_.resize(total);
    std::vector<std::uint32_t> upper_refs(count);
    {
        std::uniform_real_distribution<double> uniform(0.0, 1.0);
        const double inverse_log_connectivity = 1.0 / std::log((double)m);
        for (std::uint64_t i = first; i < total; ++i) {
            double u = uniform(generator_);
            if (u <= 0.0)
                u = 1e-300;
            const double r = -std::log(u) * inverse_log_connectivity;
            levels_[i] = (std::int16_t)std::min(r, 30.0);
            upper_refs[i - first] = levels_[i] ? (std::uint32_t)upper_lists_ : none_slot_k;
            upper_lists_ += (std::uint64_t)levels_[i];
        }
    }
    if (upper_lists_ >= none_slot_k)
        return "Too many upper-level lists for 32-bit references";
    if (keys) {
        identity_keys_ = false;
        keys_.insert(keys_.end(), keys, keys + count);
    }

    // ---- room: exact on the first call, doubling afterwards (plus the upper-level lists the spare members will need:
    //      a member has 1 / (M - 1) of them on average)
    const double t_upload = seconds_now();
    std::uint64_t capacity = total, lists_capacity = upper_lists_;
    if (!exact_capacity && total > snapshot_.build_capacity()) {
        capacity = std::max<std::uint64_t>(total, std::min<std::uint64_t>(2 * snapshot_.build_capacity(), none_slot_k - 2));
        capacity = std::max<std::uint64_t>(capacity, 1024);
    }
    if (!exact_capacity && (upper_lists_ > snapshot_.build_lists_capacity() || capacity > snapshot_.build_capacity()))
        lists_capacity = upper_lists_ + (capacity - total) / std::max<std::uint32_t>(m - 1, 1) * 3 / 2 + 1024;
    if (const char* e = snapshot_.grow_for_build(capacity, lists_capacity))
        return e;
    if (const char* e = snapshot_.append_for_build(first, count, upper_refs.data(), vectors, stride, vectors_on_device, keys))
        return e;
    stats_.seconds_upload += seconds_now() - t_upload;
    snapshot_.set_upper_lists(upper_lists_);

    if (!first)
        max_level_ = (std::uint32_t)levels_[0], entry_slot_ = 0; // the first node only becomes the entry point, index.hpp:2835-2840
    size_ = total;
    snapshot_.suspend_layout(); // lists change from here on: the rows copied next to them would go stale
    if (const char* e = link_range(std::max<std::uint64_t>(first, 1), total))
        return e;
    snapshot_.set_frontier(total, entry_slot_, max_level_);
    // removed members (key == free_key_, index_dense.hpp:513) were linked like any other node — the reference keeps them in
    // the graph too — and stop matching from here on
    bool tombstones = false;
    for (std::uint64_t key : keys_)
        tombstones |= key == free_key_k;
    snapshot_.set_tombstones(tombstones);
    if (const char* e = snapshot_.finalize_layout()) // the graph is final for now: rows of ≤ 16 bytes move next to the lists
        return e;
    stats_.max_level = max_level_;
    stats_.seconds_total += seconds_now() - t_begin;
    return nullptr;
}

const char* builder_t::update(const std::uint32_t* slots, std::uint64_t count, const void* vectors, std::size_t stride,
                              const std::uint64_t* keys) {
    if (!count)
        return nullptr;
    if (!slots || !vectors || !keys)
        return "Nothing to update";
    if (stride < bytes_per_vector(scalar_, dimensions_))
        return "Stride is smaller than one vector";
    if (identity_keys_) {
        keys_.resize(size_);
        for (std::uint64_t i = 0; i < size_; ++i)
            keys_[i] = i;
        identity_keys_ = false;
    }
    for (std::uint64_t i = 0; i < count; ++i) {
        if (slots[i] >= size_)
            return "No such member";
        keys_[slots[i]] = keys[i];
        if (const char* e = snapshot_.overwrite_member(slots[i], static_cast<const std::uint8_t*>(vectors) + i * stride, keys[i]))
            return e;
    }
    snapshot_.suspend_layout(); // lists change from here on
    if (const char* e = link_range(size_, size_, slots, count))
        return e;
    snapshot_.set_frontier(size_, entry_slot_, max_level_);
    bool tombstones = false;
    for (std::uint64_t key : keys_)
        tombstones |= key == free_key_k;
    snapshot_.set_tombstones(tombstones);
    return snapshot_.finalize_layout();
}

const char* builder_t::set_key(std::uint64_t slot, std::uint64_t key) {
    if (slot >= size_)
        return "No such member";
    if (identity_keys_) {
        keys_.resize(size_);
        for (std::uint64_t i = 0; i < size_; ++i)
            keys_[i] = i;
        identity_keys_ = false;
    }
    keys_[slot] = key;
    return snapshot_.set_key(slot, key);
}

/// Links members [begin, end) into the graph of the members before them, batch by batch.
const char* builder_t::link_range(std::uint64_t begin, const std::uint64_t end_total, const std::uint32_t* relinked,
                                  std::uint64_t relinked_count) {
    if (begin >= end_total && !relinked_count)
        return nullptr;
    const metric_kind_t metric = metric_;
    const scalar_kind_t scalar = scalar_;
    const std::uint32_t m = config_.connectivity, m0 = config_.connectivity_base;
    const std::uint32_t widest = std::max(m, m0);
    // a list and what is filed against it share one wave (a candidate per lane) while they fit 64; wider lists go through LDS
    const std::uint32_t inbox_cap = widest < 64 ? std::min<std::uint32_t>(32, 64 - widest) : build_wide_inbox_k;
    const std::uint32_t ef = std::max<std::uint32_t>(widest + 1, config_.expansion_add);
    UA_HIP(hipSetDevice(snapshot_.device()));
    hipStream_t stream = snapshot_.stream();

    // ---- workspace: per-batch arrays sized for the largest batch, per-member inboxes for the arrays' capacity; kept between
    //      calls (the link kernels leave every inbox empty), re-made when the arrays have grown
    const std::uint64_t max_batch = std::max<std::uint64_t>(1, std::min<std::uint64_t>(config_.max_batch, std::max<std::uint64_t>(end_total, 1)));
    const std::uint64_t members = snapshot_.build_capacity();
    if (workspace_nodes_ < members || workspace_batch_ < max_batch) {
        release_workspace();
        auto allocate = [&](void** out, std::size_t bytes) -> hipError_t {
            void* p = nullptr;
            const hipError_t e = hipMalloc(&p, std::max<std::size_t>(bytes, 16));
            if (e == hipSuccess)
                workspace_.push_back(p);
            *out = p;
            return e;
        };
        UA_HIP(allocate((void**)&d_nodes_, max_batch * 4));
        UA_HIP(allocate((void**)&d_cand_slots_, max_batch * ef * 8));
        UA_HIP(allocate((void**)&d_cand_distances_, max_batch * ef * 4));
        UA_HIP(allocate((void**)&d_cand_counts_, max_batch * 8));
        UA_HIP(allocate((void**)&d_visited_, max_batch * 8));
        UA_HIP(allocate((void**)&d_computed_, max_batch * 8));
        UA_HIP(allocate((void**)&d_inbox_count_, members * 4));
        UA_HIP(allocate((void**)&d_inbox_, members * inbox_cap * 8));
        UA_HIP(allocate((void**)&d_touched_, max_batch * m * 4));
        UA_HIP(allocate((void**)&d_touched_count_, 16));
        UA_HIP(allocate((void**)&d_counters_, 64));
        for (int lot = 0; lot < 2; ++lot) {
            UA_HIP(allocate((void**)&d_deferred_targets_[lot], max_batch * m * 4));
            UA_HIP(allocate(&d_deferred_requests_[lot], max_batch * m * 8));
        }
        UA_HIP(allocate((void**)&d_deferred_count_, 16));
        UA_HIP(hipMemset(d_inbox_count_, 0, members * 4));
        UA_HIP(hipMemset(d_counters_, 0, 64));
        workspace_nodes_ = members;
        workspace_batch_ = max_batch;
    }
    const snapshot_view_t& view = snapshot_.view();

    build_args_t args{};
    args.nbr0 = snapshot_.mutable_nbr0();
    args.upper = snapshot_.mutable_upper();
    args.upper_ref = view.upper_ref;
    args.needed = m;
    args.nodes = d_nodes_;
    args.cand_slots = d_cand_slots_;
    args.cand_distances = d_cand_distances_;
    args.cand_counts = d_cand_counts_;
    args.ef = ef;
    args.inbox_count = d_inbox_count_;
    args.inbox = static_cast<cand_t*>(d_inbox_);
    args.inbox_cap = inbox_cap;
    args.touched = d_touched_;
    args.touched_count = d_touched_count_;
    args.counters = d_counters_;
    args.deferred_cap = (std::uint32_t)std::min<std::uint64_t>(max_batch * m, 0xFFFFFFFFull); // every request of a pass fits
    args.candidate_cap = std::max<std::uint32_t>(widest + inbox_cap > 64 ? build_selected_k : 64, (ef + 63) / 64 * 64);

    std::vector<std::uint32_t> nodes;
    std::vector<std::uint64_t> host_counters(max_batch);
    const std::uint32_t resident = (std::uint32_t)snapshot_.compute_units() * 8;

    /// One batch: `batch` slots are linked into the graph of the first `frontier` members, level by level. `relink` = the members
    /// exist already (their slots are being recycled, index_gt::update, index.hpp:2916-2999): each keeps its level, routes through
    /// its own stale inbound links but never becomes its own candidate (`search_to_update_`).
    auto link_batch = [&](const std::vector<std::uint32_t>& batch, std::uint64_t frontier, bool relink) -> const char* {
        std::uint32_t batch_top = 0;
        for (std::uint32_t i : batch)
            batch_top = std::max<std::uint32_t>(batch_top, (std::uint32_t)levels_[i]);
        const std::uint32_t linked_top = std::min(batch_top, max_level_);
        snapshot_.set_frontier(frontier, entry_slot_, max_level_);

        // bottom-up: the search on level l reads level l and the levels above it, none of which this batch has touched yet
        for (std::uint32_t level = 0; level <= linked_top; ++level) {
            nodes.clear();
            for (std::uint32_t i : batch)
                if ((std::uint32_t)levels_[i] >= level)
                    nodes.push_back(i);
            if (nodes.empty())
                continue;
            const std::uint32_t pass_count = (std::uint32_t)nodes.size();
            UA_HIP(hipMemcpyAsync(d_nodes_, nodes.data(), (std::size_t)pass_count * 4, hipMemcpyHostToDevice, stream));
            UA_HIP(hipStreamSynchronize(stream)); // `nodes` is pageable and reused

            const double t_search = seconds_now();
            search_extras_t extras;
            extras.query_ids = d_nodes_;
            extras.beam_level = level;
            extras.emit_slots = true;
            extras.reference_frontier = true; // builds stay byte-for-byte reproducible against the reference-shaped oracle
            extras.exclude_own = relink;
            search_stats_t search_stats;
            if (const char* e = snapshot_.search_device(view.vectors, pass_count, view.row_stride, ef, ef, d_cand_slots_,
                                                        d_cand_distances_, d_cand_counts_, d_visited_, d_computed_, stream,
                                                        search_tuning_t{}, &search_stats, false, &extras))
                return e;
            stats_.seconds_search += seconds_now() - t_search;

            const double t_link = seconds_now();
            UA_HIP(hipMemsetAsync(d_touched_count_, 0, 4, stream));
            UA_HIP(hipMemsetAsync(d_deferred_count_, 0, 8, stream));
            int lot = 0;
            args.deferred_targets = d_deferred_targets_[lot];
            args.deferred_requests = static_cast<cand_t*>(d_deferred_requests_[lot]);
            args.deferred_count = d_deferred_count_ + lot;
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
            // Requests that found their target's inbox full (a hub that many of this pass's nodes picked: the inbox holds 64 - M0
            // … 32 requests, index.hpp:3848-3893 takes any number) were parked: now that the reverse kernel has applied and
            // emptied the inboxes they are filed again, round after round, until none is waiting. Nothing is dropped.
            // Every round files at least one request per target that still has some (its inbox starts the round empty), so the
            // number waiting falls strictly; should it ever not, what is left is counted as dropped instead of spinning.
            for (std::uint32_t before = 0xFFFFFFFFu;;) {
                std::uint32_t waiting = 0;
                UA_HIP(hipMemcpyAsync(&waiting, d_deferred_count_ + lot, 4, hipMemcpyDeviceToHost, stream));
                UA_HIP(hipStreamSynchronize(stream));
                if (!waiting)
                    break;
                waiting = std::min(waiting, args.deferred_cap);
                if (waiting >= before) {
                    unfiled_requests_ += waiting;
                    break;
                }
                before = waiting;
                stats_.refiled_requests += waiting;
                const std::uint32_t* from_targets = d_deferred_targets_[lot];
                const cand_t* from_requests = static_cast<const cand_t*>(d_deferred_requests_[lot]);
                lot ^= 1;
                UA_HIP(hipMemsetAsync(d_deferred_count_ + lot, 0, 4, stream));
                UA_HIP(hipMemsetAsync(d_touched_count_, 0, 4, stream));
                args.deferred_targets = d_deferred_targets_[lot];
                args.deferred_requests = static_cast<cand_t*>(d_deferred_requests_[lot]);
                args.deferred_count = d_deferred_count_ + lot;
                hipLaunchKernelGGL(build_refile_kernel, dim3((waiting + 255) / 256), dim3(256), 0, stream, args, from_targets,
                                   from_requests, waiting);
                UA_HIP(hipGetLastError());
                params.grid = (std::uint32_t)std::min<std::uint64_t>(waiting, resident);
                UA_HIP(launch_build(metric, scalar, params, view, args));
            }
            // traversal counters of this pass (tiny) while the link kernels run
            UA_HIP(hipMemcpyAsync(host_counters.data(), d_computed_, (std::size_t)pass_count * 8, hipMemcpyDeviceToHost, stream));
            UA_HIP(hipStreamSynchronize(stream));
            for (std::uint32_t i = 0; i < pass_count; ++i)
                stats_.search_distances += host_counters[i];
            UA_HIP(hipMemcpyAsync(host_counters.data(), d_visited_, (std::size_t)pass_count * 8, hipMemcpyDeviceToHost, stream));
            UA_HIP(hipStreamSynchronize(stream));
            for (std::uint32_t i = 0; i < pass_count; ++i)
                stats_.search_hops += host_counters[i];
            stats_.seconds_link += seconds_now() - t_link;
            ++stats_.passes;
        }
        if (batch_top > max_level_) { // index.hpp:2874-2877: a taller node becomes the entry point
            for (std::uint32_t i : batch)
                if ((std::uint32_t)levels_[i] == batch_top) {
                    entry_slot_ = i;
                    break;
                }
            max_level_ = batch_top;
        }
        ++stats_.batches;
        return nullptr;
    };

    std::vector<std::uint32_t> batch;
    if (relinked) { // members whose slots were recycled: they are in the graph already, and see all of it
        for (std::uint64_t offset = 0; offset < relinked_count; offset += max_batch) {
            batch.assign(relinked + offset, relinked + std::min<std::uint64_t>(relinked_count, offset + max_batch));
            if (const char* e = link_batch(batch, size_, true))
                return e;
        }
    }
    while (begin < end_total) {
        const std::uint64_t limit = std::max<std::uint64_t>(1, std::min<std::uint64_t>(max_batch, begin / config_.batch_divisor));
        const std::uint64_t end = std::min<std::uint64_t>(end_total, begin + limit);
        batch.clear();
        for (std::uint64_t i = begin; i < end; ++i)
            batch.push_back((std::uint32_t)i);
        if (const char* e = link_batch(batch, begin, false))
            return e;
        begin = end;
    }
    unsigned long long counters[4] = {0, 0, 0, 0};
    UA_HIP(hipMemcpy(counters, d_counters_, sizeof(counters), hipMemcpyDeviceToHost));
    stats_.select_distances = counters[0];
    stats_.reverse_distances = counters[1];
    stats_.repruned_lists = counters[2];
    stats_.dropped_requests = counters[3] + unfiled_requests_;
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
    p[41] = config_.multi ? 1 : 0; // head.multi = config.multi, index_dense.hpp:1046
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
    const bool identity = identity_keys_;
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
