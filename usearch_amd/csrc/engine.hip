/**
 *  usearch_amd/csrc/engine.hip — snapshot construction (flatten a v2 image into HBM arrays) and the batched search
 *  driver with its scratch-overflow retry ladder. See engine.hpp / kernels.hpp for the design.
 */
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "casts.hpp"
#include "host_util.hpp"
#include "kernels.hpp"

namespace usearch_amd {


void row_geometry(std::size_t bytes, std::uint32_t& lanes, std::uint32_t& row_stride) {
    // lanes per row (G): the smallest power of two covering the row's 16-byte chunks, at most 8 (= one 128-byte line per
    // load). USEARCH_AMD_LANES overrides (1, 2, 4, 8): fewer lanes per row = more rows per round trip.
    const std::uint32_t raw_chunks = std::max<std::uint32_t>(1, (std::uint32_t)((bytes + 15) / 16));
    lanes = std::min<std::uint32_t>(8, pow2_ceil(raw_chunks));
    if (raw_chunks <= 8) // rows of ≤ 128 bytes: 32 rows per round trip (a whole neighbour list) beat one line per load —
        lanes = std::min<std::uint32_t>(2, lanes); // 20M x 96 i8: 5.15 M QPS with G = 2 against 4.31 M with G = 8
    const std::size_t forced = env_size("USEARCH_AMD_LANES", 0);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8)
        lanes = std::min<std::uint32_t>((std::uint32_t)forced, pow2_ceil(raw_chunks));
    row_stride = (std::uint32_t)((bytes + 16 * lanes - 1) / (16 * lanes) * (16 * lanes));
}

/// Re-pitches `rows` host rows of `bytes` bytes (source stride `source_stride`) into device rows of `row_stride` bytes.
const char* upload_rows(std::uint8_t* device, std::uint32_t row_stride, const std::uint8_t* source,
                               std::size_t source_stride, std::size_t bytes, std::uint64_t rows) {
    if (!rows)
        return nullptr;
    if (row_stride == bytes && source_stride == bytes) {
        UA_HIP(hipMemcpy(device, source, (std::size_t)rows * bytes, hipMemcpyHostToDevice));
        return nullptr;
    }
    const std::uint64_t block_rows = std::max<std::uint64_t>(1, ((std::uint64_t)256 << 20) / row_stride);
    std::vector<std::uint8_t> block((std::size_t)std::min<std::uint64_t>(block_rows, rows) * row_stride);
    for (std::uint64_t first = 0; first < rows; first += block_rows) {
        const std::uint64_t count = std::min<std::uint64_t>(block_rows, rows - first);
        std::memset(block.data(), 0, (std::size_t)count * row_stride);
        parallel_ranges(count, [&](std::uint64_t begin, std::uint64_t end) {
            for (std::uint64_t r = begin; r < end; ++r)
                std::memcpy(block.data() + r * row_stride, source + (first + r) * source_stride, bytes);
        });
        UA_HIP(hipMemcpy(device + first * row_stride, block.data(), (std::size_t)count * row_stride, hipMemcpyHostToDevice));
    }
    return nullptr;
}

/// Jaccard over bit sets IS Tanimoto in the reference's dispatch (index_plugins.hpp:2003-2004): one kernel serves both.
static metric_kind_t kernel_metric(metric_kind_t metric) { return metric == metric_jaccard_k ? metric_tanimoto_k : metric; }

bool kernel_available(metric_kind_t metric, scalar_kind_t scalar) {
    metric = kernel_metric(metric);
#define UA_PAIR(m, sc, name)                                                                                           \
    if (metric == m && scalar == sc)                                                                                   \
        return true;
    USEARCH_AMD_FOR_EACH_PAIR(UA_PAIR)
#undef UA_PAIR
    return false;
}

snapshot_t::~snapshot_t() { release(); }

void snapshot_t::release() {
    if (!d_vectors_ && !d_nbr0_ && !d_status_ && !stream_)
        return;
    (void)hipSetDevice(device_);
    for (void* p : {d_vectors_, d_nbr0_, d_upper_ref_, d_upper_, d_keys_, (void*)d_status_, (void*)d_todo_,
                    (void*)d_queue_, (void*)d_peaks_, (void*)d_scratch_, (void*)d_stage_})
        if (p)
            (void)hipFree(p);
    if (h_status_)
        (void)hipHostFree(h_status_);
    if (event_begin_)
        (void)hipEventDestroy(event_begin_);
    if (event_end_)
        (void)hipEventDestroy(event_end_);
    if (stream_)
        (void)hipStreamDestroy(stream_);
    d_vectors_ = d_nbr0_ = d_upper_ref_ = d_upper_ = d_keys_ = nullptr;
    d_status_ = d_todo_ = d_queue_ = d_peaks_ = h_status_ = nullptr;
    d_scratch_ = d_stage_ = nullptr;
    workspace_queries_ = scratch_bytes_ = stage_bytes_ = 0;
    event_begin_ = event_end_ = nullptr;
    stream_ = nullptr;
}

const char* snapshot_t::build(const image_t& image, int device) {
    if (!kernel_available(image.metric, image.scalar))
        return "No MI355X kernel for this metric / scalar kind combination";
    device_ = device;
    UA_HIP(hipSetDevice(device));
    metric_ = image.metric;
    scalar_ = image.scalar;
    count_present_ = image.count_present;

    const std::uint64_t n = image.size;
    const std::uint32_t m = (std::uint32_t)image.connectivity, m0 = (std::uint32_t)image.connectivity_base;
    const std::uint32_t bpv = (std::uint32_t)image.cols;

    std::uint32_t row_stride = 0;
    row_geometry(bpv, lanes_, row_stride);

    // ---- host pass 1: tape offsets (sequential prefix) and the number of upper-level lists
    std::vector<std::uint64_t> offsets(n + 1);
    std::vector<std::uint32_t> upper_ref(n);
    std::uint64_t offset = 0, lists = 0;
    for (std::uint64_t i = 0; i < n; ++i) {
        offsets[i] = offset;
        const std::int16_t level = image.level(i);
        if (level < 0)
            return "Failed to pull nodes from the stream";
        upper_ref[i] = level ? (std::uint32_t)lists : none_slot_k;
        lists += (std::uint64_t)level;
        offset += image.node_bytes(level);
    }
    offsets[n] = offset;
    if (offset > image.tapes_length)
        return "Failed to pull nodes from the stream";
    if (lists >= none_slot_k)
        return "Too many upper-level lists for 32-bit references";
    upper_lists_ = lists;

    // ---- host pass 2: keys, level-0 rows, upper lists (parallel over nodes)
    std::vector<std::uint64_t> keys(n);
    std::vector<std::uint32_t> nbr0((std::size_t)n * m0, none_slot_k);
    std::vector<std::uint32_t> upper((std::size_t)std::max<std::uint64_t>(lists, 1) * m, none_slot_k);
    std::atomic<bool> corrupt{false}, tombstones{false};
    parallel_ranges(n, [&](std::uint64_t begin, std::uint64_t end) {
        for (std::uint64_t i = begin; i < end; ++i) {
            const std::uint8_t* tape = image.tapes + offsets[i];
            const std::uint64_t key = image_t::load<std::uint64_t>(tape);
            keys[i] = key;
            if (key == free_key_k)
                tombstones.store(true, std::memory_order_relaxed);
            const std::int16_t level = image_t::load<std::int16_t>(tape + 8);
            if (level != image.level(i)) {
                corrupt.store(true);
                return;
            }
            const std::uint8_t* list = tape + 10;
            // level 0: keep the reference's order; a slot that re-appears later in the same list could only ever be
            // seen as "already visited" there (index.hpp:4229), so dropping it preserves the traversal exactly
            std::uint32_t count = image_t::load<std::uint32_t>(list);
            if (count > m0) {
                corrupt.store(true);
                return;
            }
            std::uint32_t* row = nbr0.data() + (std::size_t)i * m0;
            std::uint32_t kept = 0;
            for (std::uint32_t j = 0; j < count; ++j) {
                const std::uint32_t s = image_t::load<std::uint32_t>(list + 4 + 4 * j);
                if (s >= n) {
                    corrupt.store(true);
                    return;
                }
                bool seen = false;
                for (std::uint32_t k = 0; k < kept && !seen; ++k)
                    seen = row[k] == s;
                if (!seen)
                    row[kept++] = s;
            }
            list += 4 + 4 * (std::size_t)m0;
            // upper levels: no visited set is consulted there (index.hpp:3976-4001): keep lists verbatim
            for (std::int16_t l = 1; l <= level; ++l, list += 4 + 4 * (std::size_t)m) {
                count = image_t::load<std::uint32_t>(list);
                if (count > m) {
                    corrupt.store(true);
                    return;
                }
                std::uint32_t* cells = upper.data() + ((std::size_t)upper_ref[i] + (l - 1)) * m;
                for (std::uint32_t j = 0; j < count; ++j) {
                    const std::uint32_t s = image_t::load<std::uint32_t>(list + 4 + 4 * j);
                    if (s >= n || image.level(s) < l) {
                        corrupt.store(true);
                        return;
                    }
                    cells[j] = s;
                }
            }
        }
    });
    if (corrupt.load())
        return "Failed to pull nodes from the stream";
    if (n && image.level(image.entry_slot) < (std::int16_t)image.max_level)
        return "Failed to pull the header from the stream";

    // ---- upload
    release();
    device_ = device;
    const std::size_t vectors_bytes = (std::size_t)n * row_stride;
    auto allocate = [&](void** p, std::size_t bytes) -> hipError_t {
        device_bytes_ += std::max<std::size_t>(bytes, 16);
        return hipMalloc(p, std::max<std::size_t>(bytes, 16));
    };
    device_bytes_ = 0;
    UA_HIP(allocate(&d_vectors_, vectors_bytes));
    UA_HIP(allocate(&d_nbr0_, nbr0.size() * 4));
    UA_HIP(allocate(&d_upper_ref_, upper_ref.size() * 4));
    UA_HIP(allocate(&d_upper_, upper.size() * 4));
    UA_HIP(allocate(&d_keys_, keys.size() * 8));
    if (n) {
        if (const char* e = upload_rows(static_cast<std::uint8_t*>(d_vectors_), row_stride, image.vectors, bpv, bpv, n))
            return e;
        UA_HIP(hipMemcpy(d_nbr0_, nbr0.data(), nbr0.size() * 4, hipMemcpyHostToDevice));
        UA_HIP(hipMemcpy(d_upper_ref_, upper_ref.data(), upper_ref.size() * 4, hipMemcpyHostToDevice));
        UA_HIP(hipMemcpy(d_keys_, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
    }
    UA_HIP(hipMemcpy(d_upper_, upper.data(), upper.size() * 4, hipMemcpyHostToDevice));

    view_.vectors = static_cast<const std::uint8_t*>(d_vectors_);
    view_.nbr0 = static_cast<const std::uint32_t*>(d_nbr0_);
    view_.upper_ref = static_cast<const std::uint32_t*>(d_upper_ref_);
    view_.upper = static_cast<const std::uint32_t*>(d_upper_);
    view_.keys = static_cast<const std::uint64_t*>(d_keys_);
    view_.size = n;
    view_.row_stride = row_stride;
    view_.chunks = row_stride / 16;
    view_.bytes_per_vector = bpv;
    view_.dimensions = (std::uint32_t)image.dimensions;
    view_.m = m;
    view_.m0 = m0;
    view_.max_level = (std::uint32_t)image.max_level;
    view_.entry_slot = (std::uint32_t)image.entry_slot;
    view_.has_tombstones = tombstones.load() ? 1u : 0u;

    hipDeviceProp_t properties;
    UA_HIP(hipGetDeviceProperties(&properties, device));
    compute_units_ = properties.multiProcessorCount > 0 ? properties.multiProcessorCount : 256;
    UA_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    UA_HIP(hipEventCreate(&event_begin_));
    UA_HIP(hipEventCreate(&event_end_));
    return nullptr;
}

const char* snapshot_t::ensure_workspace(std::size_t queries, std::size_t scratch_bytes) {
    if (queries > workspace_queries_) {
        if (d_status_)
            (void)hipFree(d_status_);
        if (d_todo_)
            (void)hipFree(d_todo_);
        if (d_peaks_)
            (void)hipFree(d_peaks_);
        if (h_status_)
            (void)hipHostFree(h_status_);
        d_status_ = d_todo_ = d_peaks_ = h_status_ = nullptr;
        workspace_queries_ = 0;
        UA_HIP(hipMalloc((void**)&d_status_, queries * 4));
        UA_HIP(hipMalloc((void**)&d_todo_, queries * 4));
        UA_HIP(hipMalloc((void**)&d_peaks_, queries * 8));
        if (!d_queue_)
            UA_HIP(hipMalloc((void**)&d_queue_, 256));
        UA_HIP(hipHostMalloc((void**)&h_status_, queries * 4, hipHostMallocDefault));
        workspace_queries_ = queries;
    }
    if (scratch_bytes > scratch_bytes_) {
        if (d_scratch_)
            (void)hipFree(d_scratch_);
        d_scratch_ = nullptr;
        scratch_bytes_ = 0;
        UA_HIP(hipMalloc((void**)&d_scratch_, scratch_bytes));
        scratch_bytes_ = scratch_bytes;
    }
    return nullptr;
}

const char* snapshot_t::ensure_staging(std::size_t query_bytes, std::size_t count, std::size_t wanted) {
    // queries | keys | distances | counts | visited | computed, each 256-byte aligned
    auto pad = [](std::size_t b) { return (b + 255) & ~(std::size_t)255; };
    const std::size_t need = pad(query_bytes * count) + pad(count * wanted * 8) + pad(count * wanted * 4) + 3 * pad(count * 8);
    if (need > stage_bytes_) {
        if (d_stage_)
            (void)hipFree(d_stage_);
        d_stage_ = nullptr;
        stage_bytes_ = 0;
        UA_HIP(hipMalloc((void**)&d_stage_, need));
        stage_bytes_ = need;
    }
    return nullptr;
}

static hipError_t launch_search(metric_kind_t metric, scalar_kind_t scalar, const launch_params_t& p,
                                const snapshot_view_t& view, const search_args_t& args) {
    metric = kernel_metric(metric);
#define UA_PAIR(m, sc, name)                                                                                           \
    if (metric == m && scalar == sc)                                                                                   \
        return launch_search_##name(p, view, args);
    USEARCH_AMD_FOR_EACH_PAIR(UA_PAIR)
#undef UA_PAIR
    return hipErrorInvalidValue;
}

static hipError_t launch_distances(metric_kind_t metric, scalar_kind_t scalar, const distances_params_t& p,
                                   const snapshot_view_t& view) {
    metric = kernel_metric(metric);
#define UA_PAIR(m, sc, name)                                                                                           \
    if (metric == m && scalar == sc)                                                                                   \
        return launch_distances_##name(p, view);
    USEARCH_AMD_FOR_EACH_PAIR(UA_PAIR)
#undef UA_PAIR
    return hipErrorInvalidValue;
}

/// Fills the outputs of queries that cannot produce anything (empty index): count 0, key 0 / signalling NaN padding.
__global__ void fill_empty_kernel(std::uint64_t* keys, std::uint32_t* distance_bits, std::uint64_t* counts,
                                  std::uint64_t* visited, std::uint64_t* computed, std::uint64_t queries,
                                  std::uint64_t wanted) {
    const std::uint64_t i = blockIdx.x * (std::uint64_t)blockDim.x + threadIdx.x;
    if (i < queries * wanted)
        keys[i] = 0, distance_bits[i] = signaling_nan_bits_k;
    if (i < queries)
        counts[i] = 0, visited[i] = 0, computed[i] = 0;
}

const char* snapshot_t::search_device(const void* queries, std::size_t count, std::size_t stride_bytes,
                                      std::size_t wanted, std::size_t expansion, std::uint64_t* keys,
                                      float* distances, std::uint64_t* counts, std::uint64_t* visited,
                                      std::uint64_t* computed, hipStream_t stream, const search_tuning_t& tuning,
                                      search_stats_t* stats, bool timed, const search_extras_t* extras) {
    if (stats)
        *stats = search_stats_t{};
    if (!count || !wanted) // index.hpp:3025-3027: nothing wanted, nothing found
        return nullptr;
    if (count >= none_slot_k || wanted >= (1u << 24))
        return "Batch is too large";
    std::lock_guard<std::mutex> lock(mutex_);
    UA_HIP(hipSetDevice(device_));
    if (!stream)
        stream = stream_;

    if (view_.size == 0) { // index.hpp:3034-3037
        const std::uint64_t cells = std::max<std::uint64_t>(count * wanted, count);
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, keys,
                           reinterpret_cast<std::uint32_t*>(distances), counts, visited, computed,
                           (std::uint64_t)count, (std::uint64_t)wanted);
        UA_HIP(hipGetLastError());
        UA_HIP(hipStreamSynchronize(stream));
        return nullptr;
    }

    if (!expansion)
        expansion = default_expansion_search_k;
    const std::uint32_t ef = (std::uint32_t)std::max(expansion, wanted); // index.hpp:3052

    // ---- scratch sizing, from measurements with the reference's own traversal (DESIGN.md "scratch sizing"): the frontier
    // peaks at 2.4-3.9 × ef and the visited set ends at 18-30 × ef entries; outliers go through the retry ladder below.
    const std::uint32_t query_lds = view_.chunks * (query_chunk_bytes_of(scalar_));
    const std::uint32_t lds_budget = (std::uint32_t)env_size("USEARCH_AMD_LDS_BUDGET", 160 * 1024);
    std::uint32_t hash_cap = tuning.hash_cap ? tuning.hash_cap : (std::uint32_t)env_size("USEARCH_AMD_HASH_CAP", 0);
    if (!hash_cap)
        hash_cap = std::max<std::uint32_t>(1024, ef * 48);
    hash_cap = pow2_ceil(hash_cap);
    std::uint32_t next_cap = tuning.next_cap ? tuning.next_cap : (std::uint32_t)env_size("USEARCH_AMD_NEXT_CAP", 0);
    if (!next_cap)
        next_cap = std::max<std::uint32_t>(512, ef * 3 + 256);
    // never larger than the index could possibly need
    hash_cap = std::min<std::uint32_t>(hash_cap, pow2_ceil((std::uint32_t)std::min<std::uint64_t>(view_.size * 2 + 128, 1u << 30)));
    next_cap = (std::uint32_t)std::min<std::uint64_t>(next_cap, view_.size + 64);
    // register/latency trade-off of the kernel (kernels.hpp kernel_variant_t); rows shorter than 8 chunks per lane have
    // nothing to unroll
    const std::uint32_t chunks_per_lane = view_.chunks / lanes_;
    std::uint32_t variant_request = tuning.variant ? tuning.variant : (std::uint32_t)env_size("USEARCH_AMD_VARIANT", 0);
    int variant = variant_u4_w4_k;
    if (lanes_ == 8 && chunks_per_lane >= 8 && all_kernel_builds(kernel_metric(metric_), scalar_)) {
        // measured on 10M x 768 f16 (profiles/): a whole row per round trip (12 loads per lane, 8 waves per CU) beats 8 loads
        // at 12 waves per CU at every expansion — the traversal is latency-bound, fewer round trips per hop win
        variant = chunks_per_lane >= 12 ? variant_u12_w2_k : variant_u8_w3_k;
    }
    if (variant_request && variant_request - 1 <= (std::uint32_t)variant_u12_w2_k && lanes_ == 8 &&
        all_kernel_builds(kernel_metric(metric_), scalar_))
        variant = (int)variant_request - 1;
    const std::uint32_t top_entries_hint = ef <= 64 ? 1u : ef <= 256 ? 4u : ef <= 512 ? 8u : ef <= 1024 ? 16u : 0u;
    const std::uint32_t variant_waves_per_cu = 4u * (std::uint32_t)kernel_waves(variant, (int)top_entries_hint);
    const std::uint32_t waves_cap = tuning.waves_per_cu ? tuning.waves_per_cu
                                                        : (std::uint32_t)env_size("USEARCH_AMD_WAVES_PER_CU", 16);
    std::uint32_t mode_request = tuning.mode ? tuning.mode : (std::uint32_t)env_size("USEARCH_AMD_MODE", 0);

    // `top` lives in registers (1 / 4 / 8 / 16 entries per lane) while the expansion allows it
    const bool top_in_memory = tuning.top_in_memory || env_size("USEARCH_AMD_TOP_IN_MEMORY", 0) != 0;
    const std::uint32_t entries_per_lane = top_in_memory ? 0u : ef <= 64 ? 1u : ef <= 256 ? 4u : ef <= 512 ? 8u : ef <= 1024 ? 16u : 0u;
    auto lds_bytes_for = [&](int mode, std::uint32_t cap_next, std::uint32_t cap_hash) -> std::uint64_t {
        if (mode == scratch_global_k)
            return query_lds;
        const scratch_layout_t l = scratch_layout(entries_per_lane ? 0 : ef, cap_next,
                                                  mode == scratch_lds_k ? (std::uint64_t)cap_hash * 4 : 0);
        return query_lds + l.total;
    };
    auto waves_for = [&](std::uint64_t lds_bytes) -> std::uint32_t {
        const std::uint64_t granule = (lds_bytes + 1023) / 1024 * 1024; // LDS is allocated in coarse granules
        return (std::uint32_t)std::max<std::uint64_t>(
            1, std::min<std::uint64_t>(std::min(waves_cap, variant_waves_per_cu), lds_budget / std::max<std::uint64_t>(granule, 1)));
    };
    // the frontier's default room has 256 cells of slack; when giving up to half of it back lets one more wave share the
    // compute unit's LDS, do (the retry ladder still catches a query that would have needed them)
    const bool default_next_cap = !tuning.next_cap && !env_size("USEARCH_AMD_NEXT_CAP", 0);
    if (default_next_cap && mode_request != 1 && mode_request != 3) {
        const std::uint32_t now = waves_for(lds_bytes_for(scratch_hash_k, next_cap, hash_cap));
        if (now < std::min(waves_cap, variant_waves_per_cu)) {
            const std::uint64_t room = lds_budget / (now + 1) / 1024 * 1024;
            const std::uint64_t fixed = lds_bytes_for(scratch_hash_k, 0, hash_cap);
            if (room > fixed) {
                const std::uint32_t trimmed = (std::uint32_t)((room - fixed) / 8 / 2 * 2);
                if (trimmed < next_cap && trimmed + 128 >= next_cap)
                    next_cap = trimmed;
            }
        }
    }
    // auto: keep the visited set in LDS only while that still leaves 8 waves per CU; otherwise move it to the global hash
    int mode = mode_request == 1 ? scratch_lds_k : mode_request == 2 ? scratch_hash_k : mode_request == 3 ? scratch_global_k
               : (waves_for(lds_bytes_for(scratch_lds_k, next_cap, hash_cap)) >= 8 ? scratch_lds_k : scratch_hash_k);

    if (const char* e = ensure_workspace(count, 0))
        return e;

    search_args_t args{};
    args.queries = static_cast<const std::uint8_t*>(queries);
    args.query_stride = stride_bytes;
    args.wanted = (std::uint32_t)wanted;
    args.ef = ef;
    args.keys = keys;
    args.distances = distances;
    args.counts = counts;
    args.visited = visited;
    args.computed = computed;
    args.status = d_status_;
    args.queue = d_queue_;
    args.peaks = d_peaks_;
    if (extras) {
        args.query_ids = extras->query_ids;
        args.beam_level = extras->beam_level;
        args.emit_slots = extras->emit_slots ? 1u : 0u;
        args.descent_only = extras->descent_only ? 1u : 0u;
        args.allow_bits = extras->allow_bits;
    }

    launch_params_t params{};
    params.metric = metric_;
    params.lanes = lanes_;
    params.variant = variant;
    params.stream = stream;

    // diagnostic: per-phase shader-clock ticks of the search kernel, printed to stderr (USEARCH_AMD_PHASES=1)
    const bool want_phases = env_size("USEARCH_AMD_PHASES", 0) != 0;
    if (want_phases) {
        args.phases = reinterpret_cast<unsigned long long*>(d_queue_) + 8; // d_queue_ is a 256-byte block
        UA_HIP(hipMemsetAsync(args.phases, 0, 128, stream));
    }

    float total_ms = 0.f;
    auto timed_launch = [&](const launch_params_t& p, const search_args_t& a) -> const char* {
        UA_HIP(hipMemsetAsync(d_queue_, 0, 4, stream));
        if (timed)
            UA_HIP(hipEventRecord(event_begin_, stream));
        UA_HIP(launch_search(metric_, scalar_, p, view_, a));
        if (timed) {
            UA_HIP(hipEventRecord(event_end_, stream));
            UA_HIP(hipEventSynchronize(event_end_));
            float ms = 0.f;
            UA_HIP(hipEventElapsedTime(&ms, event_begin_, event_end_));
            total_ms += ms;
        }
        return nullptr;
    };
    /// Collects the indices of overflowed queries (among `previous`, or all) and uploads them as the next todo list.
    auto collect_overflow = [&](const std::vector<std::uint32_t>* previous, std::vector<std::uint32_t>& todo) -> const char* {
        UA_HIP(hipMemcpyAsync(h_status_, d_status_, count * 4, hipMemcpyDeviceToHost, stream));
        UA_HIP(hipStreamSynchronize(stream));
        todo.clear();
        if (previous) {
            for (std::uint32_t q : *previous)
                if (h_status_[q] == status_overflow_k)
                    todo.push_back(q);
        } else {
            for (std::uint32_t q = 0; q < count; ++q)
                if (h_status_[q] == status_overflow_k)
                    todo.push_back(q);
        }
        if (!todo.empty())
            UA_HIP(hipMemcpyAsync(d_todo_, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, stream));
        return nullptr;
    };

    std::vector<std::uint32_t> todo, todo_next;
    std::uint32_t passes = 0;
    bool have_todo = false;

    // ---- pass 1 (+2): persistent waves, heaps in LDS; the second attempt moves the visited set to the global hash and
    //      gives both structures 4× the room
    if (mode != scratch_global_k) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (lds_bytes_for(mode, next_cap, hash_cap) > lds_budget) {
                if (mode == scratch_lds_k)
                    mode = scratch_hash_k;
                while (next_cap > 64 && lds_bytes_for(mode, next_cap, hash_cap) > lds_budget)
                    next_cap /= 2;
                if (lds_bytes_for(mode, next_cap, hash_cap) > lds_budget)
                    break; // `top` alone does not fit LDS: straight to the global fallback
            }
            const std::uint64_t lds_bytes = lds_bytes_for(mode, next_cap, hash_cap);
            const std::uint32_t pending = have_todo ? (std::uint32_t)todo.size() : (std::uint32_t)count;
            const std::uint32_t grid = (std::uint32_t)std::min<std::uint64_t>(pending, (std::uint64_t)waves_for(lds_bytes) * compute_units_);
            const std::uint64_t slab = mode == scratch_hash_k ? (std::uint64_t)hash_cap * 4 : 0;
            if (const char* e = ensure_workspace(count, slab * grid))
                return e;
            args.hash_cap = hash_cap;
            args.next_cap = next_cap;
            args.todo = have_todo ? d_todo_ : nullptr;
            args.count = pending;
            args.scratch = d_scratch_;
            args.scratch_stride = slab;
            params.mode = mode;
            params.entries_per_lane = entries_per_lane;
            params.grid = grid;
            params.lds_bytes = (std::uint32_t)lds_bytes;
            if (stats && attempt == 1)
                stats->retried_lds = pending;
            if (const char* e = timed_launch(params, args))
                return e;
            ++passes;
            if (const char* e = collect_overflow(have_todo ? &todo : nullptr, todo_next))
                return e;
            todo.swap(todo_next);
            have_todo = true;
            if (todo.empty() || attempt == 1)
                break;
            mode = scratch_hash_k;
            hash_cap = std::min<std::uint32_t>(hash_cap * 4, pow2_ceil((std::uint32_t)std::min<std::uint64_t>(view_.size * 2 + 128, 1u << 30)));
            next_cap = (std::uint32_t)std::min<std::uint64_t>((std::uint64_t)next_cap * 4, view_.size + 64);
            while (next_cap > 64 && lds_bytes_for(mode, next_cap, hash_cap) > lds_budget)
                next_cap = next_cap * 3 / 4;
        }
    }

    // ---- pass 3: global-memory scratch — exact sizes (one bit per slot, one frontier cell per slot), cannot overflow
    if (mode == scratch_global_k || (have_todo && !todo.empty())) {
        if (!have_todo) {
            todo.resize(count);
            for (std::uint32_t q = 0; q < count; ++q)
                todo[q] = q;
        }
        if (stats)
            stats->retried_global = (std::uint32_t)todo.size();
        const std::uint64_t bitmap_bytes = ((view_.size + 31) / 32) * 4;
        const std::uint32_t frontier = (std::uint32_t)std::min<std::uint64_t>(view_.size + 64, 0xFFFFFFF0u);
        const scratch_layout_t layout = scratch_layout(ef, frontier, bitmap_bytes);
        const std::size_t slab = (layout.total + 255) & ~(std::size_t)255;
        const std::size_t budget = env_size("USEARCH_AMD_GLOBAL_SCRATCH_BYTES", (std::size_t)2 << 30);
        const std::size_t waves = std::max<std::size_t>(1, std::min<std::size_t>(todo.size(), budget / slab));
        if (const char* e = ensure_workspace(count, waves * slab))
            return e;
        for (std::size_t begin = 0; begin < todo.size(); begin += waves) {
            const std::size_t chunk = std::min(waves, todo.size() - begin);
            UA_HIP(hipMemcpyAsync(d_todo_, todo.data() + begin, chunk * 4, hipMemcpyHostToDevice, stream));
            // only the bitmaps need zeroing
            UA_HIP(hipMemset2DAsync(d_scratch_ + layout.visits, slab, 0, bitmap_bytes, chunk, stream));
            args.hash_cap = 0;
            args.next_cap = frontier;
            args.todo = d_todo_;
            args.count = (std::uint32_t)chunk;
            args.scratch = d_scratch_;
            args.scratch_stride = slab;
            params.mode = scratch_global_k;
            params.entries_per_lane = 0;
            params.grid = (std::uint32_t)chunk;
            params.lds_bytes = query_lds;
            if (const char* e = timed_launch(params, args))
                return e;
            ++passes;
            UA_HIP(hipStreamSynchronize(stream));
        }
        UA_HIP(hipMemcpyAsync(h_status_, d_status_, count * 4, hipMemcpyDeviceToHost, stream));
        UA_HIP(hipStreamSynchronize(stream));
        for (std::uint32_t q : todo)
            if (h_status_[q] != status_done_k)
                return "Search scratch overflow in the global-memory pass";
    }
    if (want_phases) {
        unsigned long long ticks[16] = {0};
        UA_HIP(hipMemcpy(ticks, args.phases, 128, hipMemcpyDeviceToHost));
        double total = 0;
        for (int i = 0; i < 6; ++i)
            total += (double)ticks[i];
        std::fprintf(stderr, "[usearch_amd] phases ef=%u grid=%u: setup %.1f%% pop+list %.1f%% visited %.1f%% distances %.1f%% "
                             "commit %.1f%% [heap push %.1f%% top insert %.1f%%, %llu candidates rechecked] dump %.1f%% (%.3g ticks); "
                             "frontier pushes %llu, lists ready ahead %llu\n",
                     ef, params.grid, 100 * ticks[0] / total, 100 * ticks[1] / total, 100 * ticks[2] / total,
                     100 * ticks[3] / total, 100 * ticks[4] / total, 100 * ticks[8] / total, 100 * ticks[9] / total, ticks[10],
                     100 * ticks[5] / total, total, ticks[6], ticks[7]);
    }
    if (stats) {
        stats->passes = passes;
        stats->kernel_ms = total_ms;
        stats->mode = (std::uint32_t)mode + 1;
        stats->grid = params.grid;
        stats->lds_bytes = params.lds_bytes;
    }
    last_count_ = count;
    return nullptr;
}

const char* snapshot_t::last_peaks(std::uint32_t* out, std::size_t queries) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (!d_peaks_ || queries > last_count_)
        return "No telemetry for that many queries";
    UA_HIP(hipSetDevice(device_));
    UA_HIP(hipMemcpy(out, d_peaks_, queries * 8, hipMemcpyDeviceToHost));
    return nullptr;
}

const char* snapshot_t::search_host(const void* queries, scalar_kind_t query_kind, std::size_t count,
                                    std::size_t stride_bytes, std::size_t wanted, std::size_t expansion,
                                    std::uint64_t* keys, float* distances, std::uint64_t* counts,
                                    std::uint64_t* visited, std::uint64_t* computed, const search_tuning_t& tuning,
                                    search_stats_t* stats, const std::uint32_t* allow_bits_host,
                                    const search_extras_t* more) {
    if (stats)
        *stats = search_stats_t{};
    if (!count || !wanted)
        return nullptr;
    const std::size_t bpv = view_.bytes_per_vector ? view_.bytes_per_vector : bytes_per_vector(scalar_, view_.dimensions);
    const std::size_t dims = view_.dimensions;

    // cast (or gather strided rows) into a dense host block in the storage kind — index_dense.hpp:2058-2064
    std::vector<std::uint8_t> dense;
    const std::uint8_t* source = static_cast<const std::uint8_t*>(queries);
    if (query_kind != scalar_) {
        const std::size_t query_bytes = bytes_per_vector(query_kind, dims);
        if (query_bytes == 0)
            return "Unsupported query scalar kind";
        if (count > 1 && stride_bytes < query_bytes)
            return "Query stride is smaller than one query";
        dense.assign(count * bpv, 0);
        parallel_ranges(count, [&](std::uint64_t begin, std::uint64_t end) {
            for (std::uint64_t q = begin; q < end; ++q)
                cast_vector(query_kind, scalar_, source + q * stride_bytes, dims, dense.data() + q * bpv);
        });
        source = dense.data();
    } else if (count > 1 && stride_bytes != bpv) {
        if (stride_bytes < bpv)
            return "Query stride is smaller than one query";
        dense.resize(count * bpv);
        parallel_ranges(count, [&](std::uint64_t begin, std::uint64_t end) {
            for (std::uint64_t q = begin; q < end; ++q)
                std::memcpy(dense.data() + q * bpv, source + q * stride_bytes, bpv);
        });
        source = dense.data();
    }

    std::lock_guard<std::mutex> host_lock(host_mutex_); // the staging block below is shared
    UA_HIP(hipSetDevice(device_));
    if (const char* e = ensure_staging(bpv, count, wanted))
        return e;
    auto pad = [](std::size_t b) { return (b + 255) & ~(std::size_t)255; };
    std::uint8_t* d_queries = d_stage_;
    std::uint64_t* d_keys = reinterpret_cast<std::uint64_t*>(d_queries + pad(bpv * count));
    float* d_distances = reinterpret_cast<float*>(reinterpret_cast<std::uint8_t*>(d_keys) + pad(count * wanted * 8));
    std::uint64_t* d_counts = reinterpret_cast<std::uint64_t*>(reinterpret_cast<std::uint8_t*>(d_distances) + pad(count * wanted * 4));
    std::uint64_t* d_visited = reinterpret_cast<std::uint64_t*>(reinterpret_cast<std::uint8_t*>(d_counts) + pad(count * 8));
    std::uint64_t* d_computed = reinterpret_cast<std::uint64_t*>(reinterpret_cast<std::uint8_t*>(d_visited) + pad(count * 8));

    UA_HIP(hipMemcpy(d_queries, source, bpv * count, hipMemcpyHostToDevice));
    search_extras_t extras = more ? *more : search_extras_t{};
    std::uint32_t* d_allow = nullptr;
    if (allow_bits_host && view_.size) {
        const std::size_t words = (view_.size + 31) / 32;
        UA_HIP(hipMalloc((void**)&d_allow, words * 4));
        if (hipMemcpy(d_allow, allow_bits_host, words * 4, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_allow);
            return "Failed to upload the predicate bitmap";
        }
        extras.allow_bits = d_allow;
    }
    const char* search_error = search_device(d_queries, count, bpv, wanted, expansion, d_keys, d_distances, d_counts,
                                             d_visited, d_computed, nullptr, tuning, stats, false, &extras);
    if (d_allow)
        (void)hipFree(d_allow);
    if (search_error)
        return search_error;
    if (keys)
        UA_HIP(hipMemcpy(keys, d_keys, count * wanted * 8, hipMemcpyDeviceToHost));
    if (distances)
        UA_HIP(hipMemcpy(distances, d_distances, count * wanted * 4, hipMemcpyDeviceToHost));
    if (counts)
        UA_HIP(hipMemcpy(counts, d_counts, count * 8, hipMemcpyDeviceToHost));
    if (visited)
        UA_HIP(hipMemcpy(visited, d_visited, count * 8, hipMemcpyDeviceToHost));
    if (computed)
        UA_HIP(hipMemcpy(computed, d_computed, count * 8, hipMemcpyDeviceToHost));
    return nullptr;
}

const char* snapshot_t::cluster_host(const void* queries, scalar_kind_t query_kind, std::size_t count,
                                     std::size_t stride_bytes, std::size_t level, std::uint64_t* keys, float* distances,
                                     std::uint64_t* visited, std::uint64_t* computed) {
    // index_gt::cluster, index.hpp:3112-3114: search_for_one_ from the top level down to `level` (target level - 1, or 0)
    search_extras_t extras;
    extras.descent_only = true;
    extras.beam_level = (std::uint32_t)std::min<std::size_t>(level ? level - 1 : 0, 0xFFFFu);
    search_stats_t stats;
    return search_host(queries, query_kind, count, stride_bytes, 1, 1, keys, distances, nullptr, visited, computed,
                       search_tuning_t{}, &stats, nullptr, &extras);
}

static hipError_t launch_exact(metric_kind_t metric, scalar_kind_t scalar, const exact_params_t& p,
                               const snapshot_view_t& view) {
    metric = kernel_metric(metric);
#define UA_PAIR(m, sc, name)                                                                                           \
    if (metric == m && scalar == sc)                                                                                   \
        return launch_exact_##name(p, view);
    USEARCH_AMD_FOR_EACH_PAIR(UA_PAIR)
#undef UA_PAIR
    return hipErrorInvalidValue;
}

const char* exact_search_device(metric_kind_t metric, scalar_kind_t scalar, std::uint32_t lanes,
                                const snapshot_view_t& view, const void* queries, std::size_t count,
                                std::size_t stride_bytes, std::size_t wanted, bool map_keys, std::uint64_t* keys,
                                float* distances, std::uint64_t* counts, hipStream_t stream, float* kernel_ms) {
    if (kernel_ms)
        *kernel_ms = 0.f;
    if (!count || !wanted)
        return nullptr;
    if (count >= none_slot_k || wanted > 4096)
        return "Batch is too large";
    if (view.size == 0) {
        const std::uint64_t cells = std::max<std::uint64_t>(count * wanted, count);
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, keys,
                           reinterpret_cast<std::uint32_t*>(distances), counts, counts, counts, (std::uint64_t)count,
                           (std::uint64_t)wanted);
        UA_HIP(hipGetLastError());
        UA_HIP(hipStreamSynchronize(stream));
        return nullptr;
    }
    // enough (query, partition) waves to fill the chip, few enough candidates for one merge wave per query
    std::uint64_t partitions = std::max<std::uint64_t>(1, (8192 + count - 1) / count);
    partitions = std::min<std::uint64_t>(partitions, std::max<std::uint64_t>(1, 8192 / wanted));
    partitions = std::min<std::uint64_t>(partitions, std::max<std::uint64_t>(1, view.size / 256));
    partitions = std::min<std::uint64_t>(partitions, 65535);
    const std::uint64_t rows_per_partition = (view.size + partitions - 1) / partitions;
    partitions = (view.size + rows_per_partition - 1) / rows_per_partition;

    float* partial_distances = nullptr;
    std::uint64_t *partial_keys = nullptr, *partial_counts = nullptr;
    hipEvent_t begin = nullptr, end = nullptr;
    const char* error = nullptr;
    hipError_t e = hipMalloc((void**)&partial_distances, partitions * count * wanted * 4);
    if (e == hipSuccess)
        e = hipMalloc((void**)&partial_keys, partitions * count * wanted * 8);
    if (e == hipSuccess)
        e = hipMalloc((void**)&partial_counts, partitions * count * 8);
    if (e == hipSuccess && kernel_ms) {
        e = hipEventCreate(&begin);
        if (e == hipSuccess)
            e = hipEventCreate(&end);
        if (e == hipSuccess)
            e = hipEventRecord(begin, stream);
    }
    if (e == hipSuccess) {
        exact_params_t p{};
        p.lanes = lanes;
        p.lds_bytes = view.chunks * (query_chunk_bytes_of(scalar)) + 512 + (std::uint32_t)wanted * 8 + 16;
        p.stream = stream;
        p.queries = static_cast<const std::uint8_t*>(queries);
        p.query_stride = stride_bytes;
        p.query_count = (std::uint32_t)count;
        p.wanted = (std::uint32_t)wanted;
        p.partitions = (std::uint32_t)partitions;
        p.rows_per_partition = rows_per_partition;
        p.map_keys = map_keys ? 1u : 0u;
        p.out_distances = partial_distances;
        p.out_keys = partial_keys;
        p.out_counts = partial_counts;
        e = launch_exact(metric, scalar, p, view);
    }
    if (e == hipSuccess && kernel_ms)
        e = hipEventRecord(end, stream);
    if (e == hipSuccess)
        error = merge_shards_device(partial_distances, partial_keys, partial_counts, partitions, count, wanted, distances,
                                    keys, counts, stream, false);
    if (e == hipSuccess && !error && kernel_ms)
        e = hipEventElapsedTime(kernel_ms, begin, end);
    if (e != hipSuccess)
        error = hip_message(e);
    for (void* p : {(void*)partial_distances, (void*)partial_keys, (void*)partial_counts})
        if (p)
            (void)hipFree(p);
    if (begin)
        (void)hipEventDestroy(begin);
    if (end)
        (void)hipEventDestroy(end);
    return error;
}

const char* snapshot_t::exact_device(const void* queries, std::size_t count, std::size_t stride_bytes,
                                     std::size_t wanted, std::uint64_t* keys, float* distances, std::uint64_t* counts,
                                     hipStream_t stream, float* kernel_ms) {
    std::lock_guard<std::mutex> lock(mutex_);
    UA_HIP(hipSetDevice(device_));
    return exact_search_device(metric_, scalar_, lanes_, view_, queries, count, stride_bytes, wanted, true, keys,
                               distances, counts, stream ? stream : stream_, kernel_ms);
}

const char* snapshot_t::exact_host(const void* queries, scalar_kind_t query_kind, std::size_t count,
                                   std::size_t stride_bytes, std::size_t wanted, std::uint64_t* keys, float* distances,
                                   std::uint64_t* counts, float* kernel_ms) {
    if (!count || !wanted)
        return nullptr;
    const std::size_t bpv = view_.bytes_per_vector, dims = view_.dimensions;
    std::vector<std::uint8_t> dense(count * bpv, 0);
    const std::uint8_t* source = static_cast<const std::uint8_t*>(queries);
    if (query_kind != scalar_ && bytes_per_vector(query_kind, dims) == 0)
        return "Unsupported query scalar kind";
    parallel_ranges(count, [&](std::uint64_t begin, std::uint64_t end) {
        for (std::uint64_t q = begin; q < end; ++q)
            if (!cast_vector(query_kind, scalar_, source + q * stride_bytes, dims, dense.data() + q * bpv))
                std::memcpy(dense.data() + q * bpv, source + q * stride_bytes, bpv);
    });
    std::lock_guard<std::mutex> host_lock(host_mutex_);
    UA_HIP(hipSetDevice(device_));
    if (const char* e = ensure_staging(bpv, count, wanted))
        return e;
    auto pad = [](std::size_t b) { return (b + 255) & ~(std::size_t)255; };
    std::uint8_t* d_queries = d_stage_;
    std::uint64_t* d_keys = reinterpret_cast<std::uint64_t*>(d_queries + pad(bpv * count));
    float* d_distances = reinterpret_cast<float*>(reinterpret_cast<std::uint8_t*>(d_keys) + pad(count * wanted * 8));
    std::uint64_t* d_counts = reinterpret_cast<std::uint64_t*>(reinterpret_cast<std::uint8_t*>(d_distances) + pad(count * wanted * 4));
    UA_HIP(hipMemcpy(d_queries, dense.data(), bpv * count, hipMemcpyHostToDevice));
    if (const char* e = exact_device(d_queries, count, bpv, wanted, d_keys, d_distances, d_counts, nullptr, kernel_ms))
        return e;
    if (keys)
        UA_HIP(hipMemcpy(keys, d_keys, count * wanted * 8, hipMemcpyDeviceToHost));
    if (distances)
        UA_HIP(hipMemcpy(distances, d_distances, count * wanted * 4, hipMemcpyDeviceToHost));
    if (counts)
        UA_HIP(hipMemcpy(counts, d_counts, count * 8, hipMemcpyDeviceToHost));
    return nullptr;
}

const char* exact_search_dataset_host(metric_kind_t metric, scalar_kind_t scalar, std::size_t dimensions,
                                      const void* dataset, std::size_t dataset_count, std::size_t dataset_stride,
                                      const void* queries, std::size_t queries_count, std::size_t queries_stride,
                                      std::size_t wanted, std::uint64_t* keys, std::size_t keys_stride,
                                      float* distances, std::size_t distances_stride) {
    if (!kernel_available(metric, scalar))
        return "No MI355X kernel for this metric / scalar kind combination";
    if (!queries_count || !wanted)
        return nullptr;
    if (dataset_count >= none_slot_k)
        return "Dataset is too large for 32-bit offsets";
    const std::size_t bpv = bytes_per_vector(scalar, dimensions);
    if (dataset_stride < bpv || queries_stride < bpv)
        return "Stride is smaller than one vector";
    std::uint32_t lanes = 1, row_stride = 16;
    row_geometry(bpv, lanes, row_stride);
    std::uint8_t *d_rows = nullptr, *d_queries = nullptr;
    std::uint64_t *d_keys = nullptr, *d_counts = nullptr;
    float* d_distances = nullptr;
    const char* error = nullptr;
    hipError_t e = hipMalloc((void**)&d_rows, std::max<std::size_t>(dataset_count * row_stride, 16));
    if (e == hipSuccess)
        e = hipMalloc((void**)&d_queries, queries_count * bpv);
    if (e == hipSuccess)
        e = hipMalloc((void**)&d_keys, queries_count * wanted * 8);
    if (e == hipSuccess)
        e = hipMalloc((void**)&d_distances, queries_count * wanted * 4);
    if (e == hipSuccess)
        e = hipMalloc((void**)&d_counts, queries_count * 8);
    if (e == hipSuccess)
        error = upload_rows(d_rows, row_stride, static_cast<const std::uint8_t*>(dataset), dataset_stride, bpv, dataset_count);
    if (e == hipSuccess && !error)
        error = upload_rows(d_queries, (std::uint32_t)bpv, static_cast<const std::uint8_t*>(queries), queries_stride, bpv,
                            queries_count);
    std::vector<std::uint64_t> host_keys(queries_count * wanted);
    std::vector<float> host_distances(queries_count * wanted);
    if (e == hipSuccess && !error) {
        snapshot_view_t view{};
        view.vectors = d_rows;
        view.size = dataset_count;
        view.row_stride = row_stride;
        view.chunks = row_stride / 16;
        view.bytes_per_vector = (std::uint32_t)bpv;
        view.dimensions = (std::uint32_t)dimensions;
        error = exact_search_device(metric, scalar, lanes, view, d_queries, queries_count, bpv, wanted, false, d_keys,
                                    d_distances, d_counts, nullptr, nullptr);
    }
    if (e == hipSuccess && !error) {
        e = hipMemcpy(host_keys.data(), d_keys, host_keys.size() * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy(host_distances.data(), d_distances, host_distances.size() * 4, hipMemcpyDeviceToHost);
    }
    for (void* p : {(void*)d_rows, (void*)d_queries, (void*)d_keys, (void*)d_distances, (void*)d_counts})
        if (p)
            (void)hipFree(p);
    if (e != hipSuccess)
        return hip_message(e);
    if (error)
        return error;
    for (std::size_t q = 0; q < queries_count; ++q) {
        std::memcpy(reinterpret_cast<std::uint8_t*>(keys) + q * keys_stride, host_keys.data() + q * wanted, wanted * 8);
        std::memcpy(reinterpret_cast<std::uint8_t*>(distances) + q * distances_stride, host_distances.data() + q * wanted,
                    wanted * 4);
    }
    return nullptr;
}

const char* snapshot_t::distances_host(const void* queries, std::size_t count, std::size_t stride_bytes,
                                       const std::uint32_t* slots, std::size_t slots_per_query, float* out) {
    if (!count || !slots_per_query)
        return nullptr;
    std::lock_guard<std::mutex> lock(mutex_);
    UA_HIP(hipSetDevice(device_));
    const std::size_t bpv = view_.bytes_per_vector;
    std::uint8_t* d_queries = nullptr;
    std::uint32_t* d_slots = nullptr;
    float* d_out = nullptr;
    UA_HIP(hipMalloc((void**)&d_queries, bpv * count));
    UA_HIP(hipMalloc((void**)&d_slots, count * slots_per_query * 4));
    UA_HIP(hipMalloc((void**)&d_out, count * slots_per_query * 4));
    const char* error = nullptr;
    do {
        std::vector<std::uint8_t> dense(count * bpv);
        for (std::size_t q = 0; q < count; ++q)
            std::memcpy(dense.data() + q * bpv, static_cast<const std::uint8_t*>(queries) + q * stride_bytes, bpv);
        hipError_t e = hipMemcpy(d_queries, dense.data(), bpv * count, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(d_slots, slots, count * slots_per_query * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            error = hip_message(e);
            break;
        }
        distances_params_t p{};
        p.metric = metric_;
        p.lanes = lanes_;
        p.lds_bytes = view_.chunks * (query_chunk_bytes_of(scalar_)) + 512;
        p.stream = stream_;
        p.queries = d_queries;
        p.query_stride = bpv;
        p.slots = d_slots;
        p.slots_per_query = (std::uint32_t)slots_per_query;
        p.count = (std::uint32_t)count;
        p.out = d_out;
        if (e == hipSuccess)
            e = hipEventRecord(event_begin_, stream_);
        if (e == hipSuccess)
            e = launch_distances(metric_, scalar_, p, view_);
        if (e == hipSuccess)
            e = hipEventRecord(event_end_, stream_);
        if (e == hipSuccess)
            e = hipEventSynchronize(event_end_);
        if (e == hipSuccess)
            e = hipEventElapsedTime(&last_distances_ms_, event_begin_, event_end_);
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream_);
        if (e == hipSuccess)
            e = hipMemcpy(out, d_out, count * slots_per_query * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            error = hip_message(e);
    } while (false);
    (void)hipFree(d_queries);
    (void)hipFree(d_slots);
    (void)hipFree(d_out);
    return error;
}

} // namespace usearch_amd
