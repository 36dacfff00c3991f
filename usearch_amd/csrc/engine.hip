/**
 *  usearch_amd/csrc/engine.hip — snapshot construction (flatten a v2 image into HBM arrays) and the batched search
 *  driver with its scratch-overflow retry ladder. See engine.hpp / kernels.hpp for the design.
 */
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <hipcub/hipcub.hpp>

#include "casts.hpp"
#include "host_util.hpp"
#include "kernels.hpp"

namespace usearch_amd {

/// How the short-row walks probe their visited-set slabs unless USEARCH_AMD_PROBE_MODE says otherwise (common.hpp `probe_mode_t`).
static constexpr std::uint32_t default_probe_mode_k = probe_swap_k;
/// Rows of ≤ 128 bytes gathered next to the probe of the visited set (USEARCH_AMD_EARLY_ROWS = 0 | 1 overrides).
static constexpr std::size_t default_early_rows_k = 1; // 20M x 96 i8: +6.4 % at ef 80, +5.1 % at ef 64, same keys / bits / counters (profiles/r06_short_rows/early_rows.log)

/// The block of per-wave visited-set slabs. (Round 5's experiment — the block in uncached or fine-grained device memory, to see whether
/// the two-microsecond trip of a probe belongs to the memory type: it does not, profiles/r05_short_rows/ — compiles in only with
/// -DUSEARCH_AMD_EXPERIMENT_SCRATCH_MEMORY: USEARCH_AMD_SCRATCH_MEMORY = 1 `hipDeviceMallocUncached`, 2 `hipDeviceMallocFinegrained`.)
static hipError_t scratch_malloc(void** out, std::size_t bytes) {
#ifdef USEARCH_AMD_EXPERIMENT_SCRATCH_MEMORY
    const std::size_t kind = env_size("USEARCH_AMD_SCRATCH_MEMORY", 0);
    if (kind == 1 || kind == 2) {
        const hipError_t e = hipExtMallocWithFlags(out, bytes, kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e == hipSuccess)
            return e;
        (void)hipGetLastError();
    }
#endif
    return block_malloc(out, bytes);
}

void row_geometry(std::size_t bytes, std::uint32_t& lanes, std::uint32_t& row_stride, std::uint32_t& chunks) {
    // lanes per row (G): the smallest power of two covering the row's 16-byte chunks, at most 8 (= one 128-byte line per
    // load). USEARCH_AMD_LANES overrides (1, 2, 4, 8): fewer lanes per row = more rows per round trip.
    const std::uint32_t raw_chunks = std::max<std::uint32_t>(1, (std::uint32_t)((bytes + 15) / 16));
    lanes = std::min<std::uint32_t>(8, pow2_ceil(raw_chunks));
    if (raw_chunks <= 8) // rows of ≤ 128 bytes: 32 rows per round trip (a whole neighbour list) beat one line per load —
        lanes = std::min<std::uint32_t>(2, lanes); // 20M x 96 i8: 5.15 M QPS with G = 2 against 4.31 M with G = 8
    const std::size_t forced = env_size("USEARCH_AMD_LANES", 0);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8)
        lanes = std::min<std::uint32_t>((std::uint32_t)forced, pow2_ceil(raw_chunks));
    // what the kernels read of a row: whole 16-byte chunks, a multiple of G of them
    chunks = (std::uint32_t)((bytes + 16 * lanes - 1) / (16 * lanes) * lanes);
    row_stride = chunks * 16;
    // where rows start: a row of ≤ 128 bytes never straddles two 128-byte lines when the pitch is a power of two (96-byte
    // rows at pitch 96 touch 1.5 lines on average: half the fetched bytes wasted); longer rows start on a line boundary
    // when that costs at most 1/8 of the matrix. USEARCH_AMD_DENSE_ROWS=1 keeps the dense pitch (A/B runs).
    if (!env_size("USEARCH_AMD_DENSE_ROWS", 0)) {
        if (row_stride <= 128)
            row_stride = pow2_ceil(row_stride);
        else if (row_stride % 128 && (128 - row_stride % 128) * 8 <= row_stride)
            row_stride += 128 - row_stride % 128;
    }
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
/// gfx950 hands LDS out in blocks of 320 dwords (160 KB = 128 of them). Round 6 measured it the hard way: a wave of 8 160 bytes "fits" 20
/// times by a 1 024-byte count and runs as 18 (i8 × 96 at expansion 80 with 1 024 `seen` cells: 10.99 ms against 9.97 with 512).
constexpr std::uint64_t lds_granule_k = 1280;

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

// ---------------------------------------------------------------------------------------------------------------------
//  Workspaces: what one in-flight batch owns
// ---------------------------------------------------------------------------------------------------------------------

const char* workspace_t::create() {
    UA_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    UA_HIP(hipEventCreate(&event_begin));
    UA_HIP(hipEventCreate(&event_end));
    UA_HIP(hipMalloc((void**)&d_queue, 256));
    return nullptr;
}

const char* workspace_t::reserve(std::size_t queries_wanted, std::size_t scratch_wanted) {
    if (queries_wanted > queries) {
        for (void* p : {(void*)d_status, (void*)d_todo, (void*)d_peaks})
            if (p)
                (void)hipFree(p);
        if (h_status)
            (void)hipHostFree(h_status);
        d_status = d_todo = d_peaks = h_status = nullptr;
        queries = 0;
        const std::size_t room = std::max<std::size_t>(queries_wanted, 64);
        UA_HIP(hipMalloc((void**)&d_status, room * 4));
        UA_HIP(hipMalloc((void**)&d_todo, room * 4));
        UA_HIP(hipMalloc((void**)&d_peaks, room * 8));
        UA_HIP(hipHostMalloc((void**)&h_status, (room + 16) * 4, hipHostMallocDefault));
        queries = room;
    }
    // experiment switch (scripts/placement_study.py --scratch-redraws): a FRESH scratch block for every launch, the old ones parked
    // so that each new one lands elsewhere — does where the visited-set slabs sit decide the speed of the walk?
    if (scratch_wanted && env_size("USEARCH_AMD_SCRATCH_REDRAW", 0)) {
        static std::vector<void*> parked;
        if (d_scratch)
            parked.push_back(d_scratch);
        if (parked.size() > 12) {
            for (void* p : parked)
                placed_free(p);
            parked.clear();
        }
        d_scratch = nullptr;
        scratch_bytes = 0;
    }
    if (scratch_wanted > scratch_bytes) {
        if (d_scratch)
            placed_free(d_scratch);
        d_scratch = nullptr;
        scratch_bytes = 0;
        UA_HIP(scratch_malloc((void**)&d_scratch, scratch_wanted)); // big blocks for chip-filling launches are drawn by run_ladder
        scratch_bytes = scratch_wanted;
    }
    return nullptr;
}

const char* workspace_t::reserve_stage(std::size_t device_bytes, std::size_t host_bytes) {
    if (device_bytes > stage_bytes) {
        if (d_stage)
            (void)hipFree(d_stage);
        d_stage = nullptr;
        stage_bytes = 0;
        const std::size_t room = std::max<std::size_t>(device_bytes, 64 << 10);
        UA_HIP(hipMalloc((void**)&d_stage, room));
        stage_bytes = room;
    }
    if (host_bytes > host_stage_bytes) {
        if (h_stage)
            (void)hipHostFree(h_stage);
        h_stage = nullptr;
        host_stage_bytes = 0;
        const std::size_t room = std::max<std::size_t>(host_bytes, 64 << 10);
        UA_HIP(hipHostMalloc((void**)&h_stage, room, hipHostMallocDefault));
        host_stage_bytes = room;
    }
    return nullptr;
}

const char* workspace_t::reserve_wave_clock(std::size_t waves) {
    if (waves > wave_clock_waves) {
        if (d_wave_clock)
            (void)hipFree(d_wave_clock);
        d_wave_clock = nullptr;
        wave_clock_waves = 0;
        UA_HIP(hipMalloc((void**)&d_wave_clock, waves * 16));
        wave_clock_waves = waves;
    }
    return nullptr;
}

void workspace_t::destroy() {
    for (void* p : {(void*)d_status, (void*)d_todo, (void*)d_queue, (void*)d_peaks, (void*)d_stage, (void*)d_wave_clock})
        if (p)
            (void)hipFree(p);
    placed_free(d_scratch); // a block of `block_malloc` (possibly a mapped range)
    if (h_status)
        (void)hipHostFree(h_status);
    if (h_stage)
        (void)hipHostFree(h_stage);
    if (event_begin)
        (void)hipEventDestroy(event_begin);
    if (event_end)
        (void)hipEventDestroy(event_end);
    if (stream)
        (void)hipStreamDestroy(stream);
    *this = workspace_t{};
}

const char* snapshot_t::take(workspace_t*& out) {
    std::unique_lock<std::mutex> lock(pool_mutex_);
    for (;;) {
        if (placing_) { // a launch is trying another placement of the matrix (`try_matrix_placement`): nobody else reads it meanwhile
            pool_ready_.wait(lock);
            continue;
        }
        if (!idle_.empty()) {
            out = idle_.back();
            idle_.pop_back();
            return nullptr;
        }
        if (workspaces_.size() < std::max<std::size_t>(1, max_workspaces_)) {
            std::unique_ptr<workspace_t> fresh(new workspace_t());
            UA_HIP(hipSetDevice(device_));
            if (const char* e = fresh->create()) {
                fresh->destroy();
                return e;
            }
            out = fresh.get();
            workspaces_.push_back(std::move(fresh));
            return nullptr;
        }
        pool_ready_.wait(lock);
    }
}

void snapshot_t::give_back(workspace_t* workspace) {
    {
        std::lock_guard<std::mutex> lock(pool_mutex_);
        idle_.push_back(workspace);
        last_used_ = workspace;
    }
    pool_ready_.notify_one();
}

void snapshot_t::set_concurrency(std::size_t workspaces) {
    std::lock_guard<std::mutex> lock(pool_mutex_);
    max_workspaces_ = std::max<std::size_t>(1, std::min<std::size_t>(workspaces, 256));
}

void snapshot_t::release() {
    if (!d_vectors_ && !d_nbr0_ && workspaces_.empty() && !stream_)
        return;
    (void)hipSetDevice(device_);
    for (void** p : {&d_vectors_, &d_nbr0_, &d_nbr0_rows_}) { // the gathered arrays may be mapped memory (placement.hpp)
        placed_free(*p);
        *p = nullptr;
    }
    for (void* p : {d_upper_ref_, d_upper_, d_keys_})
        placed_free(p);
    {
        std::lock_guard<std::mutex> lock(pool_mutex_);
        for (auto& workspace : workspaces_)
            workspace->destroy();
        workspaces_.clear();
        idle_.clear();
        last_used_ = nullptr;
    }
    if (stream_)
        (void)hipStreamDestroy(stream_);
    d_vectors_ = d_nbr0_ = d_upper_ref_ = d_upper_ = d_keys_ = d_nbr0_rows_ = nullptr;
    view_.nbr0_rows = nullptr;
    stream_ = nullptr;
}

/// out[i][j] = the stored row of nbr0[i][j], one 16-byte chunk per cell (empty cells are left alone: never read).
__global__ void inline_rows_kernel(const std::uint8_t* vectors, const std::uint32_t* nbr0, uint4* out, std::uint64_t cells,
                                   std::uint32_t row_stride) {
    // grid-stride: cells = members × M0 passes 2^32 (the thread limit of one launch) at 134M members of M0 = 32
    for (std::uint64_t i = blockIdx.x * (std::uint64_t)blockDim.x + threadIdx.x; i < cells;
         i += (std::uint64_t)gridDim.x * blockDim.x) {
        const std::uint32_t slot = nbr0[i];
        if (slot != none_slot_k)
            out[i] = *reinterpret_cast<const uint4*>(vectors + (std::uint64_t)slot * row_stride);
    }
}

/// One trial (placement.hpp). `launch(view, ms)` runs the launch's first queries over `view` once and reports the milliseconds.
const char* snapshot_t::try_matrix_placement(std::uint32_t expansion, const std::function<const char*(const snapshot_view_t&, float&)>& launch, hipStream_t stream) {
    {   // the matrix must be this launch's alone: no other batch in flight, none admitted until the trial is over
        std::lock_guard<std::mutex> lock(pool_mutex_);
        if (placing_ || workspaces_.size() - idle_.size() != 1)
            return nullptr;
        placing_ = true;
        placement_last_ef_ = expansion; // a trial that actually runs: what a later, wider launch is compared with
    }
    auto end_trials = [&]() {
        std::lock_guard<std::mutex> lock(pool_mutex_);
        placement_trials_left_ = 0;
    };
    struct release_t {
        snapshot_t& owner;
        ~release_t() {
            {
                std::lock_guard<std::mutex> lock(owner.pool_mutex_);
                owner.placing_ = false;
            }
            owner.pool_ready_.notify_all();
        }
    } release{*this};
    const auto started = std::chrono::steady_clock::now();
    std::size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || free_bytes < vectors_bytes_ + ((std::size_t)4 << 30)) {
        (void)hipGetLastError();
        end_trials(); // no room for a second copy of the matrix: it stays where it is
        return nullptr;
    }
    void* candidate = nullptr;
    if (placed_malloc(&candidate, vectors_bytes_, view_.row_stride, nullptr) != hipSuccess) {
        (void)hipGetLastError();
        end_trials();
        return nullptr;
    }
    if (hipMemcpyAsync(candidate, d_vectors_, vectors_bytes_, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipGetLastError();
        placed_free(candidate);
        end_trials();
        return nullptr;
    }
    snapshot_view_t other = view_;
    other.vectors = static_cast<const std::uint8_t*>(candidate);
    // incumbent, candidate, incumbent, candidate: the first run of each also pays first touches and the clocks' ramp, the later
    // ones count (the smaller of two each)
    // (short launches — a small expansion — are judged over more rounds: their differences are tenths of a millisecond)
    float incumbent_ms = 0.f, candidate_ms = 0.f;
    const char* failure = nullptr;
    for (int round = 0; round < 6 && !failure && !(round >= 3 && incumbent_ms >= 3.f); ++round) {
        float ms = 0.f;
        failure = launch(view_, ms);
        if (!failure && round)
            incumbent_ms = incumbent_ms == 0.f ? ms : std::min(incumbent_ms, ms);
        if (!failure)
            failure = launch(other, ms);
        if (!failure && round)
            candidate_ms = candidate_ms == 0.f ? ms : std::min(candidate_ms, ms);
    }
    if (failure) {
        placed_free(candidate);
        return failure;
    }
    const std::uint32_t trial = placement_.draws;
    if (trial < (std::uint32_t)placement_max_draws_k)
        placement_.judge_ms[trial] = candidate_ms, placement_.incumbent_ms[trial] = incumbent_ms;
    ++placement_.draws;
    // a candidate has to win by more than the judge's noise — one hundredth: the smaller of the later rounds repeats within half of
    // that (profiles/r06_settled/tuned_process_*: incumbents 9.325 … 9.373 ms, candidates 9.195 … 9.214 ms in three trials of one
    // process, and round 5's two hundredths left exactly that 1.5 % on the table); three trials in a row that the incumbent wins end
    // the search — it sits on frames as good as this device hands out
    const bool swap = candidate_ms < incumbent_ms * 0.99f;
    void* loser = candidate;
    {
        std::lock_guard<std::mutex> lock(pool_mutex_);
        --placement_trials_left_;
        if (swap) {
            loser = d_vectors_;
            d_vectors_ = candidate;
            view_.vectors = other.vectors;
            ++placement_.kept;
            placement_losses_ = 0;
        } else if (++placement_losses_ >= 3) {
            placement_trials_left_ = 0;
        }
    }
    // readers that hold no workspace lease (exact search on a caller's stream, the builder's `save_buffer`) may still have work
    // enqueued over the old matrix: nothing of this device may be running when its memory goes back
    (void)hipDeviceSynchronize();
    placed_free(loser);
    placement_.probe_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - started).count();
    if (env_size("USEARCH_AMD_PLACEMENT_LOG", 0))
        std::fprintf(stderr, "[usearch_amd] matrix placement trial %u (%.2f GB): incumbent %.3f ms, fresh copy %.3f ms over the launch's first queries: %s; %u trials left\n",
                     trial, vectors_bytes_ / 1e9, incumbent_ms, candidate_ms, swap ? "moved" : "stays", placement_trials_left_);
    return nullptr;
}

const char* snapshot_t::finalize_layout() {
    if (d_nbr0_rows_) {
        placed_free(d_nbr0_rows_);
        device_bytes_ -= std::min<std::size_t>(device_bytes_, (std::size_t)view_.size * view_.m0 * 16);
        d_nbr0_rows_ = nullptr;
        view_.nbr0_rows = nullptr;
    }
    if (lanes_ != 1 || view_.chunks != 1 || view_.m0 > 64 || !view_.size || !env_size("USEARCH_AMD_INLINE_ROWS", 1))
        return nullptr;
    UA_HIP(hipSetDevice(device_));
    const std::uint64_t cells = view_.size * view_.m0;
    UA_HIP(placed_malloc(&d_nbr0_rows_, cells * 16, (std::size_t)view_.m0 * 16, nullptr)); // what a hop gathers: one block
    device_bytes_ += cells * 16;
    const std::uint64_t blocks = std::min<std::uint64_t>((cells + 255) / 256, 1u << 22);
    hipLaunchKernelGGL(inline_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream_, view_.vectors, view_.nbr0,
                       static_cast<uint4*>(d_nbr0_rows_), cells, view_.row_stride);
    UA_HIP(hipGetLastError());
    UA_HIP(hipStreamSynchronize(stream_));
    view_.nbr0_rows = static_cast<const std::uint8_t*>(d_nbr0_rows_);
    return nullptr;
}

// ---------------------------------------------------------------------------------------------------------------------
//  Loader: a serialized v2 image → the flat HBM arrays, flattened ON THE DEVICE. The host uploads three byte ranges of the
//  image as they lie (levels, node tapes, vectors); offsets are a device prefix scan of the node sizes and the graph is
//  scattered by a kernel — no host array of N × M0 cells (a 125M-member shard would need 16 GB of them).
//  Format: index_dense.hpp:995-1062, index.hpp:3277-3317, docs/format.md.
// ---------------------------------------------------------------------------------------------------------------------

/// Per node: tape bytes and number of upper-level lists (its level), to be prefix-summed.
__global__ void node_sizes_kernel(const std::int16_t* levels, std::uint64_t n, std::uint64_t node_base_bytes,
                                  std::uint64_t level_bytes, std::uint64_t* sizes, std::uint64_t* lists) {
    const std::uint64_t i = blockIdx.x * (std::uint64_t)blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const std::uint64_t level = (std::uint64_t)levels[i];
    sizes[i] = node_base_bytes + level * level_bytes;
    lists[i] = level;
}

/// Tapes start at an arbitrary byte of the image but are uploaded to the start of their own allocation, and with 4-byte slots
/// every node size is even: 2-byte loads are always aligned. With the 5-byte slots of `uint40_t` (index.hpp:969-1031) nothing is
/// aligned any more and every load goes byte by byte (`bytewise_ak`).
template <bool bytewise_ak> __device__ __forceinline__ std::uint32_t tape_u16(const std::uint8_t* p) {
    if constexpr (bytewise_ak)
        return (std::uint32_t)p[0] | ((std::uint32_t)p[1] << 8);
    else
        return *reinterpret_cast<const std::uint16_t*>(p);
}
template <bool bytewise_ak> __device__ __forceinline__ std::uint32_t tape_u32(const std::uint8_t* p) {
    return tape_u16<bytewise_ak>(p) | (tape_u16<bytewise_ak>(p + 2) << 16);
}
template <bool bytewise_ak> __device__ __forceinline__ std::uint64_t tape_u64(const std::uint8_t* p) {
    return (std::uint64_t)tape_u32<bytewise_ak>(p) | ((std::uint64_t)tape_u32<bytewise_ak>(p + 4) << 32);
}
/// Neighbour slot `j` of a list: 4 bytes, or 5 of which the fifth must be zero for the slot to fit a 32-bit cell
/// (`none_slot_k` otherwise: the caller's `slot >= n` check then flags the image).
template <bool wide_ak> __device__ __forceinline__ std::uint32_t tape_slot(const std::uint8_t* list, std::uint32_t j) {
    if constexpr (wide_ak) {
        const std::uint8_t* p = list + 4 + 5 * (std::uint64_t)j;
        return p[4] ? none_slot_k : tape_u32<true>(p);
    } else {
        return tape_u32<false>(list + 4 + 4 * (std::uint64_t)j);
    }
}

/**
 *  One thread per node: key, level-0 row (reference order, later duplicates of a slot dropped — they could only ever be seen
 *  as "already visited", index.hpp:4229 — unused cells none), upper lists verbatim (no visited set up there,
 *  index.hpp:3976-4001), and the checks the reference's loader implies. flags[0] = corrupt, flags[1] = any tombstone.
 *  `wide_ak`: 5-byte `uint40_t` slots on the tapes.
 */
template <bool wide_ak>
__global__ void flatten_kernel(const std::uint8_t* tapes, const std::uint64_t* offsets, const std::uint64_t* first_list,
                               const std::int16_t* levels, std::uint64_t n, std::uint32_t m, std::uint32_t m0,
                               std::uint64_t* keys, std::uint32_t* nbr0, std::uint32_t* upper_ref, std::uint32_t* upper,
                               std::uint32_t* flags) {
    constexpr std::uint64_t slot_bytes = wide_ak ? 5 : 4;
    const std::uint64_t i = blockIdx.x * (std::uint64_t)blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const std::uint8_t* tape = tapes + offsets[i];
    const std::uint64_t key = tape_u64<wide_ak>(tape);
    keys[i] = key;
    if (key == free_key_k)
        flags[1] = 1;
    const std::int16_t level = levels[i];
    const std::int16_t taped_level = (std::int16_t)tape_u16<wide_ak>(tape + 8);
    if (taped_level != level) {
        flags[0] = 1;
        return;
    }
    upper_ref[i] = level ? (std::uint32_t)first_list[i] : none_slot_k;
    const std::uint8_t* list = tape + 10;
    std::uint32_t count = tape_u32<wide_ak>(list);
    if (count > m0) {
        flags[0] = 1;
        return;
    }
    std::uint32_t* row = nbr0 + i * m0;
    std::uint32_t kept = 0;
    for (std::uint32_t j = 0; j < count; ++j) {
        const std::uint32_t slot = tape_slot<wide_ak>(list, j);
        if (slot >= n) {
            flags[0] = 1;
            return;
        }
        bool seen = false;
        for (std::uint32_t k = 0; k < kept && !seen; ++k)
            seen = row[k] == slot;
        if (!seen)
            row[kept++] = slot;
    }
    for (std::uint32_t j = kept; j < m0; ++j)
        row[j] = none_slot_k;
    list += 4 + slot_bytes * (std::uint64_t)m0;
    for (std::int16_t l = 1; l <= level; ++l, list += 4 + slot_bytes * (std::uint64_t)m) {
        count = tape_u32<wide_ak>(list);
        if (count > m) {
            flags[0] = 1;
            return;
        }
        std::uint32_t* cells = upper + (first_list[i] + (std::uint64_t)(l - 1)) * m;
        for (std::uint32_t j = 0; j < m; ++j) {
            std::uint32_t slot = none_slot_k;
            if (j < count) {
                slot = tape_slot<wide_ak>(list, j);
                if (slot >= n || levels[slot] < l) {
                    flags[0] = 1;
                    return;
                }
            }
            cells[j] = slot;
        }
    }
}

const char* snapshot_t::build(const image_t& image, int device) {
    if (!kernel_available(image.metric, image.scalar))
        return "No MI355X kernel for this metric / scalar kind combination";
    release();
    device_ = device;
    UA_HIP(hipSetDevice(device));
    metric_ = image.metric;
    scalar_ = image.scalar;
    count_present_ = image.count_present;

    const std::uint64_t n = image.size;
    const std::uint32_t m = (std::uint32_t)image.connectivity, m0 = (std::uint32_t)image.connectivity_base;
    const std::uint32_t bpv = (std::uint32_t)image.cols;
    std::uint32_t row_stride = 0, row_chunks = 0;
    row_geometry(bpv, lanes_, row_stride, row_chunks);
    if (n && (image.max_level > 0x7FFF || (std::int64_t)image.level(image.entry_slot) < (std::int64_t)image.max_level))
        return "Failed to pull the header from the stream";

    // ---- scratch that lives for the load only: levels, tape bytes, the two prefix sums
    struct scratch_t {
        std::vector<void*> pointers;
        ~scratch_t() {
            for (void* p : pointers)
                placed_free(p); // time-stamps the release of large blocks (placement.hpp: settle, then allocate)
        }
        hipError_t allocate(void** out, std::size_t bytes) {
            *out = nullptr;
            const hipError_t e = hipMalloc(out, std::max<std::size_t>(bytes, 16));
            if (e == hipSuccess)
                pointers.push_back(*out);
            return e;
        }
    } scratch;
    std::int16_t* d_levels = nullptr;
    std::uint8_t* d_tapes = nullptr;
    std::uint64_t *d_sizes = nullptr, *d_offsets = nullptr, *d_level_counts = nullptr, *d_first_list = nullptr;
    std::uint32_t* d_flags = nullptr;
    void* d_scan = nullptr;
    std::uint64_t tapes_bytes = 0, lists = 0;
    if (n) {
        UA_HIP(scratch.allocate((void**)&d_levels, n * 2));
        UA_HIP(scratch.allocate((void**)&d_sizes, n * 8));
        UA_HIP(scratch.allocate((void**)&d_offsets, n * 8));
        UA_HIP(scratch.allocate((void**)&d_level_counts, n * 8));
        UA_HIP(scratch.allocate((void**)&d_first_list, n * 8));
        UA_HIP(scratch.allocate((void**)&d_flags, 16));
        UA_HIP(hipMemset(d_flags, 0, 16));
        UA_HIP(hipMemcpy(d_levels, image.levels, n * 2, hipMemcpyHostToDevice)); // 2-byte aligned in its own allocation
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(node_sizes_kernel, dim3(blocks), dim3(256), 0, nullptr, d_levels, n,
                           (std::uint64_t)image.node_bytes(0), (std::uint64_t)(4 + image.slot_bytes * (std::uint64_t)m), d_sizes,
                           d_level_counts);
        UA_HIP(hipGetLastError());
        std::size_t scan_bytes = 0;
        UA_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_sizes, d_offsets, (int)n));
        UA_HIP(scratch.allocate(&d_scan, scan_bytes));
        UA_HIP(hipcub::DeviceScan::ExclusiveSum(d_scan, scan_bytes, d_sizes, d_offsets, (int)n));
        UA_HIP(hipcub::DeviceScan::ExclusiveSum(d_scan, scan_bytes, d_level_counts, d_first_list, (int)n));
        std::uint64_t last[2] = {0, 0}, last_size[2] = {0, 0};
        UA_HIP(hipMemcpy(&last[0], d_offsets + (n - 1), 8, hipMemcpyDeviceToHost));
        UA_HIP(hipMemcpy(&last[1], d_first_list + (n - 1), 8, hipMemcpyDeviceToHost));
        UA_HIP(hipMemcpy(&last_size[0], d_sizes + (n - 1), 8, hipMemcpyDeviceToHost));
        UA_HIP(hipMemcpy(&last_size[1], d_level_counts + (n - 1), 8, hipMemcpyDeviceToHost));
        tapes_bytes = last[0] + last_size[0];
        lists = last[1] + last_size[1];
        if (tapes_bytes > image.tapes_length)
            return "Failed to pull nodes from the stream";
        if (lists >= none_slot_k)
            return "Too many upper-level lists for 32-bit references";
    }
    upper_lists_ = lists;

    // ---- the index's own arrays
    const std::size_t vectors_bytes = (std::size_t)n * row_stride;
    auto allocate = [&](void** p, std::size_t bytes) -> hipError_t {
        device_bytes_ += std::max<std::size_t>(bytes, 16);
        return hipMalloc(p, std::max<std::size_t>(bytes, 16));
    };
    device_bytes_ = 0;
    // the matrix the walk gathers rows from takes the best of a few placements (placement.hpp); its size makes it THE array
    device_bytes_ += std::max<std::size_t>(vectors_bytes, 16);
    UA_HIP(placed_malloc(&d_vectors_, vectors_bytes, row_stride, &placement_));
    vectors_bytes_ = vectors_bytes;
    device_bytes_ += std::max<std::size_t>((std::size_t)n * m0 * 4, 16);
    UA_HIP(placed_malloc(&d_nbr0_, (std::size_t)n * m0 * 4, (std::size_t)m0 * 4, nullptr));
    UA_HIP(allocate(&d_upper_ref_, (std::size_t)n * 4));
    UA_HIP(allocate(&d_upper_, (std::size_t)std::max<std::uint64_t>(lists, 1) * m * 4));
    UA_HIP(allocate(&d_keys_, (std::size_t)n * 8));
    UA_HIP(hipMemset(d_upper_, 0xFF, (std::size_t)std::max<std::uint64_t>(lists, 1) * m * 4));
    bool tombstones = false;
    if (n) {
        UA_HIP(scratch.allocate((void**)&d_tapes, tapes_bytes));
        UA_HIP(hipMemcpy(d_tapes, image.tapes, tapes_bytes, hipMemcpyHostToDevice));
        const unsigned blocks = (unsigned)((n + 127) / 128);
        if (image.slot_bytes == 5)
            hipLaunchKernelGGL(flatten_kernel<true>, dim3(blocks), dim3(128), 0, nullptr, d_tapes, d_offsets, d_first_list, d_levels,
                               n, m, m0, static_cast<std::uint64_t*>(d_keys_), static_cast<std::uint32_t*>(d_nbr0_),
                               static_cast<std::uint32_t*>(d_upper_ref_), static_cast<std::uint32_t*>(d_upper_), d_flags);
        else
            hipLaunchKernelGGL(flatten_kernel<false>, dim3(blocks), dim3(128), 0, nullptr, d_tapes, d_offsets, d_first_list, d_levels,
                               n, m, m0, static_cast<std::uint64_t*>(d_keys_), static_cast<std::uint32_t*>(d_nbr0_),
                               static_cast<std::uint32_t*>(d_upper_ref_), static_cast<std::uint32_t*>(d_upper_), d_flags);
        UA_HIP(hipGetLastError());
        std::uint32_t flags[2] = {0, 0};
        UA_HIP(hipMemcpy(flags, d_flags, 8, hipMemcpyDeviceToHost));
        if (flags[0])
            return "Failed to pull nodes from the stream";
        tombstones = flags[1] != 0;
        if (const char* e = upload_rows(static_cast<std::uint8_t*>(d_vectors_), row_stride, image.vectors,
                                        (std::size_t)image.vector_stride, bpv, n))
            return e;
    }

    view_ = snapshot_view_t{};
    view_.vectors = static_cast<const std::uint8_t*>(d_vectors_);
    view_.nbr0 = static_cast<const std::uint32_t*>(d_nbr0_);
    view_.upper_ref = static_cast<const std::uint32_t*>(d_upper_ref_);
    view_.upper = static_cast<const std::uint32_t*>(d_upper_);
    view_.keys = static_cast<const std::uint64_t*>(d_keys_);
    view_.size = n;
    view_.row_stride = row_stride;
    view_.chunks = row_chunks;
    view_.bytes_per_vector = bpv;
    view_.dimensions = (std::uint32_t)image.dimensions;
    view_.m = m;
    view_.m0 = m0;
    view_.max_level = (std::uint32_t)image.max_level;
    view_.entry_slot = (std::uint32_t)image.entry_slot;
    view_.has_tombstones = tombstones ? 1u : 0u;

    hipDeviceProp_t properties;
    UA_HIP(hipGetDeviceProperties(&properties, device));
    compute_units_ = properties.multiProcessorCount > 0 ? properties.multiProcessorCount : 256;
    UA_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    return finalize_layout();
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

/// The float-valued pairs may keep their frontier as the open cells of a register `top` (kernels.hpp frontier_top_k).
static bool frontier_in_top_capable(scalar_kind_t scalar) { return scalar != scalar_b1x8_k && scalar != scalar_i8_k; }

const char* snapshot_t::search_begin(search_call_t& call, const void* queries, std::size_t count,
                                     std::size_t stride_bytes, std::size_t wanted, std::size_t expansion,
                                     std::uint64_t* keys, float* distances, std::uint64_t* counts, std::uint64_t* visited,
                                     std::uint64_t* computed, hipStream_t stream, const search_tuning_t& tuning, bool timed,
                                     const search_extras_t* extras) {
    call.count = count;
    call.timed = timed;
    if (!count || !wanted) { // index.hpp:3025-3027: nothing wanted, nothing found
        call.done = true;
        return nullptr;
    }
    if (count >= none_slot_k || wanted >= (1u << 24))
        return "Batch is too large";
    UA_HIP(hipSetDevice(device_));
    if (!call.workspace) // a caller that staged buffers in a workspace brings it along
        if (const char* e = take(call.workspace))
            return e;
    workspace_t& ws = *call.workspace;
    if (!stream)
        stream = ws.stream;
    call.stream = stream;

    if (view_.size == 0) { // index.hpp:3034-3037
        const std::uint64_t cells = std::max<std::uint64_t>(count * wanted, count);
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, keys,
                           reinterpret_cast<std::uint32_t*>(distances), counts, visited, computed,
                           (std::uint64_t)count, (std::uint64_t)wanted);
        UA_HIP(hipGetLastError());
        call.done = true;
        return nullptr;
    }

    if (!expansion)
        expansion = default_expansion_search_k;
    const std::uint32_t ef = (std::uint32_t)std::max(expansion, wanted); // index.hpp:3052
    call.ef = ef;

    // ---- scratch sizing, from measurements with the reference's own traversal (DESIGN.md "scratch sizing"): the frontier
    // peaks at 2.4-3.9 × ef; the visited set ends at 18-30 × ef entries plus what the first hops of a big index cost
    // whatever the expansion (10M × 768, ef = 64: 3 137 entries = 49 × ef) — hence the constant term. Outliers go through
    // the retry ladder.
    const std::uint32_t query_lds = view_.chunks * (query_chunk_bytes_of(scalar_));
    call.query_lds = query_lds;
    const std::uint32_t lds_budget = (std::uint32_t)env_size("USEARCH_AMD_LDS_BUDGET", 160 * 1024);
    std::uint32_t hash_cap = tuning.hash_cap ? tuning.hash_cap : (std::uint32_t)env_size("USEARCH_AMD_HASH_CAP", 0);
    if (!hash_cap) {
        // entries expected ÷ the load the set is sized for: 75 % (the kernel's limit) for short rows, whose slabs must stay
        // cache-resident (profiles/r02_visited_set.log); 50 % for rows of ≥ 128 bytes — every probe round is a two-microsecond trip
        // to the memory side for the whole wave, and the headline batch runs 1.9 % faster with 65 536 cells than with 32 768
        // (46.2 against 47.1 ms on fresh blocks, profiles/r04_placement/scratch_footprint.log); USEARCH_AMD_HASH_LOAD_PCT overrides
        const std::uint32_t load_pct = (std::uint32_t)std::min<std::size_t>(75, std::max<std::size_t>(10, env_size("USEARCH_AMD_HASH_LOAD_PCT", lanes_ >= 8 ? 50 : 75)));
        hash_cap = std::max<std::uint32_t>(1024, (std::uint32_t)((std::uint64_t)(ef * 30 + 1600) * 100 / load_pct));
    }
    hash_cap = pow2_ceil(hash_cap);
    std::uint32_t next_cap = tuning.next_cap ? tuning.next_cap : (std::uint32_t)env_size("USEARCH_AMD_NEXT_CAP", 0);
    if (!next_cap) // (peaks measured on 20M-vector slices, 100 000 queries: b1 × 128 at 64: median 158, maximum 317; i8 × 96 at 80: 220 / 360)
        next_cap = std::max<std::uint32_t>(448, ef * 3 + 256);
    // never larger than the index could possibly need
    hash_cap = std::min<std::uint32_t>(hash_cap, pow2_ceil((std::uint32_t)std::min<std::uint64_t>(view_.size * 2 + 128, 1u << 30)));
    next_cap = (std::uint32_t)std::min<std::uint64_t>(next_cap, view_.size + 64);

    std::uint32_t mode_request = tuning.mode ? tuning.mode : (std::uint32_t)env_size("USEARCH_AMD_MODE", 0);
    if (mode_request > 3)
        return "Unknown scratch mode";
    // `top` lives in registers (1 / 4 / 8 / 16 entries per lane) while the expansion allows it
    const bool top_in_memory = tuning.top_in_memory || env_size("USEARCH_AMD_TOP_IN_MEMORY", 0) != 0;
    const bool two_cells = lanes_ <= 2 && ef <= 128 && !env_size("USEARCH_AMD_NO_TWO_CELLS", 0); // short rows: see kernel_waves()
    const std::uint32_t entries_per_lane = top_in_memory ? 0u : ef <= 64 ? 1u : two_cells ? 2u : ef <= 256 ? 4u : ef <= 512 ? 8u : ef <= 1024 ? 16u : 0u;
    call.entries_per_lane = entries_per_lane;

    // ---- who holds the frontier (kernels.hpp frontier_mode_t): the open cells of `top` wherever that is exact up to ties —
    // float-valued pair, `top` in registers, every member a result candidate, slots below 2^31 — else the reference's heap
    const std::uint32_t frontier_request = tuning.frontier ? tuning.frontier : (std::uint32_t)env_size("USEARCH_AMD_FRONTIER", 0);
    const bool filtered = view_.has_tombstones || (extras && (extras->allow_bits || extras->exclude_own));
    const bool in_top_possible = frontier_in_top_capable(scalar_) && entries_per_lane && !filtered && mode_request != 3 &&
                                 view_.size < 0x80000000ull && !(extras && (extras->reference_frontier || extras->descent_only));
    if (frontier_request == 2 && !in_top_possible)
        return "The frontier cannot ride in `top` for this search (integer-valued pair, filter, tombstones or expansion > 1024)";
    const int frontier = (frontier_request == 1 || !in_top_possible) ? frontier_heap_k : frontier_top_k;
    if (frontier == frontier_top_k)
        next_cap = 0;

    // register/latency trade-off of the kernel (kernels.hpp kernel_variant_t); rows shorter than 8 chunks per lane have
    // nothing to unroll
    const std::uint32_t chunks_per_lane = view_.chunks / lanes_;
    std::uint32_t variant_request = tuning.variant ? tuning.variant : (std::uint32_t)env_size("USEARCH_AMD_VARIANT", 0);
    int variant = variant_u4_w4_k;
    const bool every_build = lanes_ == 8 && all_kernel_builds(kernel_metric(metric_), scalar_);
    if (every_build && chunks_per_lane >= 8) {
        // measured on 10M x 768 f16 (profiles/): a whole row per round trip (12 loads per lane, 8 waves per CU) beats 8 loads
        // at 12 waves per CU at every expansion — the traversal is latency-bound, fewer round trips per hop win
        variant = chunks_per_lane >= 12 ? variant_u12_w2_k : variant_u8_w3_k;
        // Without the heap (profiles/r02_sweep_variants.log, ef = 608): every build lands within 3 % of the others — the
        // kernel moves 4.6-4.9 TB/s of rows plus the visited-set traffic, which is what random 1.5-KB gathers reach on this
        // memory system at all — and what separates them is the DRAIN of the batch: with one wave per query the last queries
        // run alone, for about 0.65 × waves / queries of the launch. Few waves with many bytes in flight each (two rows per
        // lane group per round, 8 waves per CU) win while that matters; 16 waves per CU win once the batch is long enough.
        if (frontier == frontier_top_k && chunks_per_lane >= 12)
            variant = count >= 40000 ? variant_u4_w4_k : variant_u12x2_w2_k;
    }
    if (variant_request && variant_request - 1 < (std::uint32_t)variant_count_k && every_build) {
        const int requested = (int)variant_request - 1;
        if (requested == variant_u12x2_w2_k && frontier != frontier_top_k)
            return "That kernel build exists for the in-`top` frontier only";
        variant = requested;
    }
    // A batch that cannot give every CU two queries to walk (a `usearch_search` caller's single query above all) over long rows: four
    // helper waves per query take the rows of every hop, the leader walks and commits (kernels.hpp team_search_kernel)
    const bool team = every_build && chunks_per_lane >= 8 && !variant_request && mode_request != 3 &&
                      count <= 2ull * compute_units_ && !(extras && extras->descent_only) && !env_size("USEARCH_AMD_NO_TEAM", 0) &&
                      !tuning.waves_per_cu;
    if (team)
        variant = variant_u12_w2_k;
    // whether this call can run the short-row build cut for plain batches (kernels.hpp `plain_ak`) as far as that is known here; the
    // scratch mode, the `seen` cells and the early rows are settled per rung in run_ladder, which has the last word (`params.plain`)
    call.plain_possible = !team && !view_.has_tombstones && view_.m0 <= 64 && view_.nbr0 &&
                          !(extras && (extras->query_ids || extras->beam_level || extras->descent_only || extras->allow_bits || extras->exclude_own)) &&
                          (lanes_ == 1 ? view_.nbr0_rows != nullptr && view_.chunks == 1 : lanes_ == 2) &&
                          plain_build_exists(kernel_metric(metric_), scalar_, (int)lanes_, variant == variant_u4_w4_k, true, (int)entries_per_lane,
                                             frontier == frontier_heap_k) &&
                          !env_size("USEARCH_AMD_NO_PLAIN", 0);
    const std::uint32_t variant_waves_per_cu =
        4u * (std::uint32_t)kernel_waves(variant, (int)entries_per_lane, frontier, (int)lanes_, call.plain_possible);
    const std::uint32_t waves_cap = tuning.waves_per_cu ? tuning.waves_per_cu
                                                        : (std::uint32_t)env_size("USEARCH_AMD_WAVES_PER_CU", 32);
    call.waves_cap = std::min(waves_cap, variant_waves_per_cu);

    auto lds_bytes_for = [&](int mode, std::uint32_t cap_next, std::uint32_t cap_hash) -> std::uint64_t {
        if (mode == scratch_global_k)
            return query_lds;
        const scratch_layout_t l = scratch_layout(entries_per_lane ? 0 : ef, cap_next,
                                                  mode == scratch_lds_k ? (std::uint64_t)cap_hash * 4 : 0);
        return query_lds + l.total;
    };
    auto waves_for = [&](std::uint64_t lds_bytes) -> std::uint32_t {
        const std::uint64_t granule = (lds_bytes + lds_granule_k - 1) / lds_granule_k * lds_granule_k; // LDS is allocated in coarse granules
        return (std::uint32_t)std::max<std::uint64_t>(
            1, std::min<std::uint64_t>(call.waves_cap, lds_budget / std::max<std::uint64_t>(granule, 1)));
    };
    // the frontier's default room has 256 cells of slack; when giving up to half of it back lets one more wave share the
    // compute unit's LDS, do (the retry ladder still catches a query that would have needed them)
    const bool default_next_cap = !tuning.next_cap && !env_size("USEARCH_AMD_NEXT_CAP", 0);
    if (default_next_cap && next_cap && mode_request != 1 && mode_request != 3) {
        const std::uint32_t now = waves_for(lds_bytes_for(scratch_hash_k, next_cap, hash_cap));
        if (now < call.waves_cap) {
            const std::uint64_t room = lds_budget / (now + 1) / lds_granule_k * lds_granule_k;
            const std::uint64_t fixed = lds_bytes_for(scratch_hash_k, 0, hash_cap);
            if (room > fixed) {
                const std::uint32_t trimmed = (std::uint32_t)((room - fixed) / 8 / 2 * 2);
                if (trimmed < next_cap && trimmed + 128 >= next_cap)
                    next_cap = trimmed;
            }
        }
    }
    // auto: keep the visited set in LDS only while that does not cost a resident wave; otherwise move it to the global hash. A batch
    // so small that every query gets a wave of its own even at the LDS residency (a `usearch_search` caller's single query above
    // all) also takes LDS: residency buys it nothing, and every probe round of the global hash is a two-microsecond trip to the
    // memory side — half of such a query's latency (profiles/r03_short_rows/README.md §1)
    std::uint64_t lds_mode_bytes = lds_bytes_for(scratch_lds_k, next_cap, hash_cap);
    // a set sized for half load that does not fit LDS where the one sized for 75 % would (expansion 608 over long rows: 256 KB
    // against 128 KB): a small batch takes the smaller set in LDS rather than the larger one in global memory — a lone query walks
    // 3.1 ms that way and 3.5 ms the other
    if (!tuning.hash_cap && !env_size("USEARCH_AMD_HASH_CAP", 0) && mode_request == 0 && lds_mode_bytes > lds_budget) {
        const std::uint32_t tighter = std::min<std::uint32_t>(
            pow2_ceil(std::max<std::uint32_t>(1024, (ef * 30 + 1600) / 3 * 4)),
            pow2_ceil((std::uint32_t)std::min<std::uint64_t>(view_.size * 2 + 128, 1u << 30)));
        const std::uint64_t tighter_bytes = lds_bytes_for(scratch_lds_k, next_cap, tighter);
        if (tighter < hash_cap && tighter_bytes <= lds_budget && count <= (std::uint64_t)waves_for(tighter_bytes) * compute_units_ &&
            !env_size("USEARCH_AMD_NO_SMALL_BATCH_LDS", 0)) {
            hash_cap = tighter;
            lds_mode_bytes = tighter_bytes;
        }
    }
    const bool small_batch = lds_mode_bytes <= lds_budget && count <= (std::uint64_t)waves_for(lds_mode_bytes) * compute_units_ &&
                             !env_size("USEARCH_AMD_NO_SMALL_BATCH_LDS", 0);
    int mode = mode_request == 1 ? scratch_lds_k : mode_request == 2 ? scratch_hash_k : mode_request == 3 ? scratch_global_k
               : (small_batch || waves_for(lds_mode_bytes) >= std::min<std::uint32_t>(8, call.waves_cap) ? scratch_lds_k : scratch_hash_k);
    call.mode = mode;
    call.hash_cap = hash_cap;
    call.next_cap = next_cap;

    if (const char* e = ws.reserve(count, 0))
        return e;

    search_args_t& args = call.args;
    args = search_args_t{};
    args.queries = static_cast<const std::uint8_t*>(queries);
    args.query_stride = stride_bytes;
    args.wanted = (std::uint32_t)wanted;
    args.ef = ef;
    args.keys = keys;
    args.distances = distances;
    args.counts = counts;
    args.visited = visited;
    args.computed = computed;
    args.status = ws.d_status;
    args.queue = ws.d_queue;
    args.peaks = ws.d_peaks;
    if (extras) {
        args.query_ids = extras->query_ids;
        args.beam_level = extras->beam_level;
        args.emit_slots = extras->emit_slots ? 1u : 0u;
        args.descent_only = extras->descent_only ? 1u : 0u;
        args.allow_bits = extras->allow_bits;
        args.known_bits = extras->known_bits;
        args.ask_slots = extras->ask_slots, args.ask_keys = extras->ask_keys;
        args.ask_cursor = extras->ask_cursor, args.ask_cap = extras->ask_cap, args.guess_threshold = extras->guess_threshold;
        args.exclude_own = extras->exclude_own ? 1u : 0u;
    }

    launch_params_t& params = call.params;
    params = launch_params_t{};
    params.metric = metric_;
    params.lanes = lanes_;
    params.variant = variant;
    params.frontier = frontier;
    params.team = team ? 1u : 0u;
    params.stream = stream;
    call.stats = search_stats_t{};
    call.stats.frontier = frontier == frontier_top_k ? 2u : 1u;
    call.stats.variant = team ? 5u : (std::uint32_t)variant + 1; // 5 = the team build (five waves per query)
    call.stats.top_cells = entries_per_lane;

    // diagnostic: per-phase shader-clock ticks of the search kernel, printed to stderr (USEARCH_AMD_PHASES=1)
    call.want_phases = env_size("USEARCH_AMD_PHASES", 0) != 0;
    if (call.want_phases) {
        args.phases = reinterpret_cast<unsigned long long*>(ws.d_queue) + 8; // d_queue is a 256-byte block
        UA_HIP(hipMemsetAsync(args.phases, 0, 128, stream));
    }
    call.want_clock = tuning.wave_clock || env_size("USEARCH_AMD_WAVE_CLOCK", 0) != 0;

    // ---- first launch: persistent waves, heaps in LDS. Nobody waits here.
    call.passes = 0;
    call.total_ms = 0.f;
    call.have_todo = false;
    call.todo.clear();
    ws.last_count = count;
    return run_ladder(call);
}

/// One launch of the ladder's current rung over `pending` queries (all of them, or `call.todo`).
const char* snapshot_t::run_ladder(search_call_t& call) {
    workspace_t& ws = *call.workspace;
    search_args_t& args = call.args;
    launch_params_t& params = call.params;
    hipStream_t stream = call.stream;
    const std::uint32_t ef = call.ef;
    const std::uint32_t lds_budget = (std::uint32_t)env_size("USEARCH_AMD_LDS_BUDGET", 160 * 1024);
    auto lds_bytes_for = [&](int mode, std::uint32_t cap_next, std::uint32_t cap_hash) -> std::uint64_t {
        if (mode == scratch_global_k)
            return call.query_lds;
        const scratch_layout_t l = scratch_layout(call.entries_per_lane ? 0 : ef, cap_next,
                                                  mode == scratch_lds_k ? (std::uint64_t)cap_hash * 4 : 0);
        return call.query_lds + l.total;
    };
    auto waves_for = [&](std::uint64_t lds_bytes) -> std::uint32_t {
        const std::uint64_t granule = (lds_bytes + lds_granule_k - 1) / lds_granule_k * lds_granule_k;
        return (std::uint32_t)std::max<std::uint64_t>(
            1, std::min<std::uint64_t>(call.waves_cap, lds_budget / std::max<std::uint64_t>(granule, 1)));
    };
    auto timed_launch = [&](bool keep_overflows = false) -> const char* {
        // [0] the ticket counter, [1] how many queries outgrew their scratch — the latter is read once per rung, so the chunks of
        // the global rung must not erase what an earlier chunk counted
        UA_HIP(hipMemsetAsync(ws.d_queue, 0, keep_overflows ? 4 : 8, stream));
        if (call.timed)
            UA_HIP(hipEventRecord(ws.event_begin, stream));
        UA_HIP(launch_search(metric_, scalar_, params, view_, args));
        if (call.timed) {
            UA_HIP(hipEventRecord(ws.event_end, stream));
            UA_HIP(hipEventSynchronize(ws.event_end));
            float ms = 0.f;
            UA_HIP(hipEventElapsedTime(&ms, ws.event_begin, ws.event_end));
            call.total_ms += ms;
        }
        return nullptr;
    };

    const std::uint32_t pending = call.have_todo ? (std::uint32_t)call.todo.size() : (std::uint32_t)call.count;
    // a team's workgroup adds its shared control block (16-byte alignment + 64 bytes) to the leader's areas: a size that only just
    // fits the budget alone must not become a launch failure — such a batch walks with one wave per query
    const std::uint32_t team_bytes = team_block_bytes_k;
    if (params.team && call.mode != scratch_global_k &&
        (lds_bytes_for(call.mode, call.next_cap, call.hash_cap) + 15) / 16 * 16 + team_bytes > lds_budget) {
        params.team = 0;
        call.stats.variant = (std::uint32_t)params.variant + 1;
    }
    if (call.mode != scratch_global_k) {
        if (lds_bytes_for(call.mode, call.next_cap, call.hash_cap) > lds_budget) {
            if (call.mode == scratch_lds_k)
                call.mode = scratch_hash_k;
            while (call.next_cap > 64 && lds_bytes_for(call.mode, call.next_cap, call.hash_cap) > lds_budget)
                call.next_cap /= 2;
            if (lds_bytes_for(call.mode, call.next_cap, call.hash_cap) > lds_budget)
                call.mode = scratch_global_k; // `top` alone does not fit LDS: straight to the global fallback
        }
    }
    if (call.mode != scratch_global_k) {
        // a team's workgroup carries the leader's LDS areas plus the shared block; one workgroup per query of the small batch
        std::uint64_t wave_lds_bytes = lds_bytes_for(call.mode, call.next_cap, call.hash_cap);
        // short rows over a global visited set: `seen` cells in LDS in front of it (kernels.hpp `search_one`) — as many as cost no
        // resident wave (the walk lives on its residency), at most 2 048; USEARCH_AMD_SEEN_CELLS forces a number (0 = none)
        args.seen_offset = 0, args.seen_cells = 0;
        args.aside_offset = 0, args.aside_cells = 0;
        args.probe_mode = probe_swap_k, args.claim_offset = 0, args.claim_bits = 0;
        // rows of ≤ 128 bytes gathered next to the probe of the visited set instead of behind it (kernels.hpp, the hop loop)
        args.early_rows = call.mode == scratch_hash_k && !params.team && lanes_ == 2 && env_size("USEARCH_AMD_EARLY_ROWS", default_early_rows_k) ? 1u : 0u;
        call.stats.early_rows = args.early_rows;
        if (call.mode == scratch_hash_k && !params.team && lanes_ <= 2) {
            // how the slab is probed (common.hpp `probe_mode_t`): USEARCH_AMD_PROBE_MODE = 0 | 1 | 2
            // (USEARCH_AMD_PROBE_LOAD_FIRST=1, round 5's name for mode 1, still answers)
            std::size_t probe_mode = default_probe_mode_k;
#ifdef USEARCH_AMD_EXPERIMENT_PROBE_MODES // `make EXTRA=-DUSEARCH_AMD_EXPERIMENT_PROBE_MODES OUT=… OBJ=…`: the copy scripts/probe_mode_check.py loads
            probe_mode = env_size("USEARCH_AMD_PROBE_MODE", default_probe_mode_k);
            if (env_size("USEARCH_AMD_PROBE_LOAD_FIRST", 0))
                probe_mode = probe_load_first_k;
#endif
            if (probe_mode == probe_plain_k) {
                // one claim bit per cell of the slab where that costs no resident wave, else as many as do not (a smaller bitmap only
                // adds false alarms: a lane that loses a claim looks at its cell again); USEARCH_AMD_CLAIM_BITS forces a number
                std::uint32_t bits = call.hash_cap;
                const std::size_t forced = env_size("USEARCH_AMD_CLAIM_BITS", 0);
                if (forced)
                    for (bits = 64; bits * 2 <= forced && bits < call.hash_cap; bits *= 2) {}
                else
                    while (bits > 512 && waves_for((wave_lds_bytes + 15) / 16 * 16 + bits / 8) < waves_for(wave_lds_bytes))
                        bits /= 2;
                if ((wave_lds_bytes + 15) / 16 * 16 + bits / 8 <= lds_budget) {
                    args.probe_mode = (std::uint32_t)probe_mode;
                    args.claim_offset = (std::uint32_t)((wave_lds_bytes + 15) / 16 * 16);
                    args.claim_bits = bits;
                    wave_lds_bytes = args.claim_offset + bits / 8ull;
                }
            } else if (probe_mode == probe_load_first_k) {
                args.probe_mode = probe_load_first_k;
            }
            // a plain `search` batch runs the build without the features it never uses (kernels.hpp `plain_ak`); the engine vouches here
            // for everything that build takes for granted (USEARCH_AMD_NO_PLAIN=1 keeps the general build). That build never probes the
            // slab past a member's home cell and sets what collides aside in LDS: about visits² / (2 · cells of the slab) members,
            // visits ≈ 20 · expansion + 800 on the measured shapes (20M × 128 b1 at 64: median 1 521, maximum 2 225 of 100 000 queries;
            // 20M × 96 i8 at 80: 1 742 / 2 281) — 512 cells at three quarters' load must take them, and must cost no resident wave;
            // a query that outgrows them all the same is run again by the retry ladder
            args.aside_offset = 0, args.aside_cells = 0;
            const bool plain_wanted = call.plain_possible && !args.query_ids && !args.beam_level && !args.descent_only && !args.allow_bits &&
                                      !args.exclude_own && args.probe_mode == probe_swap_k && (lanes_ == 1 || args.early_rows != 0);
#ifdef USEARCH_AMD_EXPERIMENT_NO_ASIDE
            const bool aside_wanted = false;
#else
            const bool aside_wanted = plain_wanted && lanes_ == 2; // (rows that travel with the lists gain nothing from it: kernels.hpp)
#endif
            if (aside_wanted) {
                std::uint32_t aside_cells = 512;
                if (const std::size_t forced_cells = env_size("USEARCH_AMD_ASIDE_CELLS", 0)) // tests: a table that is sure to fill up
                    for (aside_cells = 64; aside_cells * 2 <= forced_cells && aside_cells < 2048; aside_cells *= 2) {}
                const std::uint64_t expected_visits = std::min<std::uint64_t>((std::uint64_t)call.ef * 20 + 800, view_.size);
                const bool room = expected_visits * expected_visits / (2ull * call.hash_cap) <= aside_cells * 3ull / 4 ||
                                  env_size("USEARCH_AMD_PLAIN_WHATEVER_THE_ROOM", 0); // tests: a query that outgrows `aside` goes up the retry ladder
                const std::uint64_t with_aside = (wave_lds_bytes + 15) / 16 * 16 + aside_cells * 4ull;
                if (room && waves_for(with_aside) >= waves_for(wave_lds_bytes) && with_aside <= lds_budget) {
                    args.aside_offset = (std::uint32_t)((wave_lds_bytes + 15) / 16 * 16);
                    args.aside_cells = aside_cells;
                    wave_lds_bytes = with_aside;
                }
            }
            const std::size_t forced = env_size("USEARCH_AMD_SEEN_CELLS", (std::size_t)-1);
            std::uint32_t cells = 0;
            if (forced != (std::size_t)-1) {
                for (cells = 1; cells * 2 <= forced && cells < 8192; cells *= 2) {}
                cells = forced ? cells : 0;
            } else {
                for (std::uint32_t candidate = 2048; candidate >= 128 && !cells; candidate /= 2)
                    if (waves_for((wave_lds_bytes + 15) / 16 * 16 + candidate * 4ull) >= waves_for(wave_lds_bytes))
                        cells = candidate;
            }
            if (cells && (wave_lds_bytes + 15) / 16 * 16 + cells * 4ull <= lds_budget) {
                args.seen_offset = (std::uint32_t)((wave_lds_bytes + 15) / 16 * 16);
                args.seen_cells = cells;
                wave_lds_bytes = args.seen_offset + cells * 4ull;
            }
            call.stats.probe_mode = args.probe_mode;
            call.stats.seen_cells = args.seen_cells;
            call.stats.claim_bits = args.claim_bits;
        }
        // decided with the LDS areas above (short rows over the global hash only): what the instantiation takes for granted must be there
        params.plain = 0;
        if (call.mode == scratch_hash_k && !params.team && lanes_ <= 2 && call.plain_possible && !args.query_ids && !args.beam_level &&
            !args.descent_only && !args.allow_bits && !args.exclude_own && args.probe_mode == probe_swap_k && (lanes_ == 1 || args.early_rows != 0)) {
#ifdef USEARCH_AMD_EXPERIMENT_NO_ASIDE
            params.plain = args.seen_cells ? 1u : 0u;
#else
            params.plain = (lanes_ == 2 ? args.aside_cells : args.seen_cells) ? 1u : 0u;
#endif
        }
        call.stats.plain = params.plain;
        call.stats.aside_cells = args.aside_cells;
        const std::uint64_t lds_bytes = params.team ? (wave_lds_bytes + 15) / 16 * 16 + team_bytes : wave_lds_bytes;
        args.team_offset = params.team ? (std::uint32_t)((wave_lds_bytes + 15) / 16 * 16) : 0u;
        const std::uint32_t grid = params.team ? pending
                                               : (std::uint32_t)std::min<std::uint64_t>(pending, (std::uint64_t)waves_for(lds_bytes) * compute_units_);
        const std::uint64_t slab = call.mode == scratch_hash_k ? (std::uint64_t)call.hash_cap * 4 : 0;
        // WHERE the block of visited-set slabs lands decides which of the walk's speeds this batch runs at (with the index arrays
        // untouched, a fresh 268-MB block flips the headline batch between 45.5 and 51.6 ms; profiles/r03_placement/), and no
        // synthetic probe tells the placements apart — only the walk itself does. So when a new block is needed for a launch that
        // fills the chip, a few placements are drawn side by side and each is timed by THIS launch over its first queries (one per
        // wave; their results are simply computed again by the launch proper); the fastest block stays with the workspace.
        const std::size_t scratch_draws = std::min<std::size_t>(8, env_size("USEARCH_AMD_SCRATCH_DRAWS", 8));
        const bool draw_scratch = slab * grid > ws.scratch_bytes && slab * grid >= ((std::uint64_t)8 << 20) && scratch_draws > 1 &&
                                  pending >= 2ull * grid && grid >= 2u * (std::uint32_t)compute_units_ &&
                                  !env_size("USEARCH_AMD_SCRATCH_REDRAW", 0);
        if (const char* e = ws.reserve(call.count, draw_scratch ? 0 : slab * grid))
            return e;
        args.status = ws.d_status, args.peaks = ws.d_peaks;
        args.hash_cap = call.hash_cap;
        args.next_cap = call.next_cap;
        args.todo = call.have_todo ? ws.d_todo : nullptr;
        args.count = pending;
        args.scratch = ws.d_scratch;
        args.scratch_stride = slab;
        args.wave_clock = nullptr;
        if (draw_scratch) {
            params.mode = call.mode;
            params.entries_per_lane = call.entries_per_lane;
            params.grid = grid;
            params.lds_bytes = (std::uint32_t)lds_bytes;
            hipEvent_t begin = nullptr, end = nullptr;
            UA_HIP(hipEventCreate(&begin));
            if (hipError_t created = hipEventCreate(&end); created != hipSuccess) {
                (void)hipEventDestroy(begin);
                return hip_message(created);
            }
            void* candidates[8] = {nullptr};
            float trial_ms[8] = {0};
            std::size_t drawn = 0;
            const char* failure = nullptr;
            auto trial = [&](void* block, float& ms) -> const char* { // the launch's first `grid` queries over this block, second run timed
                args.scratch = static_cast<std::uint8_t*>(block);
                args.count = grid;
                for (int repeat = 0; repeat < 2; ++repeat) {
                    hipError_t e = hipMemsetAsync(ws.d_queue, 0, 8, stream);
                    if (e == hipSuccess)
                        e = hipEventRecord(begin, stream);
                    if (e == hipSuccess)
                        e = launch_search(metric_, scalar_, params, view_, args);
                    if (e == hipSuccess)
                        e = hipEventRecord(end, stream);
                    if (e == hipSuccess)
                        e = hipEventSynchronize(end);
                    if (e == hipSuccess)
                        e = hipEventElapsedTime(&ms, begin, end);
                    if (e != hipSuccess)
                        return hip_message(e);
                }
                return nullptr;
            };
            // diagnostic (USEARCH_AMD_SCRATCH_REMAP = n): ONE physical block mapped at n fresh virtual ranges, each view timed the
            // same way — does the speed follow the physical pages or the mapping (its page tables)?
            if (const std::size_t views = env_size("USEARCH_AMD_SCRATCH_REMAP", 0)) {
                std::vector<float> view_ms;
                if (const char* e = remap_trial(slab * grid, views, [&](void* view, float& ms) { return trial(view, ms); }, view_ms))
                    std::fprintf(stderr, "[usearch_amd] remap trial: %s\n", e);
                std::fprintf(stderr, "[usearch_amd] one physical block of %.0f MB under %zu mappings:", slab * grid / 1e6, view_ms.size());
                for (float ms : view_ms)
                    std::fprintf(stderr, " %.3f", ms);
                std::fprintf(stderr, " ms\n");
            }
            for (; drawn < scratch_draws && !failure; ++drawn) {
                if (scratch_malloc(&candidates[drawn], slab * grid) != hipSuccess) {
                    (void)hipGetLastError();
                    candidates[drawn] = nullptr;
                    break;
                }
                // one query per wave: the launch's steady state, a single query's latency long; the first run of a block also pays
                // its first touch
                failure = trial(candidates[drawn], trial_ms[drawn]);
            }
            (void)hipEventDestroy(begin);
            (void)hipEventDestroy(end);
            std::size_t kept = 0;
            for (std::size_t i = 1; i < drawn; ++i)
                if (trial_ms[i] < trial_ms[kept])
                    kept = i;
            for (std::size_t i = 0; i < drawn; ++i)
                if (i != kept || failure || !drawn)
                    placed_free(candidates[i]);
            if (failure)
                return failure;
            if (!drawn)
                return hip_message(hipErrorOutOfMemory);
            placed_free(ws.d_scratch);
            ws.d_scratch = static_cast<std::uint8_t*>(candidates[kept]);
            ws.scratch_bytes = slab * grid;
            args.scratch = ws.d_scratch;
            args.count = pending;
            if (env_size("USEARCH_AMD_PLACEMENT_LOG", 0)) {
                std::fprintf(stderr, "[usearch_amd] scratch placement of %.0f MB, timed by the launch's first %u queries: ", slab * grid / 1e6, grid);
                for (std::size_t i = 0; i < drawn; ++i)
                    std::fprintf(stderr, "%s%.3f%s@%p", i ? " " : "", trial_ms[i], i == kept ? "*" : "", candidates[i]);
                std::fprintf(stderr, " ms\n");
            }
        }
        // ---- the matrix of stored rows: up to `placement_max_draws_k` trials over the first launches that fill the chip, judged like
        //      the scratch block — by this launch's own first queries at the caller's expansion (placement.hpp)
        //      OFF unless USEARCH_AMD_PLACEMENT_DRAWS = 2 … 8 asks for them (round 6): the matrix is placed once, at load time, after
        //      the settle window (placement.hpp) — deterministic, no second copy of the matrix in HBM during a search call, nothing
        //      swapped under a reader. The trials stay for hosts whose matrix was allocated while other gigabytes were held.
        const std::uint32_t matrix_draws = tuning_trials_ ? tuning_trials_ + 1u + placement_.draws
                                                          : (std::uint32_t)std::min<std::size_t>(placement_max_draws_k, env_size("USEARCH_AMD_PLACEMENT_DRAWS", 1));
        // A host that tunes its expansion walks up through the regimes (bench.py's recall sweep: 64, 96, 128 … 608): trials judged at a
        // small expansion — differences of hundredths of a millisecond — must not be the last word for launches several times as
        // wide. A launch more than twice as wide as the last trial's reopens a search that has ended, for three trials, twice at most.
        bool try_placement = false;
        if (matrix_draws > 1) { // the trials' bookkeeping is shared by every batch in flight: under the pool's mutex
            std::lock_guard<std::mutex> lock(pool_mutex_);
            if (!placement_trials_left_ && placement_reopens_ < 2 && placement_last_ef_ && call.ef > 2u * placement_last_ef_ &&
                call.passes == 0 && !call.have_todo && pending >= 2ull * grid)
                placement_trials_left_ = 3, placement_losses_ = 0, ++placement_reopens_;
            try_placement = placement_trials_left_ != 0;
        }
        if (try_placement && placement_.draws < matrix_draws + 3u * placement_reopens_ && call.passes == 0 && !call.have_todo &&
            !params.team && pending >= 2ull * grid && grid >= 2u * (std::uint32_t)compute_units_ && !view_.nbr0_rows && d_vectors_ &&
            vectors_bytes_ >= env_size("USEARCH_AMD_PLACEMENT_MIN_BYTES", (std::size_t)1 << 30) && !args.query_ids && !args.allow_bits &&
            !args.descent_only && !args.beam_level && !env_size("USEARCH_AMD_SCRATCH_REDRAW", 0)) {
            params.mode = call.mode;
            params.entries_per_lane = call.entries_per_lane;
            params.grid = grid;
            params.lds_bytes = (std::uint32_t)lds_bytes;
            hipEvent_t begin = nullptr, end = nullptr;
            UA_HIP(hipEventCreate(&begin));
            if (hipError_t created = hipEventCreate(&end); created != hipSuccess) {
                (void)hipEventDestroy(begin);
                return hip_message(created);
            }
            args.count = grid; // one query per wave: the launch's steady state; their results are computed again by the launch proper
            const char* failure = try_matrix_placement(
                call.ef,
                [&](const snapshot_view_t& view, float& ms) -> const char* {
                    hipError_t e = hipMemsetAsync(ws.d_queue, 0, 8, stream);
                    if (e == hipSuccess)
                        e = hipEventRecord(begin, stream);
                    if (e == hipSuccess)
                        e = launch_search(metric_, scalar_, params, view, args);
                    if (e == hipSuccess)
                        e = hipEventRecord(end, stream);
                    if (e == hipSuccess)
                        e = hipEventSynchronize(end);
                    if (e == hipSuccess)
                        e = hipEventElapsedTime(&ms, begin, end);
                    return e == hipSuccess ? nullptr : hip_message(e);
                },
                stream);
            args.count = pending;
            (void)hipEventDestroy(begin);
            (void)hipEventDestroy(end);
            if (failure)
                return failure;
        }
        if (call.want_clock && call.passes == 0) {
            if (const char* e = ws.reserve_wave_clock(grid))
                return e;
            args.wave_clock = ws.d_wave_clock;
        }
        params.mode = call.mode;
        params.entries_per_lane = call.entries_per_lane;
        params.grid = grid;
        params.lds_bytes = (std::uint32_t)lds_bytes;
        if (const char* e = timed_launch())
            return e;
        ++call.passes;
        return nullptr;
    }

    // ---- last rung: global-memory scratch — exact sizes (one bit per slot, one frontier cell per slot), cannot overflow;
    //      the reference's heap (a frontier in `top` needs `top` in registers)
    if (!call.have_todo) {
        call.todo.resize(call.count);
        for (std::uint32_t q = 0; q < call.count; ++q)
            call.todo[q] = q;
        call.have_todo = true;
    }
    call.stats.retried_global = (std::uint32_t)call.todo.size();
    const std::uint64_t bitmap_bytes = ((view_.size + 31) / 32) * 4;
    const std::uint32_t frontier_cells = (std::uint32_t)std::min<std::uint64_t>(view_.size + 64, 0xFFFFFFF0u);
    const scratch_layout_t layout = scratch_layout(ef, frontier_cells, bitmap_bytes);
    const std::size_t slab = (layout.total + 255) & ~(std::size_t)255;
    const std::size_t budget = env_size("USEARCH_AMD_GLOBAL_SCRATCH_BYTES", (std::size_t)2 << 30);
    const std::size_t waves = std::max<std::size_t>(1, std::min<std::size_t>(call.todo.size(), budget / slab));
    if (const char* e = ws.reserve(call.count, waves * slab))
        return e;
    for (std::size_t begin = 0; begin < call.todo.size(); begin += waves) {
        const std::size_t chunk = std::min(waves, call.todo.size() - begin);
        UA_HIP(hipMemcpyAsync(ws.d_todo, call.todo.data() + begin, chunk * 4, hipMemcpyHostToDevice, stream));
        // only the bitmaps need zeroing
        UA_HIP(hipMemset2DAsync(ws.d_scratch + layout.visits, slab, 0, bitmap_bytes, chunk, stream));
        args.status = ws.d_status, args.peaks = ws.d_peaks;
        args.hash_cap = 0;
        args.next_cap = frontier_cells;
        args.todo = ws.d_todo;
        args.count = (std::uint32_t)chunk;
        args.scratch = ws.d_scratch;
        args.scratch_stride = slab;
        args.wave_clock = nullptr;
        params.mode = scratch_global_k;
        params.team = 0;
        params.plain = 0;
        params.entries_per_lane = 0;
        params.frontier = frontier_heap_k;
        params.grid = (std::uint32_t)chunk;
        params.lds_bytes = call.query_lds;
        if (const char* e = timed_launch(/*keep_overflows=*/begin != 0))
            return e;
        ++call.passes;
        UA_HIP(hipStreamSynchronize(stream)); // the todo block is reused by the next chunk
    }
    return nullptr;
}

const char* snapshot_t::search_finish(search_call_t& call, search_stats_t* stats) {
    if (stats)
        *stats = search_stats_t{};
    if (!call.workspace)
        return nullptr; // nothing was wanted
    struct hand_back_t { // on every path out of here the workspace returns to the pool, unless the caller keeps it
        snapshot_t& owner;
        search_call_t& call;
        ~hand_back_t() {
            if (!call.keep_workspace)
                owner.give_back(call.workspace);
            call.workspace = nullptr;
        }
    } hand_back{*this, call};
    workspace_t& ws = *call.workspace;
    hipStream_t stream = call.stream;
    UA_HIP(hipSetDevice(device_));
    if (call.done) {
        UA_HIP(hipStreamSynchronize(stream));
        return nullptr;
    }

    const std::uint32_t first_grid = call.params.grid, first_lds = call.params.lds_bytes;
    const int first_mode = call.params.mode;
    // ---- the ladder: did any query outgrow its scratch? One 8-byte read-back answers that; the per-query status is only
    //      fetched when the answer is yes. Second rung: visited set in the global hash, 4× the room for both structures.
    //      Third rung: global-memory scratch of exact size.
    for (int rung = 0;; ++rung) {
        UA_HIP(hipMemcpyAsync(ws.h_status, ws.d_queue, 8, hipMemcpyDeviceToHost, stream));
        UA_HIP(hipStreamSynchronize(stream));
        if (call.params.mode == scratch_global_k) {
            if (ws.h_status[1])
                return "Search scratch overflow in the global-memory pass";
            break;
        }
        if (!ws.h_status[1])
            break;
        UA_HIP(hipMemcpyAsync(ws.h_status + 16, ws.d_status, call.count * 4, hipMemcpyDeviceToHost, stream));
        UA_HIP(hipStreamSynchronize(stream));
        std::vector<std::uint32_t> again;
        if (call.have_todo) {
            for (std::uint32_t q : call.todo)
                if (ws.h_status[16 + q] == status_overflow_k)
                    again.push_back(q);
        } else {
            for (std::uint32_t q = 0; q < call.count; ++q)
                if (ws.h_status[16 + q] == status_overflow_k)
                    again.push_back(q);
        }
        call.todo.swap(again);
        call.have_todo = true;
        call.reran = true;
        if (call.todo.empty())
            break;
        if (rung == 0) {
            call.stats.retried_lds = (std::uint32_t)call.todo.size();
            call.mode = scratch_hash_k;
            call.hash_cap = std::min<std::uint32_t>(call.hash_cap * 4, pow2_ceil((std::uint32_t)std::min<std::uint64_t>(view_.size * 2 + 128, 1u << 30)));
            if (call.next_cap) {
                const std::uint32_t lds_budget = (std::uint32_t)env_size("USEARCH_AMD_LDS_BUDGET", 160 * 1024);
                call.next_cap = (std::uint32_t)std::min<std::uint64_t>((std::uint64_t)call.next_cap * 4, view_.size + 64);
                while (call.next_cap > 64 &&
                       call.query_lds + scratch_layout(call.entries_per_lane ? 0 : call.ef, call.next_cap, 0).total > lds_budget)
                    call.next_cap = call.next_cap * 3 / 4;
            }
        } else {
            call.mode = scratch_global_k;
        }
        if (call.mode != scratch_global_k)
            UA_HIP(hipMemcpyAsync(ws.d_todo, call.todo.data(), call.todo.size() * 4, hipMemcpyHostToDevice, stream));
        if (const char* e = run_ladder(call))
            return e;
    }

    if (call.want_phases) {
        unsigned long long ticks[16] = {0};
        UA_HIP(hipMemcpy(ticks, call.args.phases, 128, hipMemcpyDeviceToHost));
        double total = 0;
        for (int i = 0; i < 6; ++i)
            total += (double)ticks[i];
        std::fprintf(stderr, "[usearch_amd] phases ef=%u grid=%u: setup %.1f%% pop+list %.1f%% visited %.1f%% distances %.1f%% "
                             "commit %.1f%% [heap push %.1f%% top insert %.1f%%, %llu candidates rechecked] dump %.1f%% (%.3g ticks); "
                             "frontier pushes %llu, lists ready ahead %llu\n",
                     call.ef, first_grid, 100 * ticks[0] / total, 100 * ticks[1] / total, 100 * ticks[2] / total,
                     100 * ticks[3] / total, 100 * ticks[4] / total, 100 * ticks[8] / total, 100 * ticks[9] / total, ticks[10],
                     100 * ticks[5] / total, total, ticks[6], ticks[7]);
        if (ticks[15])
            std::fprintf(stderr, "[usearch_amd] a team's helpers: %llu hops, ticks measuring per hop %.0f %.0f %.0f %.0f\n", ticks[15],
                         (double)ticks[11] / (double)ticks[15], (double)ticks[12] / (double)ticks[15],
                         (double)ticks[13] / (double)ticks[15], (double)ticks[14] / (double)ticks[15]);
    }
    if (call.want_clock && ws.d_wave_clock && first_grid) {
        // batch tail: between the first wave's start and the last wave's exit, how much wave-time was spent gone?
        std::vector<unsigned long long> clock((std::size_t)first_grid * 2);
        UA_HIP(hipMemcpy(clock.data(), ws.d_wave_clock, clock.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull, last = 0;
        for (std::uint32_t w = 0; w < first_grid; ++w)
            first = std::min(first, clock[2 * w]), last = std::max(last, clock[2 * w + 1]);
        double gone = 0;
        for (std::uint32_t w = 0; w < first_grid; ++w)
            gone += (double)(clock[2 * w] - first) + (double)(last - clock[2 * w + 1]);
        const double span = (double)(last - first);
        call.stats.tail_idle = span > 0 ? (float)(gone / (span * first_grid)) : 0.f;
        call.stats.span_ms = (float)(span / 1e5); // 100 MHz
        if (env_size("USEARCH_AMD_WAVE_CLOCK", 0) > 1) { // histogram of exit times, in tenths of the span
            unsigned long long bins[10] = {0};
            for (std::uint32_t w = 0; w < first_grid; ++w)
                ++bins[std::min<std::size_t>(9, (std::size_t)(10.0 * (double)(clock[2 * w + 1] - first) / std::max(span, 1.0)))];
            std::fprintf(stderr, "[usearch_amd] wave exits by tenth of the %.3f ms span (grid %u):", call.stats.span_ms, first_grid);
            for (unsigned long long b : bins)
                std::fprintf(stderr, " %llu", b);
            std::fprintf(stderr, "; idle share %.4f\n", call.stats.tail_idle);
        }
    }
    call.stats.passes = call.passes;
    call.stats.kernel_ms = call.total_ms;
    call.stats.mode = (std::uint32_t)first_mode + 1;
    call.stats.grid = first_grid;
    call.stats.lds_bytes = first_lds;
    if (stats)
        *stats = call.stats;
    return nullptr;
}

const char* snapshot_t::search_device(const void* queries, std::size_t count, std::size_t stride_bytes,
                                      std::size_t wanted, std::size_t expansion, std::uint64_t* keys,
                                      float* distances, std::uint64_t* counts, std::uint64_t* visited,
                                      std::uint64_t* computed, hipStream_t stream, const search_tuning_t& tuning,
                                      search_stats_t* stats, bool timed, const search_extras_t* extras) {
    if (stats)
        *stats = search_stats_t{};
    search_call_t call;
    const char* error = search_begin(call, queries, count, stride_bytes, wanted, expansion, keys, distances, counts, visited,
                                     computed, stream, tuning, timed, extras);
    if (error) {
        if (call.workspace)
            give_back(call.workspace);
        return error;
    }
    return search_finish(call, stats);
}

const char* snapshot_t::tune(const void* queries, std::size_t count, std::size_t stride_bytes, std::size_t wanted, std::size_t expansion,
                             std::uint32_t max_trials, std::uint32_t* trials_made) {
    if (trials_made)
        *trials_made = 0;
    max_trials = std::min<std::uint32_t>(max_trials, (std::uint32_t)placement_max_draws_k);
    if (!count || !wanted || !max_trials || !d_vectors_)
        return nullptr;
    UA_HIP(hipSetDevice(device_));
    // the sample's results go nowhere: a scratch block for them
    struct block_t {
        void* p = nullptr;
        ~block_t() { (void)hipFree(p); }
    } results;
    const std::size_t pad = 256, keys_bytes = (count * wanted * 8 + pad - 1) / pad * pad, distances_bytes = (count * wanted * 4 + pad - 1) / pad * pad,
                      column = (count * 8 + pad - 1) / pad * pad;
    UA_HIP(hipMalloc(&results.p, keys_bytes + distances_bytes + 3 * column));
    std::uint8_t* base = static_cast<std::uint8_t*>(results.p);
    const std::uint32_t before = placement_.draws;
    {
        std::lock_guard<std::mutex> lock(pool_mutex_);
        tuning_trials_ = max_trials;
        placement_trials_left_ = max_trials, placement_losses_ = 0, placement_reopens_ = 2; // no reopening: this call is the search
    }
    const char* failure = nullptr;
    for (std::uint32_t launch = 0; launch < max_trials + 1u && !failure; ++launch) { // every chip-filling launch makes at most one trial
        failure = search_device(queries, count, stride_bytes, wanted, expansion, reinterpret_cast<std::uint64_t*>(base),
                                reinterpret_cast<float*>(base + keys_bytes), reinterpret_cast<std::uint64_t*>(base + keys_bytes + distances_bytes),
                                reinterpret_cast<std::uint64_t*>(base + keys_bytes + distances_bytes + column),
                                reinterpret_cast<std::uint64_t*>(base + keys_bytes + distances_bytes + 2 * column), nullptr, search_tuning_t{},
                                nullptr, false);
        std::lock_guard<std::mutex> lock(pool_mutex_);
        if (!placement_trials_left_ || placement_.draws == before + launch) // over, or this launch could not make one (sample too small …)
            break;
    }
    {
        std::lock_guard<std::mutex> lock(pool_mutex_);
        tuning_trials_ = 0;
        placement_trials_left_ = 0;
    }
    if (trials_made)
        *trials_made = placement_.draws - before;
    return failure;
}

const char* snapshot_t::last_peaks(std::uint32_t* out, std::size_t queries) {
    workspace_t* ws = nullptr;
    {
        std::lock_guard<std::mutex> lock(pool_mutex_);
        ws = last_used_;
    }
    if (!ws || !ws->d_peaks || queries > ws->last_count)
        return "No telemetry for that many queries";
    UA_HIP(hipSetDevice(device_));
    UA_HIP(hipMemcpy(out, ws->d_peaks, queries * 8, hipMemcpyDeviceToHost));
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
    if (query_kind != scalar_ && bytes_per_vector(query_kind, dims) == 0)
        return "Unsupported query scalar kind";
    const std::size_t source_bytes = query_kind != scalar_ ? bytes_per_vector(query_kind, dims) : bpv;
    if (count > 1 && stride_bytes < source_bytes)
        return "Query stride is smaller than one query";

    // One block on the device — queries | keys | distances | counts | visited | computed | predicate bits — mirrored by one
    // pinned block on the host: one upload, one launch, one download, one wait per call.
    auto pad = [](std::size_t b) { return (b + 255) & ~(std::size_t)255; };
    const std::size_t allow_words = allow_bits_host && view_.size ? (view_.size + 31) / 32 : 0;
    const std::size_t o_keys = pad(bpv * count), o_distances = o_keys + pad(count * wanted * 8),
                      o_counts = o_distances + pad(count * wanted * 4), o_visited = o_counts + pad(count * 8),
                      o_computed = o_visited + pad(count * 8), o_allow = o_computed + pad(count * 8),
                      total = o_allow + pad(allow_words * 4);
    UA_HIP(hipSetDevice(device_));
    search_call_t call;
    // the lease is taken here (not inside search_begin) because the staging blocks live in the workspace
    if (const char* e = take(call.workspace))
        return e;
    workspace_t& ws = *call.workspace;
    // every way out of this function drains the workspace's stream and hands the workspace back to the pool: a kernel must not
    // outlive the lease of the scratch it runs on, and a lease that is never returned starves `take` for good
    struct return_lease_t {
        snapshot_t& owner;
        workspace_t* workspace;
        ~return_lease_t() {
            (void)hipStreamSynchronize(workspace->stream);
            owner.give_back(workspace);
        }
    } return_lease{*this, call.workspace};
    call.keep_workspace = true; // search_finish leaves it to the guard above
    if (const char* e = ws.reserve_stage(total, o_allow))
        return e;

    // cast (or gather strided rows) straight into the pinned block, in the storage kind — index_dense.hpp:2058-2064
    const std::uint8_t* source = static_cast<const std::uint8_t*>(queries);
    auto fill = [&](std::uint64_t begin, std::uint64_t end) {
        for (std::uint64_t q = begin; q < end; ++q) {
            std::uint8_t* row = ws.h_stage + q * bpv;
            if (query_kind == scalar_ || !cast_vector(query_kind, scalar_, source + q * stride_bytes, dims, row))
                std::memcpy(row, source + q * stride_bytes, bpv);
        }
    };
    if (count >= 256)
        parallel_ranges(count, fill);
    else
        fill(0, count);

    hipStream_t stream = ws.stream;
    UA_HIP(hipMemcpyAsync(ws.d_stage, ws.h_stage, bpv * count, hipMemcpyHostToDevice, stream));
    search_extras_t extras = more ? *more : search_extras_t{};
    if (allow_words) {
        UA_HIP(hipMemcpyAsync(ws.d_stage + o_allow, allow_bits_host, allow_words * 4, hipMemcpyHostToDevice, stream));
        extras.allow_bits = reinterpret_cast<const std::uint32_t*>(ws.d_stage + o_allow);
    }
    std::uint64_t* d_keys = reinterpret_cast<std::uint64_t*>(ws.d_stage + o_keys);
    float* d_distances = reinterpret_cast<float*>(ws.d_stage + o_distances);
    std::uint64_t* d_counts = reinterpret_cast<std::uint64_t*>(ws.d_stage + o_counts);
    std::uint64_t* d_visited = reinterpret_cast<std::uint64_t*>(ws.d_stage + o_visited);
    std::uint64_t* d_computed = reinterpret_cast<std::uint64_t*>(ws.d_stage + o_computed);

    // `call` already holds the workspace: search_begin uses it instead of leasing another one
    if (const char* e = search_begin(call, ws.d_stage, count, bpv, wanted, expansion, d_keys, d_distances, d_counts, d_visited,
                                     d_computed, stream, tuning, false, &extras))
        return e;
    // results ride home behind the first launch; if a rung of the ladder re-runs queries they are fetched again
    auto download = [&]() -> const char* {
        UA_HIP(hipMemcpyAsync(ws.h_stage + o_keys, ws.d_stage + o_keys, o_allow - o_keys, hipMemcpyDeviceToHost, stream));
        return nullptr;
    };
    if (const char* e = download())
        return e;
    {
        // the pinned block is read after search_finish (wait + retry ladder): the workspace stays leased until the copies are out
        search_stats_t local;
        if (const char* e = search_finish(call, &local))
            return e;
        if (stats)
            *stats = local;
        if (call.reran) {
            const char* e = download();
            if (!e && hipStreamSynchronize(stream) != hipSuccess)
                e = "Failed to fetch the results";
            if (e)
                return e;
        }
    }
    if (keys)
        std::memcpy(keys, ws.h_stage + o_keys, count * wanted * 8);
    if (distances)
        std::memcpy(distances, ws.h_stage + o_distances, count * wanted * 4);
    if (counts)
        std::memcpy(counts, ws.h_stage + o_counts, count * 8);
    if (visited)
        std::memcpy(visited, ws.h_stage + o_visited, count * 8);
    if (computed)
        std::memcpy(computed, ws.h_stage + o_computed, count * 8);
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
                                float* distances, std::uint64_t* counts, hipStream_t stream, float* kernel_ms,
                                const std::uint32_t* allow_bits) {
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
        p.allow_bits = allow_bits;
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
                                     hipStream_t stream, float* kernel_ms, bool tiled, const std::uint32_t* allow_bits) {
    if (!count || !wanted)
        return nullptr;
    UA_HIP(hipSetDevice(device_));
    lease_t lease(*this);
    if (!stream) {
        if (const char* e = lease.take())
            return e;
        stream = lease.workspace->stream;
    }
    if (tiled)
        return exact_search_tiled_device(kernel_metric(metric_), scalar_, view_, queries, count, stride_bytes, wanted, true, keys,
                                         distances, counts, stream, kernel_ms, allow_bits);
    return exact_search_device(metric_, scalar_, lanes_, view_, queries, count, stride_bytes, wanted, true, keys,
                               distances, counts, stream, kernel_ms, allow_bits);
}

const char* snapshot_t::exact_host(const void* queries, scalar_kind_t query_kind, std::size_t count,
                                   std::size_t stride_bytes, std::size_t wanted, std::uint64_t* keys, float* distances,
                                   std::uint64_t* counts, float* kernel_ms, bool tiled, const std::uint32_t* allow_bits) {
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
    UA_HIP(hipSetDevice(device_));
    lease_t lease(*this);
    if (const char* e = lease.take())
        return e;
    workspace_t& ws = *lease.workspace;
    auto pad = [](std::size_t b) { return (b + 255) & ~(std::size_t)255; };
    if (const char* e = ws.reserve_stage(pad(bpv * count) + pad(count * wanted * 8) + pad(count * wanted * 4) + pad(count * 8), 0))
        return e;
    std::uint8_t* d_queries = ws.d_stage;
    std::uint64_t* d_keys = reinterpret_cast<std::uint64_t*>(d_queries + pad(bpv * count));
    float* d_distances = reinterpret_cast<float*>(reinterpret_cast<std::uint8_t*>(d_keys) + pad(count * wanted * 8));
    std::uint64_t* d_counts = reinterpret_cast<std::uint64_t*>(reinterpret_cast<std::uint8_t*>(d_distances) + pad(count * wanted * 4));
    UA_HIP(hipMemcpy(d_queries, dense.data(), bpv * count, hipMemcpyHostToDevice));
    if (tiled) {
        if (const char* e = exact_search_tiled_device(kernel_metric(metric_), scalar_, view_, d_queries, count, bpv, wanted,
                                                      true, d_keys, d_distances, d_counts, ws.stream, kernel_ms, allow_bits))
            return e;
    } else if (const char* e = exact_search_device(metric_, scalar_, lanes_, view_, d_queries, count, bpv, wanted, true, d_keys,
                                                   d_distances, d_counts, ws.stream, kernel_ms, allow_bits)) {
        return e;
    }
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
    std::uint32_t lanes = 1, row_stride = 16, row_chunks = 1;
    row_geometry(bpv, lanes, row_stride, row_chunks);
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
        view.chunks = row_chunks;
        view.bytes_per_vector = (std::uint32_t)bpv;
        view.dimensions = (std::uint32_t)dimensions;
        // i8: the matrix-unit kernel returns the very same bits (exact integer sums, same closing arithmetic, same tie order);
        // the float kinds only reach it when asked (tolerance instead of bit equality)
        if (scalar == scalar_i8_k && exact_tiled_available(metric, scalar, wanted) && queries_count >= 32 &&
            !env_size("USEARCH_AMD_NO_TILED_EXACT", 0))
            error = exact_search_tiled_device(metric, scalar, view, d_queries, queries_count, bpv, wanted, false, d_keys,
                                              d_distances, d_counts, nullptr, nullptr);
        else
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
    UA_HIP(hipSetDevice(device_));
    lease_t lease(*this);
    if (const char* e = lease.take())
        return e;
    hipStream_t stream_ = lease.workspace->stream;
    hipEvent_t event_begin_ = lease.workspace->event_begin, event_end_ = lease.workspace->event_end;
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
