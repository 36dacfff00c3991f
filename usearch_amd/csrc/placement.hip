/**
 *  usearch_amd/csrc/placement.hip — placement draws for the arrays the walk gathers from (placement.hpp).
 */
#include "placement.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "host_util.hpp"

namespace usearch_amd {

/// Releases smaller than this do not open a settle window (placement.hpp): workspaces' status arrays, staging buffers.
constexpr std::size_t settle_counts_from_k = (std::size_t)64 << 20;

namespace {

/// One wave gathers `rows_per_wave` random rows, four rows' loads in flight at a time, every lane 16 bytes of a row per load.
/// No row depends on another: this is what the memory system gives to random `row_bytes`-sized reads of this array.
__global__ __launch_bounds__(64) void gather_probe_kernel(const uint4* base, std::uint64_t rows, std::uint32_t chunks_per_row,
                                                          std::uint32_t rows_per_wave, std::uint32_t* sink) {
    const std::uint32_t lane = threadIdx.x;
    std::uint32_t state = blockIdx.x * 2654435761u + 0x9E3779B9u;
    std::uint32_t folded = 0;
    for (std::uint32_t r = 0; r < rows_per_wave; r += 4) {
        std::uint64_t row[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            state = state * 1664525u + 1013904223u;
            row[i] = ((std::uint64_t)(state ^ (state >> 15)) * rows) >> 32;
        }
        for (std::uint32_t c = lane; c < chunks_per_row; c += 64) {
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = base[row[i] * chunks_per_row + c];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                folded ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        }
    }
    if (folded == 0x5EED5EEDu) // never, as far as the compiler can tell
        *sink = folded;
}

/// Every lane reads 16 bytes of a random 4-KB page per load, eight independent loads in flight.
__global__ __launch_bounds__(64) void translation_probe_kernel(const std::uint8_t* base, std::uint64_t pages, std::uint32_t rounds,
                                                               std::uint32_t* sink) {
    std::uint32_t state = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 0x9E3779B9u;
    std::uint32_t folded = 0;
    for (std::uint32_t r = 0; r < rounds; ++r) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            state = state * 1664525u + 1013904223u;
            const std::uint64_t page = ((std::uint64_t)(state ^ (state >> 15)) * pages) >> 32;
            v[i] = *reinterpret_cast<const uint4*>(base + page * 4096 + ((state >> 3) & 0xFF0u));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            folded ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (folded == 0x5EED5EEDu)
        *sink = folded;
}

/// Every lane follows its own chain: the next row is a function of the bytes just read (whatever they are) and a counter.
__global__ __launch_bounds__(64) void latency_probe_kernel(const std::uint8_t* base, std::uint64_t rows, std::uint64_t row_bytes,
                                                           std::uint32_t steps, std::uint32_t* sink) {
    std::uint32_t state = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 0x9E3779B9u;
    for (std::uint32_t s = 0; s < steps; ++s) {
        const std::uint64_t row = ((std::uint64_t)(state ^ (state >> 15)) * rows) >> 32;
        const uint4 v = *reinterpret_cast<const uint4*>(base + row * row_bytes);
        state = (state + (v.x ^ v.y ^ v.z ^ v.w)) * 1664525u + 1013904223u + s;
    }
    if (state == 0x5EED5EEDu)
        *sink = state;
}

/// Arrays mapped through the virtual-memory API (USEARCH_AMD_VMM_CHUNK_MB): what it takes to release them.
struct mapped_t {
    void* base;
    std::size_t bytes;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::mutex mapped_mutex;
std::vector<mapped_t> mapped_arrays;

/// `bytes` of device memory as physical chunks of `chunk` bytes, each one allocation of the driver, mapped back to back into one
/// reserved virtual range aligned to the chunk size.
hipError_t mapped_malloc(void** out, std::size_t bytes, std::size_t chunk) {
    int device = 0;
    if (hipError_t e = hipGetDevice(&device); e != hipSuccess)
        return e;
    hipMemAllocationProp properties = {};
    properties.type = hipMemAllocationTypePinned;
    properties.location.type = hipMemLocationTypeDevice;
    properties.location.id = device;
    std::size_t granularity = 0;
    if (hipError_t e = hipMemGetAllocationGranularity(&granularity, &properties, hipMemAllocationGranularityRecommended); e != hipSuccess)
        return e;
    granularity = std::max<std::size_t>(granularity, 4096);
    chunk = (std::max(chunk, granularity) + granularity - 1) / granularity * granularity;
    const std::size_t padded = (bytes + chunk - 1) / chunk * chunk;
    void* base = nullptr;
    std::size_t alignment = granularity; // the largest power of two within the chunk, at most 1 GB
    while (alignment * 2 <= std::min<std::size_t>(chunk, (std::size_t)1 << 30))
        alignment *= 2;
    if (hipError_t e = hipMemAddressReserve(&base, padded, alignment, nullptr, 0); e != hipSuccess)
        return e;
    mapped_t record{base, padded, {}};
    hipError_t result = hipSuccess;
    for (std::size_t offset = 0; offset < padded && result == hipSuccess; offset += chunk) {
        hipMemGenericAllocationHandle_t handle;
        result = hipMemCreate(&handle, chunk, &properties, 0);
        if (result != hipSuccess)
            break;
        record.handles.push_back(handle);
        result = hipMemMap(static_cast<std::uint8_t*>(base) + offset, chunk, 0, handle, 0);
    }
    if (result == hipSuccess) {
        hipMemAccessDesc access = {};
        access.location = properties.location;
        access.flags = hipMemAccessFlagsProtReadWrite;
        result = hipMemSetAccess(base, padded, &access, 1);
    }
    if (result != hipSuccess) {
        (void)hipMemUnmap(base, padded);
        for (auto handle : record.handles)
            (void)hipMemRelease(handle);
        (void)hipMemAddressFree(base, padded);
        return result;
    }
    {
        std::lock_guard<std::mutex> lock(mapped_mutex);
        mapped_arrays.push_back(std::move(record));
    }
    *out = base;
    return hipSuccess;
}

/// One placement: a `hipMalloc` block, or — USEARCH_AMD_VMM_CHUNK_MB = n — physical chunks of n MB mapped back to back into a
/// reserved range (n larger than the array: one chunk). An experiment switch: which flavour an array is made of turned out not to
/// decide its speed (profiles/r03_placement/README.md).
hipError_t draw(void** out, std::size_t bytes) {
    const std::size_t chunk_mb = env_size("USEARCH_AMD_VMM_CHUNK_MB", 0);
    if (chunk_mb) {
        if (mapped_malloc(out, bytes, chunk_mb << 20) == hipSuccess)
            return hipSuccess;
        (void)hipGetLastError();
    }
    return block_malloc(out, bytes);
}

} // namespace

hipError_t block_malloc(void** out, std::size_t bytes) {
    // experiment switch: ONE physical allocation mapped at a virtual range aligned to its own size (at most 1 GB), so that the page
    // tables may describe it with fragments as large as the physical frames allow whatever address the runtime's allocator hands out
    if (env_size("USEARCH_AMD_ALIGNED_MAP", 0) && bytes >= ((std::size_t)2 << 20)) {
        if (mapped_malloc(out, bytes, bytes) == hipSuccess)
            return hipSuccess;
        (void)hipGetLastError();
    }
    if (env_size("USEARCH_AMD_CONTIGUOUS", 0) && bytes >= ((std::size_t)2 << 20)) {
        if (hipExtMallocWithFlags(out, bytes, hipDeviceMallocContiguous) == hipSuccess) {
            if (env_size("USEARCH_AMD_PLACEMENT_LOG", 0))
                std::fprintf(stderr, "[usearch_amd] %.0f MB physically contiguous @%p\n", bytes / 1e6, *out);
            return hipSuccess;
        }
        (void)hipGetLastError();
        if (env_size("USEARCH_AMD_PLACEMENT_LOG", 0))
            std::fprintf(stderr, "[usearch_amd] no contiguous range of %.0f MB: plain allocation\n", bytes / 1e6);
    }
    const hipError_t result = hipMalloc(out, bytes);
    if (result == hipSuccess && bytes >= ((std::size_t)64 << 20) && env_size("USEARCH_AMD_PLACEMENT_LOG", 0))
        std::fprintf(stderr, "[usearch_amd] %.0f MB @%p (virtual range aligned to %zu MB)\n", bytes / 1e6, *out,
                     (std::size_t)((reinterpret_cast<std::uintptr_t>(*out) & (~reinterpret_cast<std::uintptr_t>(*out) + 1)) >> 20));
    return result;
}

const char* remap_trial(std::size_t bytes, std::size_t views, const std::function<const char*(void*, float&)>& judge,
                        std::vector<float>& view_ms) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return "no device";
    hipMemAllocationProp properties = {};
    properties.type = hipMemAllocationTypePinned;
    properties.location.type = hipMemLocationTypeDevice;
    properties.location.id = device;
    std::size_t granularity = 0;
    if (hipMemGetAllocationGranularity(&granularity, &properties, hipMemAllocationGranularityRecommended) != hipSuccess)
        return "no allocation granularity";
    granularity = std::max<std::size_t>(granularity, 4096);
    const std::size_t padded = (bytes + granularity - 1) / granularity * granularity;
    hipMemGenericAllocationHandle_t handle;
    if (hipMemCreate(&handle, padded, &properties, 0) != hipSuccess)
        return (void)hipGetLastError(), "hipMemCreate failed";
    hipMemAccessDesc access = {};
    access.location = properties.location;
    access.flags = hipMemAccessFlagsProtReadWrite;
    const char* error = nullptr;
    std::vector<void*> held; // earlier views stay mapped, so that every further one gets a range (and page tables) of its own
    for (std::size_t v = 0; v < views && !error; ++v) {
        void* base = nullptr;
        if (hipMemAddressReserve(&base, padded, granularity, nullptr, 0) != hipSuccess) {
            error = "hipMemAddressReserve failed";
            break;
        }
        if (hipMemMap(base, padded, 0, handle, 0) != hipSuccess || hipMemSetAccess(base, padded, &access, 1) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipMemAddressFree(base, padded);
            error = "hipMemMap failed (a second mapping of one allocation?)";
            break;
        }
        held.push_back(base);
        float ms = 0.f;
        error = judge(base, ms);
        view_ms.push_back(ms);
    }
    (void)hipDeviceSynchronize();
    for (void* base : held) {
        (void)hipMemUnmap(base, padded);
        (void)hipMemAddressFree(base, padded);
    }
    (void)hipMemRelease(handle);
    return error;
}

void placed_free(void* pointer) {
    if (!pointer)
        return;
    {
        std::lock_guard<std::mutex> lock(mapped_mutex);
        for (std::size_t i = 0; i < mapped_arrays.size(); ++i)
            if (mapped_arrays[i].base == pointer) {
                mapped_t record = std::move(mapped_arrays[i]);
                mapped_arrays.erase(mapped_arrays.begin() + (std::ptrdiff_t)i);
                (void)hipMemUnmap(record.base, record.bytes);
                for (auto handle : record.handles)
                    (void)hipMemRelease(handle);
                (void)hipMemAddressFree(record.base, record.bytes);
                note_release(record.bytes);
                return;
            }
    }
    std::size_t bytes = 0;
    if (hipMemPtrGetInfo(pointer, &bytes) != hipSuccess) {
        (void)hipGetLastError();
        bytes = settle_counts_from_k; // a block of unknown size counts
    }
    (void)hipFree(pointer);
    note_release(bytes);
}

namespace {
std::mutex settle_mutex;
std::chrono::steady_clock::time_point last_release{}; // epoch = nothing released yet
bool released_once = false;
float settle_waited_ms = 0.f;
std::uint32_t settle_waits = 0;
} // namespace

void note_release(std::size_t bytes) {
    if (bytes < settle_counts_from_k)
        return;
    std::lock_guard<std::mutex> lock(settle_mutex);
    last_release = std::chrono::steady_clock::now();
    released_once = true;
}

float settle_before_placing() {
    const std::size_t window_ms = env_size("USEARCH_AMD_SETTLE_MS", 1000);
    std::chrono::steady_clock::time_point since;
    {
        std::lock_guard<std::mutex> lock(settle_mutex);
        if (!released_once || !window_ms)
            return 0.f;
        since = last_release;
    }
    const auto ready = since + std::chrono::milliseconds(window_ms);
    const auto now = std::chrono::steady_clock::now();
    if (now >= ready)
        return 0.f;
    (void)hipDeviceSynchronize(); // nothing of ours may still be running on what was freed; the wait below is for the DRIVER
    std::this_thread::sleep_until(ready);
    const float waited = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - now).count();
    {
        std::lock_guard<std::mutex> lock(settle_mutex);
        settle_waited_ms += waited;
        ++settle_waits;
    }
    if (env_size("USEARCH_AMD_PLACEMENT_LOG", 0))
        std::fprintf(stderr, "[usearch_amd] waited %.0f ms for freed frames to come back before placing an array\n", waited);
    return waited;
}

void settle_totals(float* milliseconds, std::uint32_t* waits) {
    std::lock_guard<std::mutex> lock(settle_mutex);
    if (milliseconds)
        *milliseconds = settle_waited_ms;
    if (waits)
        *waits = settle_waits;
}

hipError_t translation_probe(const void* base, std::size_t bytes, float* rate) {
    *rate = 0.f;
    const std::uint64_t pages = bytes / 4096;
    if (pages < 1024)
        return hipSuccess;
    static thread_local std::uint32_t* sink = nullptr;
    if (!sink)
        if (hipError_t e = hipMalloc((void**)&sink, 4); e != hipSuccess)
            return e;
    const std::uint32_t waves = 8192, rounds = 64;
    hipEvent_t begin = nullptr, end = nullptr;
    if (hipError_t e = hipEventCreate(&begin); e != hipSuccess)
        return e;
    if (hipError_t e = hipEventCreate(&end); e != hipSuccess) {
        (void)hipEventDestroy(begin);
        return e;
    }
    float best_ms = 0.f;
    hipError_t result = hipSuccess;
    for (int repeat = 0; repeat < 4 && result == hipSuccess; ++repeat) {
        (void)hipEventRecord(begin, nullptr);
        hipLaunchKernelGGL(translation_probe_kernel, dim3(waves), dim3(64), 0, nullptr, static_cast<const std::uint8_t*>(base), pages,
                           rounds, sink);
        (void)hipEventRecord(end, nullptr);
        result = hipEventSynchronize(end);
        float ms = 0.f;
        if (result == hipSuccess)
            result = hipEventElapsedTime(&ms, begin, end);
        if (repeat && (best_ms == 0.f || ms < best_ms))
            best_ms = ms;
    }
    (void)hipEventDestroy(begin);
    (void)hipEventDestroy(end);
    if (result == hipSuccess && best_ms > 0.f)
        *rate = (float)((double)waves * 64 * rounds * 8 / best_ms / 1e3);
    return result;
}

hipError_t latency_probe(const void* base, std::size_t bytes, std::size_t row_bytes, float* nanoseconds) {
    *nanoseconds = 0.f;
    row_bytes = std::max<std::size_t>(16, row_bytes / 16 * 16);
    const std::uint64_t rows = bytes / row_bytes;
    if (rows < 1024)
        return hipSuccess;
    static thread_local std::uint32_t* sink = nullptr;
    if (!sink)
        if (hipError_t e = hipMalloc((void**)&sink, 4); e != hipSuccess)
            return e;
    const std::uint32_t waves = 512, steps = 512; // two waves per compute unit: nothing queues behind anything
    hipEvent_t begin = nullptr, end = nullptr;
    if (hipError_t e = hipEventCreate(&begin); e != hipSuccess)
        return e;
    if (hipError_t e = hipEventCreate(&end); e != hipSuccess) {
        (void)hipEventDestroy(begin);
        return e;
    }
    float best_ms = 0.f;
    hipError_t result = hipSuccess;
    for (int repeat = 0; repeat < 4 && result == hipSuccess; ++repeat) {
        (void)hipEventRecord(begin, nullptr);
        hipLaunchKernelGGL(latency_probe_kernel, dim3(waves), dim3(64), 0, nullptr, static_cast<const std::uint8_t*>(base), rows,
                           (std::uint64_t)row_bytes, steps, sink);
        (void)hipEventRecord(end, nullptr);
        result = hipEventSynchronize(end);
        float ms = 0.f;
        if (result == hipSuccess)
            result = hipEventElapsedTime(&ms, begin, end);
        if (repeat && (best_ms == 0.f || ms < best_ms))
            best_ms = ms;
    }
    (void)hipEventDestroy(begin);
    (void)hipEventDestroy(end);
    if (result == hipSuccess && best_ms > 0.f)
        *nanoseconds = best_ms * 1e6f / steps;
    return result;
}

hipError_t gather_probe(const void* base, std::size_t bytes, std::size_t row_bytes, float* gbps) {
    *gbps = 0.f;
    const std::uint32_t chunks = (std::uint32_t)std::max<std::size_t>(1, row_bytes / 16);
    const std::uint64_t rows = bytes / ((std::size_t)chunks * 16);
    if (rows < 1024)
        return hipSuccess;
    static thread_local std::uint32_t* sink = nullptr;
    if (!sink)
        if (hipError_t e = hipMalloc((void**)&sink, 4); e != hipSuccess)
            return e;
    // ≈ 2 GB of rows per launch: long enough to be a bandwidth measurement, short enough for a load path (≈ 0.4 ms)
    const std::uint32_t waves = 8192;
    const std::uint32_t rows_per_wave =
        (std::uint32_t)std::min<std::uint64_t>(4096, std::max<std::uint64_t>(16, ((2ull << 30) / waves / (chunks * 16ull) + 3) / 4 * 4));
    hipEvent_t begin = nullptr, end = nullptr;
    if (hipError_t e = hipEventCreate(&begin); e != hipSuccess)
        return e;
    if (hipError_t e = hipEventCreate(&end); e != hipSuccess) {
        (void)hipEventDestroy(begin);
        return e;
    }
    float best_ms = 0.f;
    hipError_t result = hipSuccess;
    for (int repeat = 0; repeat < 4 && result == hipSuccess; ++repeat) { // the first launch warms clocks and the kernel's code
        (void)hipEventRecord(begin, nullptr);
        hipLaunchKernelGGL(gather_probe_kernel, dim3(waves), dim3(64), 0, nullptr, static_cast<const uint4*>(base), rows, chunks,
                           rows_per_wave, sink);
        (void)hipEventRecord(end, nullptr);
        result = hipEventSynchronize(end);
        float ms = 0.f;
        if (result == hipSuccess)
            result = hipEventElapsedTime(&ms, begin, end);
        if (repeat && (best_ms == 0.f || ms < best_ms))
            best_ms = ms;
    }
    (void)hipEventDestroy(begin);
    (void)hipEventDestroy(end);
    if (result == hipSuccess && best_ms > 0.f)
        *gbps = (float)((double)waves * rows_per_wave * chunks * 16.0 / best_ms / 1e6);
    return result;
}

hipError_t placed_malloc(void** out, std::size_t bytes, std::size_t, placement_t* report) {
    if (report)
        *report = placement_t{};
    *out = nullptr;
    bytes = std::max<std::size_t>(bytes, 16);
    if (bytes < env_size("USEARCH_AMD_PLACEMENT_MIN_BYTES", (std::size_t)1 << 30))
        return block_malloc(out, bytes);
    const float waited = settle_before_placing(); // placement.hpp: settle, then allocate
    if (report)
        report->settle_ms = waited;
    return draw(out, bytes);
}

} // namespace usearch_amd
