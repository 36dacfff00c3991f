/**
 *  usearch_amd/csrc/placement.hip — placement draws for the arrays the walk gathers from (placement.hpp).
 */
#include "placement.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "host_util.hpp"

namespace usearch_amd {

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
    if (hipError_t e = hipMemAddressReserve(&base, padded, std::min<std::size_t>(chunk, (std::size_t)1 << 30), nullptr, 0); e != hipSuccess)
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

/// One placement. By default ONE physical allocation of the whole size, mapped into a reserved range (`mapped_malloc` with a
/// single chunk): five restores out of five walk at the fast speed that way, against two out of five for `hipMalloc` blocks and
/// for mappings made of 2-MB … 1-GB chunks (profiles/r03_placement/). USEARCH_AMD_VMM_CHUNK_MB: 0 = plain `hipMalloc`, n = chunks
/// of n MB. Falls back to `hipMalloc` when the mapping cannot be made.
hipError_t draw(void** out, std::size_t bytes) {
    const char* setting = std::getenv("USEARCH_AMD_VMM_CHUNK_MB");
    const std::size_t chunk_mb = setting && *setting ? (std::size_t)std::strtoull(setting, nullptr, 10) : ~(std::size_t)0;
    if (chunk_mb) {
        const std::size_t chunk = chunk_mb == ~(std::size_t)0 ? bytes : chunk_mb << 20;
        if (mapped_malloc(out, bytes, chunk) == hipSuccess)
            return hipSuccess;
        (void)hipGetLastError();
    }
    return hipMalloc(out, bytes);
}

} // namespace

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
                return;
            }
    }
    (void)hipFree(pointer);
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

hipError_t placed_malloc(void** out, std::size_t bytes, std::size_t row_bytes, placement_t* report) {
    placement_t local;
    placement_t& stats = report ? *report : local;
    stats = placement_t{};
    *out = nullptr;
    bytes = std::max<std::size_t>(bytes, 16);
    const std::size_t threshold = env_size("USEARCH_AMD_PLACEMENT_MIN_BYTES", (std::size_t)1 << 30);
    const int wanted = (int)std::min<std::size_t>(placement_max_draws_k, env_size("USEARCH_AMD_PLACEMENT_DRAWS", 1));
    if (bytes < threshold || wanted <= 1 || row_bytes < 16)
        return bytes < threshold ? hipMalloc(out, bytes) : draw(out, bytes);

    const auto started = std::chrono::steady_clock::now();
    void* candidates[placement_max_draws_k] = {nullptr};
    int drawn = 0;
    hipError_t result = hipSuccess;
    for (; drawn < wanted; ++drawn) {
        if (drawn) { // a further draw must fit NEXT to the ones held, with room to spare for the index's other arrays
            std::size_t free_bytes = 0, total_bytes = 0;
            if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || free_bytes < bytes + bytes / 2 + ((std::size_t)8 << 30))
                break;
        }
        void* p = nullptr;
        const hipError_t e = draw(&p, bytes);
        if (e != hipSuccess) {
            if (!drawn)
                result = e;
            else
                (void)hipGetLastError(); // out of room for one more: keep what there is
            break;
        }
        candidates[drawn] = p;
        float gbps = 0.f;
        if (gather_probe(p, bytes, row_bytes, &gbps) != hipSuccess)
            (void)hipGetLastError();
        stats.gather_gbps[drawn] = gbps;
    }
    if (result != hipSuccess)
        return result;
    int kept = 0;
    for (int i = 1; i < drawn; ++i)
        if (stats.gather_gbps[i] > stats.gather_gbps[kept])
            kept = i;
    for (int i = 0; i < drawn; ++i)
        if (i != kept)
            placed_free(candidates[i]);
    *out = candidates[kept];
    stats.draws = (std::uint32_t)drawn;
    stats.kept = (std::uint32_t)kept;
    stats.probe_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - started).count();
    if (std::getenv("USEARCH_AMD_PLACEMENT_LOG")) {
        std::fprintf(stderr, "[usearch_amd] placement of %.2f GB (rows of %zu B): ", bytes / 1e9, row_bytes);
        for (int i = 0; i < drawn; ++i)
            std::fprintf(stderr, "%s%.0f%s", i ? " " : "", stats.gather_gbps[i], i == kept ? "*" : "");
        std::fprintf(stderr, " GB/s, %.0f ms\n", stats.probe_ms);
    }
    return hipSuccess;
}

} // namespace usearch_amd
