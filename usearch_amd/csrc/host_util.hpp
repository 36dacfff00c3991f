/**
 *  usearch_amd/csrc/host_util.hpp — small host-side helpers shared by engine.hip and build.hip.
 */
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace usearch_amd {

inline const char* hip_message(hipError_t e) { return hipGetErrorString(e); } // static strings owned by the runtime

#define UA_HIP(call)                                                                                                   \
    do {                                                                                                               \
        hipError_t ua_error_ = (call);                                                                                 \
        if (ua_error_ != hipSuccess)                                                                                   \
            return hip_message(ua_error_);                                                                             \
    } while (0)

inline std::uint32_t pow2_ceil(std::uint32_t v) {
    std::uint32_t p = 1;
    while (p < v)
        p <<= 1;
    return p;
}

inline std::size_t env_size(const char* name, std::size_t fallback) {
    const char* v = std::getenv(name);
    return v && *v ? (std::size_t)std::strtoull(v, nullptr, 10) : fallback;
}

/// Runs `body(begin, end)` over [0, n) on the host's cores.
template <typename body_at> void parallel_ranges(std::uint64_t n, body_at&& body) {
    unsigned workers = std::thread::hardware_concurrency();
    workers = std::max(1u, std::min(workers, 64u));
    if (n < 4096 || workers == 1) {
        body(0, n);
        return;
    }
    std::vector<std::thread> pool;
    const std::uint64_t step = (n + workers - 1) / workers;
    for (unsigned w = 0; w < workers; ++w) {
        const std::uint64_t begin = std::min<std::uint64_t>(n, w * step), end = std::min<std::uint64_t>(n, begin + step);
        if (begin < end)
            pool.emplace_back([=, &body] { body(begin, end); });
    }
    for (auto& t : pool)
        t.join();
}


/// Re-pitches `rows` host rows of `bytes` bytes (source stride `source_stride`) into device rows of `row_stride` bytes.
const char* upload_rows(std::uint8_t* device, std::uint32_t row_stride, const std::uint8_t* source,
                        std::size_t source_stride, std::size_t bytes, std::uint64_t rows);

} // namespace usearch_amd
