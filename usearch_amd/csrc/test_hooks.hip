/**
 *  usearch_amd/csrc/test_hooks.hip — self-test and micro-benchmark entry points of the device-side containers. NOT part of the
 *  product: built into its own library (`usearch_amd/lib/libusearch_amd_testhooks.so`), which only tests/ and scripts/ load.
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "kernels.hpp"
#include "placement.hpp"

typedef char const* usearch_amd_error_t;
namespace {
void fail(usearch_amd_error_t* error, const char* message) {
    if (error && message)
        *error = message;
}
} // namespace

namespace usearch_amd {
/**
 *  Container self-test: replays a scripted sequence of heap pushes / pops and sorted inserts on the LDS containers
 *  and records what comes out, so the GPU tests can compare the tie behaviour with the oracle's containers directly.
 *  ops[i] = {kind, slot}: kind 0 = push(key = keys[i]), 1 = pop, 2 = sorted_insert(keys[i]) with `limit`.
 */
__global__ __launch_bounds__(64) void containers_kernel(const std::uint32_t* kinds, const float* keys,
                                                        const std::uint32_t* slots, std::uint32_t count,
                                                        std::uint32_t limit, std::uint32_t capacity,
                                                        std::uint64_t* popped, std::uint32_t* popped_count,
                                                        std::uint64_t* top_out, std::uint32_t* top_count) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    cand_t* heap = reinterpret_cast<cand_t*>(lds);
    cand_t* top = heap + capacity;
    std::uint32_t heap_size = 0, top_size = 0, pops = 0;
    for (std::uint32_t i = 0; i < count; ++i) {
        const std::uint32_t kind = kinds[i];
        if (kind == 0 && heap_size < capacity)
            heap_push<false>(heap, heap_size, keys[i], slots[i]);
        else if (kind == 1 && heap_size) {
            const cand_t c = heap_pop<false>(heap, heap_size);
            if (lane_id() == 0)
                popped[pops] = c;
            ++pops;
        } else if (kind == 2)
            sorted_insert<false>(top, top_size, limit, keys[i], slots[i]);
    }
    for (std::uint32_t i = lane_id(); i < top_size; i += 64)
        top_out[i] = top[i];
    if (lane_id() == 0)
        *popped_count = pops, *top_count = top_size;
}

/**
 *  Micro-benchmark of the register-resident `top`: every wave inserts `count` pseudo-random distances under the
 *  traversal's acceptance rule and reports its shader-clock ticks and how many were accepted (diagnostic only).
 */
template <int epl_ak>
__global__ __launch_bounds__(64) void top_bench_kernel(std::uint32_t count, std::uint32_t limit, unsigned long long* out) {
    top_gt<epl_ak, false> top;
    top.reset(nullptr);
    std::uint32_t state = 12345u + blockIdx.x * 977u;
    float radius = __builtin_inff();
    std::uint32_t accepted = 0;
    const std::uint64_t begin = __builtin_amdgcn_s_memtime();
    for (std::uint32_t i = 0; i < count; ++i) {
        state = state * 1664525u + 1013904223u;
        const float d = uniform_f32((float)(state >> 8) * (1.0f / 16777216.0f));
        if (top.size < limit || d < radius) {
            top.insert(d, i, limit, radius);
            ++accepted;
        }
    }
    const std::uint64_t end = __builtin_amdgcn_s_memtime();
    float checksum = 0.f; // keeps the buffer alive
#pragma unroll
    for (int r = 0; r < epl_ak; ++r)
        checksum += top.d[r] == __builtin_inff() ? 0.f : top.d[r];
    if (lane_id() == 0) {
        out[2 * blockIdx.x] = end - begin;
        out[2 * blockIdx.x + 1] = ((unsigned long long)accepted << 32) | __builtin_bit_cast(std::uint32_t, checksum);
    }
}

/**
 *  Micro-benchmark of the frontier heap: every wave fills a heap of `fill` pseudo-random keys in LDS, then alternates
 *  `count` pops and pushes (the traversal's steady state) and reports the shader-clock ticks spent in each (diagnostic only).
 *  `serial_ak` selects the one-level-per-round-trip pop of the reference's shape.
 */
template <bool serial_ak>
__global__ __launch_bounds__(64) void heap_bench_kernel(std::uint32_t fill, std::uint32_t count, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    cand_t* heap = reinterpret_cast<cand_t*>(lds);
    std::uint32_t size = 0, state = 4321u + blockIdx.x * 977u;
    auto next_key = [&]() {
        state = state * 1664525u + 1013904223u;
        return uniform_f32(-(float)(state >> 8) * (1.0f / 16777216.0f));
    };
    for (std::uint32_t i = 0; i < fill; ++i)
        heap_push<false>(heap, size, next_key(), i);
    unsigned long long pop_ticks = 0, push_ticks = 0, checksum = 0;
    for (std::uint32_t i = 0; i < count; ++i) {
        const std::uint64_t t0 = __builtin_amdgcn_s_memtime();
        cand_t popped;
        if constexpr (serial_ak)
            popped = heap_pop_serial<false>(heap, size);
        else
            popped = heap_pop<false>(heap, size);
        const std::uint64_t t1 = __builtin_amdgcn_s_memtime();
        heap_push<false>(heap, size, next_key(), fill + i);
        const std::uint64_t t2 = __builtin_amdgcn_s_memtime();
        pop_ticks += t1 - t0, push_ticks += t2 - t1, checksum += popped;
    }
    if (lane_id() == 0)
        out[3 * blockIdx.x] = pop_ticks, out[3 * blockIdx.x + 1] = push_ticks, out[3 * blockIdx.x + 2] = checksum;
}

} // namespace usearch_amd

using namespace usearch_amd;

extern "C" {

/** Diagnostic: per-wave ticks of `heap_bench_kernel` and a checksum of what was popped (both pops must agree on it). */
__attribute__((visibility("default"))) void usearch_amd_bench_heap(uint32_t serial, uint32_t fill, uint32_t count,
                                                                    uint32_t waves, uint64_t* pop_ticks,
                                                                    uint64_t* push_ticks, uint64_t* checksums,
                                                                    usearch_amd_error_t* error) {
    unsigned long long* d_out = nullptr;
    if (hipMalloc((void**)&d_out, (size_t)waves * 24) != hipSuccess)
        return fail(error, "hipMalloc failed");
    const size_t lds = ((size_t)fill + 8) * 8;
    if (serial)
        hipLaunchKernelGGL(heap_bench_kernel<true>, dim3(waves), dim3(64), lds, nullptr, fill, count, d_out);
    else
        hipLaunchKernelGGL(heap_bench_kernel<false>, dim3(waves), dim3(64), lds, nullptr, fill, count, d_out);
    std::vector<unsigned long long> host((size_t)waves * 3);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(host.data(), d_out, host.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
        fail(error, "heap bench failed");
    for (uint32_t w = 0; w < waves; ++w)
        pop_ticks[w] = host[3 * w], push_ticks[w] = host[3 * w + 1], checksums[w] = host[3 * w + 2];
    (void)hipFree(d_out);
}

/** Diagnostic: ticks[wave] and accepted[wave] of `top_bench_kernel` (entries per lane 4, 8 or 16). */
__attribute__((visibility("default"))) void usearch_amd_bench_top(uint32_t epl, uint32_t count, uint32_t limit,
                                                                   uint32_t waves, uint64_t* ticks, uint64_t* accepted,
                                                                   usearch_amd_error_t* error) {
    unsigned long long* d_out = nullptr;
    if (hipMalloc((void**)&d_out, (size_t)waves * 16) != hipSuccess)
        return fail(error, "hipMalloc failed");
    if (epl == 4)
        hipLaunchKernelGGL(top_bench_kernel<4>, dim3(waves), dim3(64), 0, nullptr, count, limit, d_out);
    else if (epl == 8)
        hipLaunchKernelGGL(top_bench_kernel<8>, dim3(waves), dim3(64), 0, nullptr, count, limit, d_out);
    else
        hipLaunchKernelGGL(top_bench_kernel<16>, dim3(waves), dim3(64), 0, nullptr, count, limit, d_out);
    std::vector<unsigned long long> host((size_t)waves * 2);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(host.data(), d_out, host.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
        fail(error, "top bench failed");
    for (uint32_t w = 0; w < waves; ++w)
        ticks[w] = host[2 * w], accepted[w] = host[2 * w + 1] >> 32;
    (void)hipFree(d_out);
}

/**
 *  Self-test hook: replays `count` scripted operations on the device-side containers (kind 0 = frontier push of
 *  {keys[i], slots[i]}, 1 = frontier pop, 2 = insert {keys[i], slots[i]} into the result buffer limited to `limit`) and
 *  returns what was popped and the final result buffer, each entry packed as (slot << 32 | float bits).
 */
__attribute__((visibility("default"))) void usearch_amd_test_containers(uint32_t const* kinds, float const* keys, uint32_t const* slots, size_t count,
                                 size_t limit, uint64_t* popped, size_t* popped_count, uint64_t* top, size_t* top_count,
                                 usearch_amd_error_t* error) {
    const size_t capacity = count + 1;
    uint32_t *d_kinds = nullptr, *d_slots = nullptr, *d_counts = nullptr;
    float* d_keys = nullptr;
    uint64_t *d_popped = nullptr, *d_top = nullptr;
    hipError_t e = hipSuccess;
    auto check = [&](hipError_t r) {
        if (e == hipSuccess)
            e = r;
    };
    check(hipMalloc((void**)&d_kinds, count * 4 + 4));
    check(hipMalloc((void**)&d_slots, count * 4 + 4));
    check(hipMalloc((void**)&d_keys, count * 4 + 4));
    check(hipMalloc((void**)&d_popped, capacity * 8));
    check(hipMalloc((void**)&d_top, (limit + 1) * 8));
    check(hipMalloc((void**)&d_counts, 8));
    if (e == hipSuccess) {
        check(hipMemcpy(d_kinds, kinds, count * 4, hipMemcpyHostToDevice));
        check(hipMemcpy(d_slots, slots, count * 4, hipMemcpyHostToDevice));
        check(hipMemcpy(d_keys, keys, count * 4, hipMemcpyHostToDevice));
        const size_t lds = (capacity + limit + 1) * 8;
        hipLaunchKernelGGL(containers_kernel, dim3(1), dim3(64), lds, nullptr, d_kinds, d_keys, d_slots,
                           (uint32_t)count, (uint32_t)limit, (uint32_t)capacity, d_popped, d_counts, d_top,
                           d_counts + 1);
        check(hipGetLastError());
        check(hipDeviceSynchronize());
        uint32_t host_counts[2] = {0, 0};
        check(hipMemcpy(host_counts, d_counts, 8, hipMemcpyDeviceToHost));
        if (e == hipSuccess) {
            *popped_count = host_counts[0];
            *top_count = host_counts[1];
            check(hipMemcpy(popped, d_popped, host_counts[0] * 8, hipMemcpyDeviceToHost));
            check(hipMemcpy(top, d_top, host_counts[1] * 8, hipMemcpyDeviceToHost));
        }
    }
    for (void* p : {(void*)d_kinds, (void*)d_slots, (void*)d_keys, (void*)d_popped, (void*)d_top, (void*)d_counts})
        if (p)
            (void)hipFree(p);
    if (e != hipSuccess)
        fail(error, hipGetErrorString(e));
}

} // extern "C"


// ---- diagnostics of round 3 – 5's placement studies (scripts/placement_study.py, scripts/fragment_study.py): probes over the arrays a
//      snapshot hands out through `usearch_amd_snapshot_arrays` (csrc/placement.hpp). Not product entry points.
extern "C" __attribute__((visibility("default"))) float usearch_amd_test_gather_probe(const void* base, size_t bytes, size_t row_bytes,
                                                                                      usearch_amd_error_t* error) {
    float gbps = 0.f;
    const hipError_t e = usearch_amd::gather_probe(base, bytes, row_bytes, &gbps);
    if (e != hipSuccess)
        fail(error, hipGetErrorString(e));
    return gbps;
}
extern "C" __attribute__((visibility("default"))) float usearch_amd_test_translation_probe(const void* base, size_t bytes,
                                                                                           usearch_amd_error_t* error) {
    float rate = 0.f;
    const hipError_t e = usearch_amd::translation_probe(base, bytes, &rate);
    if (e != hipSuccess)
        fail(error, hipGetErrorString(e));
    return rate;
}
extern "C" __attribute__((visibility("default"))) float usearch_amd_test_latency_probe(const void* base, size_t bytes, size_t row_bytes,
                                                                                       usearch_amd_error_t* error) {
    float nanoseconds = 0.f;
    const hipError_t e = usearch_amd::latency_probe(base, bytes, row_bytes, &nanoseconds);
    if (e != hipSuccess)
        fail(error, hipGetErrorString(e));
    return nanoseconds;
}
