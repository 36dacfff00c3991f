/**
 *  usearch_amd/csrc/c_api.hip — `extern "C"` shim over the engine: the functions declared in include/usearch_amd.h.
 *  No C++ or HIP types cross this boundary.
 */
#include "../../include/usearch_amd.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <memory>
#include <new>
#include <vector>

#include "build.hpp"
#include "casts.hpp"
#include "engine.hpp"
#include "kernels.hpp"
#include "sharded.hpp"

using namespace usearch_amd;

namespace {

scalar_kind_t scalar_from_c(int kind) { // c/lib.cpp:61-79
    switch (kind) {
    case usearch_amd_scalar_f32_k: return scalar_f32_k;
    case usearch_amd_scalar_f64_k: return scalar_f64_k;
    case usearch_amd_scalar_f16_k: return scalar_f16_k;
    case usearch_amd_scalar_i8_k: return scalar_i8_k;
    case usearch_amd_scalar_b1_k: return scalar_b1x8_k;
    case usearch_amd_scalar_bf16_k: return scalar_bf16_k;
    default: return scalar_unknown_k;
    }
}

int scalar_to_c(scalar_kind_t kind) {
    switch (kind) {
    case scalar_f32_k: return usearch_amd_scalar_f32_k;
    case scalar_f64_k: return usearch_amd_scalar_f64_k;
    case scalar_f16_k: return usearch_amd_scalar_f16_k;
    case scalar_i8_k: return usearch_amd_scalar_i8_k;
    case scalar_b1x8_k: return usearch_amd_scalar_b1_k;
    case scalar_bf16_k: return usearch_amd_scalar_bf16_k;
    default: return 0;
    }
}

metric_kind_t metric_from_c(int kind) { // c/lib.cpp:26-42
    switch (kind) {
    case 1: return metric_cos_k;
    case 2: return metric_ip_k;
    case 3: return metric_l2sq_k;
    case 4: return metric_haversine_k;
    case 5: return metric_divergence_k;
    case 6: return metric_pearson_k;
    case 7: return metric_jaccard_k;
    case 8: return metric_hamming_k;
    case 9: return metric_tanimoto_k;
    case 10: return metric_sorensen_k;
    default: return metric_unknown_k;
    }
}

int metric_to_c(metric_kind_t kind) { // c/usearch.h:40-52 ← c/lib.cpp:26-59
    switch (kind) {
    case metric_cos_k: return 1;
    case metric_ip_k: return 2;
    case metric_l2sq_k: return 3;
    case metric_haversine_k: return 4;
    case metric_divergence_k: return 5;
    case metric_pearson_k: return 6;
    case metric_jaccard_k: return 7;
    case metric_hamming_k: return 8;
    case metric_tanimoto_k: return 9;
    case metric_sorensen_k: return 10;
    default: return 0;
    }
}

search_tuning_t tuning_from_c(usearch_amd_tuning_t const* t) {
    search_tuning_t out;
    if (t) {
        out.hash_cap = t->hash_cap;
        out.next_cap = t->next_cap;
        out.variant = t->variant;
        out.mode = t->mode;
        out.waves_per_cu = t->waves_per_cu;
        out.frontier = t->frontier;
        out.wave_clock = t->wave_clock;
    }
    return out;
}

void stats_to_c(const search_stats_t& s, usearch_amd_stats_t* out) {
    if (!out)
        return;
    out->passes = s.passes;
    out->retried_lds = s.retried_lds;
    out->retried_global = s.retried_global;
    out->kernel_ms = s.kernel_ms;
    out->mode = s.mode;
    out->grid = s.grid;
    out->lds_bytes = s.lds_bytes;
    out->frontier = s.frontier;
    out->variant = s.variant;
    out->tail_idle = s.tail_idle;
    out->span_ms = s.span_ms;
    out->top_cells = s.top_cells;
}

void fail(usearch_amd_error_t* error, const char* message) {
    if (error && message)
        *error = message;
}

/// What the function-try-block of an entry point ends in: an exception never crosses the C ABI, it becomes the error string.
void fail_from_exception(usearch_amd_error_t* error) {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        fail(error, "Out of memory!");
    } catch (...) {
        fail(error, "Unexpected failure inside the engine");
    }
}

snapshot_t* as_snapshot(usearch_amd_snapshot_t handle) { return static_cast<snapshot_t*>(handle); }

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

extern "C" {

/** Diagnostic: per-wave ticks of `heap_bench_kernel` and a checksum of what was popped (both pops must agree on it). Not in
 *  the public header. */
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

/** Diagnostic: ticks[wave] and accepted[wave] of `top_bench_kernel` (entries per lane 4, 8 or 16). Not in the public header. */
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

int usearch_amd_device_count(usearch_amd_error_t* error) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) {
        fail(error, hipGetErrorString(e));
        return 0;
    }
    if (count == 0)
        fail(error, "No HIP device is visible: the MI355X search engine has no CPU fallback");
    return count;
}

usearch_amd_snapshot_t usearch_amd_snapshot_from_buffer(void const* buffer, size_t length, int device,
                                                        usearch_amd_error_t* error) try {
    image_t image;
    if (const char* e = image.open(buffer, length)) {
        fail(error, e);
        return nullptr;
    }
    std::unique_ptr<snapshot_t> snapshot(new snapshot_t());
    if (const char* e = snapshot->build(image, device)) {
        fail(error, e);
        return nullptr;
    }
    return snapshot.release();
} catch (...) {
    fail_from_exception(error);
    return nullptr;
}

usearch_amd_snapshot_t usearch_amd_snapshot_from_parts(void const* graph, size_t graph_length, void const* vectors,
                                                       size_t vectors_stride, int device, usearch_amd_error_t* error) try {
    if (!vectors) {
        fail(error, "No vectors");
        return nullptr;
    }
    image_t image;
    image.external_vectors = static_cast<const std::uint8_t*>(vectors);
    image.external_stride = vectors_stride;
    if (const char* e = image.open(graph, graph_length)) {
        fail(error, e);
        return nullptr;
    }
    std::unique_ptr<snapshot_t> snapshot(new snapshot_t());
    if (const char* e = snapshot->build(image, device)) {
        fail(error, e);
        return nullptr;
    }
    return snapshot.release();
} catch (...) {
    fail_from_exception(error);
    return nullptr;
}

usearch_amd_snapshot_t usearch_amd_snapshot_from_file(char const* path, int device, usearch_amd_error_t* error) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        fail(error, "Can't open file!"); // index_plugins.hpp memory_mapped_file_t wording
        return nullptr;
    }
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size <= 0) {
        ::close(fd);
        fail(error, "Can't infer file size");
        return nullptr;
    }
    void* mapped = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (mapped == MAP_FAILED) {
        fail(error, "Can't memory-map the file");
        return nullptr;
    }
    usearch_amd_snapshot_t snapshot = usearch_amd_snapshot_from_buffer(mapped, (size_t)st.st_size, device, error);
    ::munmap(mapped, (size_t)st.st_size);
    return snapshot;
}

void usearch_amd_snapshot_free(usearch_amd_snapshot_t snapshot, usearch_amd_error_t*) { delete as_snapshot(snapshot); }

size_t usearch_amd_snapshot_size(usearch_amd_snapshot_t s) { return (size_t)as_snapshot(s)->view().size; }
size_t usearch_amd_snapshot_dimensions(usearch_amd_snapshot_t s) { return as_snapshot(s)->view().dimensions; }
size_t usearch_amd_snapshot_connectivity(usearch_amd_snapshot_t s) { return as_snapshot(s)->view().m; }
size_t usearch_amd_snapshot_max_level(usearch_amd_snapshot_t s) { return as_snapshot(s)->view().max_level; }
size_t usearch_amd_snapshot_bytes_per_vector(usearch_amd_snapshot_t s) { return as_snapshot(s)->view().bytes_per_vector; }
size_t usearch_amd_snapshot_row_stride(usearch_amd_snapshot_t s) { return as_snapshot(s)->view().row_stride; }
size_t usearch_amd_snapshot_device_bytes(usearch_amd_snapshot_t s) { return as_snapshot(s)->device_bytes(); }
int usearch_amd_snapshot_scalar_kind(usearch_amd_snapshot_t s) { return scalar_to_c(as_snapshot(s)->scalar()); }
int usearch_amd_snapshot_metric_kind(usearch_amd_snapshot_t s) { return metric_to_c(as_snapshot(s)->metric()); }
size_t usearch_amd_snapshot_lanes_per_row(usearch_amd_snapshot_t s) { return as_snapshot(s)->lanes_per_row(); }
size_t usearch_amd_snapshot_inline_rows(usearch_amd_snapshot_t s) { return as_snapshot(s)->view().nbr0_rows ? 1 : 0; }

void usearch_amd_search_many(usearch_amd_snapshot_t snapshot, void const* queries, int query_kind,
                             size_t queries_count, size_t queries_stride, size_t wanted, size_t expansion,
                             usearch_amd_key_t* keys, usearch_amd_distance_t* distances, uint64_t* counts,
                             uint64_t* visited, uint64_t* computed, usearch_amd_tuning_t const* tuning,
                             usearch_amd_stats_t* stats, usearch_amd_error_t* error) try {
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!"); // c/lib.cpp:120
    search_stats_t s;
    if (const char* e = as_snapshot(snapshot)->search_host(queries, kind, queries_count, queries_stride, wanted,
                                                           expansion, keys, distances, counts, visited, computed,
                                                           tuning_from_c(tuning), &s))
        return fail(error, e);
    stats_to_c(s, stats);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_search_many_device(usearch_amd_snapshot_t snapshot, void const* queries, size_t queries_count,
                                    size_t queries_stride, size_t wanted, size_t expansion, usearch_amd_key_t* keys,
                                    usearch_amd_distance_t* distances, uint64_t* counts, uint64_t* visited,
                                    uint64_t* computed, void* stream, usearch_amd_tuning_t const* tuning, int timed,
                                    usearch_amd_stats_t* stats, usearch_amd_error_t* error) try {
    if (queries_count && wanted && (!queries || !keys || !distances || !counts || !visited || !computed))
        return fail(error, "Device entry point needs every buffer");
    search_stats_t s;
    if (const char* e = as_snapshot(snapshot)->search_device(queries, queries_count, queries_stride, wanted, expansion,
                                                             keys, distances, counts, visited, computed,
                                                             static_cast<hipStream_t>(stream), tuning_from_c(tuning),
                                                             &s, timed != 0))
        return fail(error, e);
    stats_to_c(s, stats);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_cluster_many(usearch_amd_snapshot_t snapshot, void const* queries, int query_kind, size_t queries_count,
                              size_t queries_stride, size_t level, usearch_amd_key_t* keys,
                              usearch_amd_distance_t* distances, uint64_t* visited, uint64_t* computed,
                              usearch_amd_error_t* error) try {
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!");
    if (queries_count && (!queries || !keys || !distances))
        return fail(error, "Cluster search needs the query, key and distance buffers");
    if (const char* e = as_snapshot(snapshot)->cluster_host(queries, kind, queries_count, queries_stride, level, keys,
                                                            distances, visited, computed))
        return fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_exact_search_many(usearch_amd_snapshot_t snapshot, void const* queries, int query_kind,
                                   size_t queries_count, size_t queries_stride, size_t wanted, usearch_amd_key_t* keys,
                                   usearch_amd_distance_t* distances, uint64_t* counts, float* kernel_ms,
                                   usearch_amd_error_t* error) try {
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!");
    if (const char* e = as_snapshot(snapshot)->exact_host(queries, kind, queries_count, queries_stride, wanted, keys,
                                                          distances, counts, kernel_ms))
        fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_exact_search_many_tiled(usearch_amd_snapshot_t snapshot, void const* queries, int query_kind,
                                         size_t queries_count, size_t queries_stride, size_t wanted,
                                         usearch_amd_key_t* keys, usearch_amd_distance_t* distances, uint64_t* counts,
                                         float* kernel_ms, usearch_amd_error_t* error) try {
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!");
    if (const char* e = as_snapshot(snapshot)->exact_host(queries, kind, queries_count, queries_stride, wanted, keys,
                                                          distances, counts, kernel_ms, true))
        fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_exact_search_dataset(void const* dataset, size_t dataset_count, size_t dataset_stride,
                                      void const* queries, size_t queries_count, size_t queries_stride,
                                      int scalar_kind, size_t dimensions, int metric_kind, size_t wanted,
                                      usearch_amd_key_t* keys, size_t keys_stride, usearch_amd_distance_t* distances,
                                      size_t distances_stride, usearch_amd_error_t* error) try {
    if (const char* e = exact_search_dataset_host(metric_from_c(metric_kind), scalar_from_c(scalar_kind), dimensions,
                                                  dataset, dataset_count, dataset_stride, queries, queries_count,
                                                  queries_stride, wanted, keys, keys_stride, distances,
                                                  distances_stride))
        fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_comm_unique_id(void* out, usearch_amd_error_t* error) {
    if (const char* e = comm_t::unique_id(out))
        fail(error, e);
}

usearch_amd_comm_t usearch_amd_comm_init_rccl(void const* unique_id, int rank, int world, int device,
                                              usearch_amd_error_t* error) try {
    std::unique_ptr<comm_t> comm(new comm_t());
    if (const char* e = comm->init_rccl(unique_id, rank, world, device)) {
        fail(error, e);
        return nullptr;
    }
    return comm.release();
} catch (...) {
    fail_from_exception(error);
    return nullptr;
}

usearch_amd_comm_t usearch_amd_comm_init_custom(usearch_amd_transport_t const* transport, int rank, int world, int device,
                                                usearch_amd_error_t* error) try {
    if (!transport) {
        fail(error, "No transport");
        return nullptr;
    }
    std::unique_ptr<comm_t> comm(new comm_t());
    transport_t inner;
    inner.context = transport->context;
    inner.all_gather = transport->all_gather;
    inner.broadcast = transport->broadcast;
    inner.buffers_on_host = transport->buffers_on_host;
    inner.local_search = transport->local_search;
    if (const char* e = comm->init_custom(inner, rank, world, device)) {
        fail(error, e);
        return nullptr;
    }
    return comm.release();
} catch (...) {
    fail_from_exception(error);
    return nullptr;
}

void usearch_amd_comm_free(usearch_amd_comm_t comm) { delete static_cast<comm_t*>(comm); }
int usearch_amd_comm_rank(usearch_amd_comm_t comm) { return comm ? static_cast<comm_t*>(comm)->rank() : 0; }
int usearch_amd_comm_world(usearch_amd_comm_t comm) { return comm ? static_cast<comm_t*>(comm)->world() : 1; }

void usearch_amd_comm_broadcast(usearch_amd_comm_t comm, void* buffer, size_t bytes, int root, void* stream,
                                usearch_amd_error_t* error) try {
    if (!comm)
        return fail(error, "No communicator");
    if (const char* e = static_cast<comm_t*>(comm)->broadcast(buffer, bytes, root, static_cast<hipStream_t>(stream)))
        fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_sharded_search_many(usearch_amd_snapshot_t snapshot, usearch_amd_comm_t comm, void* queries,
                                     size_t queries_count, size_t queries_stride, size_t wanted, size_t expansion,
                                     int broadcast_root, usearch_amd_key_t* keys, usearch_amd_distance_t* distances,
                                     uint64_t* counts, uint64_t* visited, uint64_t* computed, void* stream,
                                     usearch_amd_tuning_t const* tuning, int timed, usearch_amd_stats_t* stats,
                                     usearch_amd_sharded_stats_t* sharded_stats, usearch_amd_error_t* error) try {
    if (!comm)
        return fail(error, "No communicator");
    search_stats_t s;
    sharded_stats_t step;
    if (const char* e = static_cast<comm_t*>(comm)->search(
            snapshot ? as_snapshot(snapshot) : nullptr, queries, queries_count, queries_stride, wanted, expansion,
            broadcast_root, keys, distances, counts, visited, computed, static_cast<hipStream_t>(stream),
            tuning_from_c(tuning), timed != 0, &s, &step))
        return fail(error, e);
    stats_to_c(s, stats);
    if (sharded_stats) {
        sharded_stats->block_bytes = step.block_bytes;
        sharded_stats->gathered_bytes = step.gathered_bytes;
        sharded_stats->exchange_ms = step.exchange_ms;
        sharded_stats->exchanges = step.exchanges;
    }
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_merge_many_device(usearch_amd_distance_t const* distances, usearch_amd_key_t const* keys,
                                   uint64_t const* counts, size_t shards, size_t queries_count, size_t wanted,
                                   usearch_amd_distance_t* out_distances, usearch_amd_key_t* out_keys,
                                   uint64_t* out_counts, void* stream, usearch_amd_error_t* error) try {
    if (const char* e = merge_shards_device(distances, keys, counts, shards, queries_count, wanted, out_distances,
                                            out_keys, out_counts, static_cast<hipStream_t>(stream)))
        fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_merge_many(usearch_amd_distance_t const* distances, usearch_amd_key_t const* keys,
                            uint64_t const* counts, size_t shards, size_t queries_count, size_t wanted,
                            usearch_amd_distance_t* out_distances, usearch_amd_key_t* out_keys, uint64_t* out_counts,
                            usearch_amd_error_t* error) try {
    const size_t cells = shards * queries_count * wanted, rows = shards * queries_count;
    if (!cells)
        return;
    float *d_distances = nullptr, *d_out_distances = nullptr;
    uint64_t *d_keys = nullptr, *d_counts = nullptr, *d_out_keys = nullptr, *d_out_counts = nullptr;
    hipError_t e = hipSuccess;
    auto check = [&](hipError_t r) {
        if (e == hipSuccess)
            e = r;
    };
    check(hipMalloc((void**)&d_distances, cells * 4));
    check(hipMalloc((void**)&d_keys, cells * 8));
    check(hipMalloc((void**)&d_counts, rows * 8));
    check(hipMalloc((void**)&d_out_distances, queries_count * wanted * 4));
    check(hipMalloc((void**)&d_out_keys, queries_count * wanted * 8));
    check(hipMalloc((void**)&d_out_counts, queries_count * 8));
    if (e == hipSuccess) {
        check(hipMemcpy(d_distances, distances, cells * 4, hipMemcpyHostToDevice));
        check(hipMemcpy(d_keys, keys, cells * 8, hipMemcpyHostToDevice));
        check(hipMemcpy(d_counts, counts, rows * 8, hipMemcpyHostToDevice));
    }
    if (e == hipSuccess) {
        if (const char* message = merge_shards_device(d_distances, d_keys, d_counts, shards, queries_count, wanted,
                                                      d_out_distances, d_out_keys, d_out_counts, nullptr))
            fail(error, message);
        check(hipMemcpy(out_distances, d_out_distances, queries_count * wanted * 4, hipMemcpyDeviceToHost));
        check(hipMemcpy(out_keys, d_out_keys, queries_count * wanted * 8, hipMemcpyDeviceToHost));
        check(hipMemcpy(out_counts, d_out_counts, queries_count * 8, hipMemcpyDeviceToHost));
    }
    for (void* p : {(void*)d_distances, (void*)d_keys, (void*)d_counts, (void*)d_out_distances, (void*)d_out_keys,
                    (void*)d_out_counts})
        if (p)
            (void)hipFree(p);
    if (e != hipSuccess)
        fail(error, hipGetErrorString(e));
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_last_peaks(usearch_amd_snapshot_t snapshot, uint32_t* out, size_t queries_count,
                            usearch_amd_error_t* error) {
    if (const char* e = as_snapshot(snapshot)->last_peaks(out, queries_count))
        fail(error, e);
}

void usearch_amd_distances(usearch_amd_snapshot_t snapshot, void const* queries, size_t queries_count,
                           size_t queries_stride, uint32_t const* slots, size_t slots_per_query,
                           usearch_amd_distance_t* out, usearch_amd_error_t* error) try {
    if (const char* e = as_snapshot(snapshot)->distances_host(queries, queries_count, queries_stride, slots,
                                                              slots_per_query, out))
        fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

float usearch_amd_last_distances_ms(usearch_amd_snapshot_t snapshot) { return as_snapshot(snapshot)->last_distances_ms(); }

void usearch_amd_test_containers(uint32_t const* kinds, float const* keys, uint32_t const* slots, size_t count,
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

usearch_amd_builder_t usearch_amd_build(void const* vectors, size_t count, size_t stride, int scalar_kind,
                                        size_t dimensions, int metric_kind, usearch_amd_key_t const* keys,
                                        usearch_amd_build_config_t const* config, int device, int vectors_on_device,
                                        usearch_amd_error_t* error) try {
    build_config_t c;
    if (config) {
        if (config->connectivity)
            c.connectivity = config->connectivity;
        c.connectivity_base = config->connectivity_base;
        if (config->expansion_add)
            c.expansion_add = config->expansion_add;
        if (config->batch_divisor)
            c.batch_divisor = config->batch_divisor;
        if (config->max_batch)
            c.max_batch = config->max_batch;
        if (config->seed)
            c.seed = config->seed;
    }
    std::unique_ptr<builder_t> builder(new builder_t());
    if (const char* e = builder->build(metric_from_c(metric_kind), scalar_from_c(scalar_kind), dimensions, vectors, count,
                                       stride, vectors_on_device != 0, keys, c, device)) {
        fail(error, e);
        return nullptr;
    }
    return builder.release();
} catch (...) {
    fail_from_exception(error);
    return nullptr;
}

void usearch_amd_build_free(usearch_amd_builder_t builder, usearch_amd_error_t*) { delete static_cast<builder_t*>(builder); }

usearch_amd_snapshot_t usearch_amd_build_snapshot(usearch_amd_builder_t builder) {
    return &static_cast<builder_t*>(builder)->snapshot();
}

size_t usearch_amd_build_serialized_length(usearch_amd_builder_t builder) {
    return static_cast<builder_t*>(builder)->serialized_length();
}

void usearch_amd_build_save_buffer(usearch_amd_builder_t builder, void* buffer, size_t length,
                                   usearch_amd_error_t* error) try {
    if (const char* e = static_cast<builder_t*>(builder)->save_buffer(buffer, length))
        fail(error, e);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_build_stats(usearch_amd_builder_t builder, usearch_amd_build_stats_t* out) {
    const build_stats_t& s = static_cast<builder_t*>(builder)->stats();
    *out = usearch_amd_build_stats_t{};
    out->batches = s.batches, out->passes = s.passes;
    out->search_distances = s.search_distances, out->search_hops = s.search_hops;
    out->select_distances = s.select_distances, out->reverse_distances = s.reverse_distances;
    out->repruned_lists = s.repruned_lists, out->dropped_requests = s.dropped_requests;
    out->seconds_total = s.seconds_total, out->seconds_search = s.seconds_search;
    out->seconds_link = s.seconds_link, out->seconds_upload = s.seconds_upload;
    out->max_level = s.max_level;
}

int usearch_amd_cast(int from_kind, int to_kind, void const* input, size_t dimensions, void* output) {
    return cast_vector(scalar_from_c(from_kind), scalar_from_c(to_kind), static_cast<const std::uint8_t*>(input),
                       dimensions, static_cast<std::uint8_t*>(output))
               ? 1
               : 0;
}

} // extern "C"
