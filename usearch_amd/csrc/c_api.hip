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
#include "filter.hpp"
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
    out->probe_mode = s.probe_mode;
    out->seen_cells = s.seen_cells;
    out->claim_bits = s.claim_bits;
    out->early_rows = s.early_rows;
    out->plain = s.plain;
    out->aside_cells = s.aside_cells;
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


extern "C" {

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
void usearch_amd_snapshot_arrays(usearch_amd_snapshot_t s, usearch_amd_arrays_t* out) {
    if (!out)
        return;
    const snapshot_view_t& view = as_snapshot(s)->view();
    *out = usearch_amd_arrays_t{};
    out->vectors = view.vectors, out->level0 = view.nbr0, out->keys = view.keys;
    out->size = view.size, out->row_stride = view.row_stride, out->level0_cells = view.m0;
    out->device = as_snapshot(s)->device();
}
void usearch_amd_snapshot_placement(usearch_amd_snapshot_t s, uint32_t* draws, uint32_t* kept, float* judge_ms, float* incumbent_ms,
                                    float* probe_ms) {
    const placement_t& placement = as_snapshot(s)->placement();
    if (draws)
        *draws = placement.draws;
    if (kept)
        *kept = placement.kept;
    if (judge_ms)
        for (int i = 0; i < placement_max_draws_k; ++i)
            judge_ms[i] = placement.judge_ms[i];
    if (incumbent_ms)
        for (int i = 0; i < placement_max_draws_k; ++i)
            incumbent_ms[i] = placement.incumbent_ms[i];
    if (probe_ms)
        *probe_ms = placement.probe_ms;
}
uint32_t usearch_amd_snapshot_tune(usearch_amd_snapshot_t s, void const* queries, size_t count, size_t stride, size_t wanted,
                                   size_t expansion, uint32_t max_trials, usearch_amd_error_t* error) try {
    std::uint32_t made = 0;
    if (const char* e = as_snapshot(s)->tune(queries, count, stride, wanted, expansion, max_trials, &made))
        fail(error, e);
    return made;
} catch (...) {
    return fail_from_exception(error), 0u;
}
void usearch_amd_note_device_free(void) { note_release((std::size_t)1 << 40); }
float usearch_amd_settle(void) { return settle_before_placing(); }
float usearch_amd_snapshot_settle_ms(usearch_amd_snapshot_t s) { return as_snapshot(s)->placement().settle_ms; }
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

// ---- filters (filter.hpp)
static filter_t* as_filter(usearch_amd_filter_t f) { return static_cast<filter_t*>(f); }

usearch_amd_filter_t usearch_amd_filter_from_key_range(usearch_amd_snapshot_t snapshot, usearch_amd_key_t first_key,
                                                       usearch_amd_key_t last_key, usearch_amd_error_t* error) try {
    std::unique_ptr<filter_t> filter;
    if (const char* e = filter_t::from_key_range(*as_snapshot(snapshot), first_key, last_key, filter))
        return fail(error, e), nullptr;
    return filter.release();
} catch (...) {
    return fail_from_exception(error), nullptr;
}

usearch_amd_filter_t usearch_amd_filter_from_keys(usearch_amd_snapshot_t snapshot, usearch_amd_key_t const* keys, size_t keys_count,
                                                  int allow, usearch_amd_error_t* error) try {
    std::unique_ptr<filter_t> filter;
    if (const char* e = filter_t::from_keys(*as_snapshot(snapshot), keys, keys_count, allow != 0, filter))
        return fail(error, e), nullptr;
    return filter.release();
} catch (...) {
    return fail_from_exception(error), nullptr;
}

usearch_amd_filter_t usearch_amd_filter_from_bits(usearch_amd_snapshot_t snapshot, uint32_t const* bits, size_t words,
                                                  usearch_amd_error_t* error) try {
    std::unique_ptr<filter_t> filter;
    if (const char* e = filter_t::from_bits(*as_snapshot(snapshot), bits, words, filter))
        return fail(error, e), nullptr;
    return filter.release();
} catch (...) {
    return fail_from_exception(error), nullptr;
}

size_t usearch_amd_filter_allowed(usearch_amd_filter_t filter) { return filter ? (size_t)as_filter(filter)->allowed() : 0; }
void const* usearch_amd_filter_device_bits(usearch_amd_filter_t filter) { return filter ? as_filter(filter)->bits() : nullptr; }
void usearch_amd_filter_free(usearch_amd_filter_t filter, usearch_amd_error_t*) { delete as_filter(filter); }

void usearch_amd_filtered_search_many(usearch_amd_snapshot_t snapshot, usearch_amd_filter_t filter, void const* queries,
                                      int query_kind, size_t queries_count, size_t queries_stride, size_t wanted,
                                      size_t expansion, usearch_amd_key_t* keys, usearch_amd_distance_t* distances,
                                      uint64_t* counts, uint64_t* visited, uint64_t* computed,
                                      usearch_amd_tuning_t const* tuning, usearch_amd_stats_t* stats,
                                      usearch_amd_error_t* error) try {
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!");
    search_extras_t extras;
    if (filter) {
        if (const char* e = as_filter(filter)->check(*as_snapshot(snapshot)))
            return fail(error, e);
        extras.allow_bits = as_filter(filter)->bits();
    }
    search_stats_t s;
    if (const char* e = as_snapshot(snapshot)->search_host(queries, kind, queries_count, queries_stride, wanted, expansion, keys,
                                                           distances, counts, visited, computed, tuning_from_c(tuning), &s,
                                                           nullptr, &extras))
        return fail(error, e);
    stats_to_c(s, stats);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_filtered_search_many_device(usearch_amd_snapshot_t snapshot, usearch_amd_filter_t filter, void const* queries,
                                             size_t queries_count, size_t queries_stride, size_t wanted, size_t expansion,
                                             usearch_amd_key_t* keys, usearch_amd_distance_t* distances, uint64_t* counts,
                                             uint64_t* visited, uint64_t* computed, void* stream,
                                             usearch_amd_tuning_t const* tuning, int timed, usearch_amd_stats_t* stats,
                                             usearch_amd_error_t* error) try {
    if (queries_count && wanted && (!queries || !keys || !distances || !counts || !visited || !computed))
        return fail(error, "Device entry point needs every buffer");
    search_extras_t extras;
    if (filter) {
        if (const char* e = as_filter(filter)->check(*as_snapshot(snapshot)))
            return fail(error, e);
        extras.allow_bits = as_filter(filter)->bits();
    }
    search_stats_t s;
    if (const char* e = as_snapshot(snapshot)->search_device(queries, queries_count, queries_stride, wanted, expansion, keys,
                                                             distances, counts, visited, computed,
                                                             static_cast<hipStream_t>(stream), tuning_from_c(tuning), &s,
                                                             timed != 0, &extras))
        return fail(error, e);
    stats_to_c(s, stats);
} catch (...) {
    fail_from_exception(error);
}

void usearch_amd_filtered_exact_search_many(usearch_amd_snapshot_t snapshot, usearch_amd_filter_t filter, void const* queries,
                                            int query_kind, size_t queries_count, size_t queries_stride, size_t wanted,
                                            usearch_amd_key_t* keys, usearch_amd_distance_t* distances, uint64_t* counts,
                                            int tiled, float* kernel_ms, usearch_amd_error_t* error) try {
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!");
    const std::uint32_t* bits = nullptr;
    if (filter) {
        if (const char* e = as_filter(filter)->check(*as_snapshot(snapshot)))
            return fail(error, e);
        bits = as_filter(filter)->bits();
    }
    if (const char* e = as_snapshot(snapshot)->exact_host(queries, kind, queries_count, queries_stride, wanted, keys, distances,
                                                          counts, kernel_ms, tiled != 0, bits))
        fail(error, e);
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

void usearch_amd_exact_search_many_device(usearch_amd_snapshot_t snapshot, void const* queries, size_t queries_count,
                                          size_t queries_stride, size_t wanted, usearch_amd_key_t* keys,
                                          usearch_amd_distance_t* distances, uint64_t* counts, void* stream, int tiled,
                                          float* kernel_ms, usearch_amd_error_t* error) try {
    if (const char* e = as_snapshot(snapshot)->exact_device(queries, queries_count, queries_stride, wanted, keys, distances,
                                                            counts, static_cast<hipStream_t>(stream), kernel_ms, tiled != 0))
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
    out->refiled_requests = s.refiled_requests;
}

int usearch_amd_cast(int from_kind, int to_kind, void const* input, size_t dimensions, void* output) {
    return cast_vector(scalar_from_c(from_kind), scalar_from_c(to_kind), static_cast<const std::uint8_t*>(input),
                       dimensions, static_cast<std::uint8_t*>(output))
               ? 1
               : 0;
}

} // extern "C"
