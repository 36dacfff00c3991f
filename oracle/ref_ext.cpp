/**
 *  oracle/ref_ext.cpp — TEST INFRASTRUCTURE, not product code.
 *
 *  A thin batch driver over the *real* reference (`/root/reference/include/usearch/index_dense.hpp`),
 *  compiled by `oracle/Makefile` into `oracle/_ref/libusearch_ref.so` together with the reference's own
 *  `c/lib.cpp`. Nothing of the reference is copied here: this file only *calls* its public C++ surface.
 *
 *  It exists because the reference C ABI (`c/usearch.h`) has no batched search and does not expose the
 *  per-query counters (`visited_members`, `computed_distances`) that double as parity checks, nor
 *  `exact=true` search (ground truth for recall). The loops below have the shape of the reference's own
 *  batch drivers: `cpp/bench.cpp:329-350` (index_many) and `cpp/bench.cpp:352-377` (search_many).
 *
 *  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may load the resulting library.
 */
#include <cstdint>
#include <cstring>

#if defined(_OPENMP)
#include <omp.h>
#endif

#include <limits>

#include <usearch/index_dense.hpp>

extern "C" {
#include "usearch.h"
}

using namespace unum::usearch;

namespace {

scalar_kind_t to_cpp(usearch_scalar_kind_t kind) {
    switch (kind) {
    case usearch_scalar_f32_k: return scalar_kind_t::f32_k;
    case usearch_scalar_f64_k: return scalar_kind_t::f64_k;
    case usearch_scalar_f16_k: return scalar_kind_t::f16_k;
    case usearch_scalar_bf16_k: return scalar_kind_t::bf16_k;
    case usearch_scalar_i8_k: return scalar_kind_t::i8_k;
    case usearch_scalar_b1_k: return scalar_kind_t::b1x8_k;
    default: return scalar_kind_t::unknown_k;
    }
}

using dense_search_result_t = typename index_dense_t::search_result_t;

dense_search_result_t search_one(index_dense_t& index, void const* q, scalar_kind_t kind, std::size_t k,
                                 std::size_t thread, bool exact) {
    switch (kind) {
    case scalar_kind_t::f32_k: return index.search((f32_t const*)q, k, thread, exact);
    case scalar_kind_t::f64_k: return index.search((f64_t const*)q, k, thread, exact);
    case scalar_kind_t::f16_k: return index.search((f16_t const*)q, k, thread, exact);
    case scalar_kind_t::bf16_k: return index.search((bf16_t const*)q, k, thread, exact);
    case scalar_kind_t::i8_k: return index.search((i8_t const*)q, k, thread, exact);
    default: return index.search((b1x8_t const*)q, k, thread, exact);
    }
}

bool add_one(index_dense_t& index, std::uint64_t key, void const* v, scalar_kind_t kind, std::size_t thread) {
    switch (kind) {
    case scalar_kind_t::f32_k: return bool(index.add(key, (f32_t const*)v, thread));
    case scalar_kind_t::f64_k: return bool(index.add(key, (f64_t const*)v, thread));
    case scalar_kind_t::f16_k: return bool(index.add(key, (f16_t const*)v, thread));
    case scalar_kind_t::bf16_k: return bool(index.add(key, (bf16_t const*)v, thread));
    case scalar_kind_t::i8_k: return bool(index.add(key, (i8_t const*)v, thread));
    default: return bool(index.add(key, (b1x8_t const*)v, thread));
    }
}

void ensure_threads(index_dense_t& index, std::size_t threads) {
    index_limits_t limits = index.limits();
    if (limits.threads_add >= threads && limits.threads_search >= threads)
        return;
    limits.threads_add = (std::max)(limits.threads_add, threads);
    limits.threads_search = (std::max)(limits.threads_search, threads);
    limits.members = (std::max)(limits.members, index.size());
    index.try_reserve(limits);
}

} // namespace

extern "C" {

/// Number of OpenMP threads a `threads == 0` request resolves to.
int uref_max_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/// Parallel insertion: loop shape of `cpp/bench.cpp:329-350`. Returns the number of successful adds.
size_t uref_add_many(usearch_index_t handle, usearch_key_t const* keys, void const* vectors,
                     usearch_scalar_kind_t kind, size_t n, size_t stride_bytes, size_t threads) {
    index_dense_t& index = *reinterpret_cast<index_dense_t*>(handle);
    scalar_kind_t cpp_kind = to_cpp(kind);
    if (!threads)
        threads = (size_t)uref_max_threads();
    ensure_threads(index, threads);
    size_t done = 0;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static, 32) num_threads(threads) reduction(+ : done)
#endif
    for (std::size_t i = 0; i < n; ++i) {
        std::size_t thread = 0;
#if defined(_OPENMP)
        thread = (std::size_t)omp_get_thread_num();
#endif
        done += add_one(index, keys[i], (char const*)vectors + i * stride_bytes, cpp_kind, thread);
    }
    return done;
}

/**
 *  Batched search: loop shape of `cpp/bench.cpp:352-377`
 *  (`#pragma omp parallel for schedule(static,32)` → `index.search(q, k, thread).dump_to(keys, dists, k)`).
 *  `visited`/`computed` (optional) receive the per-query counters of `index.hpp:3071-3072`.
 */
void uref_search_many(usearch_index_t handle, void const* queries, usearch_scalar_kind_t kind, size_t count,
                      size_t stride_bytes, size_t k, int exact, size_t threads, //
                      usearch_key_t* keys, usearch_distance_t* distances, uint64_t* counts, uint64_t* visited,
                      uint64_t* computed) {
    index_dense_t& index = *reinterpret_cast<index_dense_t*>(handle);
    scalar_kind_t cpp_kind = to_cpp(kind);
    if (!threads)
        threads = (size_t)uref_max_threads();
    ensure_threads(index, threads);
#if defined(_OPENMP)
#pragma omp parallel for schedule(static, 32) num_threads(threads)
#endif
    for (std::size_t i = 0; i < count; ++i) {
        std::size_t thread = 0;
#if defined(_OPENMP)
        thread = (std::size_t)omp_get_thread_num();
#endif
        dense_search_result_t r =
            search_one(index, (char const*)queries + i * stride_bytes, cpp_kind, k, thread, exact != 0);
        std::size_t found = r.dump_to(keys + i * k, distances + i * k, k);
        if (counts)
            counts[i] = found;
        if (visited)
            visited[i] = r.visited_members;
        if (computed)
            computed[i] = r.computed_distances;
    }
}

/**
 *  Batched `index_dense_gt::cluster(query, level)` (index_dense.hpp:788-793 → index_gt::cluster, index.hpp:3089-3125),
 *  one query per OpenMP task like the search loop above. keys / distances / counters are [count] arrays; a failed call
 *  ("No clusters to identify" on an empty index) leaves key 0 and a signalling NaN.
 */
void uref_cluster_many(usearch_index_t handle, void const* queries, usearch_scalar_kind_t kind, size_t count,
                       size_t stride_bytes, size_t level, size_t threads, usearch_key_t* keys,
                       usearch_distance_t* distances, uint64_t* visited, uint64_t* computed) {
    index_dense_t& index = *reinterpret_cast<index_dense_t*>(handle);
    scalar_kind_t cpp_kind = to_cpp(kind);
    if (!threads)
        threads = (size_t)uref_max_threads();
    ensure_threads(index, threads);
#if defined(_OPENMP)
#pragma omp parallel for schedule(static, 32) num_threads(threads)
#endif
    for (std::size_t i = 0; i < count; ++i) {
        std::size_t thread = 0;
#if defined(_OPENMP)
        thread = (std::size_t)omp_get_thread_num();
#endif
        char const* q = (char const*)queries + i * stride_bytes;
        index_dense_t::cluster_result_t r;
        switch (cpp_kind) {
        case scalar_kind_t::f32_k: r = index.cluster((f32_t const*)q, level, thread); break;
        case scalar_kind_t::f64_k: r = index.cluster((f64_t const*)q, level, thread); break;
        case scalar_kind_t::f16_k: r = index.cluster((f16_t const*)q, level, thread); break;
        case scalar_kind_t::bf16_k: r = index.cluster((bf16_t const*)q, level, thread); break;
        case scalar_kind_t::i8_k: r = index.cluster((i8_t const*)q, level, thread); break;
        default: r = index.cluster((b1x8_t const*)q, level, thread); break;
        }
        keys[i] = 0;
        distances[i] = std::numeric_limits<usearch_distance_t>::signaling_NaN();
        if (r) {
            keys[i] = r.cluster.member.key;
            distances[i] = r.cluster.distance;
        }
        if (visited)
            visited[i] = r.visited_members;
        if (computed)
            computed[i] = r.computed_distances;
    }
}

/// Graph shape, to cross-check the flattener: max level, entry slot and nodes per level (up to 32 levels).
void uref_graph_shape(usearch_index_t handle, uint64_t* max_level, uint64_t* connectivity,
                      uint64_t* connectivity_base, uint64_t* nodes_per_level) {
    index_dense_t& index = *reinterpret_cast<index_dense_t*>(handle);
    *max_level = index.max_level();
    *connectivity = index.connectivity();
    *connectivity_base = index.config().connectivity_base;
    for (std::size_t level = 0; level <= index.max_level() && level < 32; ++level)
        nodes_per_level[level] = index.stats(level).nodes;
}

} // extern "C"
