/**
 *  usearch_amd/csrc/launch_impl.hpp — template dispatch (metric × lanes-per-row × unroll × scratch space) for one
 *  storage scalar kind. Each `search_<kind>.hip` includes this once, so the kinds compile in parallel.
 */
#pragma once
#include "engine.hpp"
#include "kernels.hpp"

namespace usearch_amd {

/// The instantiations that exist in the `plain_ak` cut as well: the short-row walks of the common pairs (expansion ≤ 128 over the
/// global hash, the reference's heap) — what BASELINE's configs 4 and 5 run. `plain_build_exists` (engine.hpp) says the same to the engine.
template <int metric_ak, int scalar_ak, int lanes_ak, int variant_ak, int mode_ak, int epl_ak, int frontier_ak>
constexpr bool plain_build() {
    return plain_build_exists((metric_kind_t)metric_ak, (scalar_kind_t)scalar_ak, lanes_ak, variant_ak == variant_u4_w4_k,
                              mode_ak == scratch_hash_k, epl_ak, frontier_ak == frontier_heap_k);
}

template <int metric_ak, int scalar_ak, int lanes_ak, int variant_ak, int mode_ak, int epl_ak, int frontier_ak = frontier_heap_k>
hipError_t launch_search_one(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    if (p.team) { // five waves per query: rows of ≥ 128 bytes, the 12-deep build of the common pairs, heaps in LDS
        if constexpr (lanes_ak == 8 && variant_ak == variant_u12_w2_k && mode_ak != scratch_global_k) {
            auto team_kernel = team_search_kernel<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, epl_ak, frontier_ak>;
            if (p.lds_bytes > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(team_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
                if (e != hipSuccess)
                    return e;
            }
            hipLaunchKernelGGL(team_kernel, dim3(p.grid), dim3(64 * team_waves_k), p.lds_bytes, p.stream, view, args);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (p.plain) { // short rows, a plain `search` batch: the build without the features such a batch never uses (kernels.hpp `plain_ak`)
        if constexpr (plain_build<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, epl_ak, frontier_ak>()) {
            hipLaunchKernelGGL((search_kernel<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, epl_ak, frontier_ak, true>), dim3(p.grid),
                               dim3(64), p.lds_bytes, p.stream, view, args);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    auto kernel = search_kernel<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, epl_ak, frontier_ak>;
    if (p.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(kernel, dim3(p.grid), dim3(64), p.lds_bytes, p.stream, view, args);
    return hipGetLastError();
}

/// The frontier as the open cells of a register `top` (kernels.hpp frontier_top_k): float-valued pairs only — the
/// integer-valued ones (b1, i8) tie all the time and keep the reference's heap and its pop order.
template <int scalar_ak> constexpr bool frontier_in_top_pair() { return scalar_ak != scalar_b1x8_k && scalar_ak != scalar_i8_k; }

template <int metric_ak, int scalar_ak, int lanes_ak, int variant_ak, int mode_ak, int epl_ak>
hipError_t launch_search_frontier(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    if constexpr (epl_ak > 0 && frontier_in_top_pair<scalar_ak>()) {
        if (p.frontier == frontier_top_k)
            return launch_search_one<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, epl_ak, frontier_top_k>(p, view, args);
    }
    if (p.frontier != frontier_heap_k)
        return hipErrorInvalidValue;
    if constexpr (variant_ak == variant_u12x2_w2_k)
        return hipErrorInvalidValue;
    else
        return launch_search_one<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, epl_ak, frontier_heap_k>(p, view, args);
}

/// `top` in registers with 1 / 2 / 4 / 8 / 16 entries per lane (expansion ≤ 64 / 128 (short rows) / 256 / 512 / 1024), or in scratch
/// memory (0).
template <int metric_ak, int scalar_ak, int lanes_ak, int variant_ak, int mode_ak>
hipError_t launch_search_epl(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    switch (p.entries_per_lane) {
    case 0: return launch_search_frontier<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, 0>(p, view, args);
    case 1: return launch_search_frontier<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, 1>(p, view, args);
    case 2: // rows of ≤ 128 bytes at 64 < expansion ≤ 128: two cells per lane keep the build inside the short-row register budget
        if constexpr (lanes_ak <= 2 && variant_ak == variant_u4_w4_k)
            return launch_search_frontier<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, 2>(p, view, args);
        else
            return hipErrorInvalidValue;
    case 4: return launch_search_frontier<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, 4>(p, view, args);
    case 8: return launch_search_frontier<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, 8>(p, view, args);
    case 16: return launch_search_frontier<metric_ak, scalar_ak, lanes_ak, variant_ak, mode_ak, 16>(p, view, args);
    default: return hipErrorInvalidValue;
    }
}

template <int metric_ak, int scalar_ak, int lanes_ak, int variant_ak>
hipError_t launch_search_mode(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    switch (p.mode) {
    case scratch_lds_k: return launch_search_epl<metric_ak, scalar_ak, lanes_ak, variant_ak, scratch_lds_k>(p, view, args);
    case scratch_hash_k: return launch_search_epl<metric_ak, scalar_ak, lanes_ak, variant_ak, scratch_hash_k>(p, view, args);
    default: return hipErrorInvalidValue;
    }
}

template <int metric_ak, int scalar_ak, int lanes_ak>
hipError_t launch_search_lanes(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    if (p.mode == scratch_global_k)
        return launch_search_one<metric_ak, scalar_ak, lanes_ak, variant_u4_w4_k, scratch_global_k, 0>(p, view, args);
    // rows of ≥ 128 bytes: the unroll depth matters; pairs outside the common set only carry the 4-deep build
    if constexpr (lanes_ak == 8 && all_kernel_builds((metric_kind_t)metric_ak, (scalar_kind_t)scalar_ak)) {
        switch (p.variant) {
        case variant_u8_w3_k: return launch_search_mode<metric_ak, scalar_ak, lanes_ak, variant_u8_w3_k>(p, view, args);
        case variant_u12_w2_k: return launch_search_mode<metric_ak, scalar_ak, lanes_ak, variant_u12_w2_k>(p, view, args);
        case variant_u12x2_w2_k: // two rows per lane group per round: the registers only the heap-less builds have
            if constexpr (frontier_in_top_pair<scalar_ak>()) {
                if (p.frontier != frontier_top_k || !p.entries_per_lane)
                    return hipErrorInvalidValue;
                return launch_search_mode<metric_ak, scalar_ak, lanes_ak, variant_u12x2_w2_k>(p, view, args);
            }
            return hipErrorInvalidValue;
        default: break;
        }
    }
    return launch_search_mode<metric_ak, scalar_ak, lanes_ak, variant_u4_w4_k>(p, view, args);
}

template <int metric_ak, int scalar_ak>
hipError_t launch_search_metric(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    switch (p.lanes) {
    case 1: return launch_search_lanes<metric_ak, scalar_ak, 1>(p, view, args);
    case 2: return launch_search_lanes<metric_ak, scalar_ak, 2>(p, view, args);
    case 4: // never chosen by `row_geometry`, only forced (USEARCH_AMD_LANES): kept for the common pairs' tuning runs
        if constexpr (all_kernel_builds((metric_kind_t)metric_ak, (scalar_kind_t)scalar_ak))
            return launch_search_lanes<metric_ak, scalar_ak, 4>(p, view, args);
        else
            return hipErrorInvalidValue;
    case 8: return launch_search_lanes<metric_ak, scalar_ak, 8>(p, view, args);
    default: return hipErrorInvalidValue;
    }
}

template <int metric_ak, int scalar_ak, int lanes_ak>
hipError_t launch_distances_one(const distances_params_t& p, const snapshot_view_t& view) {
    auto kernel = distances_kernel<metric_ak, scalar_ak, lanes_ak, lanes_ak == 8 ? 8 : 4>;
    hipLaunchKernelGGL(kernel, dim3(p.count), dim3(64), p.lds_bytes, p.stream, view, p.queries, p.query_stride,
                       p.slots, p.slots_per_query, p.out);
    return hipGetLastError();
}

template <int metric_ak, int scalar_ak>
hipError_t launch_distances_metric(const distances_params_t& p, const snapshot_view_t& view) {
    switch (p.lanes) {
    case 1: return launch_distances_one<metric_ak, scalar_ak, 1>(p, view);
    case 2: return launch_distances_one<metric_ak, scalar_ak, 2>(p, view);
    case 4: // never chosen by `row_geometry`, only forced (USEARCH_AMD_LANES): kept for the common pairs' tuning runs
        if constexpr (all_kernel_builds((metric_kind_t)metric_ak, (scalar_kind_t)scalar_ak))
            return launch_distances_one<metric_ak, scalar_ak, 4>(p, view);
        else
            return hipErrorInvalidValue;
    case 8: return launch_distances_one<metric_ak, scalar_ak, 8>(p, view);
    default: return hipErrorInvalidValue;
    }
}

template <int metric_ak, int scalar_ak, int lanes_ak>
hipError_t launch_exact_one(const exact_params_t& p, const snapshot_view_t& view) {
    auto kernel = exact_kernel<metric_ak, scalar_ak, lanes_ak, lanes_ak == 8 ? 8 : 4>;
    hipLaunchKernelGGL(kernel, dim3(p.query_count, p.partitions), dim3(64), p.lds_bytes, p.stream, view, p.queries,
                       p.query_stride, p.query_count, p.wanted, p.rows_per_partition, p.map_keys, p.allow_bits,
                       p.out_distances, p.out_keys, p.out_counts);
    return hipGetLastError();
}

template <int metric_ak, int scalar_ak>
hipError_t launch_exact_metric(const exact_params_t& p, const snapshot_view_t& view) {
    switch (p.lanes) {
    case 1: return launch_exact_one<metric_ak, scalar_ak, 1>(p, view);
    case 2: return launch_exact_one<metric_ak, scalar_ak, 2>(p, view);
    case 4: // never chosen by `row_geometry`, only forced (USEARCH_AMD_LANES): kept for the common pairs' tuning runs
        if constexpr (all_kernel_builds((metric_kind_t)metric_ak, (scalar_kind_t)scalar_ak))
            return launch_exact_one<metric_ak, scalar_ak, 4>(p, view);
        else
            return hipErrorInvalidValue;
    case 8: return launch_exact_one<metric_ak, scalar_ak, 8>(p, view);
    default: return hipErrorInvalidValue;
    }
}

} // namespace usearch_amd

#include "build_launch_impl.hpp" // the link kernels' launchers (not part of the walk's sources)

namespace usearch_amd {

/// One (metric, scalar) pair per translation unit, so that the pairs compile in parallel.
#define USEARCH_AMD_DEFINE_LAUNCHERS(name, metric_kind, scalar_kind)                                                   \
    hipError_t launch_search_##name(const launch_params_t& p, const snapshot_view_t& view,                             \
                                    const search_args_t& args) {                                                       \
        return launch_search_metric<metric_kind, scalar_kind>(p, view, args);                                          \
    }                                                                                                                  \
    hipError_t launch_distances_##name(const distances_params_t& p, const snapshot_view_t& view) {                     \
        return launch_distances_metric<metric_kind, scalar_kind>(p, view);                                             \
    }                                                                                                                  \
    hipError_t launch_exact_##name(const exact_params_t& p, const snapshot_view_t& view) {                             \
        return launch_exact_metric<metric_kind, scalar_kind>(p, view);                                                 \
    }                                                                                                                  \
    hipError_t launch_build_##name(const build_params_t& p, const snapshot_view_t& view, const build_args_t& args) {   \
        return launch_build_metric<metric_kind, scalar_kind>(p, view, args);                                           \
    }

} // namespace usearch_amd
