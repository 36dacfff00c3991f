/**
 *  usearch_amd/csrc/launch_impl.hpp — template dispatch (metric × lanes-per-row × unroll × scratch space) for one
 *  storage scalar kind. Each `search_<kind>.hip` includes this once, so the kinds compile in parallel.
 */
#pragma once
#include "engine.hpp"
#include "kernels.hpp"

namespace usearch_amd {

template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak, int mode_ak>
hipError_t launch_search_one(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    auto kernel = search_kernel<metric_ak, scalar_ak, lanes_ak, unroll_ak, mode_ak>;
    if (p.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(kernel, dim3(p.grid), dim3(64), p.lds_bytes, p.stream, view, args);
    return hipGetLastError();
}

template <int metric_ak, int scalar_ak, int lanes_ak, int unroll_ak>
hipError_t launch_search_mode(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    switch (p.mode) {
    case scratch_lds_k: return launch_search_one<metric_ak, scalar_ak, lanes_ak, unroll_ak, scratch_lds_k>(p, view, args);
    case scratch_hash_k: return launch_search_one<metric_ak, scalar_ak, lanes_ak, unroll_ak, scratch_hash_k>(p, view, args);
    default: return hipErrorInvalidValue;
    }
}

template <int metric_ak, int scalar_ak, int lanes_ak>
hipError_t launch_search_lanes(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    if (p.mode == scratch_global_k)
        return launch_search_one<metric_ak, scalar_ak, lanes_ak, 4, scratch_global_k>(p, view, args);
    if constexpr (lanes_ak == 8) {
        if (p.unroll >= 8)
            return launch_search_mode<metric_ak, scalar_ak, lanes_ak, 8>(p, view, args);
    }
    return launch_search_mode<metric_ak, scalar_ak, lanes_ak, 4>(p, view, args);
}

template <int metric_ak, int scalar_ak>
hipError_t launch_search_metric(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    switch (p.lanes) {
    case 1: return launch_search_lanes<metric_ak, scalar_ak, 1>(p, view, args);
    case 2: return launch_search_lanes<metric_ak, scalar_ak, 2>(p, view, args);
    case 4: return launch_search_lanes<metric_ak, scalar_ak, 4>(p, view, args);
    case 8: return launch_search_lanes<metric_ak, scalar_ak, 8>(p, view, args);
    default: return hipErrorInvalidValue;
    }
}

template <int metric_ak, int scalar_ak, int lanes_ak>
hipError_t launch_distances_one(const distances_params_t& p, const snapshot_view_t& view) {
    auto kernel = distances_kernel<metric_ak, scalar_ak, lanes_ak, 4>;
    hipLaunchKernelGGL(kernel, dim3(p.count), dim3(64), p.lds_bytes, p.stream, view, p.queries, p.query_stride,
                       p.slots, p.slots_per_query, p.out);
    return hipGetLastError();
}

template <int metric_ak, int scalar_ak>
hipError_t launch_distances_metric(const distances_params_t& p, const snapshot_view_t& view) {
    switch (p.lanes) {
    case 1: return launch_distances_one<metric_ak, scalar_ak, 1>(p, view);
    case 2: return launch_distances_one<metric_ak, scalar_ak, 2>(p, view);
    case 4: return launch_distances_one<metric_ak, scalar_ak, 4>(p, view);
    case 8: return launch_distances_one<metric_ak, scalar_ak, 8>(p, view);
    default: return hipErrorInvalidValue;
    }
}

/// ip / cos / l2sq over one numeric scalar kind.
#define USEARCH_AMD_DEFINE_NUMERIC_LAUNCHERS(suffix, scalar_kind)                                                      \
    hipError_t launch_search_##suffix(const launch_params_t& p, const snapshot_view_t& view,                           \
                                      const search_args_t& args) {                                                     \
        switch (p.metric) {                                                                                            \
        case metric_ip_k: return launch_search_metric<metric_ip_k, scalar_kind>(p, view, args);                        \
        case metric_cos_k: return launch_search_metric<metric_cos_k, scalar_kind>(p, view, args);                      \
        case metric_l2sq_k: return launch_search_metric<metric_l2sq_k, scalar_kind>(p, view, args);                    \
        default: return hipErrorInvalidValue;                                                                          \
        }                                                                                                              \
    }                                                                                                                  \
    hipError_t launch_distances_##suffix(const distances_params_t& p, const snapshot_view_t& view) {                   \
        switch (p.metric) {                                                                                            \
        case metric_ip_k: return launch_distances_metric<metric_ip_k, scalar_kind>(p, view);                           \
        case metric_cos_k: return launch_distances_metric<metric_cos_k, scalar_kind>(p, view);                         \
        case metric_l2sq_k: return launch_distances_metric<metric_l2sq_k, scalar_kind>(p, view);                       \
        default: return hipErrorInvalidValue;                                                                          \
        }                                                                                                              \
    }

} // namespace usearch_amd
