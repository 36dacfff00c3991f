/**
 *  usearch_amd/csrc/build_launch_impl.hpp — template dispatch of the construction kernels (build_kernels.hpp) for one
 *  (metric, scalar) pair; included by launch_impl.hpp so that every pair's translation unit carries them next to the walk.
 */
#pragma once
#include "build_kernels.hpp"
#include "engine.hpp"

namespace usearch_amd {

template <int metric_ak, int scalar_ak, int lanes_ak>
hipError_t launch_build_one(const build_params_t& p, const snapshot_view_t& view, const build_args_t& args) {
    constexpr int unroll_ak = lanes_ak == 8 ? 8 : 4;
    const std::uint32_t lds_bytes = query_lds_bytes<scalar_ak>(view.chunks) + build_lds_bytes(args.candidate_cap);
    if (p.reverse)
        hipLaunchKernelGGL((build_reverse_kernel<metric_ak, scalar_ak, lanes_ak, unroll_ak>), dim3(p.grid), dim3(64),
                           lds_bytes, p.stream, view, args);
    else
        hipLaunchKernelGGL((build_select_kernel<metric_ak, scalar_ak, lanes_ak, unroll_ak>), dim3(p.grid), dim3(64),
                           lds_bytes, p.stream, view, args);
    return hipGetLastError();
}

template <int metric_ak, int scalar_ak>
hipError_t launch_build_metric(const build_params_t& p, const snapshot_view_t& view, const build_args_t& args) {
    switch (p.lanes) {
    case 1: return launch_build_one<metric_ak, scalar_ak, 1>(p, view, args);
    case 2: return launch_build_one<metric_ak, scalar_ak, 2>(p, view, args);
    case 4: // never chosen by `row_geometry`, only forced (USEARCH_AMD_LANES): kept for the common pairs' tuning runs
        if constexpr (all_kernel_builds((metric_kind_t)metric_ak, (scalar_kind_t)scalar_ak))
            return launch_build_one<metric_ak, scalar_ak, 4>(p, view, args);
        else
            return hipErrorInvalidValue;
    case 8: return launch_build_one<metric_ak, scalar_ak, 8>(p, view, args);
    default: return hipErrorInvalidValue;
    }
}

} // namespace usearch_amd
