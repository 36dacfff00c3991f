// usearch_amd/csrc/search_b1.hip — kernel instantiations for b1x8 (bit) storage: Hamming (see launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
hipError_t launch_search_b1(const launch_params_t& p, const snapshot_view_t& view, const search_args_t& args) {
    if (p.metric != metric_hamming_k)
        return hipErrorInvalidValue;
    return launch_search_metric<metric_hamming_k, scalar_b1x8_k>(p, view, args);
}
hipError_t launch_distances_b1(const distances_params_t& p, const snapshot_view_t& view) {
    if (p.metric != metric_hamming_k)
        return hipErrorInvalidValue;
    return launch_distances_metric<metric_hamming_k, scalar_b1x8_k>(p, view);
}
} // namespace usearch_amd
