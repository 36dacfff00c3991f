// usearch_amd/csrc/search_divergence_f64.hip — kernel instantiations for metric divergence over f64 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(divergence_f64, metric_divergence_k, scalar_f64_k)
}
