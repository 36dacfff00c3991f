// usearch_amd/csrc/search_divergence_f16.hip — kernel instantiations for metric divergence over f16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(divergence_f16, metric_divergence_k, scalar_f16_k)
}
