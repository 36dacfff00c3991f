// usearch_amd/csrc/search_divergence_f32.hip — kernel instantiations for metric divergence over f32 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(divergence_f32, metric_divergence_k, scalar_f32_k)
}
