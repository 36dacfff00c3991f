// usearch_amd/csrc/search_divergence_bf16.hip — kernel instantiations for metric divergence over bf16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(divergence_bf16, metric_divergence_k, scalar_bf16_k)
}
