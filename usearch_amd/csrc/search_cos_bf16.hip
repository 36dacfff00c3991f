// usearch_amd/csrc/search_cos_bf16.hip — kernel instantiations for metric cos over bf16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(cos_bf16, metric_cos_k, scalar_bf16_k)
}
