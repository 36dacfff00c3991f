// usearch_amd/csrc/search_l2sq_bf16.hip — kernel instantiations for metric l2sq over bf16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(l2sq_bf16, metric_l2sq_k, scalar_bf16_k)
}
