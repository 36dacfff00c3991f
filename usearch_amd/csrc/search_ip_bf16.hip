// usearch_amd/csrc/search_ip_bf16.hip — kernel instantiations for metric ip over bf16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(ip_bf16, metric_ip_k, scalar_bf16_k)
}
