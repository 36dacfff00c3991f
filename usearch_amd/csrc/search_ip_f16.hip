// usearch_amd/csrc/search_ip_f16.hip — kernel instantiations for metric ip over f16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(ip_f16, metric_ip_k, scalar_f16_k)
}
