// usearch_amd/csrc/search_ip_i8.hip — kernel instantiations for metric ip over i8 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(ip_i8, metric_ip_k, scalar_i8_k)
}
