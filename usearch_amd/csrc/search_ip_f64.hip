// usearch_amd/csrc/search_ip_f64.hip — kernel instantiations for metric ip over f64 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(ip_f64, metric_ip_k, scalar_f64_k)
}
