// usearch_amd/csrc/search_l2sq_f64.hip — kernel instantiations for metric l2sq over f64 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(l2sq_f64, metric_l2sq_k, scalar_f64_k)
}
