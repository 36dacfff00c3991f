// usearch_amd/csrc/search_l2sq_f16.hip — kernel instantiations for metric l2sq over f16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(l2sq_f16, metric_l2sq_k, scalar_f16_k)
}
