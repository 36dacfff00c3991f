// usearch_amd/csrc/search_l2sq_i8.hip — kernel instantiations for metric l2sq over i8 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(l2sq_i8, metric_l2sq_k, scalar_i8_k)
}
