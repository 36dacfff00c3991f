// usearch_amd/csrc/search_cos_f64.hip — kernel instantiations for metric cos over f64 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(cos_f64, metric_cos_k, scalar_f64_k)
}
