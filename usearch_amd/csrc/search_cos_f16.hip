// usearch_amd/csrc/search_cos_f16.hip — kernel instantiations for metric cos over f16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(cos_f16, metric_cos_k, scalar_f16_k)
}
