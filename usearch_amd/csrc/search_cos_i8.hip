// usearch_amd/csrc/search_cos_i8.hip — kernel instantiations for metric cos over i8 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(cos_i8, metric_cos_k, scalar_i8_k)
}
