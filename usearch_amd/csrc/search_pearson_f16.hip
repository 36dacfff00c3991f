// usearch_amd/csrc/search_pearson_f16.hip — kernel instantiations for metric pearson over f16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(pearson_f16, metric_pearson_k, scalar_f16_k)
}
