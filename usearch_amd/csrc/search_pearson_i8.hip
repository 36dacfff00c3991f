// usearch_amd/csrc/search_pearson_i8.hip — kernel instantiations for metric pearson over i8 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(pearson_i8, metric_pearson_k, scalar_i8_k)
}
