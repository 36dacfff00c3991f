// usearch_amd/csrc/search_pearson_bf16.hip — kernel instantiations for metric pearson over bf16 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(pearson_bf16, metric_pearson_k, scalar_bf16_k)
}
