// usearch_amd/csrc/search_pearson_f32.hip — kernel instantiations for metric pearson over f32 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(pearson_f32, metric_pearson_k, scalar_f32_k)
}
