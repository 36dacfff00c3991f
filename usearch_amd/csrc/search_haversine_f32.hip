// usearch_amd/csrc/search_haversine_f32.hip — kernel instantiations for metric haversine over f32 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(haversine_f32, metric_haversine_k, scalar_f32_k)
}
