// usearch_amd/csrc/search_haversine_f64.hip — kernel instantiations for metric haversine over f64 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(haversine_f64, metric_haversine_k, scalar_f64_k)
}
