// usearch_amd/csrc/search_sorensen_b1.hip — kernel instantiations for metric sorensen over b1 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(sorensen_b1, metric_sorensen_k, scalar_b1x8_k)
}
