// usearch_amd/csrc/search_tanimoto_b1.hip — kernel instantiations for metric tanimoto over b1 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(tanimoto_b1, metric_tanimoto_k, scalar_b1x8_k)
}
