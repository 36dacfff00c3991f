// usearch_amd/csrc/search_hamming_b1.hip — kernel instantiations for metric hamming over b1x8 storage (launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_LAUNCHERS(hamming_b1, metric_hamming_k, scalar_b1x8_k)
}
