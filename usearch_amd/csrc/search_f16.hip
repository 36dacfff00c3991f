// usearch_amd/csrc/search_f16.hip — kernel instantiations for f16 storage (see launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_NUMERIC_LAUNCHERS(f16, scalar_f16_k)
}
