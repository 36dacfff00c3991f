// usearch_amd/csrc/search_i8.hip — kernel instantiations for i8 storage (see launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_NUMERIC_LAUNCHERS(i8, scalar_i8_k)
}
