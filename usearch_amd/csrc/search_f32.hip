// usearch_amd/csrc/search_f32.hip — kernel instantiations for f32 storage (see launch_impl.hpp).
#include "launch_impl.hpp"
namespace usearch_amd {
USEARCH_AMD_DEFINE_NUMERIC_LAUNCHERS(f32, scalar_f32_k)
}
