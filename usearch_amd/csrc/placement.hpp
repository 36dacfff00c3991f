/**
 *  usearch_amd/csrc/placement.hpp — where a big, randomly gathered array lands in HBM decides how fast it can be walked.
 *
 *  The same 15.4-GB matrix of 1.5-KB rows is searched in 45.3 ms or in 51.9 ms depending on which physical memory `hipMalloc`
 *  happened to hand out (stable for the life of the allocation; profiles/r02_placement.log, profiles/r03_placement/). What
 *  round 3 found: the address-translation path sees the slow placements differently — same UTCL1 misses, but the requests stay
 *  in flight 36 % longer (TCP_CLIENT_UTCL1_INFLIGHT ÷ TCP_UTCL1_TRANSLATION_MISS) — and ONE physical allocation of the whole
 *  array, created with `hipMemCreate` and mapped into a reserved virtual range, walks at the fast speed every time (5 of 5
 *  restores, 45.30-45.52 ms), where `hipMalloc` blocks and mappings stitched from 2-MB … 1-GB chunks draw fast and slow alike.
 *  So the arrays the walk gathers from are allocated that way (`placed_malloc`); every path that creates a snapshot — loader,
 *  builder, drop-in, sharded step — goes through it and nothing above the engine has to know. The draw-and-probe machinery of
 *  the first attempt (several placements held side by side, a dependency-free gather timed on each, the fastest kept) is still
 *  here behind USEARCH_AMD_PLACEMENT_DRAWS for memory the mapping cannot serve; its probe turned out to be a weak predictor of
 *  the walk (±3 % against the walk's ±7 %), which is why it is no longer the default.
 */
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace usearch_amd {

constexpr int placement_max_draws_k = 8;

struct placement_t {
    std::uint32_t draws = 0;                        ///< placements tried (0 = the array was too small to bother)
    std::uint32_t kept = 0;                         ///< which one was kept
    float gather_gbps[placement_max_draws_k] = {0}; ///< random-row gather rate measured on each draw
    float probe_ms = 0.f;                           ///< wall time the draws cost, allocation included
};

/**
 *  Device memory for an array that will be read `row_bytes` at a time at random offsets. Arrays below
 *  USEARCH_AMD_PLACEMENT_MIN_BYTES (default 1 GiB) are plain `hipMalloc` blocks; larger ones are one physical allocation mapped
 *  into a reserved range (USEARCH_AMD_VMM_CHUNK_MB = 0 turns that off, n = chunks of n MB). USEARCH_AMD_PLACEMENT_DRAWS = n > 1
 *  additionally draws n placements and keeps the one whose gather probe is fastest. Release with `placed_free`.
 */
hipError_t placed_malloc(void** out, std::size_t bytes, std::size_t row_bytes, placement_t* report);

/// Releases what `placed_malloc` returned (some placements are mapped through the virtual-memory API, not `hipMalloc`).
void placed_free(void* pointer);

/// The probe alone: GB/s of a dependency-free gather of random `row_bytes`-byte rows of `base[0 .. bytes)`.
hipError_t gather_probe(const void* base, std::size_t bytes, std::size_t row_bytes, float* gbps);

/// A probe of the address-translation path: every lane of every load reads 16 bytes of another random 4-KB page of
/// `base[0 .. bytes)`. Reports million page touches per second. (What separates a slow placement from a fast one is the latency
/// of translations, profiles/r03_placement/README.md; rows of 1.5 KB dilute it, single 16-byte reads do not.)
hipError_t translation_probe(const void* base, std::size_t bytes, float* mega_touches_per_second);

} // namespace usearch_amd
