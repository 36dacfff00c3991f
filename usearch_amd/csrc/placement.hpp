/**
 *  usearch_amd/csrc/placement.hpp — where arrays land in HBM decides how fast the walk runs over them.
 *
 *  The headline batch (10M x 768 f16, 10 000 queries, ef 608) takes 45.4, ≈ 47.5, ≈ 49.5 or 51.8 ms on the same bytes, stable for
 *  the life of the allocations (profiles/r02_placement.log). Round 3 took the effect apart (profiles/r03_placement/README.md):
 *
 *    * TWO allocations decide, independently, about 6 % each: the block of per-wave visited-set slabs of the workspace (268 MB;
 *      with the index untouched, re-allocating it alone flips the batch between 45.5 and 51.6 ms) and the matrix of stored rows.
 *      Fast + fast = 45.4 ms, one slow = 47.5 … 49.5, both slow = 51.8: the four levels.
 *    * No synthetic probe tells a slow placement from a fast one — not a dependency-free gather of rows (± 3 %, uncorrelated), not
 *      random page touches, not chains of dependent reads, not the slabs' own pattern of scattered compare-and-swaps (2.43 … 2.45
 *      µs per round everywhere). The counters see the same UTCL1 misses but translation requests that stay in flight 36 % longer
 *      (TCP_CLIENT_UTCL1_INFLIGHT ÷ TCP_UTCL1_TRANSLATION_MISS) — only the walk itself, gathers and atomics interleaved, feels it.
 *    * How an array is allocated (`hipMalloc`, `hipMemCreate` in chunks of 2 MB … the whole array) does not decide its speed.
 *
 *  So the engine draws placements and lets THE WALK judge them: the scratch block is drawn when a chip-filling launch needs a new
 *  one, each candidate timed by that very launch over its first queries (engine.hip `run_ladder`); the matrix is drawn once the
 *  index is resident, each candidate — a device-to-device copy — timed by a short self-search of stored rows
 *  (`snapshot_t::tune_placement`). Every path that creates a snapshot (loader, builder, drop-in, sharded step) gets both.
 *  This file keeps the allocation flavours and the probes used to rule the suspects out (diagnostics:
 *  scripts/placement_study.py).
 */
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace usearch_amd {

constexpr int placement_max_draws_k = 8;

struct placement_t {
    std::uint32_t draws = 0;                        ///< placements of the matrix tried (0 = too small to bother, or not tuned)
    std::uint32_t kept = 0;                         ///< which one was kept
    float judge_ms[placement_max_draws_k] = {0};    ///< milliseconds the judging self-search took on each draw (lower is better)
    float probe_ms = 0.f;                           ///< wall time the draws cost, allocation and copies included
};

/**
 *  Device memory for an array the walk gathers from: a `hipMalloc` block, or (USEARCH_AMD_VMM_CHUNK_MB = n, arrays of at least
 *  USEARCH_AMD_PLACEMENT_MIN_BYTES = 1 GiB) physical chunks of n MB mapped into one reserved range. Release with `placed_free`.
 */
hipError_t placed_malloc(void** out, std::size_t bytes, std::size_t row_bytes, placement_t* report);

/// One block of device memory for something the walk hits at random (the matrix, the lists, the visited-set slabs):
/// USEARCH_AMD_CONTIGUOUS = 1 asks the driver for PHYSICALLY CONTIGUOUS memory first (`hipDeviceMallocContiguous`: one range of
/// frames, so the page tables can describe it with the largest fragments the alignment allows and a translation-cache entry covers
/// far more than 2 MB), plain `hipMalloc` when that is refused. Release with `hipFree` / `placed_free`.
hipError_t block_malloc(void** out, std::size_t bytes);

/// Releases what `placed_malloc` returned (some placements are mapped through the virtual-memory API, not `hipMalloc`).
void placed_free(void* pointer);

/// The probe alone: GB/s of a dependency-free gather of random `row_bytes`-byte rows of `base[0 .. bytes)`.
hipError_t gather_probe(const void* base, std::size_t bytes, std::size_t row_bytes, float* gbps);

/// A probe of the address-translation path: every lane of every load reads 16 bytes of another random 4-KB page of
/// `base[0 .. bytes)`. Reports million page touches per second. (What separates a slow placement from a fast one is the latency
/// of translations, profiles/r03_placement/README.md; rows of 1.5 KB dilute it, single 16-byte reads do not.)
hipError_t translation_probe(const void* base, std::size_t bytes, float* mega_touches_per_second);

/// A probe of LATENCY: chains of dependent reads — where a chain goes next depends on what it just read — at random row starts
/// of `base[0 .. bytes)`, few enough chains in flight that nothing queues. Reports nanoseconds per dependent read. The walk is a
/// chain of dependent reads too; the throughput probes above run thousands of independent loads deep and hide what it feels.
hipError_t latency_probe(const void* base, std::size_t bytes, std::size_t row_bytes, float* nanoseconds);

/// Diagnostic: one physical allocation of `bytes` mapped, one after the other, at `views` fresh virtual ranges; `judge(view, ms)`
/// times the walk over each (returns an error message or null). Everything is unmapped and released before returning.
const char* remap_trial(std::size_t bytes, std::size_t views, const std::function<const char*(void*, float&)>& judge,
                        std::vector<float>& view_ms);

} // namespace usearch_amd
