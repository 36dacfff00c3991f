/**
 *  usearch_amd/csrc/exact_tiled.hip — many-to-many EXACT search as a tiled matrix product on the MFMA units.
 *
 *  The reference's `exact_search_t` (/root/reference/include/usearch/index_plugins.hpp:2071-2164: a queries × dataset distance
 *  matrix, then a partial sort per query) and `index_gt::search_exact_` (index.hpp:4252-4268) are the one place on this path
 *  where operands ARE reused — every query meets every row — so unlike the graph walk this is a GEMM: a workgroup owns a tile
 *  of 64 queries and streams its share of the dataset through in tiles of 128 rows, rows are read once per 64 queries instead
 *  of once per query, the products run on `v_mfma_f32_32x32x16_{f16,bf16}` / `v_mfma_i32_32x32x32_i8`, and the per-query
 *  top-k is folded in the epilogue of every tile (a candidate survives only if it beats the query's current k-th best).
 *
 *  Pairs: cos, ip and l2sq over f16 / bf16 (f32 accumulation inside the matrix unit: results within the float tolerance of the
 *  wave-per-query kernel of kernels.hpp, which stays THE bit-exact path), and ip / cos / l2sq over i8 (exact int32 sums and
 *  the same closing arithmetic as `finalize_distance`: bit-identical to that kernel, ties included — selection is the total
 *  order (distance ↑, slot ↓) that `search_exact_`'s lower_bound inserts produce).
 *
 *  Both operands are row-major with the summation index contiguous, and the A and B fragments of these MFMA shapes use the
 *  same (lane half, element) → k assignment, so every lane simply loads 16 consecutive bytes of "its" query row and of "its"
 *  dataset row: no transposition anywhere. C/D: column (dataset row) = lane & 31, row (query) = (reg & 3) + 8·(reg >> 2) +
 *  4·(lane >> 5).
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>

#include "common.hpp"
#include "engine.hpp"
#include "host_util.hpp"

namespace usearch_amd {

namespace {

constexpr int tile_queries_k = 64;  ///< queries per workgroup
constexpr int tile_rows_k = 128;    ///< dataset rows per inner tile
constexpr int chunk_bytes_k = 128;  ///< bytes of every row staged per step of the summation loop (4 MFMA steps of 32 bytes)
constexpr int pitch_k = chunk_bytes_k + 16; ///< LDS row pitch: keeps 16-byte reads of consecutive rows off the same banks
constexpr int max_wanted_k = 64;    ///< per-query results the epilogue keeps (one lane per entry)

using f32x16_t = float __attribute__((ext_vector_type(16)));
using i32x16_t = int __attribute__((ext_vector_type(16)));
using f16x8_t = _Float16 __attribute__((ext_vector_type(8)));
using bf16x8_t = __bf16 __attribute__((ext_vector_type(8)));
using i32x4_t = int __attribute__((ext_vector_type(4)));

template <int scalar_ak> struct accumulator_gt {
    using type = f32x16_t;
};
template <> struct accumulator_gt<scalar_i8_k> {
    using type = i32x16_t;
};

template <int scalar_ak>
__device__ __forceinline__ typename accumulator_gt<scalar_ak>::type multiply(uint4 a, uint4 b,
                                                                             typename accumulator_gt<scalar_ak>::type c) {
    if constexpr (scalar_ak == scalar_f16_k)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else if constexpr (scalar_ak == scalar_bf16_k)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), c, 0, 0, 0);
}

/// Σx² of every row, as the closing arithmetic wants it: f32 for the float kinds, exact int32 for i8 (stored as bits).
template <int scalar_ak>
__global__ __launch_bounds__(256) void row_norms_kernel(const std::uint8_t* rows, std::uint64_t count, std::uint64_t stride,
                                                        std::uint32_t bytes, std::uint32_t* out) {
    // one wave per row, grid-stride: a launch may not exceed 2^32 threads, and 100M rows × 64 lanes would
    const std::uint32_t lane = threadIdx.x % 64;
    for (std::uint64_t row = blockIdx.x * 4ull + threadIdx.x / 64; row < count; row += (std::uint64_t)gridDim.x * 4) {
        const std::uint8_t* p = rows + row * stride;
        float sum = 0.f;
        int exact = 0;
        for (std::uint32_t b = lane * 2; b < bytes; b += 128) {
            if constexpr (scalar_ak == scalar_i8_k) {
                const int x = (std::int8_t)p[b], y = b + 1 < bytes ? (std::int8_t)p[b + 1] : 0;
                exact += x * x + y * y;
            } else {
                const std::uint32_t bits = (std::uint32_t)p[b] | ((std::uint32_t)p[b + 1] << 8);
                const float x = scalar_ak == scalar_bf16_k ? __builtin_bit_cast(float, bits << 16)
                                                           : (float)__builtin_bit_cast(_Float16, (std::uint16_t)bits);
                sum = __builtin_fmaf(x, x, sum);
            }
        }
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) {
            sum += __shfl_xor(sum, offset, 64);
            exact += __shfl_xor(exact, offset, 64);
        }
        if (lane == 0)
            out[row] = scalar_ak == scalar_i8_k ? (std::uint32_t)exact : __builtin_bit_cast(std::uint32_t, sum);
    }
}

/// The metric's closing arithmetic from the matrix unit's sum Σab and the two stored Σx² (bits of an f32, or an exact int32 for
/// i8) — the same expressions as `finalize_distance` of kernels.hpp.
template <int metric_ak, int scalar_ak, typename sum_at>
__device__ __forceinline__ float closing_distance(sum_at sum, std::uint32_t a2_bits, std::uint32_t b2_bits) {
    if constexpr (scalar_ak == scalar_i8_k) {
        const int ab = sum, a2 = (int)a2_bits, b2 = (int)b2_bits;
        if constexpr (metric_ak == metric_cos_k) { // metric_cos_i8_t, index_plugins.hpp:1583-1607
            const float a2f = __builtin_sqrtf((float)a2), b2f = __builtin_sqrtf((float)b2);
            return ab != 0 ? 1.f - (float)ab / (a2f * b2f) : 0.f;
        } else if constexpr (metric_ak == metric_ip_k) {
            return 1.f - (float)ab;
        } else { // metric_l2sq_i8_t 1613-1630
            return (float)(a2 + b2 - 2 * ab);
        }
    } else {
        const float ab = sum, a2 = __builtin_bit_cast(float, a2_bits), b2 = __builtin_bit_cast(float, b2_bits);
        if constexpr (metric_ak == metric_cos_k) { // metric_cos_gt, index_plugins.hpp:1334-1359
            if (a2 == 0.f && b2 == 0.f)
                return 0.f;
            if (a2 == 0.f || b2 == 0.f)
                return 1.f;
            return 1.f - ab / (__builtin_sqrtf(a2) * __builtin_sqrtf(b2));
        } else if constexpr (metric_ak == metric_l2sq_k) { // Σ(a−b)² (index_plugins.hpp:1365-1385) as Σa² + Σb² − 2Σab, never below 0
            const float d = a2 + b2 - 2.f * ab;
            return d > 0.f ? d : 0.f;
        } else {
            return 1.f - ab;
        }
    }
}

/// 1/√Σx² for the conservative bound of the wide kernel's epilogue; NaN (= "take the exact path") when the norm is zero or so
/// small that the reciprocal would overflow.
template <bool integers_ak> __device__ __forceinline__ float bound_scale(std::uint32_t norm_bits) {
    const float x = integers_ak ? (float)(int)norm_bits : __builtin_bit_cast(float, norm_bits);
    return x > 1e-30f ? __builtin_amdgcn_rsqf(x) : __builtin_nanf("");
}

/// (distance, slot) `a` goes before `b` in what `search_exact_` returns: closer first, the later slot first among equals.
__device__ __forceinline__ bool goes_before(float da, std::uint32_t sa, float db, std::uint32_t sb) {
    return da < db || (da == db && sa > sb);
}

/**
 *  grid = (query tiles, row partitions), 256 threads = 4 waves. Wave w multiplies all 64 queries of the tile with rows
 *  [32w, 32w + 32) of every row tile: two 32 × 32 accumulators.
 */
template <int metric_ak, int scalar_ak>
__global__ __launch_bounds__(256) void exact_tiled_kernel(const snapshot_view_t ix, const std::uint8_t* queries,
                                                          std::uint64_t query_stride, std::uint32_t query_count,
                                                          std::uint32_t wanted, std::uint64_t rows_per_partition,
                                                          const std::uint32_t* row_norms, const std::uint32_t* query_norms,
                                                          std::uint32_t map_keys, const std::uint32_t* allow_bits,
                                                          float* out_distances, std::uint64_t* out_keys,
                                                          std::uint64_t* out_counts) {
    using accumulator_t = typename accumulator_gt<scalar_ak>::type;
    extern __shared__ __attribute__((aligned(16))) std::uint8_t lds[];
    std::uint8_t* stage_q = lds;                                      // [64][pitch]
    std::uint8_t* stage_r = stage_q + tile_queries_k * pitch_k;       // [128][pitch]
    float* tile_d = reinterpret_cast<float*>(lds);                    // [64][129], aliases the staging area after the products
    constexpr std::uint32_t staging_bytes = (tile_queries_k + tile_rows_k) * pitch_k;
    constexpr std::uint32_t tile_bytes = tile_queries_k * (tile_rows_k + 1) * 4;
    constexpr std::uint32_t shared_bytes = staging_bytes > tile_bytes ? staging_bytes : tile_bytes;
    float* top_d = reinterpret_cast<float*>(lds + shared_bytes);      // [64][max_wanted_k]
    std::uint32_t* top_s = reinterpret_cast<std::uint32_t*>(top_d + tile_queries_k * max_wanted_k);
    std::uint32_t* top_n = top_s + tile_queries_k * max_wanted_k;      // [64]
    std::uint32_t* norms_q = top_n + tile_queries_k;                   // [64]
    std::uint32_t* norms_r = norms_q + tile_queries_k;                 // [128]
    std::uint32_t* valid_r = norms_r + tile_rows_k;                    // [128] 1 = a live member

    const std::uint32_t thread = threadIdx.x, wave = thread / 64, lane = thread % 64;
    const std::uint32_t first_query = blockIdx.x * tile_queries_k;
    const std::uint64_t first_row = (std::uint64_t)blockIdx.y * rows_per_partition;
    const std::uint64_t last_row = first_row + rows_per_partition < ix.size ? first_row + rows_per_partition : ix.size;
    const std::uint32_t bytes = ix.bytes_per_vector;
    const std::uint32_t chunks = (bytes + chunk_bytes_k - 1) / chunk_bytes_k;

    if (thread < tile_queries_k) {
        top_n[thread] = 0;
        norms_q[thread] = first_query + thread < query_count ? query_norms[first_query + thread] : 0u;
    }
    __syncthreads();

    for (std::uint64_t tile_row = first_row; tile_row < last_row; tile_row += tile_rows_k) {
        accumulator_t acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[t][r] = 0;
        if (thread < tile_rows_k) {
            const std::uint64_t row = tile_row + thread;
            const bool inside = row < last_row;
            norms_r[thread] = inside ? row_norms[row] : 0u;
            bool member = inside && (!ix.has_tombstones || ix.keys[row] != free_key_k);
            if (allow_bits && member) // the caller's predicate, one bit per slot (index.hpp:4260-4263)
                member = ((allow_bits[row >> 5] >> (row & 31)) & 1u) != 0;
            valid_r[thread] = member ? 1u : 0u;
        }
        for (std::uint32_t chunk = 0; chunk < chunks; ++chunk) {
            // ---- stage 128 bytes of every query and of every row of the tile: 8 consecutive threads fetch one row's 128 bytes
            const std::uint32_t segment = thread % 8, byte = chunk * chunk_bytes_k + segment * 16;
            {
                for (std::uint32_t pass = 0; pass < tile_queries_k / 32; ++pass) {
                    const std::uint32_t i = pass * 32 + thread / 8;
                    uint4 value = {0u, 0u, 0u, 0u};
                    if (first_query + i < query_count && byte < bytes) {
                        const std::uint8_t* source = queries + (std::uint64_t)(first_query + i) * query_stride + byte;
                        if (byte + 16 <= bytes && (query_stride % 16 == 0) && ((std::uintptr_t)queries % 16 == 0))
                            value = *reinterpret_cast<const uint4*>(source);
                        else { // ragged end of a row, or an unaligned batch: byte by byte
                            std::uint8_t parts[16] = {0};
                            for (std::uint32_t b = 0; b < 16 && byte + b < bytes; ++b)
                                parts[b] = source[b];
                            value = *reinterpret_cast<const uint4*>(parts);
                        }
                    }
                    *reinterpret_cast<uint4*>(stage_q + i * pitch_k + segment * 16) = value;
                }
                for (std::uint32_t pass = 0; pass < tile_rows_k / 32; ++pass) {
                    const std::uint32_t j = pass * 32 + thread / 8;
                    const std::uint64_t row = tile_row + j;
                    uint4 value = {0u, 0u, 0u, 0u};
                    // stored rows are 16-byte aligned and zero padded to whole 16-byte chunks (`ix.chunks`)
                    if (row < last_row && byte < ix.chunks * 16u)
                        value = *reinterpret_cast<const uint4*>(ix.vectors + row * ix.row_stride + byte);
                    *reinterpret_cast<uint4*>(stage_r + j * pitch_k + segment * 16) = value;
                }
            }
            __syncthreads();
            // ---- four steps of 32 bytes: lane half h of every fragment owns bytes [32·step + 16h, +16) of its row
#pragma unroll
            for (std::uint32_t step = 0; step < chunk_bytes_k / 32; ++step) {
                const std::uint32_t offset = step * 32 + (lane >> 5) * 16;
                const uint4 b = *reinterpret_cast<const uint4*>(stage_r + (wave * 32 + (lane & 31)) * pitch_k + offset);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint4 a = *reinterpret_cast<const uint4*>(stage_q + (t * 32 + (lane & 31)) * pitch_k + offset);
                    acc[t] = multiply<scalar_ak>(a, b, acc[t]);
                }
            }
            __syncthreads();
        }

        // ---- distances of the tile into LDS (the staging area is free now): D[query][row of the tile]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const std::uint32_t i = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const std::uint32_t j = wave * 32 + (lane & 31);
                const float distance = closing_distance<metric_ak, scalar_ak>(acc[t][r], norms_q[i], norms_r[j]);
                tile_d[i * (tile_rows_k + 1) + j] = valid_r[j] ? distance : __builtin_inff();
            }
        }
        __syncthreads();

        // ---- fold: wave w owns queries [16w, 16w + 16); a candidate enters a query's list if the list is not full or it goes
        //      before the list's last entry. Lists are kept in order, one lane per entry.
        for (std::uint32_t local = 0; local < tile_queries_k / 4; ++local) {
            const std::uint32_t i = wave * (tile_queries_k / 4) + local;
            if (first_query + i >= query_count)
                break;
            float* list_d = top_d + i * max_wanted_k;
            std::uint32_t* list_s = top_s + i * max_wanted_k;
            std::uint32_t size = top_n[i];
#pragma unroll
            for (std::uint32_t half = 0; half < tile_rows_k / 64; ++half) {
                const std::uint32_t j = half * 64 + lane;
                const float candidate = tile_d[i * (tile_rows_k + 1) + j];
                const std::uint32_t slot = (std::uint32_t)(tile_row + j);
                float worst_d = size ? list_d[size - 1] : 0.f;
                std::uint32_t worst_s = size ? list_s[size - 1] : 0u;
                std::uint64_t pending = __ballot(candidate != __builtin_inff() &&
                                                 (size < wanted || goes_before(candidate, slot, worst_d, worst_s)));
                while (pending) {
                    const std::uint32_t source = (std::uint32_t)__ffsll((long long)pending) - 1;
                    pending &= pending - 1;
                    const float d = __shfl(candidate, (int)source, 64);
                    const std::uint32_t s = (std::uint32_t)(tile_row + half * 64 + source);
                    if (size == wanted && !goes_before(d, s, worst_d, worst_s))
                        continue;
                    // position = entries that go before the newcomer; the ones at and after it move one cell down
                    const bool mine = lane < size;
                    const float my_d = mine ? list_d[lane] : 0.f;
                    const std::uint32_t my_s = mine ? list_s[lane] : 0u;
                    const std::uint32_t position = (std::uint32_t)__popcll(__ballot(mine && goes_before(my_d, my_s, d, s)));
                    const std::uint32_t grown = size < wanted ? size + 1 : size;
                    if (mine && lane >= position && lane + 1 < grown)
                        list_d[lane + 1] = my_d, list_s[lane + 1] = my_s;
                    if (lane == position)
                        list_d[position] = d, list_s[position] = s;
                    size = grown;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                    worst_d = list_d[size - 1], worst_s = list_s[size - 1];
                }
            }
            if (lane == 0)
                top_n[i] = size;
        }
        __syncthreads(); // the tile's distances are consumed: the area becomes staging again
    }

    // ---- this partition's lists, laid out [partition][query][wanted] like the wave-per-query kernel's
    for (std::uint32_t cell = thread; cell < tile_queries_k * wanted; cell += 256) {
        const std::uint32_t i = cell / wanted, position = cell % wanted;
        const std::uint32_t q = first_query + i;
        if (q >= query_count)
            continue;
        const std::uint64_t out = ((std::uint64_t)blockIdx.y * query_count + q) * wanted + position;
        std::uint64_t key = 0;
        std::uint32_t bits = signaling_nan_bits_k;
        if (position < top_n[i]) {
            const std::uint32_t slot = top_s[i * max_wanted_k + position];
            key = map_keys ? ix.keys[slot] : (std::uint64_t)slot;
            bits = __builtin_bit_cast(std::uint32_t, top_d[i * max_wanted_k + position]);
        }
        out_keys[out] = key;
        reinterpret_cast<std::uint32_t*>(out_distances)[out] = bits;
    }
    if (thread < tile_queries_k && first_query + thread < query_count)
        out_counts[(std::uint64_t)blockIdx.y * query_count + first_query + thread] = top_n[thread];
}

// ---------------------------------------------------------------------------------------------------------------------
//  The wide tile: 256 queries × 256 rows per workgroup of 8 waves; both operands go global → LDS by LDS-DMA
//  (`global_load_lds_dwordx4`, no staging registers, no `ds_write`) into two XOR-swizzled buffers; register epilogue
// ---------------------------------------------------------------------------------------------------------------------

constexpr int wide_queries_k = 256; ///< queries per workgroup: a row is read once per 256 queries
constexpr int wide_rows_k = 256;    ///< dataset rows per tile: a query chunk is read once per 256 rows
constexpr int wide_blocks_k = wide_rows_k / 32; ///< 32-row blocks of a tile = accumulators of a wave
constexpr int wide_threads_k = 512; ///< 8 waves; wave w multiplies queries [32w, 32w + 32) with the tile's 256 rows
constexpr int wide_wanted_k = 16;   ///< results per query this kernel keeps (a lane per entry in the ordered insert, entries in the output arrays)
constexpr int wide_buffers_k = 2;   ///< staging buffers: one being multiplied, one being filled
constexpr int wide_stage_rows_k = wide_queries_k + wide_rows_k;
constexpr std::uint32_t wide_stage_bytes_k = wide_stage_rows_k * chunk_bytes_k;         ///< a buffer: [512][128] bytes, no padding
constexpr int wide_fills_k = wide_stage_rows_k / 8 / (wide_threads_k / 64);              ///< fill instructions per wave per chunk: 8

inline std::uint64_t wide_padded_stride(std::uint64_t bytes_per_vector) {
    return (bytes_per_vector + chunk_bytes_k - 1) / chunk_bytes_k * chunk_bytes_k;
}
/// Results per query up to which the lists fit LDS next to the two staging buffers (an insert is then an LDS affair of a few
/// hundred cycles instead of a global round trip and a fence: the epilogue and, through it, the barrier shrink).
constexpr int wide_lds_lists_k = 10;
constexpr std::uint32_t wide_lds_bytes(bool lds_lists, std::uint32_t wanted) {
    return wide_buffers_k * wide_stage_bytes_k + wide_queries_k * 4 * 5 + 2 * wide_rows_k * 4 + 2 * 512 * 4 +
           (lds_lists ? wide_queries_k * wanted * 8 : 0);
}

/// A float as an unsigned integer of the same order (negative distances exist: 1 − Σab), so that `atomicMin` keeps the smallest.
__device__ __forceinline__ std::uint32_t ordered_bits(float x) {
    const std::uint32_t bits = __builtin_bit_cast(std::uint32_t, x);
    return bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
/// All ones ("nothing published yet") decode to a NaN.
__device__ __forceinline__ float from_ordered_bits(std::uint32_t ordered) {
    return __builtin_bit_cast(float, ordered ^ ((ordered >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

/// Where the 16-byte piece `piece` of staged row `row` sits inside the row's 128 bytes. A `ds_read_b128` is served in groups of 16
/// lanes — 16 different rows at one piece index — and rows 128 bytes apart fall on the same banks every second row; XOR-ing the
/// piece index with bits 1…3 of the row spreads each group over all sixteen 16-byte bank slots (no conflict). The DMA writes lanes
/// linearly, so the swizzle is applied to the SOURCE: lane i of a fill instruction fetches the piece that belongs in slot i.
__device__ __forceinline__ std::uint32_t wide_swizzle(std::uint32_t row, std::uint32_t piece) { return piece ^ ((row >> 1) & 7u); }

#ifdef USEARCH_AMD_EXACT_PHASES
/// Diagnostic build (`make EXTRA=-DUSEARCH_AMD_EXACT_PHASES`): shader-clock ticks of wave 0 of every workgroup, summed — [0] multiply
/// (fragment reads, MFMAs, the fills issued between them), [1] fold, [2] waiting for the own fills, [3] at the barrier, [4] chunks
__device__ unsigned long long exact_phase_ticks[8];
#define UA_PHASE_TICK(phase)                                                                                                           \
    {                                                                                                                                  \
        const std::uint64_t now = __builtin_amdgcn_s_memtime();                                                                        \
        phase_ticks[phase] += now - phase_mark;                                                                                        \
        phase_mark = now;                                                                                                              \
    }
#else
#define UA_PHASE_TICK(phase)
#endif

using global_bytes_t = const __attribute__((address_space(1))) void*;
using lds_bytes_t = __attribute__((address_space(3))) void*;

/**
 *  grid = 1-D, ONE round of the chip where the batch allows (`wide_plan`): 10 000 queries = 40 query tiles = 5 per XCD × 6
 *  partitions = 240 workgroups; what an XCD's L2 holds is its 5 query tiles (2 MB for 768-d f16) and the row tiles its 6
 *  partitions are streaming, each shared by 5 workgroups. Few partitions matter: every partition rebuilds the top-k race of its
 *  queries from scratch — k·ln(rows/k) inserts per query AND partition (the 64 partitions of round 3: 6 200 inserts per query, 5 per
 *  tile and wave at ≈ 3 500 cycles each = three quarters of the epilogue; 6 partitions: 720).
 *
 *  What bounds this kernel is the path global → LDS of a compute unit (PMC, profiles/r04_exact/: matrix unit 31 % busy, vector
 *  ALU 26 %, LDS 21 %, the texture path waiting on L2 43 % of the time with a 256 × 128 tile). Hence the shape: the tile is as
 *  square as two LDS buffers allow — 256 queries × 256 rows stage 256 bytes per MFMA where 256 × 128 staged 384 — every wave
 *  multiplies ITS 32 queries with all 256 rows (eight 32 × 32 accumulators: nothing about a query is shared between waves), and
 *  nothing but the staged bytes crosses that path:
 *    · fills are LDS-DMA (8 rows × 128 bytes per instruction, 8 instructions per wave and chunk), issued in four parts in the
 *      shadow of the chunk's MFMA groups, sources kept as running pointers;
 *    · fragment reads are inline assembly with hand-counted `lgkmcnt` waits: the compiler cannot tell a read of THIS buffer from the
 *      fill of the OTHER one and would put `s_waitcnt vmcnt(0)` in front of every chunk's first read;
 *    · the accumulators never visit LDS: a lane tests its 128 sums against the queries' bounds with one multiply and one compare
 *      per sum (cos: Σab·rsq(Σb²) against (1 − bound − 10⁻⁵)·√Σa², kept per query register), the compares' lane masks are ORed on
 *      the scalar unit, and only sums that may enter a list take the exact closing arithmetic and the ordered insert;
 *    · a query's bound is SHARED by all partitions (`shared_bounds`: the smallest k-th best any partition has reached, atomicMin
 *      of ordered bits, fetched by DMA with every tile): a row farther than k rows some partition already holds cannot be among
 *      the query's k nearest, whichever partition it lies in. Inserts drop from k·ln(rows/k) per query AND partition to per query,
 *      which is what lets the lists live where they end up — this partition's cells of the output arrays — instead of in LDS.
 *  Same lists as the 64-query kernel: the best `wanted` under (distance ↑, slot ↓), whatever the order of arrival; a partition may
 *  hold fewer than `wanted` of them (the merge takes counts).
 */
template <int metric_ak, int scalar_ak, bool lds_lists_ak>
__global__ __launch_bounds__(wide_threads_k) void exact_wide_kernel(const snapshot_view_t ix, const std::uint8_t* padded_queries,
                                                                    std::uint64_t padded_stride, std::uint32_t query_count,
                                                                    std::uint32_t wanted, std::uint64_t rows_per_partition,
                                                                    std::uint32_t query_tiles, std::uint32_t tiles_per_xcd,
                                                                    const std::uint32_t* row_norms, const std::uint32_t* query_norms,
                                                                    std::uint32_t map_keys, const std::uint32_t* allow_bits,
                                                                    std::uint32_t* shared_bounds, float* out_distances,
                                                                    std::uint64_t* out_keys, std::uint64_t* out_counts,
                                                                    std::uint32_t knock) {
#ifndef USEARCH_AMD_EXPERIMENT_EXACT_KNOCKOUT
    knock = 0u; // product builds: every knock-out below compiles away (`make EXTRA=-DUSEARCH_AMD_EXPERIMENT_EXACT_KNOCKOUT OUT=… OBJ=…`
                // builds the copy scripts/exact_knockout.py loads through USEARCH_AMD_LIBRARY)
#endif
    // `knock` (USEARCH_AMD_EXACT_KNOCKOUT, timing experiments only — results are wrong with any bit set): 1 = no fold, 2 = no fills
    // after the prologue's, 4 = no wait for the fills and no barrier, 8 = the fold's thresholds refreshed for the first tile only, 16 = the
    // fold without its per-block tests, 32 = the general fold where the fused one would run (results stay right), 64 = the fused fold without what follows a block's test
    using accumulator_t = typename accumulator_gt<scalar_ak>::type;
    constexpr bool integers = scalar_ak == scalar_i8_k;
    using sum_t = typename std::conditional<integers, int, float>::type;
    using u32x4_t = std::uint32_t __attribute__((ext_vector_type(4))); // a native 128-bit register operand for the assembler
    extern __shared__ __attribute__((aligned(1024))) std::uint8_t lds[];
    std::uint32_t* top_n = reinterpret_cast<std::uint32_t*>(lds + wide_buffers_k * wide_stage_bytes_k); // [256] entries per list
    std::uint32_t* norms_q = top_n + wide_queries_k;   // [256] Σa²
    float* roots_q = reinterpret_cast<float*>(norms_q + wide_queries_k); // [256] cos: √Σa², NaN for a zero norm ("always the exact path")
    float* limit = roots_q + wide_queries_k; // [256] THIS partition's k-th best per query (+inf while its list is filling): felt from
                                             // the next tile on, where the shared bound arrives a tile or two late
    std::uint32_t* norms_r = reinterpret_cast<std::uint32_t*>(limit + wide_queries_k); // [2][256] Σb² of the rows of the tile being multiplied / being fetched
                                                       // (fills run one chunk ahead: the next tile at most): slot = tile mod 2
    std::uint32_t* others = norms_r + 2 * wide_rows_k; // [2][8][64] the queries' shared bounds as fetched with the tile (same
                                                       // slots): a fill writes 64 cells, a wave's 32 queries twice
    float* folded = reinterpret_cast<float*>(others + 2 * 512); // [256] what the fused fold put into the sums per query (f16 cos / ip)
    float* lists_d = folded + wide_queries_k;                                           // with `lds_lists_ak`: [256][wanted] distances
    std::uint32_t* lists_s = reinterpret_cast<std::uint32_t*>(lists_d + wide_queries_k * wanted); // … and slots

    // the wave's number as a SCALAR: everything that only depends on it — the LDS targets of the fills (they travel in M0), the
    // wave's slices of the per-query arrays — is then scalar arithmetic instead of vector registers held (and spilled) across the loop
    const std::uint32_t thread = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(thread / 64), lane = thread % 64;
    // ---- which (query tile, partition) this workgroup is (wide_plan): workgroups go to the XCDs round-robin by their linear
    //      index; an XCD's `tiles_per_xcd` consecutive workgroups work on ONE partition for different query tiles (the row tile
    //      they stream is shared through the XCD's L2). A batch of ≥ 8 query tiles is dealt over the XCDs (tile = 8·g + xcd, every
    //      XCD sees every partition); a smaller one is handled whole by every XCD (partitions ≡ xcd mod 8).
    const std::uint32_t xcd = blockIdx.x % 8, sequence = blockIdx.x / 8;
    const std::uint32_t query_tile = query_tiles >= 8 ? (sequence % tiles_per_xcd) * 8 + xcd : sequence % tiles_per_xcd;
    const std::uint32_t partition = query_tiles >= 8 ? sequence / tiles_per_xcd : sequence / tiles_per_xcd * 8 + xcd;
    if (query_tile >= query_tiles) // uniform: the last round of tiles may be short
        return;
    const std::uint32_t first_query = query_tile * wide_queries_k;
    const std::uint64_t first_row = (std::uint64_t)partition * rows_per_partition;
    const std::uint64_t last_row = first_row + rows_per_partition < ix.size ? first_row + rows_per_partition : ix.size;
    const std::uint32_t bytes = ix.bytes_per_vector, row_bytes = ix.chunks * 16u;
    const std::uint32_t chunks = (bytes + chunk_bytes_k - 1) / chunk_bytes_k;
    const std::uint32_t tiles = first_row < last_row ? (std::uint32_t)((last_row - first_row + wide_rows_k - 1) / wide_rows_k) : 0u;
    const std::uint32_t total = tiles * chunks;

    for (std::uint32_t i = thread; i < wide_queries_k; i += wide_threads_k) {
        top_n[i] = 0;
        limit[i] = __builtin_inff();
        norms_q[i] = first_query + i < query_count ? query_norms[first_query + i] : 0u;
        const float a2 = integers ? (float)(int)norms_q[i] : __builtin_bit_cast(float, norms_q[i]);
        roots_q[i] = a2 > 1e-30f ? __builtin_sqrtf(a2) : __builtin_nanf("");
    }
    __syncthreads();

    // ---- per lane: the 16 queries its accumulator registers belong to (register r ↔ query 32·wave + (r&3) + 8(r>>2) + 4(lane>>5))
    std::uint32_t exists = 0;   // bit r: that query is inside the batch
    std::uint32_t query_norm[16];
    float bound[16];            // the query's shared bound as of the last tile head (+inf: nobody has k results yet)
    float threshold[16];        // cos: (1 − bound − 10⁻⁵)·√Σa², what Σab·rsq(Σb²) has to reach; +inf for a query outside the batch
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const std::uint32_t i = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const bool inside = first_query + i < query_count;
        exists |= (inside ? 1u : 0u) << r;
        query_norm[r] = norms_q[i];
        bound[r] = __builtin_inff();
        threshold[r] = inside ? -__builtin_inff() : __builtin_inff();
    }
    /// `shared[r]`: the ordered bits of the shared bound of register r's query, as fetched with the tile; `own[r]`: this partition's
    /// k-th best; `roots[r]`: the query's √Σa²
    auto refresh_thresholds = [&](const std::uint32_t (&shared)[16], const float (&own)[16], const float (&roots)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float other = from_ordered_bits(shared[r]);
            bound[r] = other < own[r] ? other : own[r]; // a NaN ("nothing published") keeps our own
            if constexpr (metric_ak == metric_cos_k) { // −inf while nobody has k results yet: everything may enter; −inf too for a
                const float product = (1.f - (bound[r] + 1e-5f)) * roots[r]; // query of zero norm (its root is a NaN): the exact path
                threshold[r] = (exists >> r) & 1u ? (product == product ? product : -__builtin_inff()) : __builtin_inff();
            }
        }
    };

    // ---- LDS-DMA fills. A fill instruction of a wave writes 1 024 consecutive bytes = 8 staged rows; wave w fills rows
    //      64·pass + 8w … + 8 (4 passes of queries, 4 of dataset rows); lane i brings the piece that belongs in slot i of "its" row
    //      (wide_swizzle; the same piece in every pass: row bases are multiples of 64). The BUFFER form of the instruction
    //      (`buffer_load_dwordx4 … offen lds`): the 256 padded queries of the workgroup are one resource for the whole launch, the
    //      256 rows of a tile one resource per tile (scalar arithmetic); a lane's share of an address is ONE 32-bit offset that
    //      never changes — its row inside the block and its piece — and pass and chunk travel in the instruction's scalar offset.
    //      A fill is then two scalar instructions (M0, the offset) and the load: no 64-bit vector address per fill and chunk (the
    //      global form moved eight of them per wave and chunk through the vector ALU and twice the address registers to the
    //      texture unit). No branch on the data path: a row past the partition's end lies outside its tile's resource (whatever
    //      the staging cell then holds, the epilogue drops that column: `live`), bytes past a row's last 16-byte chunk re-read that
    //      chunk (ragged rows only: one subtraction on the last chunk) and meet the zeros the padded queries hold there.
    constexpr int resource_flags = 0x00020000; // raw buffer, 32-bit data format: what the range check needs on gfx94x / gfx950
    const std::uint32_t fill_row = wave * 8 + lane / 8; // + 64·pass
    const std::uint32_t fill_piece = wide_swizzle(fill_row, lane & 7u);
    const std::uint32_t row_stride = (std::uint32_t)ix.row_stride, query_stride = (std::uint32_t)padded_stride; // × 256 fits 31 bits (launch_wide)
    const __amdgpu_buffer_rsrc_t query_resource = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<std::uint8_t*>(padded_queries + (std::uint64_t)first_query * padded_stride), 0, (int)(wide_queries_k * query_stride),
        resource_flags);
    // (the two pass strides as scalars the compiler cannot re-derive from the kernel's arguments: short of scalar registers it
    // would otherwise RE-LOAD an argument in the middle of a chunk, and the wait for that load — the counter it shares with LDS —
    // sits out the fragment reads in flight)
    std::uint32_t query_pass_bytes = 64u * query_stride, row_pass_bytes = 64u * row_stride;
    asm volatile("" : "+s"(query_pass_bytes), "+s"(row_pass_bytes));
    const std::uint32_t query_offset = fill_row * query_stride + fill_piece * 16; // + (64·pass rows + the chunk) in the scalar offset
    const std::uint32_t row_offset = fill_row * row_stride + fill_piece * 16;
    const bool ragged = row_bytes % chunk_bytes_k != 0; // the last chunk of a row ends before 128 bytes
    // the tile's resource as three scalars (the compiler must SEE that they are: a resource it takes for lane-dependent is
    // applied lane by lane, in a loop around every fill)
    std::uint32_t row_base_low = 0, row_base_high = 0, row_records = 0;
    std::uint32_t fetch_tile = 0, fetch_chunk = 0;
    auto begin_tile = [&]() {
        const std::uint64_t tile_row = first_row + (std::uint64_t)fetch_tile * wide_rows_k;
        const std::uint64_t rows_here = tile_row < last_row ? (last_row - tile_row < wide_rows_k ? last_row - tile_row : wide_rows_k) : 0;
        const std::uint64_t base = (std::uint64_t)(ix.vectors + (tile_row < last_row ? tile_row : 0) * ix.row_stride);
        row_base_low = __builtin_amdgcn_readfirstlane((std::uint32_t)base);
        row_base_high = __builtin_amdgcn_readfirstlane((std::uint32_t)(base >> 32));
        row_records = __builtin_amdgcn_readfirstlane((std::uint32_t)rows_here * row_stride);
    };
    auto fill_queries = [&](std::uint32_t buffer, int first_pass) { // two of the four query passes
        std::uint8_t* stage = lds + buffer * wide_stage_bytes_k;
#pragma unroll
        for (int pass = first_pass; pass < first_pass + 2; ++pass)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(query_resource, (lds_bytes_t)(stage + (pass * 64 + wave * 8) * chunk_bytes_k), 16,
                                                     (int)query_offset,
                                                     (int)__builtin_amdgcn_readfirstlane(pass * query_pass_bytes + fetch_chunk * chunk_bytes_k), 0, 0);
    };
    auto fill_rows = [&](std::uint32_t buffer, int first_pass) { // two of the four row passes
        std::uint8_t* stage = lds + buffer * wide_stage_bytes_k;
        const __amdgpu_buffer_rsrc_t row_resource = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((std::uint64_t)row_base_high << 32) | row_base_low), 0, (int)row_records, resource_flags);
        std::uint32_t offset = row_offset;
        if (ragged && fetch_chunk + 1 == chunks) { // bytes past the row's last 16-byte chunk: that chunk again
            const std::uint32_t byte = fetch_chunk * chunk_bytes_k + fill_piece * 16;
            offset -= byte < row_bytes ? 0u : byte - (row_bytes - 16);
        }
#pragma unroll
        for (int pass = first_pass; pass < first_pass + 2; ++pass)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(row_resource, (lds_bytes_t)(stage + (wide_queries_k + pass * 64 + wave * 8) * chunk_bytes_k),
                                                     16, (int)offset,
                                                     (int)__builtin_amdgcn_readfirstlane(pass * row_pass_bytes + fetch_chunk * chunk_bytes_k), 0, 0);
    };
    auto fill_tile_head_and_advance = [&]() {
        if (fetch_chunk == 0) { // this wave's 32 queries' shared bounds as they stand now (lanes 32 … 63 repeat them), same DMA
            std::uint32_t lane_here = lane; // opaque: once per tile, two addresses computed on the spot instead of four registers
            asm volatile("" : "+v"(lane_here)); // held across the loop (and spilled: a reload waits for every fill in flight)
            const std::uint32_t q = first_query + wave * 32 + (lane_here & 31);
            __builtin_amdgcn_global_load_lds((global_bytes_t)(shared_bounds + (q < query_count ? q : query_count - 1)),
                                             (lds_bytes_t)(others + (fetch_tile & 1u) * 512 + wave * 64), 4, 0, 0);
            if (wave < wide_rows_k / 64) { // the tile's Σb², 64 per wave
                const std::uint64_t wanted_row = first_row + (std::uint64_t)fetch_tile * wide_rows_k + wave * 64 + lane_here;
                __builtin_amdgcn_global_load_lds((global_bytes_t)(row_norms + (wanted_row < last_row ? wanted_row : last_row - 1)),
                                                 (lds_bytes_t)(norms_r + (fetch_tile & 1u) * wide_rows_k + wave * 64), 4, 0, 0);
            }
        }
        if (++fetch_chunk == chunks) {
            fetch_chunk = 0, ++fetch_tile;
            begin_tile();
        }
    };
    begin_tile();

    accumulator_t acc[wide_blocks_k];
#pragma unroll
    for (int u = 0; u < wide_blocks_k; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[u][r] = 0;
    std::uint32_t work_tile = 0, work_chunk = 0;

    // ---- fragment reads. Lane L reads row (base + (L & 31)), piece 2·step + (L >> 5), through the swizzle of that row; row bases
    //      are multiples of 32, so the swizzled piece only depends on the lane and the step: four byte offsets, computed once.
    //      Per step a wave needs its query fragment and eight row fragments for eight MFMAs. Registers: the query fragment twice
    //      (the next step's is requested while this step's is still being multiplied), the row fragments once, in two halves of
    //      four — a half is requested again as soon as its four MFMAs are issued, and lands while the other half is multiplied.
    const std::uint32_t lane_swizzle = ((lane & 31u) >> 1) & 7u, lane_half = lane >> 5;
    const std::uint32_t lds_base = (std::uint32_t)(std::uintptr_t)(lds_bytes_t)lds;
    const std::uint32_t query_fragments = lds_base + (wave * 32 + (lane & 31)) * chunk_bytes_k;
    const std::uint32_t row_fragments = lds_base + (wide_queries_k + (lane & 31)) * chunk_bytes_k;
    std::uint32_t step_offset[4];
#pragma unroll
    for (int step = 0; step < 4; ++step)
        step_offset[step] = ((2u * step + lane_half) ^ lane_swizzle) * 16u;
    static_assert(chunk_bytes_k / 32 == 4 && wide_blocks_k == 8, "multiply_chunk is written out for four steps of eight row blocks");
    auto product = [&](u32x4_t a, u32x4_t b, accumulator_t c) -> accumulator_t {
        return multiply<scalar_ak>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, b), c);
    };
#define UA_REQUEST_QUERY(target, step)                                                                                                 \
    asm volatile("ds_read_b128 %0, %1" : "=&v"(target) : "v"(query_fragments + buffer_bytes + step_offset[step]));
#define UA_REQUEST_ROWS(b0, b1, b2, b3, first_block, step)                                                                            \
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\t"                                                                                   \
                 "ds_read_b128 %1, %4 offset:%6\n\t"                                                                                   \
                 "ds_read_b128 %2, %4 offset:%7\n\t"                                                                                   \
                 "ds_read_b128 %3, %4 offset:%8"                                                                                       \
                 : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)                                                                          \
                 : "v"(row_fragments + buffer_bytes + step_offset[step]), "n"((first_block)*4096), "n"((first_block)*4096 + 4096),     \
                   "n"((first_block)*4096 + 8192), "n"((first_block)*4096 + 12288));
#define UA_AWAIT(still_in_flight, a, b0, b1, b2, b3)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #still_in_flight ")" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    /// Multiplies the chunk in `buffer`; with `filling`, the next chunk goes into `target` meanwhile, a part of its fills behind each
    /// step's MFMAs. `fresh` (a tile's first chunk): the first step's MFMAs take a literal zero as their addend — the accumulators
    /// are never cleared by moves.
    ///
    /// The workgroup's rendezvous sits in the MIDDLE of a chunk's last step. Once a wave's last fragment reads of the buffer have
    /// landed (the wait in front of the step's second half) it has no business left with the buffer: it waits for its own fills of
    /// the next chunk and meets the others there, the step's last four MFMAs still to come — their operands are in registers. On
    /// the far side of the barrier (the top of the next call, `fresh` = false) it first requests the next chunk's first fragments,
    /// then issues those four MFMAs, and the requests fly meanwhile. Before (all of a chunk, barrier, requests) every chunk opened
    /// with eight waves waiting for their first LDS reads and the matrix pipes empty. A tile's LAST chunk is finished in place — the
    /// fold needs the sums, and it has to stay in front of the barrier (behind it, a fast wave's next tile head would land on the
    /// Σb² a slow wave's fold is still reading) — so a tile's first chunk (`fresh`) has nothing deferred to issue.
    /// What crosses the barrier and the loop's back-edge are five LANDED fragments (`a_odd`, `h0` … `h3`); a register with a read
    /// still in flight never leaves a straight run of these statements — the compiler knows nothing of the read behind an asm
    /// statement's output and is free to copy or spill such a register anywhere else (it did: results went wrong).
    u32x4_t a_odd, h0, h1, h2, h3;
    auto multiply_chunk = [&](auto fresh_tag, std::uint32_t buffer, bool filling, std::uint32_t target, bool last_of_tile) {
        constexpr bool fresh = decltype(fresh_tag)::value;
        const accumulator_t zero = {};
        const std::uint32_t buffer_bytes = buffer * wide_stage_bytes_k;
        u32x4_t a_even, l0, l1, l2, l3;
        __builtin_amdgcn_s_setprio(1);
#define UA_LOW_HALF(a_now, step)                                                                                                       \
        UA_AWAIT(4, a_now, l0, l1, l2, l3) /* the query fragment and the low half are there; the high half may be in flight */       \
        acc[0] = product(a_now, l0, fresh && (step) == 0 ? zero : acc[0]);                                                             \
        acc[1] = product(a_now, l1, fresh && (step) == 0 ? zero : acc[1]);                                                             \
        acc[2] = product(a_now, l2, fresh && (step) == 0 ? zero : acc[2]);                                                             \
        acc[3] = product(a_now, l3, fresh && (step) == 0 ? zero : acc[3]);
#define UA_HIGH_HALF(a_now, first_of_tile)                                                                                             \
        acc[4] = product(a_now, h0, (first_of_tile) ? zero : acc[4]);                                                                  \
        acc[5] = product(a_now, h1, (first_of_tile) ? zero : acc[5]);                                                                  \
        acc[6] = product(a_now, h2, (first_of_tile) ? zero : acc[6]);                                                                  \
        acc[7] = product(a_now, h3, (first_of_tile) ? zero : acc[7]);
#define UA_STEP(a_now, a_next, step, fill_statement)                                                                                   \
        UA_LOW_HALF(a_now, step)                                                                                                       \
        UA_REQUEST_QUERY(a_next, (step) + 1)                                                                                           \
        UA_REQUEST_ROWS(l0, l1, l2, l3, 0, (step) + 1)                                                                                 \
        UA_AWAIT(5, a_now, h0, h1, h2, h3) /* the high half is there; the five just requested may be in flight */                      \
        UA_HIGH_HALF(a_now, fresh && (step) == 0)                                                                                      \
        UA_REQUEST_ROWS(h0, h1, h2, h3, 4, (step) + 1)                                                                                 \
        if (filling) {                                                                                                                 \
            fill_statement;                                                                                                            \
        }
        UA_REQUEST_QUERY(a_even, 0)
        UA_REQUEST_ROWS(l0, l1, l2, l3, 0, 0)
        if constexpr (!fresh) { // what the chunk before left for this side of the barrier
            UA_HIGH_HALF(a_odd, false)
        }
        UA_REQUEST_ROWS(h0, h1, h2, h3, 4, 0)
        // the dataset rows first — they come from HBM, the queries from L2 — and nothing behind the third step: the wave waits for
        // its fills in the middle of the fourth
        UA_STEP(a_even, a_odd, 0, fill_rows(target, 0); fill_rows(target, 2))
        UA_STEP(a_odd, a_even, 1, fill_queries(target, 0))
        UA_STEP(a_even, a_odd, 2, fill_queries(target, 2); fill_tile_head_and_advance())
        UA_LOW_HALF(a_odd, 3)
        UA_AWAIT(0, a_odd, h0, h1, h2, h3) // every read this wave makes of the buffer has landed
        if (last_of_tile) {
            UA_HIGH_HALF(a_odd, false)
        }
#undef UA_STEP
#undef UA_LOW_HALF
#undef UA_HIGH_HALF
        __builtin_amdgcn_s_setprio(0);
    };
#undef UA_REQUEST_QUERY
#undef UA_REQUEST_ROWS
#undef UA_AWAIT

    // ---- LDS words behind the compiler's back. A read or write it can see makes it wait for every LDS-DMA fill in flight first
    //      (`s_waitcnt vmcnt(0)`: for all it knows the fill writes the very cell) — a wave that finds a candidate for a list would
    //      sit out the landing of the next tile's first chunk, and the seven others meet it at the barrier. The cells these touch
    //      (limits, list sizes, lists, what the fused fold put in) are never a fill's target.
    auto lds_address = [&](const void* cell) -> std::uint32_t { return lds_base + (std::uint32_t)((const std::uint8_t*)cell - lds); };
    auto lds_read = [&](const void* cell) -> std::uint32_t {
        std::uint32_t value;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(value) : "v"(lds_address(cell)) : "memory");
        return value;
    };
    auto lds_write = [&](void* cell, std::uint32_t value) { asm volatile("ds_write_b32 %0, %1" : : "v"(lds_address(cell)), "v"(value) : "memory"); };

    /// Offers (distance `d`, slot `s`) to query i's list — the whole wave calls it together, uniformly: the ordered insert that the
    /// rare path of either fold ends in.
    auto offer = [&](std::uint32_t i, std::uint32_t s, float d) {
        if (d > __builtin_bit_cast(float, lds_read(limit + i))) // farther than this partition's k-th best (the test in the registers is a tile old): no list access
            return;
        const std::uint32_t size = lds_read(top_n + i);
        const bool mine = lane < size;
        float my_d = 0.f;
        std::uint32_t my_s = 0u;
        // the list: in LDS when it fits there, else where it ends up — this partition's cells of the output arrays
        // (distances as they are, the slot in the low word of the key cell until the end)
        const std::uint64_t cells = ((std::uint64_t)partition * query_count + first_query + i) * wanted;
        float* entries_d = lds_lists_ak ? lists_d + i * wanted : out_distances + cells;
        std::uint32_t* entries_s = lds_lists_ak ? lists_s + i * wanted : reinterpret_cast<std::uint32_t*>(out_keys + cells);
        constexpr std::uint32_t slot_pitch = lds_lists_ak ? 1u : 2u; // entry e of a key cell list: word 2e
        // the whole list in registers, a lane per entry (read once: what follows never re-reads what it wrote)
        if constexpr (lds_lists_ak) {
            const std::uint32_t entry = mine ? lane : 0u;
            const std::uint32_t read_d = lds_read(entries_d + entry), read_s = lds_read(entries_s + entry);
            my_d = mine ? __builtin_bit_cast(float, read_d) : 0.f, my_s = mine ? read_s : 0u;
        } else if (mine) {
            // past the compute unit's vector cache: this wave's own stores of the insert before went THROUGH that cache to L2, and a
            // line it still holds from the read before that is stale (the fills that stream through the cache used to push such
            // lines out before the next insert came; with the rendezvous moved they no longer do)
            my_d = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<std::uint32_t*>(entries_d) + lane, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
            my_s = __hip_atomic_load(entries_s + slot_pitch * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (size == wanted) { // full: the newcomer has to beat the last entry
            const float last_d = __shfl(my_d, (int)(size - 1), 64);
            const std::uint32_t last_s = (std::uint32_t)__shfl((int)my_s, (int)(size - 1), 64);
            if (!goes_before(d, s, last_d, last_s))
                return;
        }
        // position = entries that go before the newcomer; the ones at and after it move one cell down
        const std::uint32_t position = (std::uint32_t)__popcll(__ballot(mine && goes_before(my_d, my_s, d, s)));
        const std::uint32_t grown = size < wanted ? size + 1 : size;
        // the new k-th best: the newcomer if it went last, else what was second to last
        const std::uint32_t new_last = grown - 1;
        const float shifted = __shfl(my_d, (int)(new_last ? new_last - 1 : 0), 64);
        const float kth = position == new_last ? d : shifted;
        if constexpr (lds_lists_ak) {
            if (mine && lane >= position && lane + 1 < grown)
                lds_write(entries_d + lane + 1, __builtin_bit_cast(std::uint32_t, my_d)), lds_write(entries_s + lane + 1, my_s);
            if (lane == position)
                lds_write(entries_d + position, __builtin_bit_cast(std::uint32_t, d)), lds_write(entries_s + position, s);
            if (lane == 0) {
                lds_write(top_n + i, grown);
                if (grown == wanted) { // k rows at most this far exist: no partition needs anything farther for this query
                    lds_write(limit + i, __builtin_bit_cast(std::uint32_t, kth));
                    atomicMin(shared_bounds + first_query + i, ordered_bits(kth));
                }
            }
            // this wave's next insert into the list reads these cells: the LDS pipe of a compute unit serves one wave's accesses in
            // the order it issued them, and nobody else touches this wave's queries
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            if (mine && lane >= position && lane + 1 < grown)
                entries_d[lane + 1] = my_d, entries_s[slot_pitch * (lane + 1)] = my_s;
            if (lane == position)
                entries_d[position] = d, entries_s[slot_pitch * position] = s;
            if (lane == 0) {
                top_n[i] = grown;
                if (grown == wanted) {
                    limit[i] = kth;
                    atomicMin(shared_bounds + first_query + i, ordered_bits(kth));
                }
            }
            // this wave's next insert into the list reads these cells
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    };

    /// The tile's 32 × 256 sums of this wave against its queries' lists; the accumulators are cleared for the next tile.
    auto fold_tile = [&]() {
        const std::uint64_t tile_row = first_row + (std::uint64_t)work_tile * wide_rows_k;
        std::uint32_t lane_here = lane; // opaque: the addresses below are computed here, not held (and spilled) across the chunk loop
        asm volatile("" : "+v"(lane_here));
        // the tile's Σb² and the shared bounds by hand-waited reads, like the fragments: a compiler-visible LDS read here would
        // drain the fills in flight. Registers 4j … 4j + 3 ↔ queries 8j + 4·(lane >> 5) + 0 … 3
        {
            const std::uint32_t cells = lds_base + (std::uint32_t)((std::uint8_t*)(others + (work_tile & 1u) * 512 + wave * 64 + 4 * (lane_here >> 5)) - lds);
            u32x4_t s0, s1, s2, s3;
            asm volatile("ds_read_b128 %0, %4\n\t"
                         "ds_read_b128 %1, %4 offset:32\n\t"
                         "ds_read_b128 %2, %4 offset:64\n\t"
                         "ds_read_b128 %3, %4 offset:96\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                         : "v"(cells));
            const std::uint32_t shared[16] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3],
                                              s2[0], s2[1], s2[2], s2[3], s3[0], s3[1], s3[2], s3[3]};
            float own[16];
            {
                const std::uint32_t own_cells = lds_base + (std::uint32_t)((std::uint8_t*)(limit + wave * 32 + 4 * (lane_here >> 5)) - lds);
                u32x4_t o0, o1, o2, o3;
                asm volatile("ds_read_b128 %0, %4\n\t"
                             "ds_read_b128 %1, %4 offset:32\n\t"
                             "ds_read_b128 %2, %4 offset:64\n\t"
                             "ds_read_b128 %3, %4 offset:96\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                             : "v"(own_cells));
                const std::uint32_t bits[16] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3],
                                                o2[0], o2[1], o2[2], o2[3], o3[0], o3[1], o3[2], o3[3]};
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    own[r] = __builtin_bit_cast(float, bits[r]);
            }
            float roots[16] = {};
            if constexpr (metric_ak == metric_cos_k) {
                const std::uint32_t root_cells = lds_base + (std::uint32_t)((std::uint8_t*)(roots_q + wave * 32 + 4 * (lane_here >> 5)) - lds);
                u32x4_t q0, q1, q2, q3;
                asm volatile("ds_read_b128 %0, %4\n\t"
                             "ds_read_b128 %1, %4 offset:32\n\t"
                             "ds_read_b128 %2, %4 offset:64\n\t"
                             "ds_read_b128 %3, %4 offset:96\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                             : "v"(root_cells));
                const std::uint32_t bits[16] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3],
                                                q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    roots[r] = __builtin_bit_cast(float, bits[r]);
            }
            if (!(knock & 8u) || work_tile == 0)
                refresh_thresholds(shared, own, roots);
        }
        const std::uint32_t tile_norms = lds_base + (std::uint32_t)((std::uint8_t*)(norms_r + (work_tile & 1u) * wide_rows_k + (lane_here & 31)) - lds);
        std::uint32_t tile_b2[wide_blocks_k];
        asm volatile("ds_read_b32 %0, %8\n\t"
                     "ds_read_b32 %1, %8 offset:128\n\t"
                     "ds_read_b32 %2, %8 offset:256\n\t"
                     "ds_read_b32 %3, %8 offset:384\n\t"
                     "ds_read_b32 %4, %8 offset:512\n\t"
                     "ds_read_b32 %5, %8 offset:640\n\t"
                     "ds_read_b32 %6, %8 offset:768\n\t"
                     "ds_read_b32 %7, %8 offset:896\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(tile_b2[0]), "=&v"(tile_b2[1]), "=&v"(tile_b2[2]), "=&v"(tile_b2[3]), "=&v"(tile_b2[4]), "=&v"(tile_b2[5]),
                       "=&v"(tile_b2[6]), "=&v"(tile_b2[7])
                     : "v"(tile_norms));
        const bool check_members = ix.has_tombstones || allow_bits != nullptr;
#pragma unroll
        for (int u = 0; u < wide_blocks_k; ++u) {
            if (knock & 16u)
                continue;
            const std::uint64_t my_row = tile_row + u * 32 + (lane & 31);
            bool live = my_row < last_row;
            if (check_members) {
                if (ix.has_tombstones && live)
                    live = ix.keys[my_row] != free_key_k;
                if (allow_bits && live) // the caller's predicate, one bit per slot (index.hpp:4260-4263)
                    live = ((allow_bits[my_row >> 5] >> (my_row & 31)) & 1u) != 0;
            }
            const std::uint32_t b2 = tile_b2[u];
            const float row_scale = metric_ak == metric_cos_k ? bound_scale<integers>(b2) : 1.f;
            // ---- the fast test, conservative: never false for a sum whose exact distance is ≤ the bound (NaN — a zero norm —
            //      compares "may"). One multiply and one compare per sum; the compares' lane masks are ORed on the scalar unit.
            auto may_enter = [&](int r) -> bool {
                if constexpr (metric_ak == metric_l2sq_k)
                    return closing_distance<metric_ak, scalar_ak>(acc[u][r], query_norm[r], b2) <= bound[r];
                else if constexpr (metric_ak == metric_ip_k)
                    return 1.f - (float)acc[u][r] <= bound[r];
                else // cos: 1 − Σab/(√Σa²·√Σb²) ≤ bound + 10⁻⁵; an integer Σab = 0 closes to 0 whatever the norms
                    return !((float)acc[u][r] * row_scale < threshold[r]) || (integers && acc[u][r] == 0);
            };
            std::uint64_t any = 0;
            if constexpr (metric_ak == metric_cos_k && !integers) {
                // 16 fused multiply-adds Σab·rsq(Σb²) − threshold, their maximum by eight three-way maxima, ONE compare: a NaN — a
                // row of zero norm — is "may" (`!(m < 0)`); queries of zero norm carry −inf as their threshold
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    t[r] = __builtin_fmaf(acc[u][r], row_scale, -threshold[r]);
                const float m = __builtin_fmaxf(
                    __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t[0], t[1]), __builtin_fmaxf(t[2], t[3])),
                                    __builtin_fmaxf(__builtin_fmaxf(t[4], t[5]), __builtin_fmaxf(t[6], t[7]))),
                    __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t[8], t[9]), __builtin_fmaxf(t[10], t[11])),
                                    __builtin_fmaxf(__builtin_fmaxf(t[12], t[13]), __builtin_fmaxf(t[14], t[15]))));
                any = __ballot(!(m < 0.f) && live);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    any |= __ballot(may_enter(r));
                any &= __ballot(live);
            }
            if (any == 0)
                continue;
            // ---- the rare path: which sums exactly, then one ordered insert at a time
            std::uint32_t may = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                may |= (may_enter(r) ? 1u : 0u) << r;
            may = live ? may & exists : 0u;
            for (int r = 0; r < 16; ++r) { // kept rolled
                std::uint64_t pending = __ballot((may >> r) & 1u);
                while (pending) {
                    const std::uint32_t source = (std::uint32_t)__ffsll((long long)pending) - 1;
                    pending &= pending - 1;
                    const std::uint32_t i = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (source >> 5);
                    const std::uint32_t s = (std::uint32_t)(tile_row + u * 32 + (source & 31));
                    sum_t sum = 0;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) // the register index is a loop variable: select, then broadcast
                        sum = rr == r ? acc[u][rr] : sum;
                    sum = __shfl(sum, (int)source, 64);
                    const std::uint32_t source_b2 = (std::uint32_t)__shfl((int)b2, (int)source, 64);
                    const float d = closing_distance<metric_ak, scalar_ak>(sum, lds_read(norms_q + i), source_b2);
                    offer(i, s, d);
                }
            }
        }
    };

    /**
     *  The fold of f16 cos / ip with the thresholds INSIDE the sums. The test "may this sum enter its query's list" is
     *  Σab ≥ threshold(query) · √Σb²(row) (ip: · 1): one more step of the summation with −threshold in the query operand and √Σb² in
     *  the row operand — eight MFMAs per tile and wave next to the 384 of the product — leaves Σab − threshold·√Σb² in the
     *  accumulators, and a block of 32 × 32 sums is tested by the maximum of a lane's sixteen (eight three-way maxima) and ONE
     *  compare; the general fold spends 16 multiply-adds per block on the same question, and a threshold refresh per REGISTER where
     *  this one has it per LANE (a lane ↔ a query in the operand layout). The vector ALU was what the epilogue cost
     *  (profiles/r05_exact: 33 of 174 ms, 28 of them the per-block tests). Both factors are rounded to f16; the threshold is
     *  loosened by 2⁻⁹ of itself first, which covers the two roundings (2⁻¹¹ each) whatever the signs: the test stays conservative.
     *  A sum that passes gets what was put in taken out again (Σab = sum + threshold₁₆·norm₁₆, the product exact in f32) and goes
     *  through the closing arithmetic and the list as in the general fold. Rows whose norm f16 cannot carry (zero, < 2⁻¹², > 60 000)
     *  get a zero in the operand and always pass. Returns false — nothing touched — while a query of the wave has no finite
     *  threshold yet (its list and everybody else's still filling: the first tile of a launch; a query of zero norm): the general
     *  fold takes the tile.
     *  NOT BIT-REPRODUCIBLE ACROSS FOLDS: the accumulator was rounded once more with the threshold's term in it, so the Σab recovered
     *  here can differ in its last bits from the one the general fold (or the 64-row tile) closes — which fold a tile takes depends
     *  on where the launch's tiles begin and on zero-norm queries in the wave. Distances of the f16 cos / ip wide tile therefore
     *  agree with the bit-exact wave kernel to float tolerance (tests/test_gpu_exact.py: 2e-6 relative; keys > 0.97 equal, the rest
     *  are such near-ties), never bit for bit; i8 takes integer sums through the general fold and stays bit-identical. A sum that is
     *  NaN fails every comparison: its row is dropped by the maximum — the general fold would offer it and the list would refuse
     *  it (NaN orders before nothing), the same result.
     */
    auto fold_tile_fused = [&]() -> bool {
        const std::uint64_t tile_row = first_row + (std::uint64_t)work_tile * wide_rows_k;
        // (the lane number through an opaque move: what is computed from it here is computed HERE, not hoisted out of the chunk loop
        // into registers that the loop then has to spill — a spill's reload waits for every fill in flight)
        std::uint32_t lane_here = lane;
        asm volatile("" : "+v"(lane_here));
        const std::uint32_t my_query = wave * 32 + (lane_here & 31);
        std::uint32_t shared_bits, own_bits, root_bits;
        {
            const std::uint32_t shared_cell = lds_base + (std::uint32_t)((std::uint8_t*)(others + (work_tile & 1u) * 512 + wave * 64 + (lane_here & 31)) - lds);
            const std::uint32_t own_cell = lds_base + (std::uint32_t)((std::uint8_t*)(limit + my_query) - lds);
            const std::uint32_t root_cell = lds_base + (std::uint32_t)((std::uint8_t*)(roots_q + my_query) - lds);
            asm volatile("ds_read_b32 %0, %3\n\t"
                         "ds_read_b32 %1, %4\n\t"
                         "ds_read_b32 %2, %5\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(shared_bits), "=&v"(own_bits), "=&v"(root_bits)
                         : "v"(shared_cell), "v"(own_cell), "v"(root_cell));
        }
        const float other = from_ordered_bits(shared_bits), own = __builtin_bit_cast(float, own_bits);
        const float my_bound = other < own ? other : own; // a NaN ("nothing published") keeps our own
        float my_threshold = metric_ak == metric_cos_k ? (1.f - (my_bound + 1e-5f)) * __builtin_bit_cast(float, root_bits) : 1.f - my_bound;
        my_threshold -= __builtin_fabsf(my_threshold) * 0x1p-9f;                                  // room for the two roundings
        my_threshold = __builtin_fabsf(my_threshold) < 0x1p-13f ? -0x1p-13f : my_threshold;       // clear of f16's subnormals
        my_threshold = first_query + my_query < query_count ? my_threshold : 60000.f;            // a query outside the batch: never
        if (__ballot(!(__builtin_fabsf(my_threshold) <= 60000.f))) // infinite, a NaN, or beyond f16
            return false;
        const _Float16 threshold16 = (_Float16)my_threshold;
        {
            const float put_in = (float)threshold16;
            const std::uint32_t cell = lds_base + (std::uint32_t)((std::uint8_t*)(folded + my_query) - lds);
            asm volatile("ds_write_b32 %0, %1" : : "v"(cell), "v"(put_in) : "memory");
        }
        const std::uint32_t low_half = lane_here < 32 ? 0xFFFFu : 0u; // element 0 of the summation step lives in lanes 0 … 31
        const u32x4_t query_operand = {((std::uint32_t)__builtin_bit_cast(std::uint16_t, threshold16) ^ 0x8000u) & low_half, 0u, 0u, 0u};
        std::uint32_t tile_b2[wide_blocks_k] = {};
        if constexpr (metric_ak == metric_cos_k) {
            const std::uint32_t tile_norms = lds_base + (std::uint32_t)((std::uint8_t*)(norms_r + (work_tile & 1u) * wide_rows_k + (lane_here & 31)) - lds);
            asm volatile("ds_read_b32 %0, %8\n\t"
                         "ds_read_b32 %1, %8 offset:128\n\t"
                         "ds_read_b32 %2, %8 offset:256\n\t"
                         "ds_read_b32 %3, %8 offset:384\n\t"
                         "ds_read_b32 %4, %8 offset:512\n\t"
                         "ds_read_b32 %5, %8 offset:640\n\t"
                         "ds_read_b32 %6, %8 offset:768\n\t"
                         "ds_read_b32 %7, %8 offset:896\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(tile_b2[0]), "=&v"(tile_b2[1]), "=&v"(tile_b2[2]), "=&v"(tile_b2[3]), "=&v"(tile_b2[4]), "=&v"(tile_b2[5]),
                           "=&v"(tile_b2[6]), "=&v"(tile_b2[7])
                         : "v"(tile_norms));
        }
        float put_in_row[wide_blocks_k]; // the f16 norm that went into the row operand, as a float (0: a row that always passes)
        std::uint32_t forced = 0;        // bit u: this lane's row of block u always passes
#pragma unroll
        for (int u = 0; u < wide_blocks_k; ++u) {
            float norm = 1.f;
            if constexpr (metric_ak == metric_cos_k) {
                norm = __builtin_amdgcn_sqrtf(__builtin_bit_cast(float, tile_b2[u])); // the bare instruction (1 ulp): the margin is 2⁻⁹
                const bool carried = norm >= 0x1p-12f && norm <= 60000.f;
                forced |= (carried ? 0u : 1u) << u;
                norm = carried ? norm : 0.f;
            }
            const _Float16 norm16 = (_Float16)norm;
            put_in_row[u] = (float)norm16;
            const u32x4_t row_operand = {(std::uint32_t)__builtin_bit_cast(std::uint16_t, norm16) & low_half, 0u, 0u, 0u};
            acc[u] = multiply<scalar_ak>(__builtin_bit_cast(uint4, query_operand), __builtin_bit_cast(uint4, row_operand), acc[u]);
        }
        const bool check_members = ix.has_tombstones || allow_bits != nullptr;
        // rows of the tile that exist: 256 but for a partition's last tile (one 32-bit compare per block instead of a 64-bit one)
        const std::uint32_t rows_here = last_row - tile_row < wide_rows_k ? (std::uint32_t)(last_row - tile_row) : (std::uint32_t)wide_rows_k;
#pragma unroll
        for (int u = 0; u < wide_blocks_k; ++u) {
            const std::uint64_t my_row = tile_row + u * 32 + (lane_here & 31);
            bool live = u * 32 + (lane_here & 31) < rows_here;
            if (check_members) {
                if (ix.has_tombstones && live)
                    live = ix.keys[my_row] != free_key_k;
                if (allow_bits && live) // the caller's predicate, one bit per slot (index.hpp:4260-4263)
                    live = ((allow_bits[my_row >> 5] >> (my_row & 31)) & 1u) != 0;
            }
            const bool always = (forced >> u) & 1u;
            const float m = __builtin_fmaxf(
                __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(acc[u][0], acc[u][1]), __builtin_fmaxf(acc[u][2], acc[u][3])),
                                __builtin_fmaxf(__builtin_fmaxf(acc[u][4], acc[u][5]), __builtin_fmaxf(acc[u][6], acc[u][7]))),
                __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(acc[u][8], acc[u][9]), __builtin_fmaxf(acc[u][10], acc[u][11])),
                                __builtin_fmaxf(__builtin_fmaxf(acc[u][12], acc[u][13]), __builtin_fmaxf(acc[u][14], acc[u][15]))));
            if (__ballot((!(m < 0.f) || always) && live) == 0 || (knock & 64u))
                continue;
            // ---- the rare path, lane-parallel up to the list: every lane with a sum that passed takes what was put in out again,
            //      closes the distance and holds it against its query's limit — all of them at once, a round per passed sum of a
            //      lane (seldom a second one) — and only what is still in the running goes to the lists, one insert at a time
            std::uint32_t may = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                may |= ((!(acc[u][r] < 0.f) || always) ? 1u : 0u) << r;
            may = live ? may & exists : 0u;
            while (__ballot(may != 0u)) {
                const bool candidate = may != 0u;
                const std::uint32_t r = candidate ? (std::uint32_t)__builtin_ctz(may) : 0u;
                may &= may - 1u;
                float sum = acc[u][0];
#pragma unroll
                for (int rr = 1; rr < 16; ++rr)
                    sum = r == (std::uint32_t)rr ? acc[u][rr] : sum;
                const std::uint32_t i = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane_here >> 5);
                std::uint32_t put_in_bits, norm_bits, limit_bits;
                asm volatile("ds_read_b32 %0, %3\n\t"
                             "ds_read_b32 %1, %4\n\t"
                             "ds_read_b32 %2, %5\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(put_in_bits), "=&v"(norm_bits), "=&v"(limit_bits)
                             : "v"(lds_address(folded + i)), "v"(lds_address(norms_q + i)), "v"(lds_address(limit + i))
                             : "memory");
                sum = __builtin_fmaf(__builtin_bit_cast(float, put_in_bits), put_in_row[u], sum); // Σab again
                const float d = closing_distance<metric_ak, scalar_ak>(sum, norm_bits, tile_b2[u]);
                std::uint64_t pending = __ballot(candidate && !(d > __builtin_bit_cast(float, limit_bits)));
                while (pending) {
                    const std::uint32_t source = (std::uint32_t)__ffsll((long long)pending) - 1;
                    pending &= pending - 1;
                    offer((std::uint32_t)__shfl((int)i, (int)source, 64), (std::uint32_t)(tile_row + u * 32 + (source & 31)),
                          __shfl(d, (int)source, 64));
                }
            }
        }
        return true;
    };

    // ---- the pipeline: chunk c is multiplied out of buffer c & 1 while the DMA fills the other buffer with chunk c + 1. A wave
    //      waits for ITS fills (the VM counter), then meets the others at a bare barrier: what a buffer holds is read one iteration
    //      after the wait + barrier that completed it, and refilled one barrier after its last read.
    if (total) {
        fill_queries(0, 0);
        fill_queries(0, 2);
        fill_rows(0, 0);
        fill_rows(0, 2);
        fill_tile_head_and_advance();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#ifdef USEARCH_AMD_EXACT_PHASES
    std::uint64_t phase_ticks[4] = {0, 0, 0, 0}, phase_mark = __builtin_amdgcn_s_memtime();
#endif
    for (std::uint32_t c = 0; c < total; ++c) {
        const bool more = c + 1 < total, filling = more && !(knock & 2u);
        const bool last_of_tile = work_chunk + 1 == chunks;
        if (work_chunk == 0)
            multiply_chunk(std::true_type{}, c & 1u, filling, (c + 1) & 1u, last_of_tile);
        else
            multiply_chunk(std::false_type{}, c & 1u, filling, (c + 1) & 1u, last_of_tile);
        UA_PHASE_TICK(0)
        if (++work_chunk == chunks) {
            if (!(knock & 1u)) {
                bool folded_already = false;
                if constexpr (scalar_ak == scalar_f16_k && (metric_ak == metric_cos_k || metric_ak == metric_ip_k))
                    folded_already = !(knock & 32u) && fold_tile_fused();
#ifdef USEARCH_AMD_EXACT_PHASES
                if (thread == 0)
                    atomicAdd(exact_phase_ticks + (folded_already ? 5 : 6), 1ull); // tiles the fused / the general fold took
#endif
                if (!folded_already)
                    fold_tile();
            }
            work_chunk = 0, ++work_tile;
        }
        UA_PHASE_TICK(1)
        if (!(knock & 4u)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of the next chunk is in LDS
            UA_PHASE_TICK(2)
            __builtin_amdgcn_s_barrier();
        }
        UA_PHASE_TICK(3)
    }
#ifdef USEARCH_AMD_EXACT_PHASES
    if (thread == 0) {
        for (int phase = 0; phase < 4; ++phase)
            atomicAdd(exact_phase_ticks + phase, (unsigned long long)phase_ticks[phase]);
        atomicAdd(exact_phase_ticks + 4, (unsigned long long)total);
    }
#endif

    // ---- this partition's lists are already where the merge reads them, [partition][query][wanted] like the wave-per-query
    //      kernel's: slots become keys, unused cells the padding
    __syncthreads();
    for (std::uint32_t cell = thread; cell < wide_queries_k * wanted; cell += wide_threads_k) {
        const std::uint32_t i = cell / wanted, position = cell % wanted;
        const std::uint32_t q = first_query + i;
        if (q >= query_count)
            continue;
        const std::uint64_t out = ((std::uint64_t)partition * query_count + q) * wanted + position;
        if (position < top_n[i]) {
            const std::uint32_t slot = lds_lists_ak ? lists_s[i * wanted + position] : (std::uint32_t)out_keys[out];
            out_keys[out] = map_keys ? ix.keys[slot] : (std::uint64_t)slot;
            if constexpr (lds_lists_ak)
                out_distances[out] = lists_d[i * wanted + position];
        } else {
            out_keys[out] = 0;
            reinterpret_cast<std::uint32_t*>(out_distances)[out] = signaling_nan_bits_k;
        }
    }
    for (std::uint32_t i = thread; i < wide_queries_k; i += wide_threads_k)
        if (first_query + i < query_count)
            out_counts[(std::uint64_t)partition * query_count + first_query + i] = top_n[i];
}

/// Queries as the wide kernel stages them: whole 128-byte chunks of every row, whole tiles of 256 rows, zeros where the batch ends.
__global__ __launch_bounds__(256) void pad_queries_kernel(const std::uint8_t* queries, std::uint64_t stride, std::uint32_t count,
                                                          std::uint32_t bytes, std::uint8_t* padded, std::uint64_t padded_stride,
                                                          std::uint64_t padded_rows) {
    const std::uint64_t total = padded_rows * padded_stride;
    for (std::uint64_t cell = (std::uint64_t)blockIdx.x * 256 + threadIdx.x; cell < total; cell += (std::uint64_t)gridDim.x * 256) {
        const std::uint64_t row = cell / padded_stride, byte = cell % padded_stride;
        padded[cell] = row < count && byte < bytes ? queries[row * stride + byte] : (std::uint8_t)0;
    }
}

template <int metric_ak, int scalar_ak>
hipError_t launch_wide(const snapshot_view_t& view, const std::uint8_t* queries, std::uint64_t query_stride,
                       std::uint8_t* padded, std::uint32_t query_count, std::uint32_t wanted, std::uint32_t tiles_per_xcd, std::uint32_t workgroups,
                       std::uint64_t rows_per_partition, const std::uint32_t* row_norms, const std::uint32_t* query_norms,
                       bool map_keys, const std::uint32_t* allow_bits, std::uint32_t* shared_bounds, float* out_distances,
                       std::uint64_t* out_keys, std::uint64_t* out_counts, hipStream_t stream) {
    const bool lds_lists = wanted <= (std::uint32_t)wide_lds_lists_k && !env_size("USEARCH_AMD_EXACT_GLOBAL_LISTS", 0);
    auto kernel = lds_lists ? exact_wide_kernel<metric_ak, scalar_ak, true> : exact_wide_kernel<metric_ak, scalar_ak, false>;
    const std::uint32_t lds_bytes = wide_lds_bytes(lds_lists, wanted);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds_bytes);
    if (e != hipSuccess)
        return e;
    const std::uint32_t query_tiles = (query_count + wide_queries_k - 1) / wide_queries_k;
    const std::uint64_t padded_stride = wide_padded_stride(view.bytes_per_vector);
    const std::uint64_t padded_rows = (std::uint64_t)query_tiles * wide_queries_k;
    hipLaunchKernelGGL(pad_queries_kernel, dim3((unsigned)std::min<std::uint64_t>((padded_rows * padded_stride + 255) / 256, 1u << 20)),
                       dim3(256), 0, stream, queries, query_stride, query_count, (std::uint32_t)view.bytes_per_vector, padded,
                       padded_stride, padded_rows);
    hipLaunchKernelGGL(kernel, dim3(workgroups), dim3(wide_threads_k), lds_bytes, stream, view,
                       (const std::uint8_t*)padded, padded_stride, query_count, wanted, rows_per_partition, query_tiles, tiles_per_xcd, row_norms,
                       query_norms, map_keys ? 1u : 0u, allow_bits, shared_bounds, out_distances, out_keys, out_counts,
#ifdef USEARCH_AMD_EXPERIMENT_EXACT_KNOCKOUT
                       (std::uint32_t)env_size("USEARCH_AMD_EXACT_KNOCKOUT", 0));
#else
                       0u);
#endif
    return hipGetLastError();
}

constexpr std::uint32_t tiled_lds_bytes() {
    constexpr std::uint32_t staging = (tile_queries_k + tile_rows_k) * pitch_k;
    constexpr std::uint32_t tile = tile_queries_k * (tile_rows_k + 1) * 4;
    return (staging > tile ? staging : tile) + tile_queries_k * max_wanted_k * 8 + tile_queries_k * 4 + tile_queries_k * 4 +
           tile_rows_k * 4 + tile_rows_k * 4;
}

template <int metric_ak, int scalar_ak>
hipError_t launch_tiled(const snapshot_view_t& view, const std::uint8_t* queries, std::uint64_t query_stride,
                        std::uint32_t query_count, std::uint32_t wanted, std::uint32_t partitions,
                        std::uint64_t rows_per_partition, const std::uint32_t* row_norms, const std::uint32_t* query_norms,
                        bool map_keys, const std::uint32_t* allow_bits, float* out_distances, std::uint64_t* out_keys,
                        std::uint64_t* out_counts, hipStream_t stream) {
    auto kernel = exact_tiled_kernel<metric_ak, scalar_ak>;
    const std::uint32_t lds = tiled_lds_bytes();
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    const dim3 grid((query_count + tile_queries_k - 1) / tile_queries_k, partitions);
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, view, queries, query_stride, query_count, wanted,
                       rows_per_partition, row_norms, query_norms, map_keys ? 1u : 0u, allow_bits, out_distances, out_keys,
                       out_counts);
    return hipGetLastError();
}

template <int scalar_ak>
hipError_t launch_norms(const std::uint8_t* rows, std::uint64_t count, std::uint64_t stride, std::uint32_t bytes,
                        std::uint32_t* out, hipStream_t stream) {
    if (!count)
        return hipSuccess;
    hipLaunchKernelGGL(row_norms_kernel<scalar_ak>, dim3((unsigned)std::min<std::uint64_t>((count + 3) / 4, 1u << 20)), dim3(256), 0,
                       stream, rows, count, stride, bytes, out);
    return hipGetLastError();
}

} // namespace

bool exact_tiled_available(metric_kind_t metric, scalar_kind_t scalar, std::size_t wanted) {
    if (!wanted || wanted > (std::size_t)max_wanted_k)
        return false;
    if (scalar == scalar_f16_k || scalar == scalar_bf16_k)
        return metric == metric_cos_k || metric == metric_ip_k || metric == metric_l2sq_k;
    if (scalar == scalar_i8_k)
        return metric == metric_cos_k || metric == metric_ip_k || metric == metric_l2sq_k;
    return false;
}

const char* exact_search_tiled_device(metric_kind_t metric, scalar_kind_t scalar, const snapshot_view_t& view,
                                      const void* queries, std::size_t count, std::size_t stride_bytes, std::size_t wanted,
                                      bool map_keys, std::uint64_t* keys, float* distances, std::uint64_t* counts,
                                      hipStream_t stream, float* kernel_ms, const std::uint32_t* allow_bits) {
    if (kernel_ms)
        *kernel_ms = 0.f;
    if (!exact_tiled_available(metric, scalar, wanted))
        return "No tiled exact-search kernel for this metric / scalar kind / result count";
    if (!count)
        return nullptr;
    if (count >= none_slot_k || view.size >= none_slot_k)
        return "Batch is too large";
    if (!view.size)
        return "Nothing to scan";
    // The wide tile (256 queries per workgroup: a quarter of the dataset passes) takes batches that fill the chip with it; the
    // 64-query tile keeps the small batches and the long result lists. USEARCH_AMD_EXACT_TILE=64 / 256 forces either.
    const std::size_t forced_tile = env_size("USEARCH_AMD_EXACT_TILE", 0);
    // (a tile of 256 rows, stored or padded, must lie inside one buffer resource with 32-bit offsets: rows of < 8 MB)
    const bool wide = forced_tile != 64 && wanted <= (std::size_t)wide_wanted_k && view.size >= 8u * 4u * wide_rows_k &&
                      (forced_tile == 256 || count > 512) && (std::uint64_t)view.row_stride * wide_rows_k < (1ull << 31) &&
                      wide_padded_stride(view.bytes_per_vector) * wide_queries_k < (1ull << 31);
    std::uint64_t partitions, rows_per_partition, tiles_per_xcd = 0, workgroups = 0;
    if (wide) {
        // ONE round of the chip (32 compute units per XCD) with as few partitions as that takes (exact_wide_kernel)
        const std::uint64_t query_tiles = (count + wide_queries_k - 1) / wide_queries_k;
        const std::uint64_t most = std::max<std::uint64_t>(1, view.size / (4 * wide_rows_k)); // a partition is at least four tiles
        if (query_tiles >= 8) { // tiles dealt over the XCDs, every XCD sees every partition
            tiles_per_xcd = (query_tiles + 7) / 8;
            partitions = std::min<std::uint64_t>(std::max<std::uint64_t>(1, 32 / tiles_per_xcd), most);
            workgroups = 8 * tiles_per_xcd * partitions;
        } else { // every XCD works on all the tiles, partitions ≡ xcd (mod 8)
            tiles_per_xcd = query_tiles;
            partitions = std::max<std::uint64_t>(8, 32 / query_tiles * 8);
            while (partitions > 8 && partitions > most)
                partitions -= 8;
            workgroups = tiles_per_xcd * partitions;
        }
        rows_per_partition = (view.size + partitions - 1) / partitions;
        rows_per_partition = (rows_per_partition + wide_rows_k - 1) / wide_rows_k * wide_rows_k;
    } else {
        // enough (query tile, partition) workgroups to fill the chip several times over; few enough lists per query to fold
        const std::uint64_t query_tiles = (count + tile_queries_k - 1) / tile_queries_k;
        partitions = std::max<std::uint64_t>(1, (2048 + query_tiles - 1) / query_tiles);
        partitions = std::min<std::uint64_t>(partitions, std::max<std::uint64_t>(1, 8192 / wanted));
        partitions = std::min<std::uint64_t>(partitions, std::max<std::uint64_t>(1, view.size / (4 * tile_rows_k)));
        partitions = std::min<std::uint64_t>(partitions, 65535);
        rows_per_partition = (view.size + partitions - 1) / partitions;
        rows_per_partition = (rows_per_partition + tile_rows_k - 1) / tile_rows_k * tile_rows_k;
        partitions = (view.size + rows_per_partition - 1) / rows_per_partition;
    }

    struct scratch_t {
        std::vector<void*> pointers;
        hipEvent_t begin = nullptr, end = nullptr;
        ~scratch_t() {
            for (void* p : pointers)
                if (p)
                    (void)hipFree(p);
            if (begin)
                (void)hipEventDestroy(begin);
            if (end)
                (void)hipEventDestroy(end);
        }
        hipError_t allocate(void** out, std::size_t bytes) {
            *out = nullptr;
            const hipError_t e = hipMalloc(out, std::max<std::size_t>(bytes, 16));
            if (e == hipSuccess)
                pointers.push_back(*out);
            return e;
        }
    } scratch;
    std::uint32_t *row_norms = nullptr, *query_norms = nullptr;
    float* partial_distances = nullptr;
    std::uint64_t *partial_keys = nullptr, *partial_counts = nullptr;
    UA_HIP(scratch.allocate((void**)&row_norms, view.size * 4));
    UA_HIP(scratch.allocate((void**)&query_norms, count * 4));
    UA_HIP(scratch.allocate((void**)&partial_distances, partitions * count * wanted * 4));
    UA_HIP(scratch.allocate((void**)&partial_keys, partitions * count * wanted * 8));
    UA_HIP(scratch.allocate((void**)&partial_counts, partitions * count * 8));
    std::uint8_t* padded_queries = nullptr;
    std::uint32_t* shared_bounds = nullptr;
    if (wide) {
        UA_HIP(scratch.allocate((void**)&padded_queries, (count + wide_queries_k - 1) / wide_queries_k * wide_queries_k *
                                                              wide_padded_stride(view.bytes_per_vector)));
        // per query, the smallest k-th best any partition has reached so far, as ordered bits: all ones = nobody has k results yet
        UA_HIP(scratch.allocate((void**)&shared_bounds, count * 4));
        UA_HIP(hipMemsetAsync(shared_bounds, 0xFF, count * 4, stream));
    }
    if (kernel_ms) {
        UA_HIP(hipEventCreate(&scratch.begin));
        UA_HIP(hipEventCreate(&scratch.end));
        UA_HIP(hipEventRecord(scratch.begin, stream));
    }
    const std::uint8_t* query_bytes = static_cast<const std::uint8_t*>(queries);
    hipError_t e = hipSuccess;
#define UA_TILED(m, sc)                                                                                                 \
    if (e == hipSuccess && metric == m && scalar == sc) {                                                              \
        e = launch_norms<sc>(view.vectors, view.size, view.row_stride, view.bytes_per_vector, row_norms, stream);      \
        if (e == hipSuccess)                                                                                           \
            e = launch_norms<sc>(query_bytes, count, stride_bytes, view.bytes_per_vector, query_norms, stream);        \
        if (e == hipSuccess && wide)                                                                                   \
            e = launch_wide<m, sc>(view, query_bytes, stride_bytes, padded_queries, (std::uint32_t)count,              \
                                   (std::uint32_t)wanted, (std::uint32_t)tiles_per_xcd, (std::uint32_t)workgroups, rows_per_partition, \
                                   row_norms, query_norms,                                                              \
                                   map_keys, allow_bits, shared_bounds, partial_distances, partial_keys, partial_counts,  \
                                   stream);                                                                         \
        else if (e == hipSuccess)                                                                                      \
            e = launch_tiled<m, sc>(view, query_bytes, stride_bytes, (std::uint32_t)count, (std::uint32_t)wanted,      \
                                    (std::uint32_t)partitions, rows_per_partition, row_norms, query_norms, map_keys,   \
                                    allow_bits, partial_distances, partial_keys, partial_counts, stream);              \
    }
    UA_TILED(metric_cos_k, scalar_f16_k)
    UA_TILED(metric_ip_k, scalar_f16_k)
    UA_TILED(metric_cos_k, scalar_bf16_k)
    UA_TILED(metric_ip_k, scalar_bf16_k)
    UA_TILED(metric_l2sq_k, scalar_f16_k)
    UA_TILED(metric_l2sq_k, scalar_bf16_k)
    UA_TILED(metric_cos_k, scalar_i8_k)
    UA_TILED(metric_ip_k, scalar_i8_k)
    UA_TILED(metric_l2sq_k, scalar_i8_k)
#undef UA_TILED
    if (e != hipSuccess)
        return hip_message(e);
#ifdef USEARCH_AMD_EXACT_PHASES
    if (wide) {
        unsigned long long ticks[8] = {0};
        UA_HIP(hipStreamSynchronize(stream));
        UA_HIP(hipMemcpyFromSymbol(ticks, HIP_SYMBOL(exact_phase_ticks), sizeof(ticks)));
        const double chunks_done = (double)std::max<unsigned long long>(ticks[4], 1);
        std::fprintf(stderr, "[usearch_amd] exact phases, shader-clock ticks per chunk (wave 0 of every workgroup): multiply %.0f fold %.0f "
                             "own fills %.0f barrier %.0f; tiles folded with the thresholds inside the sums %llu, by the general fold %llu\n", ticks[0] / chunks_done, ticks[1] / chunks_done, ticks[2] / chunks_done, ticks[3] / chunks_done, ticks[5], ticks[6]);
        unsigned long long zeros[8] = {0};
        UA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(exact_phase_ticks), zeros, sizeof(zeros)));
    }
#endif
    if (kernel_ms)
        UA_HIP(hipEventRecord(scratch.end, stream));
    if (const char* error = merge_shards_device(partial_distances, partial_keys, partial_counts, partitions, count, wanted,
                                                distances, keys, counts, stream, false))
        return error;
    if (kernel_ms)
        UA_HIP(hipEventElapsedTime(kernel_ms, scratch.begin, scratch.end));
    return nullptr;
}

} // namespace usearch_amd
