/**
 *  usearch_amd/csrc/casts.hpp — host-side query casts into the index's storage scalar kind.
 *
 *  `index_dense_gt::search_` casts the query ONCE before the traversal (/root/reference/include/usearch/
 *  index_dense.hpp:2058-2064) with the `cast_gt` family of index_plugins.hpp:1105-1224; this file mirrors those rules
 *  (same rounding, same quirks) so that a batch handed over in any scalar kind searches exactly like the reference:
 *    same kind            → no cast (1115-1137)
 *    any → b1x8           → bit `128 >> (i & 7)` of byte i/8 set when x > 0; only dim/8 bytes are cleared first (1139-1158)
 *    b1x8 → any           → set bits become 1, others 0 (1160-1170)
 *    float kinds → i8     → L2-normalise in double, × 127, clamp to ±127, truncate toward zero (1172-1191)
 *    i8 → float kinds     → x / 127.f in the target type (1193-1201)
 *    float ↔ float kinds  → element-wise conversion through float/double, IEEE round-to-nearest-even (1105-1113)
 *  It runs once per query over `dimensions` scalars — O(Q·d) host work against O(Q·2000·d) device work.
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "common.hpp"

namespace usearch_amd {

inline float f16_bits_to_f32(std::uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
inline std::uint16_t f32_to_f16_bits(float f) { return __builtin_bit_cast(std::uint16_t, (_Float16)f); }
inline float bf16_bits_to_f32(std::uint16_t h) { return __builtin_bit_cast(float, (std::uint32_t)h << 16); }
/// bf16_bits_t(float) of index_plugins.hpp:430-470 truncates (keeps the upper 16 bits) — no rounding.
inline std::uint16_t f32_to_bf16_bits(float f) { return (std::uint16_t)(__builtin_bit_cast(std::uint32_t, f) >> 16); }

inline double load_scalar(scalar_kind_t kind, const std::uint8_t* p, std::size_t i) {
    switch (kind) {
    case scalar_f64_k: { double v; std::memcpy(&v, p + 8 * i, 8); return v; }
    case scalar_f32_k: { float v; std::memcpy(&v, p + 4 * i, 4); return v; }
    case scalar_f16_k: { std::uint16_t v; std::memcpy(&v, p + 2 * i, 2); return f16_bits_to_f32(v); }
    case scalar_bf16_k: { std::uint16_t v; std::memcpy(&v, p + 2 * i, 2); return bf16_bits_to_f32(v); }
    case scalar_i8_k: return (double)(std::int8_t)p[i];
    default: return 0;
    }
}

inline void store_scalar(scalar_kind_t kind, std::uint8_t* p, std::size_t i, float as_float, double as_double) {
    switch (kind) {
    case scalar_f64_k: std::memcpy(p + 8 * i, &as_double, 8); break;
    case scalar_f32_k: std::memcpy(p + 4 * i, &as_float, 4); break;
    case scalar_f16_k: { std::uint16_t v = f32_to_f16_bits(as_float); std::memcpy(p + 2 * i, &v, 2); break; }
    case scalar_bf16_k: { std::uint16_t v = f32_to_bf16_bits(as_float); std::memcpy(p + 2 * i, &v, 2); break; }
    default: break;
    }
}

/// Returns false when `from == to` (nothing written), true after writing `bytes_per_vector(to, dimensions)` bytes.
/// `output` must be zero-initialised for `to == b1x8` with dimensions not divisible by 8 to be deterministic.
inline bool cast_vector(scalar_kind_t from, scalar_kind_t to, const std::uint8_t* input, std::size_t dimensions,
                        std::uint8_t* output) {
    if (from == to)
        return false;
    if (to == scalar_b1x8_k) {
        std::memset(output, 0, dimensions / 8);
        for (std::size_t i = 0; i != dimensions; ++i)
            if (load_scalar(from, input, i) > 0)
                output[i / 8] |= (std::uint8_t)(128 >> (i & 7));
        return true;
    }
    if (from == scalar_b1x8_k) {
        for (std::size_t i = 0; i != dimensions; ++i) {
            const bool bit = (input[i / 8] & (128 >> (i & 7))) != 0;
            if (to == scalar_i8_k)
                output[i] = (std::uint8_t)bit;
            else
                store_scalar(to, output, i, (float)bit, (double)bit);
        }
        return true;
    }
    if (to == scalar_i8_k) {
        double magnitude = 0.0;
        for (std::size_t i = 0; i != dimensions; ++i) {
            const double x = load_scalar(from, input, i);
            magnitude += x * x;
        }
        magnitude = std::sqrt(magnitude);
        for (std::size_t i = 0; i != dimensions; ++i) {
            double v = load_scalar(from, input, i) * 127.0 / magnitude;
            v = v < -127.0 ? -127.0 : (v > 127.0 ? 127.0 : v);
            output[i] = (std::uint8_t)(std::int8_t)v;
        }
        return true;
    }
    if (from == scalar_i8_k) {
        for (std::size_t i = 0; i != dimensions; ++i) {
            const std::int8_t x = (std::int8_t)input[i];
            if (to == scalar_f64_k) {
                store_scalar(to, output, i, 0.f, (double)x / 127.f);
            } else if (to == scalar_f32_k) {
                store_scalar(to, output, i, (float)x / 127.f, 0.0);
            } else if (to == scalar_f16_k) { // f16_t(int) then a float division, rounded back to f16
                store_scalar(to, output, i, f16_bits_to_f32(f32_to_f16_bits((float)x)) / 127.f, 0.0);
            } else { // bf16
                store_scalar(to, output, i, bf16_bits_to_f32(f32_to_bf16_bits((float)x)) / 127.f, 0.0);
            }
        }
        return true;
    }
    for (std::size_t i = 0; i != dimensions; ++i) {
        const double wide = load_scalar(from, input, i);
        store_scalar(to, output, i, (float)wide, wide);
    }
    return true;
}

} // namespace usearch_amd
