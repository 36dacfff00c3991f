/**
 *  oracle/usearch_oracle.c — CPU restatement of the USearch HNSW search hot path in plain C11.
 *
 *  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see usearch_oracle.h). Parity status: PINNED against the compiled
 *  reference (`oracle/_ref`) and `tests/golden/` — see `tests/test_oracle_vs_reference.py`.
 *
 *  Citations are to files under /root/reference. Nothing is copied: each routine re-states the *algorithm* of the
 *  cited lines (container semantics, comparison strictness, tie placement) because those decide label parity.
 */
#include "usearch_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------------
 *  Unaligned little-endian loads (the reference uses misaligned_load everywhere: index.hpp:200-211)
 * ---------------------------------------------------------------------------------------------------------------- */
static uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* ------------------------------------------------------------------------------------------------------------------
 *  Scalars. f16 <-> f32 as `_Float16` casts do it (index_plugins.hpp:398-428): IEEE binary16, round-to-nearest-even.
 * ---------------------------------------------------------------------------------------------------------------- */
static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7FFFFFFFu;
    if (absx >= 0x7F800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | (absx > 0x7F800000u ? (0x200u | ((absx >> 13) & 0x3FFu)) : 0));
    if (absx >= 0x477FF000u) /* rounds to >= 65520 → inf */
        return (uint16_t)(sign | 0x7C00u);
    if (absx < 0x33000001u) /* < 2^-25 (or == 2^-25, ties to even → 0) */
        return (uint16_t)sign;
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t man = (absx & 0x7FFFFFu) | 0x800000u;
    if (e < -14) { /* subnormal result: shift so that the unit is 2^-24 */
        int shift = (-14 - e) + 13;
        uint32_t halfway = 1u << (shift - 1);
        uint32_t rest = man & ((1u << shift) - 1u);
        uint32_t q = man >> shift;
        if (rest > halfway || (rest == halfway && (q & 1u)))
            q++;
        return (uint16_t)(sign | q);
    }
    uint32_t q = ((uint32_t)(e + 15) << 10) | ((man >> 13) & 0x3FFu);
    uint32_t rest = man & 0x1FFFu;
    if (rest > 0x1000u || (rest == 0x1000u && (q & 1u)))
        q++; /* may carry into the exponent, which is the correct rounding */
    return (uint16_t)(sign | q);
}

size_t uo_bytes_per_vector(uint8_t scalar_kind, uint64_t dimensions) {
    /* index_plugins.hpp:1853-1855 */
    switch (scalar_kind) {
    case UO_SCALAR_B1: return (size_t)((dimensions + 7) / 8);
    case UO_SCALAR_I8: return (size_t)dimensions;
    case UO_SCALAR_F16:
    case UO_SCALAR_BF16: return (size_t)dimensions * 2;
    case UO_SCALAR_F32: return (size_t)dimensions * 4;
    case UO_SCALAR_F64: return (size_t)dimensions * 8;
    default: return 0;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 *  Image parsing: index_dense.hpp:995-1062 (matrix + 64-byte head) and index.hpp:3277-3317 (graph).
 * ---------------------------------------------------------------------------------------------------------------- */
int uo_open(uo_index_t* ix, const void* image, size_t length, const char** error) {
    static const char* e_short = "File is corrupted and lacks matrix dimensions";
    memset(ix, 0, sizeof(*ix));
    const uint8_t* p = (const uint8_t*)image;
    const uint8_t* end = p + length;
    if (length < 8) { *error = e_short; return 1; }
    ix->image = p;
    ix->image_length = length;
    ix->rows = ld32(p);
    ix->cols = ld32(p + 4);
    p += 8;
    /* `use_64_bit_dimensions` images announce the matrix with two u64 instead; sniffed the way the reference does it
     * (index_dense_metadata_from_buffer, index_dense.hpp:321-371): where does the "usearch" magic turn up? */
    if (length >= 16 && !(ix->rows * ix->cols + 8 + 64 <= length && memcmp(ix->image + ix->rows * ix->cols + 8, "usearch", 7) == 0)) {
        uint64_t rows64 = ld64(ix->image), cols64 = ld64(ix->image + 8);
        if (cols64 && rows64 <= length / cols64 && rows64 * cols64 + 16 + 64 <= length &&
            memcmp(ix->image + rows64 * cols64 + 16, "usearch", 7) == 0)
            ix->rows = rows64, ix->cols = cols64, p = ix->image + 16;
    }
    if ((uint64_t)(end - p) < ix->rows * ix->cols + 64) { *error = "File is corrupted and lacks a header"; return 1; }
    ix->vectors = p;
    p += ix->rows * ix->cols;
    /* head: magic[7] "usearch", u16 x3 version, metric u8, scalar u8, key kind u8, slot kind u8, u64 present,
     * u64 deleted, u64 dimensions, u8 multi (index_dense.hpp:42-79) */
    if (memcmp(p, "usearch", 7) != 0) { *error = "Magic header mismatch - the file isn't an index"; return 1; }
    ix->version[0] = ld16(p + 7);
    ix->version[1] = ld16(p + 9);
    ix->version[2] = ld16(p + 11);
    ix->metric_kind = p[13];
    ix->scalar_kind = p[14];
    ix->key_kind = p[15];
    ix->slot_kind = p[16];
    ix->count_present = ld64(p + 17);
    ix->count_deleted = ld64(p + 25);
    ix->dimensions = ld64(p + 33);
    ix->multi = p[41];
    if (ix->key_kind != 14) { *error = "Key type doesn't match, consider rebuilding"; return 1; }   /* u64_k */
    if (ix->slot_kind != 15) { *error = "Slot type doesn't match, consider rebuilding"; return 1; } /* u32_k */
    p += 64;
    if ((size_t)(end - p) < 40) { *error = "Failed to pull the header from the stream"; return 1; }
    ix->size = ld64(p);
    ix->connectivity = ld64(p + 8);
    ix->connectivity_base = ld64(p + 16);
    ix->max_level = ld64(p + 24);
    ix->entry_slot = ld64(p + 32);
    p += 40;
    if (ix->size != ix->rows) { *error = "Index size and the number of vectors doesn't match"; return 1; }
    if ((uint64_t)(end - p) < ix->size * 2) { *error = "Failed to pull nodes levels from the stream"; return 1; }
    ix->levels = p;
    p += ix->size * 2;
    ix->node_offsets = (uint64_t*)malloc((ix->size + 1) * sizeof(uint64_t));
    if (!ix->node_offsets) { *error = "Out of memory"; return 1; }
    uint64_t off = (uint64_t)(p - ix->image);
    /* node bytes = 10 + (4 + 4*M0) + level * (4 + 4*M)   (index.hpp:2085, 3731-3748) */
    for (uint64_t i = 0; i < ix->size; ++i) {
        ix->node_offsets[i] = off;
        int16_t level = (int16_t)ld16(ix->levels + 2 * i);
        off += 10 + (4 + 4 * ix->connectivity_base) + (uint64_t)level * (4 + 4 * ix->connectivity);
    }
    ix->node_offsets[ix->size] = off;
    if (off > length) { free(ix->node_offsets); ix->node_offsets = NULL; *error = "Failed to pull nodes from the stream"; return 1; }
    return 0;
}

void uo_close(uo_index_t* ix) {
    free(ix->node_offsets);
    ix->node_offsets = NULL;
}

uint64_t uo_key(const uo_index_t* ix, uint64_t slot) { return ld64(ix->image + ix->node_offsets[slot]); }
int uo_level(const uo_index_t* ix, uint64_t slot) { return (int16_t)ld16(ix->levels + 2 * slot); }

static const uint8_t* neighbors_tape(const uo_index_t* ix, uint64_t slot, int level) {
    /* index.hpp:3786-3795: base list first, then one fixed-size list per upper level */
    const uint8_t* tape = ix->image + ix->node_offsets[slot] + 10;
    if (level == 0)
        return tape;
    return tape + (4 + 4 * ix->connectivity_base) + (uint64_t)(level - 1) * (4 + 4 * ix->connectivity);
}

uint32_t uo_neighbors(const uo_index_t* ix, uint64_t slot, int level, uint32_t* out, uint32_t cap) {
    const uint8_t* tape = neighbors_tape(ix, slot, level);
    uint32_t n = ld32(tape);
    for (uint32_t i = 0; i < n && i < cap; ++i)
        out[i] = ld32(tape + 4 + 4 * i);
    return n;
}

static const uint8_t* vector_at(const uo_index_t* ix, uint64_t slot) { return ix->vectors + slot * ix->cols; }

/* ------------------------------------------------------------------------------------------------------------------
 *  Metrics (index_plugins.hpp:1309-1657, dispatch 1930-2008). Float accumulation layout is explicit — see header.
 *
 *  Which struct the reference instantiates per (metric, scalar) — `configure_with_autovec`, 1930-2008:
 *    ip / cos / l2sq / pearson   bf16, f16, f32 → result f32;  f64 → result f64;  i8 → metric_{cos,l2sq}_i8_t (int32) or
 *                                metric_{ip,pearson}_gt<i8_t, f32_t> (float accumulation of small integers)
 *    divergence                  bf16, f16, f32 (f32 arithmetic), f64;   haversine  f32, f64 (first two scalars)
 *    hamming / tanimoto (= jaccard) / sorensen   b1x8 bytes
 *  Every result is narrowed to `float` (distance_punned_t, 1660; `equidimensional_`, 2010-2014).
 * ---------------------------------------------------------------------------------------------------------------- */
#define UO_CHUNK 16 /* bytes dealt to one lane at a time */
#define UO_MAX_LANES 64

static float load_float(uint8_t scalar_kind, const uint8_t* p, uint64_t i) {
    switch (scalar_kind) {
    case UO_SCALAR_F32: { float f; memcpy(&f, p + 4 * i, 4); return f; }
    case UO_SCALAR_F16: return f16_to_f32(ld16(p + 2 * i));
    case UO_SCALAR_BF16: { uint32_t b = (uint32_t)ld16(p + 2 * i) << 16; float f; memcpy(&f, &b, 4); return f; }
    case UO_SCALAR_I8: return (float)(int8_t)p[i];
    default: return 0.f;
    }
}
static double load_f64(const uint8_t* p, uint64_t i) { double d; memcpy(&d, p + 8 * i, 8); return d; }

/* The running sums of every equidimensional metric: ab = Σab, a2 = Σa², b2 = Σb², l2 = Σ(a-b)² (ip, cos, l2sq);
 * sa = Σa, sb = Σb (pearson, 1478-1520); kp, kq = the two Kullback-Leibler sums of divergence (1526-1551). Generated
 * twice: f32 arithmetic (result_t = f32_t) and f64 arithmetic (result_t = f64_t).
 *
 * lanes <= 0: the reference loop order — one chain per accumulator in element order, unfused multiply-add. The
 * reference lets the compiler reassociate (`omp simd reduction`), so last-bit agreement with one particular build of
 * it is neither promised nor needed.
 * lanes = G:  16-byte chunks dealt round-robin to G lanes, one fused chain per lane in chunk order, XOR butterfly. */
enum { UO_AB, UO_A2, UO_B2, UO_L2, UO_SA, UO_SB, UO_KP, UO_KQ, UO_SUMS };

#define UO_DEFINE_SUMS(NAME, T, FMA, LOG, EPSILON, LOAD_A, LOAD_B, PER_CHUNK)                                           \
    typedef struct { T sum[UO_SUMS]; } NAME##_acc_t;                                                                    \
    static void NAME##_step(NAME##_acc_t* acc, T a, T b, int with_logs, int fused) {                                    \
        T* s = acc->sum;                                                                                                \
        T t = a - b;                                                                                                    \
        if (fused) {                                                                                                    \
            s[UO_AB] = FMA(a, b, s[UO_AB]), s[UO_A2] = FMA(a, a, s[UO_A2]), s[UO_B2] = FMA(b, b, s[UO_B2]);             \
            s[UO_L2] = FMA(t, t, s[UO_L2]);                                                                             \
        } else {                                                                                                        \
            s[UO_AB] += a * b, s[UO_A2] += a * a, s[UO_B2] += b * b, s[UO_L2] += t * t;                                 \
        }                                                                                                               \
        s[UO_SA] += a, s[UO_SB] += b;                                                                                   \
        if (with_logs) {                                                                                                \
            T m = (a + b) / 2 + EPSILON;                                                                                \
            s[UO_KP] += a * LOG((a + EPSILON) / m);                                                                     \
            s[UO_KQ] += b * LOG((b + EPSILON) / m);                                                                     \
        }                                                                                                               \
    }                                                                                                                   \
    /* XOR butterfly, offsets lanes/2 … 1: every lane ends with the same bit pattern (fp add is commutative). */        \
    static T NAME##_butterfly(T* v, int lanes) {                                                                        \
        T tmp[UO_MAX_LANES];                                                                                            \
        for (int off = lanes / 2; off >= 1; off >>= 1) {                                                                \
            for (int l = 0; l < lanes; ++l)                                                                             \
                tmp[l] = v[l] + v[l ^ off];                                                                             \
            memcpy(v, tmp, sizeof(T) * (size_t)lanes);                                                                  \
        }                                                                                                               \
        return v[0];                                                                                                    \
    }                                                                                                                   \
    static NAME##_acc_t NAME##_sums(uint8_t scalar_kind, const uint8_t* a, const uint8_t* b, uint64_t dims, int lanes,  \
                                    int with_logs) {                                                                    \
        (void)scalar_kind;                                                                                              \
        NAME##_acc_t total;                                                                                             \
        memset(&total, 0, sizeof(total));                                                                               \
        if (lanes <= 0) {                                                                                               \
            for (uint64_t i = 0; i < dims; ++i)                                                                         \
                NAME##_step(&total, LOAD_A, LOAD_B, with_logs, 0);                                                      \
            return total;                                                                                               \
        }                                                                                                               \
        uint64_t per_chunk = PER_CHUNK;                                                                                 \
        uint64_t chunks = (dims + per_chunk - 1) / per_chunk;                                                           \
        NAME##_acc_t lane[UO_MAX_LANES];                                                                                \
        memset(lane, 0, sizeof(lane));                                                                                  \
        for (uint64_t c = 0; c < chunks; ++c)                                                                           \
            for (uint64_t i = c * per_chunk; i < (c + 1) * per_chunk && i < dims; ++i)                                  \
                NAME##_step(&lane[c % (uint64_t)lanes], LOAD_A, LOAD_B, with_logs, 1);                                  \
        T v[UO_MAX_LANES];                                                                                              \
        for (int f = 0; f < UO_SUMS; ++f) {                                                                             \
            for (int l = 0; l < lanes; ++l)                                                                             \
                v[l] = lane[l].sum[f];                                                                                  \
            total.sum[f] = NAME##_butterfly(v, lanes);                                                                  \
        }                                                                                                               \
        return total;                                                                                                   \
    }                                                                                                                   \
    /* pearson: metric_pearson_gt 1478-1520 */                                                                          \
    static T NAME##_pearson(const NAME##_acc_t* acc, uint64_t dims) {                                                   \
        const T* s = acc->sum;                                                                                          \
        if (dims <= 1)                                                                                                  \
            return 0;                                                                                                   \
        T n = (T)dims;                                                                                                  \
        T denom = (n * s[UO_A2] - s[UO_SA] * s[UO_SA]) * (n * s[UO_B2] - s[UO_SB] * s[UO_SB]);                          \
        if (denom == 0)                                                                                                 \
            return 0;                                                                                                   \
        T corr = n * s[UO_AB] - s[UO_SA] * s[UO_SB];                                                                    \
        return 1 - corr / (T)sqrt((double)denom); /* sqrt in double then narrowed = the correctly rounded sqrtf */       \
    }

#define UO_FLOAT_PER_CHUNK (UO_CHUNK / (uo_bytes_per_vector(scalar_kind, 8) / 8))
UO_DEFINE_SUMS(f32, float, fmaf, logf, 1.1920928955078125e-7f, load_float(scalar_kind, a, i), load_float(scalar_kind, b, i),
               UO_FLOAT_PER_CHUNK)
UO_DEFINE_SUMS(f64, double, fma, log, 2.220446049250313e-16, load_f64(a, i), load_f64(b, i), 2)

/* metric_haversine_gt 1636-1657: latitude, longitude in degrees; angle_to_radians 203 */
static float haversine_f32(const uint8_t* a, const uint8_t* b) {
    const float pi = (float)3.14159265358979323846;
    float lat_a = load_float(UO_SCALAR_F32, a, 0), lon_a = load_float(UO_SCALAR_F32, a, 1);
    float lat_b = load_float(UO_SCALAR_F32, b, 0), lon_b = load_float(UO_SCALAR_F32, b, 1);
    float lat_delta = ((lat_b - lat_a) * pi / 180.f) / 2, lon_delta = ((lon_b - lon_a) * pi / 180.f) / 2;
    float cla = lat_a * pi / 180.f, clb = lat_b * pi / 180.f;
    float s1 = sinf(lat_delta), s2 = sinf(lon_delta);
    float x = s1 * s1 + cosf(cla) * cosf(clb) * (s2 * s2);
    return 2 * asinf(sqrtf(x));
}
static float haversine_f64(const uint8_t* a, const uint8_t* b) {
    const double pi = 3.14159265358979323846;
    double lat_a = load_f64(a, 0), lon_a = load_f64(a, 1), lat_b = load_f64(b, 0), lon_b = load_f64(b, 1);
    double lat_delta = ((lat_b - lat_a) * pi / 180.0) / 2, lon_delta = ((lon_b - lon_a) * pi / 180.0) / 2;
    double cla = lat_a * pi / 180.0, clb = lat_b * pi / 180.0;
    double s1 = sin(lat_delta), s2 = sin(lon_delta);
    double x = s1 * s1 + cos(cla) * cos(clb) * (s2 * s2);
    return (float)(2 * asin(sqrt(x)));
}

float uo_distance(uint8_t metric_kind, uint8_t scalar_kind, const void* av, const void* bv, uint64_t dims,
                  int lanes) {
    const uint8_t* a = (const uint8_t*)av;
    const uint8_t* b = (const uint8_t*)bv;
    if (scalar_kind == UO_SCALAR_B1) {
        /* bit-set metrics over ceil(d/8) bytes (1744): metric_hamming_gt 1392-1414, metric_tanimoto_gt 1420-1445 (jaccard
         * maps to it, 2003-2004), metric_sorensen_gt 1451-1476. The counts are exact; the fractions are f32 divisions. */
        uint64_t words = (dims + 7) / 8, differ = 0, both = 0, either = 0, total = 0;
        for (uint64_t i = 0; i < words; ++i) {
            differ += (uint64_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
            both += (uint64_t)__builtin_popcount((unsigned)(a[i] & b[i]));
            either += (uint64_t)__builtin_popcount((unsigned)(a[i] | b[i]));
            total += (uint64_t)__builtin_popcount((unsigned)a[i]) + (uint64_t)__builtin_popcount((unsigned)b[i]);
        }
        switch (metric_kind) {
        case UO_METRIC_HAMMING: return (float)differ;
        case UO_METRIC_JACCARD:
        case UO_METRIC_TANIMOTO: return 1 - (float)both / (float)either;
        case UO_METRIC_SORENSEN: return 1 - 2 * (float)both / (float)total;
        default: return NAN;
        }
    }
    if (metric_kind == UO_METRIC_HAVERSINE) {
        if (scalar_kind == UO_SCALAR_F32) return haversine_f32(a, b);
        if (scalar_kind == UO_SCALAR_F64) return haversine_f64(a, b);
        return NAN;
    }
    if (scalar_kind == UO_SCALAR_I8 && (metric_kind == UO_METRIC_L2SQ || metric_kind == UO_METRIC_COS)) {
        int32_t ab = 0, a2 = 0, b2 = 0, l2 = 0;
        for (uint64_t i = 0; i < dims; ++i) {
            int16_t ai = (int8_t)a[i], bi = (int8_t)b[i];
            ab += ai * bi, a2 += ai * ai, b2 += bi * bi;
            l2 += (ai - bi) * (ai - bi);
        }
        if (metric_kind == UO_METRIC_L2SQ) /* metric_l2sq_i8_t 1613-1630 */
            return (float)l2;
        /* metric_cos_i8_t 1583-1607, incl. the `ab == 0 → 0` quirk */
        float a2f = sqrtf((float)a2), b2f = sqrtf((float)b2);
        return (ab != 0) ? (1.f - (float)ab / (a2f * b2f)) : 0.f;
    }
    if (scalar_kind == UO_SCALAR_I8 && metric_kind == UO_METRIC_DIVERGENCE)
        return NAN; /* no such instantiation (1991-2001) */
    if (scalar_kind == UO_SCALAR_F64) {
        f64_acc_t s = f64_sums(scalar_kind, a, b, dims, lanes, metric_kind == UO_METRIC_DIVERGENCE);
        switch (metric_kind) {
        case UO_METRIC_IP: return (float)(1 - s.sum[UO_AB]);
        case UO_METRIC_COS:
            if (s.sum[UO_A2] == 0 && s.sum[UO_B2] == 0) return 0.f;
            if (s.sum[UO_A2] == 0 || s.sum[UO_B2] == 0) return 1.f;
            return (float)(1 - s.sum[UO_AB] / (sqrt(s.sum[UO_A2]) * sqrt(s.sum[UO_B2])));
        case UO_METRIC_L2SQ: return (float)s.sum[UO_L2];
        case UO_METRIC_PEARSON: return (float)f64_pearson(&s, dims);
        case UO_METRIC_DIVERGENCE: return (float)((s.sum[UO_KP] + s.sum[UO_KQ]) / 2);
        default: return NAN;
        }
    }
    f32_acc_t s = f32_sums(scalar_kind, a, b, dims, lanes, metric_kind == UO_METRIC_DIVERGENCE);
    switch (metric_kind) {
    case UO_METRIC_IP: return 1.f - s.sum[UO_AB]; /* metric_ip_gt 1309-1326 */
    case UO_METRIC_COS: {                 /* metric_cos_gt 1334-1359 */
        if (s.sum[UO_A2] == 0.f && s.sum[UO_B2] == 0.f) return 0.f;
        if (s.sum[UO_A2] == 0.f || s.sum[UO_B2] == 0.f) return 1.f;
        return 1.f - s.sum[UO_AB] / (sqrtf(s.sum[UO_A2]) * sqrtf(s.sum[UO_B2]));
    }
    case UO_METRIC_L2SQ: return s.sum[UO_L2]; /* metric_l2sq_gt 1365-1385 */
    case UO_METRIC_PEARSON: return f32_pearson(&s, dims);
    case UO_METRIC_DIVERGENCE: return (s.sum[UO_KP] + s.sum[UO_KQ]) / 2; /* metric_divergence_gt 1526-1551 */
    default: return NAN;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 *  Casts (index_plugins.hpp:1105-1224)
 * ---------------------------------------------------------------------------------------------------------------- */
static double load_double(uint8_t kind, const uint8_t* p, uint64_t i) {
    if (kind == UO_SCALAR_F64) { double d; memcpy(&d, p + 8 * i, 8); return d; }
    return (double)load_float(kind, p, i);
}

int uo_cast(uint8_t from, uint8_t to, const void* inv, uint64_t dims, void* outv) {
    const uint8_t* in = (const uint8_t*)inv;
    uint8_t* out = (uint8_t*)outv;
    if (from == to)
        return 0; /* cast_gt<T,T>::try_ returns false (1115-1137) */
    if (to == UO_SCALAR_B1) {
        /* cast_to_b1x8_gt 1139-1158: clears dim/8 bytes only, MSB-first bits, `x > 0` */
        memset(out, 0, (size_t)(dims / 8));
        for (uint64_t i = 0; i < dims; ++i) {
            int positive = (from == UO_SCALAR_I8) ? ((int8_t)in[i] > 0) : (load_double(from, in, i) > 0);
            out[i / 8] |= positive ? (uint8_t)(128 >> (i & 7)) : 0;
        }
        return 1;
    }
    if (from == UO_SCALAR_B1) {
        /* cast_from_b1x8_gt 1160-1170: set bits → 1, others → 0 */
        for (uint64_t i = 0; i < dims; ++i) {
            int bit = (in[i / 8] & (128 >> (i & 7))) != 0;
            if (to == UO_SCALAR_F32) { float f = (float)bit; memcpy(out + 4 * i, &f, 4); }
            else if (to == UO_SCALAR_F16) { uint16_t h = f32_to_f16((float)bit); memcpy(out + 2 * i, &h, 2); }
            else if (to == UO_SCALAR_I8) out[i] = (uint8_t)bit;
            else if (to == UO_SCALAR_F64) { double d = bit; memcpy(out + 8 * i, &d, 8); }
            else if (to == UO_SCALAR_BF16) { uint16_t h = bit ? 0x3F80u : 0u; memcpy(out + 2 * i, &h, 2); }
        }
        return 1;
    }
    if (to == UO_SCALAR_I8) {
        /* cast_to_i8_gt 1172-1191: L2-normalise in double, scale by 127, clamp, truncate toward zero */
        double magnitude = 0.0;
        for (uint64_t i = 0; i < dims; ++i) {
            double x = load_double(from, in, i);
            magnitude += x * x;
        }
        magnitude = sqrt(magnitude);
        for (uint64_t i = 0; i < dims; ++i) {
            double v = load_double(from, in, i) * 127.0 / magnitude;
            v = v < -127.0 ? -127.0 : (v > 127.0 ? 127.0 : v);
            out[i] = (uint8_t)(int8_t)v;
        }
        return 1;
    }
    if (from == UO_SCALAR_I8) {
        /* cast_from_i8_gt 1193-1201: x / 127.f in the target type */
        for (uint64_t i = 0; i < dims; ++i) {
            int8_t x = (int8_t)in[i];
            if (to == UO_SCALAR_F32) { float f = (float)x / 127.f; memcpy(out + 4 * i, &f, 4); }
            else if (to == UO_SCALAR_F64) { double d = (double)x / 127.f; memcpy(out + 8 * i, &d, 8); }
            else if (to == UO_SCALAR_F16) {
                /* f16_bits_t(int)/float → float division, then f16_bits_t(float) (index_plugins.hpp:486-496, 1198) */
                float q = f16_to_f32(f32_to_f16((float)x)) / 127.f;
                uint16_t h = f32_to_f16(q);
                memcpy(out + 2 * i, &h, 2);
            } else if (to == UO_SCALAR_BF16) {
                /* bf16_bits_t(int) is exact for |x| <= 127 (8 significant bits); the quotient is truncated (f32_to_bf16, 453-469) */
                float q = (float)x / 127.f;
                uint32_t bits;
                memcpy(&bits, &q, 4);
                uint16_t h = (uint16_t)(bits >> 16);
                memcpy(out + 2 * i, &h, 2);
            }
        }
        return 1;
    }
    /* generic cast_gt 1105-1113: to_scalar(from) element-wise; f16 goes through float (488-489) */
    for (uint64_t i = 0; i < dims; ++i) {
        if (to == UO_SCALAR_F64) { double d = load_double(from, in, i); memcpy(out + 8 * i, &d, 8); continue; }
        float f = (from == UO_SCALAR_F64) ? (float)load_double(from, in, i) : load_float(from, in, i);
        if (to == UO_SCALAR_F32) memcpy(out + 4 * i, &f, 4);
        else if (to == UO_SCALAR_F16) { uint16_t h = f32_to_f16(f); memcpy(out + 2 * i, &h, 2); }
        else if (to == UO_SCALAR_BF16) { uint32_t b; memcpy(&b, &f, 4); uint16_t h = (uint16_t)(b >> 16); memcpy(out + 2 * i, &h, 2); }
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------------------------
 *  Containers. candidate_t = {float distance; u32 slot}, ordered by distance only (index.hpp:2097-2101).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct { float distance; uint32_t slot; } cand_t;

/* sorted_buffer_gt (index.hpp:845-956) — `top` */
typedef struct { cand_t* e; size_t size, cap; } sorted_t;

static size_t sorted_lower_bound(const sorted_t* s, float d) {
    /* std::lower_bound by `a.distance < b.distance`: first index whose distance is NOT < d (index.hpp:916, 929) */
    size_t lo = 0, hi = s->size;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (s->e[mid].distance < d) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

static int sorted_insert(sorted_t* s, cand_t c, size_t limit) {
    /* index.hpp:928-939: new element goes BEFORE equal ones; when full the last (worst) falls off; a new element
     * that is >= all while full is rejected */
    size_t slot = s->size ? sorted_lower_bound(s, c.distance) : 0;
    if (slot == limit)
        return 0;
    int full = s->size == limit;
    size_t to_move = s->size - slot - (size_t)full;
    for (size_t i = 0; i < to_move; ++i) {
        size_t src = s->size - 1 - (size_t)full - i;
        s->e[src + 1] = s->e[src];
    }
    s->e[slot] = c;
    s->size += full ? 0 : 1;
    return 1;
}

/* max_heap_gt on {-distance, slot} (index.hpp:664-835) — `next`. Stored here with the NEGATED distance exactly as
 * the reference stores it, so every comparison below is literally the reference's `less`. */
typedef struct { cand_t* e; size_t size, cap; } heap_t;

static int heap_reserve(heap_t* h, size_t n) {
    if (n <= h->cap) return 1;
    size_t cap = h->cap ? h->cap : 16;
    while (cap < n) cap *= 2;
    cand_t* e = (cand_t*)realloc(h->e, cap * sizeof(cand_t));
    if (!e) return 0;
    h->e = e, h->cap = cap;
    return 1;
}

/* Sizing statistics of the last uo_search call on this thread (how big `next` and `visits` got): used by the
 * tests / DESIGN.md to size the device scratch, not part of the restated algorithm. */
static __thread size_t stat_peak_next = 0, stat_visits = 0;
/* Shape of the last traversal on this thread (design studies for the device kernel, not part of the restated algorithm):
 * [0] hops on level 0, [1] hops whose node had been pushed during the hop right before (a "newcomer": the next hop was not
 * the frontier's best at pop time), [2] candidates admitted (pushed), [3] fresh neighbours measured, [4] frontier size at the
 * end, [5] of those, entries that could still be popped (distance <= radius), [6] sum over hops of the frontier size,
 * [7] sum over hops of the frontier's live entries (sampled every 16th hop). */
static __thread double stat_shape[8];
size_t uo_last_peak_next(void) { return stat_peak_next; }
size_t uo_last_visits(void) { return stat_visits; }

static void heap_insert(heap_t* h, cand_t c) {
    /* insert_reserved + shift_up (index.hpp:765-770, 808-811): swap while parent < child, strictly */
    heap_reserve(h, h->size + 1);
    size_t i = h->size++;
    if (h->size > stat_peak_next) stat_peak_next = h->size;
    h->e[i] = c;
    while (i && h->e[(i - 1) / 2].distance < h->e[i].distance) {
        cand_t t = h->e[(i - 1) / 2];
        h->e[(i - 1) / 2] = h->e[i];
        h->e[i] = t;
        i = (i - 1) / 2;
    }
}

static cand_t heap_pop(heap_t* h) {
    /* pop + shift_down (index.hpp:786-794, 819-834): left child preferred unless right is strictly greater */
    cand_t result = h->e[0];
    h->e[0] = h->e[h->size - 1];
    h->size--;
    size_t i = 0;
    for (;;) {
        size_t max_idx = i, left = 2 * i + 1, right = 2 * i + 2;
        if (left < h->size && h->e[max_idx].distance < h->e[left].distance) max_idx = left;
        if (right < h->size && h->e[max_idx].distance < h->e[right].distance) max_idx = right;
        if (max_idx == i) break;
        cand_t t = h->e[i];
        h->e[i] = h->e[max_idx];
        h->e[max_idx] = t;
        i = max_idx;
    }
    return result;
}

/* growing_hash_set_gt (index.hpp:1085-1211) is an EXACT set; only membership is observable, so a bitmap restates it. */
typedef struct { uint8_t* bits; uint32_t* touched; size_t touched_count, touched_cap; } visits_t;

static int visits_set(visits_t* v, uint32_t slot) {
    /* returns the previous state, like growing_hash_set_gt::set (index.hpp:1163-1175) */
    uint8_t mask = (uint8_t)(1u << (slot & 7));
    if (v->bits[slot >> 3] & mask) return 1;
    v->bits[slot >> 3] |= mask;
    if (v->touched_count == v->touched_cap) {
        v->touched_cap = v->touched_cap ? v->touched_cap * 2 : 1024;
        v->touched = (uint32_t*)realloc(v->touched, v->touched_cap * sizeof(uint32_t));
    }
    v->touched[v->touched_count++] = slot;
    return 0;
}

static void visits_clear(visits_t* v) {
    for (size_t i = 0; i < v->touched_count; ++i)
        v->bits[v->touched[i] >> 3] = 0;
    v->touched_count = 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 *  The search itself
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const uo_index_t* ix;
    const uint8_t* query; /* storage kind */
    int lanes;
    uo_filter_t filter;
    void* filter_state;
    uint64_t computed_distances, iteration_cycles; /* context_t counters (index.hpp:2208-2211) */
    sorted_t top;
    heap_t next;
    visits_t visits;
    int has_tombstones; /* any key == free_key_ (only looked up when the in-`top` frontier mode is requested) */
} ctx_t;

static float measure(ctx_t* c, uint32_t slot) {
    /* context_t::measure (index.hpp:2215-2222) → metric_proxy_t (index_dense.hpp:419-438) */
    c->computed_distances++;
    return uo_distance(c->ix->metric_kind, c->ix->scalar_kind, c->query, vector_at(c->ix, slot), c->ix->dimensions,
                       c->lanes);
}

static int allow(ctx_t* c, uint32_t slot) {
    /* index_dense.hpp:2071-2081: `key != free_key_` (UINT64_MAX, index_dense.hpp:513) [&& user predicate(key)] */
    uint64_t key = uo_key(c->ix, slot);
    if (key == UINT64_MAX) return 0;
    return c->filter ? (c->filter(key, c->filter_state) != 0) : 1;
}

static uint32_t search_for_one(ctx_t* c, uint32_t closest, int begin_level, int end_level) {
    /* index.hpp:3964-4003 — greedy descent; ALL neighbours scanned in list order, strict `<`, repeat while changed */
    float closest_dist = measure(c, closest);
    uint32_t nbrs[4096];
    for (int level = begin_level; level > end_level; --level) {
        int changed;
        do {
            changed = 0;
            uint32_t n = uo_neighbors(c->ix, closest, level, nbrs, 4096);
            for (uint32_t i = 0; i < n; ++i) {
                float d = measure(c, nbrs[i]);
                if (d < closest_dist) closest_dist = d, closest = nbrs[i], changed = 1;
            }
            c->iteration_cycles++;
        } while (changed);
    }
    return closest;
}

/* The frontier mode of the device kernels (usearch_amd/csrc/kernels.hpp, frontier_top_k), restated so that the GPU can be
 * checked bit for bit in it: there is no `next` container; the frontier is the not-yet-expanded part of `top` (bit 31 of a
 * kept candidate's slot says "expanded"). A candidate enters `next` and `top` together (index.hpp:4233-4240) and what `top`
 * evicts is farther than the radius for good, so the reference's loop only ever expands members still in `top`: same hops,
 * same counters, same results whenever the distances that meet in the frontier are distinct (tests/test_oracle_frontier.py
 * checks this mode against the reference-shaped one and documents the tie cases). Valid only when every member is a result
 * candidate (no predicate, no tombstones) and slots stay below 2^31 — the caller checks, as the engine does. */
#define UO_CLOSED 0x80000000u
static __thread int frontier_in_top_requested = 0;
void uo_set_frontier_in_top(int enabled) { frontier_in_top_requested = enabled; }

static void search_to_find_in_base_frontier_in_top(ctx_t* c, uint32_t start, size_t top_limit) {
    visits_clear(&c->visits);
    c->next.size = 0;
    c->top.size = 0;
    float radius = measure(c, start);
    visits_set(&c->visits, start);
    sorted_insert(&c->top, (cand_t){radius, start}, top_limit);
    uint32_t nbrs[4096];
    memset(stat_shape, 0, sizeof(stat_shape));
    for (;;) {
        size_t first_open = 0;
        while (first_open < c->top.size && (c->top.e[first_open].slot & UO_CLOSED))
            ++first_open;
        if (first_open == c->top.size)
            break;
        uint32_t expanded = c->top.e[first_open].slot;
        c->top.e[first_open].slot |= UO_CLOSED;
        c->iteration_cycles++;
        stat_shape[0] += 1;
        uint32_t n = uo_neighbors(c->ix, expanded, 0, nbrs, 4096);
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t successor = nbrs[i];
            if (visits_set(&c->visits, successor))
                continue;
            float d = measure(c, successor);
            stat_shape[3] += 1;
            if (c->top.size < top_limit || d < radius) {
                sorted_insert(&c->top, (cand_t){d, successor}, top_limit);
                radius = c->top.e[c->top.size - 1].distance;
                stat_shape[2] += 1;
            }
        }
    }
    for (size_t i = 0; i < c->top.size; ++i)
        c->top.e[i].slot &= ~UO_CLOSED;
}

static int frontier_in_top_applies(const ctx_t* c, size_t top_limit) {
    /* the engine's rule (usearch_amd/csrc/engine.hip, search_begin): float-valued pair, `top` in registers (expansion
     * <= 1024), no predicate, no tombstones, slots < 2^31 */
    if (!frontier_in_top_requested || c->filter || top_limit > 1024 || c->ix->size >= 0x80000000ull)
        return 0;
    if (c->ix->scalar_kind == UO_SCALAR_B1 || c->ix->scalar_kind == UO_SCALAR_I8)
        return 0;
    return !c->has_tombstones;
}

static void search_to_find_in_base(ctx_t* c, uint32_t start, size_t top_limit) {
    /* index.hpp:4176-4246 */
    if (frontier_in_top_applies(c, top_limit)) {
        search_to_find_in_base_frontier_in_top(c, start, top_limit);
        return;
    }
    visits_clear(&c->visits);
    c->next.size = 0;
    c->top.size = 0;
    float radius = measure(c, start);
    heap_insert(&c->next, (cand_t){-radius, start});
    visits_set(&c->visits, start);
    if (allow(c, start))
        sorted_insert(&c->top, (cand_t){radius, start}, top_limit); /* insert_reserved ≡ insert when empty */

    uint32_t nbrs[4096];
    memset(stat_shape, 0, sizeof(stat_shape));
    size_t previous_hop_first_push = (size_t)-1, pushes = 0; /* telemetry only: pushes are numbered to tell newcomers */
    uint32_t* pushed_at = (uint32_t*)calloc((size_t)c->ix->size + 1, sizeof(uint32_t));
    while (c->next.size) {
        cand_t candidate = c->next.e[0];
        if ((-candidate.distance) > radius && c->top.size == top_limit) /* 4210: strict `>` */
            break;
        heap_pop(&c->next);
        c->iteration_cycles++;
        stat_shape[0] += 1;
        if (pushed_at && previous_hop_first_push != (size_t)-1 && pushed_at[candidate.slot] > previous_hop_first_push)
            stat_shape[1] += 1;
        stat_shape[6] += (double)c->next.size;
        if (((size_t)stat_shape[0] & 15) == 0)
            for (size_t i = 0; i < c->next.size; ++i)
                stat_shape[7] += (-c->next.e[i].distance) <= radius || c->top.size < top_limit;
        previous_hop_first_push = pushes;
        uint32_t n = uo_neighbors(c->ix, candidate.slot, 0, nbrs, 4096);
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t successor = nbrs[i];
            if (visits_set(&c->visits, successor)) /* 4229 */
                continue;
            float d = measure(c, successor);
            stat_shape[3] += 1;
            if (c->top.size < top_limit || d < radius) { /* 4233: strict `<` */
                heap_insert(&c->next, (cand_t){-d, successor});
                stat_shape[2] += 1;
                if (pushed_at)
                    pushed_at[successor] = (uint32_t)++pushes;
                if (allow(c, successor)) {
                    sorted_insert(&c->top, (cand_t){d, successor}, top_limit);
                    radius = c->top.e[c->top.size - 1].distance; /* top.top() = last = worst kept (891) */
                }
            }
        }
    }
    stat_shape[4] = (double)c->next.size;
    for (size_t i = 0; i < c->next.size; ++i)
        stat_shape[5] += (-c->next.e[i].distance) <= radius;
    free(pushed_at);
}

static void search_exact(ctx_t* c, size_t count) {
    /* index.hpp:4252-4268 */
    c->top.size = 0;
    for (uint64_t i = 0; i < c->ix->size; ++i) {
        if (!allow(c, (uint32_t)i)) continue;
        float d = measure(c, (uint32_t)i);
        sorted_insert(&c->top, (cand_t){d, (uint32_t)i}, count);
    }
}

static float signaling_nan(void) {
    /* std::numeric_limits<float>::signaling_NaN() on x86-64 gcc/clang (index.hpp:2717-2719) */
    uint32_t bits = 0x7FA00000u;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static size_t search_with_ctx(ctx_t* c, const void* query, uint8_t query_kind, size_t wanted, size_t expansion,
                              int exact, uint64_t* keys, float* distances, uint64_t* visited, uint64_t* computed) {
    const uo_index_t* ix = c->ix;
    size_t count = 0;
    c->computed_distances = c->iteration_cycles = 0;
    stat_peak_next = 0;
    /* index_dense.hpp:2058-2064: cast the query into the storage kind first */
    size_t bpv = uo_bytes_per_vector(ix->scalar_kind, ix->dimensions);
    uint8_t* casted = (uint8_t*)calloc(bpv + 16, 1);
    c->query = uo_cast(query_kind, ix->scalar_kind, query, ix->dimensions, casted) ? casted : (const uint8_t*)query;

    if (wanted && ix->size) { /* index.hpp:3025-3037 */
        if (!expansion) expansion = 64; /* default_expansion_search(), index.hpp:3029-3030 */
        size_t limit = exact ? wanted : (expansion > wanted ? expansion : wanted); /* 3052 */
        if (c->top.cap < limit + 1) {
            c->top.e = (cand_t*)realloc(c->top.e, (limit + 1) * sizeof(cand_t));
            c->top.cap = limit + 1;
        }
        if (exact) {
            search_exact(c, wanted);
        } else {
            uint32_t closest = search_for_one(c, (uint32_t)ix->entry_slot, (int)ix->max_level, 0);
            search_to_find_in_base(c, closest, limit);
        }
        count = c->top.size < wanted ? c->top.size : wanted; /* shrink(wanted), 3067-3073 */
    }
    /* dump_to(keys, distances, capacity = wanted): index.hpp:2707-2722 */
    for (size_t i = 0; i < count; ++i) {
        if (keys) keys[i] = uo_key(ix, c->top.e[i].slot);
        if (distances) distances[i] = c->top.e[i].distance;
    }
    for (size_t i = count; i < wanted; ++i) {
        if (keys) keys[i] = 0;
        if (distances) distances[i] = signaling_nan();
    }
    if (visited) *visited = c->iteration_cycles;
    if (computed) *computed = c->computed_distances;
    stat_visits = c->visits.touched_count;
    free(casted);
    return count;
}

static int ctx_init(ctx_t* c, const uo_index_t* ix, int lanes, uo_filter_t filter, void* state) {
    memset(c, 0, sizeof(*c));
    c->ix = ix, c->lanes = lanes, c->filter = filter, c->filter_state = state;
    c->visits.bits = (uint8_t*)calloc((size_t)(ix->size / 8 + 1), 1);
    if (frontier_in_top_requested)
        for (uint64_t i = 0; i < ix->size && !c->has_tombstones; ++i)
            c->has_tombstones = uo_key(ix, i) == UINT64_MAX;
    return c->visits.bits != NULL;
}

static void ctx_free(ctx_t* c) {
    free(c->top.e);
    free(c->next.e);
    free(c->visits.bits);
    free(c->visits.touched);
}

size_t uo_search(const uo_index_t* ix, const void* query, uint8_t query_kind, size_t wanted, size_t expansion,
                 int exact, int lanes, uo_filter_t filter, void* filter_state, uint64_t* keys, float* distances,
                 uint64_t* visited, uint64_t* computed) {
    ctx_t c;
    if (!ctx_init(&c, ix, lanes, filter, filter_state)) return 0;
    size_t n = search_with_ctx(&c, query, query_kind, wanted, expansion, exact, keys, distances, visited, computed);
    ctx_free(&c);
    return n;
}

void uo_search_many(const uo_index_t* ix, const void* queries, uint8_t query_kind, size_t count, size_t stride,
                    size_t wanted, size_t expansion, int exact, int lanes, uint64_t* keys, float* distances,
                    uint64_t* counts, uint64_t* visited, uint64_t* computed) {
    ctx_t c;
    if (!ctx_init(&c, ix, lanes, NULL, NULL)) return;
    for (size_t q = 0; q < count; ++q) {
        size_t n = search_with_ctx(&c, (const uint8_t*)queries + q * stride, query_kind, wanted, expansion, exact,
                                   keys ? keys + q * wanted : NULL, distances ? distances + q * wanted : NULL,
                                   visited ? visited + q : NULL, computed ? computed + q : NULL);
        if (counts) counts[q] = n;
    }
    ctx_free(&c);
}

void uo_cluster_many(const uo_index_t* ix, const void* queries, uint8_t query_kind, size_t count, size_t stride,
                     size_t level, int lanes, uint64_t* keys, float* distances, uint64_t* visited, uint64_t* computed) {
    /* index_gt::cluster (index.hpp:3089-3125) behind index_dense_gt::cluster_ (index_dense.hpp:788-793): the greedy descent
     * of search_for_one_ from the top level down to `level` (target level = level - 1, or 0), then ONE more evaluation of
     * the winner's distance (3115), which the counters include. An empty index fails with "No clusters to identify":
     * key 0 / signalling NaN here. */
    ctx_t c;
    if (!ctx_init(&c, ix, lanes, NULL, NULL)) return;
    size_t bpv = uo_bytes_per_vector(ix->scalar_kind, ix->dimensions);
    uint8_t* casted = (uint8_t*)calloc(bpv + 16, 1);
    for (size_t q = 0; q < count; ++q) {
        const uint8_t* query = (const uint8_t*)queries + q * stride;
        c.computed_distances = c.iteration_cycles = 0;
        memset(casted, 0, bpv + 16);
        c.query = uo_cast(query_kind, ix->scalar_kind, query, ix->dimensions, casted) ? casted : query;
        keys[q] = 0, distances[q] = signaling_nan();
        if (ix->size) {
            uint32_t member = search_for_one(&c, (uint32_t)ix->entry_slot, (int)ix->max_level, level ? (int)level - 1 : 0);
            keys[q] = uo_key(ix, member);
            distances[q] = measure(&c, member);
        }
        if (visited) visited[q] = c.iteration_cycles;
        if (computed) computed[q] = c.computed_distances;
    }
    free(casted);
    ctx_free(&c);
}

void uo_last_traversal_shape(double* out) { memcpy(out, stat_shape, sizeof(stat_shape)); }

size_t uo_merge_into(uint64_t* keys, float* distances, size_t old_count, size_t max_count, const uint64_t* new_keys,
                     const float* new_distances, size_t new_count) {
    /* index.hpp:2650-2670: std::lower_bound on distance ⇒ a later-merged equal goes BEFORE the earlier ones */
    size_t merged = old_count;
    for (size_t i = 0; i < new_count; ++i) {
        float d = new_distances[i];
        size_t lo = 0, hi = merged;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (distances[mid] < d) lo = mid + 1;
            else hi = mid;
        }
        if (lo == max_count) continue;
        size_t count_worse = merged - lo - (max_count == merged ? 1 : 0);
        memmove(keys + lo + 1, keys + lo, count_worse * sizeof(uint64_t));
        memmove(distances + lo + 1, distances + lo, count_worse * sizeof(float));
        keys[lo] = new_keys[i];
        distances[lo] = d;
        merged += merged != max_count;
    }
    return merged;
}
