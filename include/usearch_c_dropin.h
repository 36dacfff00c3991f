/**
 *  include/usearch_c_dropin.h — the reference's C99 ABI as exported by THIS repository's drop-in `libusearch_c.so`.
 *
 *  Same 38 symbols, signatures, enum values and error convention as `/root/reference/c/usearch.h:116-481`
 *  (implemented there by c/lib.cpp): a program or binding linked against the reference's `libusearch_c` (Go:
 *  golang/lib.go:29-30, C#: csharp/src/Cloud.Unum.USearch/NativeMethods.cs:16, plain C: c/test.c) links against this
 *  library unchanged. What differs is WHO does the work:
 *
 *    search        `usearch_search`, `usearch_filtered_search`, `usearch_exact_search` and the additive batched
 *                  `usearch_search_many` run on the MI355X through an HBM snapshot of the index (include/usearch_amd.h).
 *                  A filter callback is a host function: it is evaluated lazily, for the members the walk wants to admit —
 *                  about as many calls as the reference makes (index.hpp:4200-4205, 4236-4240; see "filtered search" below).
 *    construction  `usearch_add` stages the vector on the host (and, like the reference, fails without reserved room:
 *                  index.hpp:2812-2818) and links it on the device before it returns WHEN SEARCHES INTERLEAVE WITH ADDS (any
 *                  search arrived since the add before: a reader racing a writer finds a member the moment its `add` is back,
 *                  as in the reference); adds in a row with nobody searching are linked together by the next search / save —
 *                  everything in one batched build the first time, afterwards only the members added since (the graph is
 *                  extended in place, the batch-deferred form of index.hpp:2780-2879). USEARCH_AMD_IMMEDIATE_ADD = 1 / 0
 *                  forces either behaviour. `usearch_remove` writes the tombstone and `usearch_rename` the new key
 *                  in place in HBM; only `usearch_change_metric_kind` costs a rebuild. Limits of the device builder are said
 *                  at `usearch_init`: connectivity ≤ 64 (base layer ≤ 128), expansion_add ≤ 1024.
 *    concurrency   searches share the index (one engine workspace per call in flight, `usearch_change_threads_search` sizes
 *                  the pool); mutations and the deferred linking take it alone.
 *    persistence   `usearch_save*` writes and `usearch_load* / view*` read the reference's v2 format: files move freely
 *                  between the two libraries.
 *    refused       by name, never computed elsewhere: a user-defined metric function (`usearch_change_metric`,
 *                  `usearch_init_options_t::metric`); a metric / scalar pair outside the reference's own dispatch table
 *                  (index_plugins.hpp:1930-2008 — e.g. haversine over half floats, divergence over i8): every pair inside it
 *                  has a kernel.
 *  Nothing is forwarded to the reference and there is no CPU search or build path: without a HIP device every call that
 *  needs one fails with an error string.
 *
 *  Errors: the callee stores a pointer to a static NUL-terminated string in `*error` on failure and leaves it untouched
 *  on success (c/usearch.h:24-28).
 */
#ifndef USEARCH_AMD_DROPIN_USEARCH_H
#define USEARCH_AMD_DROPIN_USEARCH_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef USEARCH_EXPORT
#define USEARCH_EXPORT __attribute__((visibility("default")))
#endif

/* The 38 entry points and their types are the reference's own ABI. A translation unit that also includes the reference's
 * `c/usearch.h` (its `c/lib.cpp` does, after `include/usearch/index_dense.hpp` pulled this header in) must see them once:
 * whichever of the two headers comes first declares them, under the reference's own include guard. */
#ifndef UNUM_USEARCH_H
#define UNUM_USEARCH_H

typedef void* usearch_index_t;       /* c/usearch.h:20 */
typedef uint64_t usearch_key_t;      /* c/usearch.h:21 */
typedef float usearch_distance_t;    /* c/usearch.h:22 */
typedef char const* usearch_error_t; /* c/usearch.h:28 */
typedef usearch_distance_t (*usearch_metric_t)(void const*, void const*); /* c/usearch.h:34 */

typedef enum usearch_metric_kind_t { /* c/usearch.h:40-52 */
    usearch_metric_unknown_k = 0,
    usearch_metric_cos_k = 1,
    usearch_metric_ip_k = 2,
    usearch_metric_l2sq_k = 3,
    usearch_metric_haversine_k = 4,
    usearch_metric_divergence_k = 5,
    usearch_metric_pearson_k = 6,
    usearch_metric_jaccard_k = 7,
    usearch_metric_hamming_k = 8,
    usearch_metric_tanimoto_k = 9,
    usearch_metric_sorensen_k = 10,
} usearch_metric_kind_t;

typedef enum usearch_scalar_kind_t { /* c/usearch.h:54-62 */
    usearch_scalar_unknown_k = 0,
    usearch_scalar_f32_k = 1,
    usearch_scalar_f64_k = 2,
    usearch_scalar_f16_k = 3,
    usearch_scalar_i8_k = 4,
    usearch_scalar_b1_k = 5,
    usearch_scalar_bf16_k = 6,
} usearch_scalar_kind_t;

typedef struct usearch_init_options_t { /* c/usearch.h:64-110, same member order */
    usearch_metric_kind_t metric_kind;
    usearch_metric_t metric;
    usearch_scalar_kind_t quantization;
    size_t dimensions;
    size_t connectivity;
    size_t expansion_add;
    size_t expansion_search;
    bool multi;
} usearch_init_options_t;

/* ---- the 38 reference entry points (c/usearch.h line of each in the comment) ---- */
USEARCH_EXPORT char const* usearch_version(void);                                                             /* 116 */
USEARCH_EXPORT usearch_index_t usearch_init(usearch_init_options_t* options, usearch_error_t* error);         /* 124 */
USEARCH_EXPORT void usearch_free(usearch_index_t index, usearch_error_t* error);                              /* 131 */
USEARCH_EXPORT size_t usearch_memory_usage(usearch_index_t index, usearch_error_t* error);                    /* 139 */
USEARCH_EXPORT char const* usearch_hardware_acceleration(usearch_index_t index, usearch_error_t* error);      /* 147 */
USEARCH_EXPORT size_t usearch_serialized_length(usearch_index_t index, usearch_error_t* error);               /* 154 */
USEARCH_EXPORT void usearch_save(usearch_index_t index, char const* path, usearch_error_t* error);            /* 162 */
USEARCH_EXPORT void usearch_load(usearch_index_t index, char const* path, usearch_error_t* error);            /* 170 */
USEARCH_EXPORT void usearch_view(usearch_index_t index, char const* path, usearch_error_t* error);            /* 178 */
USEARCH_EXPORT void usearch_metadata(char const* path, usearch_init_options_t* options, usearch_error_t* error); /* 186 */
USEARCH_EXPORT void usearch_save_buffer(usearch_index_t index, void* buffer, size_t length, usearch_error_t* error); /* 195 */
USEARCH_EXPORT void usearch_load_buffer(usearch_index_t index, void const* buffer, size_t length, usearch_error_t* error); /* 204 */
USEARCH_EXPORT void usearch_view_buffer(usearch_index_t index, void const* buffer, size_t length, usearch_error_t* error); /* 214 */
USEARCH_EXPORT void usearch_metadata_buffer(void const* buffer, size_t length, usearch_init_options_t* options,
                                            usearch_error_t* error);                                          /* 223 */
USEARCH_EXPORT size_t usearch_size(usearch_index_t index, usearch_error_t* error);                            /* 231 */
USEARCH_EXPORT size_t usearch_capacity(usearch_index_t index, usearch_error_t* error);                        /* 238 */
USEARCH_EXPORT size_t usearch_dimensions(usearch_index_t index, usearch_error_t* error);                      /* 245 */
USEARCH_EXPORT size_t usearch_connectivity(usearch_index_t index, usearch_error_t* error);                    /* 252 */
USEARCH_EXPORT void usearch_reserve(usearch_index_t index, size_t capacity, usearch_error_t* error);          /* 260 */
USEARCH_EXPORT size_t usearch_expansion_add(usearch_index_t index, usearch_error_t* error);                   /* 268 */
USEARCH_EXPORT size_t usearch_expansion_search(usearch_index_t index, usearch_error_t* error);                /* 276 */
USEARCH_EXPORT void usearch_change_expansion_add(usearch_index_t index, size_t expansion, usearch_error_t* error);    /* 284 */
USEARCH_EXPORT void usearch_change_expansion_search(usearch_index_t index, size_t expansion, usearch_error_t* error); /* 292 */
USEARCH_EXPORT void usearch_change_threads_add(usearch_index_t index, size_t threads, usearch_error_t* error);        /* 300 */
USEARCH_EXPORT void usearch_change_threads_search(usearch_index_t index, size_t threads, usearch_error_t* error);     /* 308 */
USEARCH_EXPORT void usearch_change_metric_kind(usearch_index_t index, usearch_metric_kind_t kind, usearch_error_t* error); /* 316 */
USEARCH_EXPORT void usearch_change_metric(usearch_index_t index, usearch_metric_t metric, void* state,
                                          usearch_metric_kind_t kind, usearch_error_t* error);                /* 327 */
USEARCH_EXPORT void usearch_add(usearch_index_t index, usearch_key_t key, void const* vector,
                                usearch_scalar_kind_t vector_kind, usearch_error_t* error);                   /* 338 */
USEARCH_EXPORT bool usearch_contains(usearch_index_t index, usearch_key_t key, usearch_error_t* error);       /* 349 */
USEARCH_EXPORT size_t usearch_count(usearch_index_t index, usearch_key_t key, usearch_error_t* error);        /* 358 */
USEARCH_EXPORT size_t usearch_search(usearch_index_t index, void const* query, usearch_scalar_kind_t query_kind,
                                     size_t count, usearch_key_t* keys, usearch_distance_t* distances,
                                     usearch_error_t* error);                                                 /* 371 */
USEARCH_EXPORT size_t usearch_filtered_search(usearch_index_t index, void const* query,
                                              usearch_scalar_kind_t query_kind, size_t count,
                                              int (*filter)(usearch_key_t key, void* filter_state), void* filter_state,
                                              usearch_key_t* keys, usearch_distance_t* distances,
                                              usearch_error_t* error);                                        /* 391 */
USEARCH_EXPORT size_t usearch_get(usearch_index_t index, usearch_key_t key, size_t count, void* vector,
                                  usearch_scalar_kind_t vector_kind, usearch_error_t* error);                 /* 407 */
USEARCH_EXPORT size_t usearch_remove(usearch_index_t index, usearch_key_t key, usearch_error_t* error);       /* 418 */
USEARCH_EXPORT size_t usearch_rename(usearch_index_t index, usearch_key_t from, usearch_key_t to,
                                     usearch_error_t* error);                                                 /* 428 */
USEARCH_EXPORT usearch_distance_t usearch_distance(void const* vector_first, void const* vector_second,
                                                   usearch_scalar_kind_t scalar_kind, size_t dimensions,
                                                   usearch_metric_kind_t metric_kind, usearch_error_t* error); /* 441 */
USEARCH_EXPORT void usearch_exact_search(void const* dataset, size_t dataset_size, size_t dataset_stride,
                                         void const* queries, size_t queries_size, size_t queries_stride,
                                         usearch_scalar_kind_t scalar_kind, size_t dimensions,
                                         usearch_metric_kind_t metric_kind, size_t count, size_t threads,
                                         usearch_key_t* keys, size_t keys_stride, usearch_distance_t* distances,
                                         size_t distances_stride, usearch_error_t* error);                    /* 467 */
USEARCH_EXPORT void usearch_clear(usearch_index_t index, usearch_error_t* error);                             /* 481 */

#endif /* UNUM_USEARCH_H */

/* ---- additive, MI355X-specific ---- */

/**
 *  `usearch_search` for `queries_count` queries in one call — the batch the reference leaves to its callers
 *  (cpp/bench.cpp:352-377, python/lib.cpp:261-319). Strides in bytes; `keys`/`distances` rows hold exactly `count`
 *  cells (unused: key 0 / signalling NaN); `counts`, `visited_members`, `computed_distances` may be NULL, the last two
 *  receive the batch totals of `search_result_t::{visited_members, computed_distances}` (index.hpp:3071-3072).
 */
USEARCH_EXPORT void usearch_search_many(usearch_index_t index, void const* queries, usearch_scalar_kind_t query_kind,
                                        size_t queries_count, size_t queries_stride, size_t count,
                                        usearch_key_t* keys, size_t keys_stride, usearch_distance_t* distances,
                                        size_t distances_stride, size_t* counts, size_t* visited_members,
                                        size_t* computed_distances, usearch_error_t* error);
/**
 *  `index_dense_gt::cluster(query, level)` (index_dense.hpp:788-793 → index_gt::cluster, index.hpp:3089-3125) for a batch:
 *  per query the member the greedy descent reaches on `level` of the hierarchy and its distance — the C ABI of the reference
 *  has no entry point for it, its C++ class does. `keys` and `distances` hold one cell per query.
 */
USEARCH_EXPORT void usearch_cluster_many(usearch_index_t index, void const* queries, usearch_scalar_kind_t query_kind,
                                         size_t queries_count, size_t queries_stride, size_t level,
                                         usearch_key_t* keys, usearch_distance_t* distances, usearch_error_t* error);
/**
 *  `index_dense_gt::search(query, count, thread, exact = true)` (index_dense.hpp:767-772 → `search_exact_`, index.hpp:4252-4268)
 *  for a batch: brute force over every member of the index, top-`count` under (distance, slot descending). The reference's C ABI
 *  only offers exact search over raw datasets (`usearch_exact_search`); its C++ class offers this one. Layout as
 *  `usearch_search_many`.
 */
USEARCH_EXPORT void usearch_search_exact_many(usearch_index_t index, void const* queries, usearch_scalar_kind_t query_kind,
                                              size_t queries_count, size_t queries_stride, size_t count, usearch_key_t* keys,
                                              size_t keys_stride, usearch_distance_t* distances, size_t distances_stride,
                                              size_t* counts, usearch_error_t* error);
/* ---- filtered search without a callback per member ------------------------------------------------------------------------
 *  `usearch_filtered_search` (c/usearch.h:391-395 → c/lib.cpp:413-429 → index_dense.hpp:774-779, 2071-2084) takes a host callback.
 *  The reference runs it inside the traversal, for the members it is about to admit to the result buffer — a few hundred to a few
 *  thousand calls per query (index.hpp:4200-4205, 4236-4240). A host function cannot run on the device; this library evaluates it
 *  LAZILY (round 6, csrc/dropin.hip `lazy_predicate_t`): the walk runs with two bits per slot in HBM — "the host has answered for
 *  this member" and its answer — admits a member it has no answer for yet with the share of "yes" among the answers so far while posting its key to an ask
 *  list; the host answers what was asked and the query runs again, until a run asks nothing. That run has seen the true predicate
 *  wherever it looked: it is the reference's traversal (same keys, distances, counters). Every member is asked about at most once
 *  per call, and about as many as the reference asks about (tests/test_gpu_dropin.py: no more than twice its count at
 *  selectivities 1 … 1/20). A predicate that rejects nearly everything keeps pushing the walk outward: after 48 runs (or a run that asks about more than 65 536 members) the rest is
 *  evaluated for every member (`USEARCH_AMD_FILTER_LAZY=0`, read at `usearch_init`, does that from the start: one callback per
 *  member per call, the behaviour up to round 5). A caller that searches more than once under one predicate, or that can say what
 *  the predicate IS, still does better with a filter made once: a bitmap (built by a kernel over the keys in HBM for ranges and key
 *  sets) that any number of batches reuse. A filter describes the index as it was when the filter was made: after `usearch_add /
 *  remove / rename / load / view / clear / usearch_gpu_release` searches under it fail with "The index changed since the filter was
 *  made".
 *
 *  What the callback must be: a PURE function of the key for the duration of the call (a provisional run may show it a member the
 *  reference's walk would not have; it must not call back into the index). For binaries that cannot be changed and whose predicate
 *  stays the same between calls, `USEARCH_AMD_FILTER_MEMO=1` in the environment (read at `usearch_init`) evaluates it for every
 *  member once and keeps the bitmap of the last eight (callback, filter_state pointer) pairs per index version: a repeated
 *  `usearch_filtered_search` under the same pair then makes no callback at all. That asks for MORE than purity per call — the
 *  predicate must not change while its state pointer stays the same — hence opt-in. Every mutation of the index forgets the
 *  remembered bitmaps.
 * ---------------------------------------------------------------------------------------------------------------------------- */
typedef void* usearch_filter_t;

/** Members whose key lies in [first_key, last_key], both ends included. */
USEARCH_EXPORT usearch_filter_t usearch_filter_from_key_range(usearch_index_t index, usearch_key_t first_key, usearch_key_t last_key,
                                                             usearch_error_t* error);
/** Members whose key is among `keys[0 .. keys_count)` (`allow`) or is not among them (`!allow`: a deny list). */
USEARCH_EXPORT usearch_filter_t usearch_filter_from_keys(usearch_index_t index, usearch_key_t const* keys, size_t keys_count,
                                                        bool allow, usearch_error_t* error);
/** The callback of `usearch_filtered_search`, evaluated over every member NOW, once. */
USEARCH_EXPORT usearch_filter_t usearch_filter_from_callback(usearch_index_t index, int (*filter)(usearch_key_t key, void* filter_state),
                                                            void* filter_state, usearch_error_t* error);
/** Members that pass. */
USEARCH_EXPORT size_t usearch_filter_allowed(usearch_filter_t filter, usearch_error_t* error);
USEARCH_EXPORT void usearch_filter_free(usearch_filter_t filter, usearch_error_t* error);
/** `usearch_filtered_search` for a batch under a made filter; layout as `usearch_search_many`. */
USEARCH_EXPORT void usearch_filtered_search_many(usearch_index_t index, usearch_filter_t filter, void const* queries,
                                                 usearch_scalar_kind_t query_kind, size_t queries_count, size_t queries_stride,
                                                 size_t count, usearch_key_t* keys, size_t keys_stride, usearch_distance_t* distances,
                                                 size_t distances_stride, size_t* counts, size_t* visited_members,
                                                 size_t* computed_distances, usearch_error_t* error);
/** `index_dense_gt::filtered_search(query, count, predicate, thread, exact = true)` for a batch (index_dense.hpp:774-779 →
 *  `search_exact_` with the predicate, index.hpp:4252-4268): brute force over the members that pass. Layout as
 *  `usearch_search_exact_many`. */
USEARCH_EXPORT void usearch_filtered_search_exact_many(usearch_index_t index, usearch_filter_t filter, void const* queries,
                                                       usearch_scalar_kind_t query_kind, size_t queries_count, size_t queries_stride,
                                                       size_t count, usearch_key_t* keys, size_t keys_stride,
                                                       usearch_distance_t* distances, size_t distances_stride, size_t* counts,
                                                       usearch_error_t* error);
/** Threads `usearch_change_threads_add / _search` recorded (the reference's `index_limits_t`, index.hpp:1338-1357). */
USEARCH_EXPORT size_t usearch_threads_search(usearch_index_t index, usearch_error_t* error);
/** Takes (or refreshes) the HBM snapshot now instead of at the next search. */
USEARCH_EXPORT void usearch_gpu_sync(usearch_index_t index, usearch_error_t* error);
/** Drops the HBM snapshot (it is re-taken on demand). */
USEARCH_EXPORT void usearch_gpu_release(usearch_index_t index, usearch_error_t* error);

/**
 *  Every entry point above as one table of function pointers. For code that must call THIS library while defining functions
 *  of the same names itself: `include/usearch/index_dense.hpp` — the `unum::usearch::index_dense_gt` surface over this
 *  library — is what the reference's own `c/lib.cpp` is compiled against in the tests, and that file defines `usearch_*`.
 */
typedef struct usearch_amd_c_api_t {
    size_t entries; /**< number of pointers below (append-only) */
    char const* (*version)(void);
    usearch_index_t (*init)(usearch_init_options_t*, usearch_error_t*);
    void (*free)(usearch_index_t, usearch_error_t*);
    size_t (*memory_usage)(usearch_index_t, usearch_error_t*);
    char const* (*hardware_acceleration)(usearch_index_t, usearch_error_t*);
    size_t (*serialized_length)(usearch_index_t, usearch_error_t*);
    void (*save)(usearch_index_t, char const*, usearch_error_t*);
    void (*load)(usearch_index_t, char const*, usearch_error_t*);
    void (*view)(usearch_index_t, char const*, usearch_error_t*);
    void (*metadata)(char const*, usearch_init_options_t*, usearch_error_t*);
    void (*save_buffer)(usearch_index_t, void*, size_t, usearch_error_t*);
    void (*load_buffer)(usearch_index_t, void const*, size_t, usearch_error_t*);
    void (*view_buffer)(usearch_index_t, void const*, size_t, usearch_error_t*);
    void (*metadata_buffer)(void const*, size_t, usearch_init_options_t*, usearch_error_t*);
    size_t (*size)(usearch_index_t, usearch_error_t*);
    size_t (*capacity)(usearch_index_t, usearch_error_t*);
    size_t (*dimensions)(usearch_index_t, usearch_error_t*);
    size_t (*connectivity)(usearch_index_t, usearch_error_t*);
    void (*reserve)(usearch_index_t, size_t, usearch_error_t*);
    size_t (*expansion_add)(usearch_index_t, usearch_error_t*);
    size_t (*expansion_search)(usearch_index_t, usearch_error_t*);
    void (*change_expansion_add)(usearch_index_t, size_t, usearch_error_t*);
    void (*change_expansion_search)(usearch_index_t, size_t, usearch_error_t*);
    void (*change_threads_add)(usearch_index_t, size_t, usearch_error_t*);
    void (*change_threads_search)(usearch_index_t, size_t, usearch_error_t*);
    void (*change_metric_kind)(usearch_index_t, usearch_metric_kind_t, usearch_error_t*);
    void (*change_metric)(usearch_index_t, usearch_metric_t, void*, usearch_metric_kind_t, usearch_error_t*);
    void (*add)(usearch_index_t, usearch_key_t, void const*, usearch_scalar_kind_t, usearch_error_t*);
    bool (*contains)(usearch_index_t, usearch_key_t, usearch_error_t*);
    size_t (*count)(usearch_index_t, usearch_key_t, usearch_error_t*);
    size_t (*search)(usearch_index_t, void const*, usearch_scalar_kind_t, size_t, usearch_key_t*, usearch_distance_t*,
                     usearch_error_t*);
    size_t (*filtered_search)(usearch_index_t, void const*, usearch_scalar_kind_t, size_t, int (*)(usearch_key_t, void*), void*,
                              usearch_key_t*, usearch_distance_t*, usearch_error_t*);
    size_t (*get)(usearch_index_t, usearch_key_t, size_t, void*, usearch_scalar_kind_t, usearch_error_t*);
    size_t (*remove)(usearch_index_t, usearch_key_t, usearch_error_t*);
    size_t (*rename)(usearch_index_t, usearch_key_t, usearch_key_t, usearch_error_t*);
    usearch_distance_t (*distance)(void const*, void const*, usearch_scalar_kind_t, size_t, usearch_metric_kind_t,
                                   usearch_error_t*);
    void (*exact_search)(void const*, size_t, size_t, void const*, size_t, size_t, usearch_scalar_kind_t, size_t,
                         usearch_metric_kind_t, size_t, size_t, usearch_key_t*, size_t, usearch_distance_t*, size_t,
                         usearch_error_t*);
    void (*clear)(usearch_index_t, usearch_error_t*);
    void (*search_many)(usearch_index_t, void const*, usearch_scalar_kind_t, size_t, size_t, size_t, usearch_key_t*, size_t,
                        usearch_distance_t*, size_t, size_t*, size_t*, size_t*, usearch_error_t*);
    void (*cluster_many)(usearch_index_t, void const*, usearch_scalar_kind_t, size_t, size_t, size_t, usearch_key_t*,
                         usearch_distance_t*, usearch_error_t*);
    void (*search_exact_many)(usearch_index_t, void const*, usearch_scalar_kind_t, size_t, size_t, size_t, usearch_key_t*, size_t,
                              usearch_distance_t*, size_t, size_t*, usearch_error_t*);
    size_t (*threads_search)(usearch_index_t, usearch_error_t*);
    void (*gpu_sync)(usearch_index_t, usearch_error_t*);
    void (*gpu_release)(usearch_index_t, usearch_error_t*);
    usearch_filter_t (*filter_from_key_range)(usearch_index_t, usearch_key_t, usearch_key_t, usearch_error_t*);
    usearch_filter_t (*filter_from_keys)(usearch_index_t, usearch_key_t const*, size_t, bool, usearch_error_t*);
    usearch_filter_t (*filter_from_callback)(usearch_index_t, int (*)(usearch_key_t, void*), void*, usearch_error_t*);
    size_t (*filter_allowed)(usearch_filter_t, usearch_error_t*);
    void (*filter_free)(usearch_filter_t, usearch_error_t*);
    void (*filtered_search_many)(usearch_index_t, usearch_filter_t, void const*, usearch_scalar_kind_t, size_t, size_t, size_t,
                                 usearch_key_t*, size_t, usearch_distance_t*, size_t, size_t*, size_t*, size_t*, usearch_error_t*);
    void (*filtered_search_exact_many)(usearch_index_t, usearch_filter_t, void const*, usearch_scalar_kind_t, size_t, size_t, size_t,
                                       usearch_key_t*, size_t, usearch_distance_t*, size_t, size_t*, usearch_error_t*);
} usearch_amd_c_api_t;
USEARCH_EXPORT usearch_amd_c_api_t const* usearch_amd_c_api(void);

#ifdef __cplusplus
}
#endif
#endif
