/**
 *  usearch_amd/csrc/dropin.hip — the reference's C99 ABI (`/root/reference/c/usearch.h`, implemented there by c/lib.cpp)
 *  on top of the MI355X engine: `usearch_amd/lib/libusearch_c.so`, declared in include/usearch_c_dropin.h.
 *
 *  Every one of the 38 entry points is implemented natively — nothing is forwarded to the reference and nothing runs on
 *  the CPU that the reference would run as its hot path:
 *    search            `usearch_search`, `_filtered_search`, `_exact_search`, `usearch_search_many` → the search kernels
 *    construction      `usearch_add` stages the vector on the host; the next search / save / size-dependent call links ALL
 *                      staged vectors on the device (build.hip). Bulk-load-then-search is the intended pattern; every
 *                      mutation after a build costs a rebuild at the next search.
 *    persistence       `usearch_save*` writes, `usearch_load* / view*` read the reference's v2 format (docs/format.md)
 *    `usearch_distance` one launch of the exact-search kernel over a 1-row dataset
 *  What the device cannot do is refused by name: a user-defined metric function (`usearch_change_metric`,
 *  `usearch_init_options_t::metric`), metrics / scalar kinds without a kernel.
 *
 *  The host keeps what the reference keeps in `index_dense_gt`: keys and vectors per slot (for `get`, `contains`,
 *  `count`, `rename`, `remove`), either inside a serialized image (after load / view) or in staging arrays (after add).
 */
#include "../../include/usearch_c_dropin.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

#include "build.hpp"
#include "casts.hpp"
#include "combiner.hpp"
#include "engine.hpp"
#include "filter.hpp"
#include "host_util.hpp"

using namespace usearch_amd;

namespace {

metric_kind_t metric_from_c(usearch_metric_kind_t kind) { // c/lib.cpp:26-42
    switch (kind) {
    case usearch_metric_cos_k: return metric_cos_k;
    case usearch_metric_ip_k: return metric_ip_k;
    case usearch_metric_l2sq_k: return metric_l2sq_k;
    case usearch_metric_haversine_k: return metric_haversine_k;
    case usearch_metric_divergence_k: return metric_divergence_k;
    case usearch_metric_pearson_k: return metric_pearson_k;
    case usearch_metric_jaccard_k: return metric_jaccard_k;
    case usearch_metric_hamming_k: return metric_hamming_k;
    case usearch_metric_tanimoto_k: return metric_tanimoto_k;
    case usearch_metric_sorensen_k: return metric_sorensen_k;
    default: return metric_unknown_k;
    }
}
usearch_metric_kind_t metric_to_c(metric_kind_t kind) { // c/lib.cpp:44-59
    switch (kind) {
    case metric_cos_k: return usearch_metric_cos_k;
    case metric_ip_k: return usearch_metric_ip_k;
    case metric_l2sq_k: return usearch_metric_l2sq_k;
    case metric_haversine_k: return usearch_metric_haversine_k;
    case metric_divergence_k: return usearch_metric_divergence_k;
    case metric_pearson_k: return usearch_metric_pearson_k;
    case metric_jaccard_k: return usearch_metric_jaccard_k;
    case metric_hamming_k: return usearch_metric_hamming_k;
    case metric_tanimoto_k: return usearch_metric_tanimoto_k;
    case metric_sorensen_k: return usearch_metric_sorensen_k;
    default: return usearch_metric_unknown_k;
    }
}
scalar_kind_t scalar_from_c(usearch_scalar_kind_t kind) { // c/lib.cpp:61-71
    switch (kind) {
    case usearch_scalar_f32_k: return scalar_f32_k;
    case usearch_scalar_f64_k: return scalar_f64_k;
    case usearch_scalar_f16_k: return scalar_f16_k;
    case usearch_scalar_i8_k: return scalar_i8_k;
    case usearch_scalar_b1_k: return scalar_b1x8_k;
    case usearch_scalar_bf16_k: return scalar_bf16_k;
    default: return scalar_unknown_k;
    }
}
usearch_scalar_kind_t scalar_to_c(scalar_kind_t kind) { // c/lib.cpp:73-83
    switch (kind) {
    case scalar_f32_k: return usearch_scalar_f32_k;
    case scalar_f64_k: return usearch_scalar_f64_k;
    case scalar_f16_k: return usearch_scalar_f16_k;
    case scalar_i8_k: return usearch_scalar_i8_k;
    case scalar_b1x8_k: return usearch_scalar_b1_k;
    case scalar_bf16_k: return usearch_scalar_bf16_k;
    default: return usearch_scalar_unknown_k;
    }
}

void fail(usearch_error_t* error, const char* message) {
    if (error && message)
        *error = message;
}

/// Runs `body`; an allocation failure or any other exception becomes an error string instead of crossing the C ABI.
template <typename body_at> void guarded(usearch_error_t* error, body_at&& body) {
    try {
        body();
    } catch (const std::bad_alloc&) {
        fail(error, "Out of memory!");
    } catch (const std::exception&) {
        fail(error, "Unexpected failure inside the index");
    }
}

/// Same for an entry point that returns something: `fallback` is what the caller gets next to the error string.
template <typename result_at, typename body_at> result_at guarded(usearch_error_t* error, result_at fallback, body_at&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        fail(error, "Out of memory!");
    } catch (const std::exception&) {
        fail(error, "Unexpected failure inside the index");
    }
    return fallback;
}

/// Callers of this library loop single queries from many threads (Go routines, C# tasks: the reference leases a context per thread,
/// index_dense.hpp:1984-2000). Every call is a launch on its own stream, and the HIP runtime multiplexes streams onto FOUR hardware
/// queues unless told otherwise — kernels sharing a queue run one after the other. `GPU_MAX_HW_QUEUES=16` in the process
/// environment lifts that (16 callers on the 10M×768 f16 index, expansion 64: 11.4 k → 23.8 k calls per second,
/// scripts/threads_check.py). The library does not touch the environment of its host process on its own: a
/// process that cannot set the variable itself opts in with USEARCH_AMD_HW_QUEUES=n, which is copied over at load time (before
/// the HIP runtime starts; a GPU_MAX_HW_QUEUES already set wins).
struct hardware_queues_t {
    hardware_queues_t() {
        if (const char* wanted = std::getenv("USEARCH_AMD_HW_QUEUES"))
            if (*wanted)
                (void)setenv("GPU_MAX_HW_QUEUES", wanted, /*overwrite=*/0);
    }
} hardware_queues;

using shared_lock_t = std::shared_lock<std::shared_mutex>;
using unique_lock_t = std::unique_lock<std::shared_mutex>;

/// One `usearch_index_t`.
struct index_t {
    /// Searches share it (the engine leases one workspace per call in flight, as the reference leases one context per thread,
    /// index_dense.hpp:1984-2000); everything that changes the index, or links pending members, takes it alone.
    std::shared_mutex mutex;
    std::size_t threads_search = 0; ///< `usearch_change_threads_search`: batches in flight at once (0 = the engine's default)
    /// `usearch_search` calls in flight share a launch (combiner.hpp): whoever finds nobody launching takes every compatible call
    /// that is waiting and runs them as one batch, after giving the callers of the launch before up to `USEARCH_AMD_COALESCE_WINDOW_US`
    /// (200) microseconds — never more than an eighth of that launch, nothing for a lone caller — to call again: looping callers then go
    /// out in one launch instead of two alternating halves. A lone caller is untouched (2.7 ms / 0.35 ms per call at ef 608 / 64 on the
    /// headline index either way); 16 native callers at ef 608: 1 640 → 5 530 calls per second (9.8 → 2.9 ms per call), 64 callers:
    /// 2 068 → 20 546 (30.9 → 3.1 ms), 64 callers at ef 64: 9.1 k → 80.3 k (profiles/r04_single_query/). `USEARCH_AMD_COALESCE=0`
    /// (read when the index is created) turns it off.
    bool coalesce = env_size("USEARCH_AMD_COALESCE", 1) != 0;
    combiner_t combiner;
    /// `usearch_filtered_search` under USEARCH_AMD_FILTER_MEMO=1 (read at `usearch_init`): the bitmap a callback produced is kept per
    /// (callback, state pointer, index version) and the next call with the same three makes NO callbacks and uploads nothing. Opt-in,
    /// because it needs more than the ABI promises: the predicate must be a pure function of the key for as long as the state
    /// POINTER stays the same (the reference calls it afresh, a few thousand times per query: c/lib.cpp:413-429, index.hpp:4200-4205).
    bool filter_memo = env_size("USEARCH_AMD_FILTER_MEMO", 0) != 0;
    /// `usearch_filtered_search` evaluates its callback LAZILY (the default; USEARCH_AMD_FILTER_LAZY=0, read at `usearch_init`, goes
    /// back to one callback per member per call): only for the members the walk wants to admit to `top`, as the reference does
    /// (c/lib.cpp:413-429 → index.hpp:4200-4205, 4236-4240) — see `lazy_predicate_t`.
    bool filter_lazy = env_size("USEARCH_AMD_FILTER_LAZY", 1) != 0;
    /// When `usearch_add` links the member on the device (USEARCH_AMD_IMMEDIATE_ADD, read at `usearch_init`). The reference's `add`
    /// links before it returns (index.hpp:2780-2879): a reader racing a writer finds the member the moment `add` is back.
    ///   unset / "auto": an `add` links at once WHEN SEARCHES INTERLEAVE WITH ADDS — any search arrived on this index since the add
    ///                   before (or is waiting for the lock right now) — so a host that mixes readers and writers sees the
    ///                   reference's visibility without any switch; a bulk loader (adds in a row, nobody searching) keeps the
    ///                   batch-deferred form, thousands of members per launch (what the device is for);
    ///   "1": every add links at once;   "0": never (the next search / save links everything pending).
    int immediate_add = [] {
        const char* text = std::getenv("USEARCH_AMD_IMMEDIATE_ADD");
        return !text || !*text || !std::strcmp(text, "auto") ? 2 : std::atoi(text) != 0 ? 1 : 0;
    }();
    std::atomic<std::uint64_t> searches_arrived{0}; ///< bumped by every search BEFORE it asks for the lock
    std::uint64_t searches_at_last_add = 0;         ///< … as of the add before (under the unique lock)
    struct filter_memo_t {
        int (*filter)(usearch_key_t, void*) = nullptr;
        void* state = nullptr;
        std::uint64_t version = 0;
        std::shared_ptr<filter_t> bitmap;
    };
    std::mutex memo_mutex;
    std::vector<filter_memo_t> memos; ///< most recent first, at most `memo_limit_k`
    static constexpr std::size_t memo_limit_k = 8;
    // configuration — `usearch_init_options_t`, c/usearch.h:64-110
    metric_kind_t metric = metric_cos_k;
    scalar_kind_t scalar = scalar_f32_k;
    std::size_t dimensions = 0, connectivity = 16, expansion_add = 128, expansion_search = 64, capacity = 0;
    bool multi = false;
    int device = 0;

    // (1) a serialized image: owned bytes, a borrowed buffer (`view_buffer`) or a mapped file (`view`)
    std::vector<std::uint8_t> image_owned;
    const std::uint8_t* image_bytes = nullptr;
    std::size_t image_length = 0;
    void* mapping = nullptr;
    std::size_t mapping_length = 0;
    image_t image;      ///< parsed header of `image_bytes` when `has_image`
    bool has_image = false;
    snapshot_t* snapshot = nullptr; ///< HBM copy of the image, taken at the first search

    // (2) staging: keys and vectors per slot once anything was added / removed / renamed since the image
    bool staged = false;
    std::vector<std::uint64_t> keys;
    std::vector<std::uint8_t> vectors; ///< storage scalar kind, `bytes_per_vector` per slot
    builder_t* builder = nullptr;      ///< device index linked from the staging arrays (null = stale)
    /// Slots `usearch_remove` freed, oldest first — the reference's `free_keys_` ring (index_dense.hpp:507, 1479-1511): the next
    /// `usearch_add` takes the oldest instead of a new slot, and the member is linked anew in place (`builder_t::update`).
    std::vector<std::uint32_t> free_slots;
    std::size_t free_head = 0;
    std::vector<std::uint32_t> recycled; ///< reused slots whose device copy still holds the member that was removed

    // key → slots, rebuilt lazily
    std::unordered_multimap<std::uint64_t, std::uint32_t> lookup;
    bool lookup_valid = false;

    /// Bumped by everything that changes the members or drops their HBM copy: a `usearch_filter_t` describes one version.
    std::uint64_t version = 0;

    std::size_t bpv() const { return bytes_per_vector(scalar, dimensions); }
    std::size_t size() const { return staged ? keys.size() : has_image ? (std::size_t)image.size : 0; }

    void drop_device() {
        ++version;
        memos.clear(); // the callers hold the index alone: no search is reading one
        delete builder, builder = nullptr;
        delete snapshot, snapshot = nullptr;
    }
    void drop_image() {
        if (mapping)
            ::munmap(mapping, mapping_length), mapping = nullptr, mapping_length = 0;
        image_owned.clear(), image_owned.shrink_to_fit();
        image_bytes = nullptr, image_length = 0, has_image = false;
        image = image_t{};
    }
    ~index_t() {
        drop_device();
        drop_image();
    }

    /// Adopts `bytes` as the current image. Configuration comes from its header (c/lib.cpp `usearch_load`: the index takes
    /// the file's metric, scalar kind and dimensions).
    const char* open_image(const void* bytes, std::size_t length) {
        image_t parsed;
        if (const char* e = parsed.open(bytes, length))
            return e;
        if (!kernel_available(parsed.metric, parsed.scalar))
            return "No MI355X kernel for this metric / scalar kind combination";
        image = parsed;
        image_bytes = static_cast<const std::uint8_t*>(bytes);
        image_length = length;
        has_image = true;
        metric = parsed.metric, scalar = parsed.scalar, dimensions = (std::size_t)parsed.dimensions;
        if (parsed.connectivity)
            connectivity = (std::size_t)parsed.connectivity;
        multi = parsed.multi;
        staged = false;
        keys.clear(), vectors.clear();
        free_slots.clear(), free_head = 0, recycled.clear();
        lookup_valid = false;
        capacity = std::max<std::size_t>(capacity, (std::size_t)parsed.size);
        return nullptr;
    }

    /// Key of every slot of the image, in slot order (the node tapes are variable-length: one sequential pass).
    void image_keys(std::vector<std::uint64_t>& out) const {
        out.resize((std::size_t)image.size);
        std::size_t offset = 0;
        for (std::uint64_t i = 0; i < image.size; ++i) {
            out[i] = image_t::load<std::uint64_t>(image.tapes + offset);
            offset += image.node_bytes(image.level(i));
        }
    }

    /// Moves keys and vectors out of the image into the staging arrays: the index is about to be mutated.
    void materialize() {
        if (staged)
            return;
        if (has_image) {
            image_keys(keys);
            vectors.assign(image.vectors, image.vectors + (std::size_t)image.size * image.cols);
        }
        free_slots.clear(), free_head = 0, recycled.clear();
        for (std::size_t slot = 0; slot < keys.size(); ++slot) // what `reindex_keys_` does after a load (index_dense.hpp:2162-2200)
            if (keys[slot] == free_key_k)
                free_slots.push_back((std::uint32_t)slot);
        staged = true;
        drop_device();
        drop_image();
        lookup_valid = false;
    }

    const std::unordered_multimap<std::uint64_t, std::uint32_t>& key_lookup() {
        if (!lookup_valid) {
            lookup.clear();
            std::vector<std::uint64_t> from_image;
            const std::vector<std::uint64_t>* source = &keys;
            if (!staged && has_image)
                image_keys(from_image), source = &from_image;
            lookup.reserve(source->size());
            for (std::size_t slot = 0; slot < source->size(); ++slot)
                if ((*source)[slot] != free_key_k)
                    lookup.emplace((*source)[slot], (std::uint32_t)slot);
            lookup_valid = true;
        }
        return lookup;
    }

    const std::uint8_t* vector_of(std::uint32_t slot) const {
        return staged ? vectors.data() + (std::size_t)slot * bpv() : image.vectors + (std::size_t)slot * image.cols;
    }

    /// Is the device index up to date with the host-side content? (Shared lock suffices to ask.)
    bool device_current() const {
        if (staged)
            return keys.empty() || (builder && builder->size() == keys.size() && recycled.empty());
        return !has_image || snapshot != nullptr;
    }
    snapshot_t* device_index() { return staged ? (builder ? &builder->snapshot() : nullptr) : snapshot; }

    /// Brings the device index up to date (unique lock): uploads the image, builds the staged members, or — after a build —
    /// links only the members added since (`builder_t::extend`, the batch-deferred form of index.hpp:2780-2879).
    const char* ready(snapshot_t** out) {
        *out = nullptr;
        if (staged) {
            if (!builder && !keys.empty()) {
                builder_t* fresh = new (std::nothrow) builder_t();
                if (!fresh)
                    return "Out of memory!";
                build_config_t config;
                config.connectivity = (std::uint32_t)connectivity;
                config.expansion_add = (std::uint32_t)expansion_add;
                config.multi = multi;
                if (const char* e = fresh->build(metric, scalar, dimensions, vectors.data(), keys.size(), bpv(), false,
                                                 keys.data(), config, device)) {
                    delete fresh;
                    return e;
                }
                builder = fresh;
                recycled.clear(); // a build from the staging arrays has every member as it is now
            } else if (builder && !recycled.empty()) {
                // slots a removal freed and an add took over: the members the device still holds there are replaced and linked anew
                // (index_gt::update); slots beyond what is linked are simply part of the members still to come
                std::vector<std::uint32_t> slots;
                std::vector<std::uint64_t> new_keys;
                std::vector<std::uint8_t> rows;
                std::sort(recycled.begin(), recycled.end());
                recycled.erase(std::unique(recycled.begin(), recycled.end()), recycled.end());
                for (std::uint32_t slot : recycled)
                    if (slot < builder->size() && keys[slot] != free_key_k) {
                        slots.push_back(slot);
                        new_keys.push_back(keys[slot]);
                        rows.insert(rows.end(), vectors.data() + (std::size_t)slot * bpv(), vectors.data() + (std::size_t)(slot + 1) * bpv());
                    }
                if (const char* e = builder->update(slots.data(), slots.size(), rows.data(), bpv(), new_keys.data())) {
                    delete builder, builder = nullptr;
                    return e;
                }
                recycled.clear();
            }
            if (builder && builder->size() < keys.size()) {
                const std::size_t linked = (std::size_t)builder->size();
                if (const char* e = builder->extend(vectors.data() + linked * bpv(), keys.size() - linked, bpv(), false,
                                                    keys.data() + linked)) {
                    delete builder, builder = nullptr; // the arrays may be half grown: start over at the next call
                    return e;
                }
            }
            *out = builder ? &builder->snapshot() : nullptr;
            if (*out && threads_search)
                (*out)->set_concurrency(threads_search);
            return nullptr;
        }
        if (has_image && !snapshot) {
            snapshot_t* fresh = new (std::nothrow) snapshot_t();
            if (!fresh)
                return "Out of memory!";
            if (const char* e = fresh->build(image, device)) {
                delete fresh;
                return e;
            }
            snapshot = fresh;
        }
        *out = snapshot;
        if (*out && threads_search)
            (*out)->set_concurrency(threads_search);
        return nullptr;
    }

};

index_t* as_index(usearch_index_t handle) { return static_cast<index_t*>(handle); }

/// A `usearch_filter_t`: the predicate as a bitmap in HBM, and the version of the index it describes.
struct made_filter_t {
    index_t* index = nullptr;
    std::uint64_t version = 0;
    std::unique_ptr<filter_t> bitmap; ///< null while the index has no members
};

/// The host callback over every member, as one bit per slot (what `usearch_filtered_search` has to do per call, and
/// `usearch_filter_from_callback` once). Any lock on the index will do.
static void callback_bits(index_t& index, int (*filter)(usearch_key_t key, void* filter_state), void* filter_state,
                          std::vector<std::uint32_t>& bits) {
    std::vector<std::uint64_t> from_image;
    const std::vector<std::uint64_t>* member_keys = &index.keys;
    if (!index.staged && index.has_image)
        index.image_keys(from_image), member_keys = &from_image;
    bits.assign((member_keys->size() + 31) / 32 + 1, 0);
    for (std::size_t slot = 0; slot < member_keys->size(); ++slot)
        if ((*member_keys)[slot] != free_key_k && filter((*member_keys)[slot], filter_state))
            bits[slot >> 5] |= 1u << (slot & 31);
}

/**
 *  The host callback of `usearch_filtered_search`, evaluated lazily. The reference calls the predicate inside the traversal, for the
 *  members it is about to admit to `top` — a few hundred to a few thousand per query (index.hpp:4200-4205, 4236-4240). A host
 *  function cannot run inside a kernel, and evaluating it for EVERY member per call (what `callback_bits` does: 10 M callbacks on
 *  the headline index) is not what an unchanged binary expects. So the device keeps two bits per slot — `known` (the host has
 *  answered for this member) and `allow` (its answer) — and the walk treats a member it wants to admit that is not known yet as
 *  allowed while posting (slot, key) to an ask list. The host answers what was asked and runs the query AGAIN; a run that asks
 *  nothing has seen the true predicate wherever it looked: it is the reference's traversal, bit for bit (keys, distances, both
 *  counters). Runs before it are provisional and discarded. Every member is asked about once at most; the runs converge because
 *  each one is exact up to the first member it had to guess, and quickly because the guess is informed: an unknown member is
 *  admitted with the share of "yes" among the answers so far (a fixed pseudo-random draw per slot), so a provisional walk fills its
 *  `top` about as fast as the true one and reaches about as far — three to six runs whatever the selectivity, where "unknown =
 *  allowed" needs ln(ef) / selectivity of them. After `rounds_limit_k` runs, or when a run asks more than the list holds (a
 *  predicate that rejects everything makes the walk visit every member, the reference's too), every member is evaluated, as before.
 */
struct lazy_predicate_t {
    static constexpr std::uint32_t ask_cap_k = 1u << 16;
    static constexpr int rounds_limit_k = 48;
    std::size_t allowed_so_far = 0;
    std::uint32_t *d_allow = nullptr, *d_known = nullptr, *d_ask_slots = nullptr, *d_cursor = nullptr;
    std::uint64_t* d_ask_keys = nullptr;
    std::vector<std::uint32_t> allow, known, ask_slots;
    std::vector<std::uint64_t> ask_keys;
    std::size_t words = 0;
    std::size_t callbacks = 0;

    ~lazy_predicate_t() {
        for (void* p : {(void*)d_allow, (void*)d_known, (void*)d_ask_slots, (void*)d_cursor, (void*)d_ask_keys})
            if (p)
                (void)hipFree(p);
    }
    const char* open(std::size_t members) {
        words = (members + 31) / 32 + 1;
        allow.assign(words, 0u), known.assign(words, 0u);
        if (hipMalloc((void**)&d_allow, words * 4) != hipSuccess || hipMalloc((void**)&d_known, words * 4) != hipSuccess ||
            hipMalloc((void**)&d_ask_slots, ask_cap_k * 4) != hipSuccess || hipMalloc((void**)&d_ask_keys, ask_cap_k * 8) != hipSuccess ||
            hipMalloc((void**)&d_cursor, 4) != hipSuccess || hipMemset(d_allow, 0, words * 4) != hipSuccess ||
            hipMemset(d_known, 0, words * 4) != hipSuccess)
            return (void)hipGetLastError(), "Out of device memory for the predicate's bitmaps";
        return nullptr;
    }
    void fill(search_extras_t& extras) const {
        // a member nobody has answered for yet is GUESSED by the run: admitted with the share of "yes" among the answers so far
        // (everything, before the first answer), so that a provisional walk reaches about as far as the true one
        const double share = callbacks ? std::max(1.0 / 4096, (double)allowed_so_far / (double)callbacks) : 1.0;
        extras.guess_threshold = share >= 1.0 ? 0xFFFFFFFFu : (std::uint32_t)(share * 4294967295.0);
        extras.allow_bits = d_allow, extras.known_bits = d_known;
        extras.ask_slots = d_ask_slots, extras.ask_keys = d_ask_keys, extras.ask_cursor = d_cursor, extras.ask_cap = ask_cap_k;
    }
    const char* before_run() { return hipMemset(d_cursor, 0, 4) == hipSuccess ? nullptr : "hipMemset failed"; }
    /// What the run asked: answers it on the host, uploads the two bitmaps. `*asked` = members newly answered (0: the run is final);
    /// `*overflow`: more was asked than the list holds.
    const char* after_run(int (*filter)(usearch_key_t, void*), void* state, std::size_t* asked, bool* overflow) {
        std::uint32_t cursor = 0;
        if (hipMemcpy(&cursor, d_cursor, 4, hipMemcpyDeviceToHost) != hipSuccess)
            return "Failed to read the ask list";
        *asked = 0, *overflow = cursor > ask_cap_k;
        if (!cursor)
            return nullptr;
        const std::uint32_t listed = std::min(cursor, ask_cap_k);
        ask_slots.resize(listed), ask_keys.resize(listed);
        if (hipMemcpy(ask_slots.data(), d_ask_slots, listed * 4ull, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ask_keys.data(), d_ask_keys, listed * 8ull, hipMemcpyDeviceToHost) != hipSuccess)
            return "Failed to read the ask list";
        for (std::uint32_t i = 0; i < listed; ++i) {
            const std::uint32_t slot = ask_slots[i], word = slot >> 5, bit = 1u << (slot & 31);
            if (word >= words || (known[word] & bit))
                continue; // two queries of a batch (or a re-run rung) asked about the same member
            known[word] |= bit;
            ++callbacks, ++*asked;
            if (filter(ask_keys[i], state))
                allow[word] |= bit, ++allowed_so_far;
        }
        if (hipMemcpy(d_allow, allow.data(), words * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_known, known.data(), words * 4, hipMemcpyHostToDevice) != hipSuccess)
            return "Failed to upload the predicate's bitmaps";
        return nullptr;
    }
};

/// Brings the device index up to date and wraps what `make` builds over it. `make` gets the snapshot (never null).
template <typename make_at> usearch_filter_t make_filter(usearch_index_t handle, usearch_error_t* error, make_at&& make) {
    index_t& index = *as_index(handle);
    return guarded(error, usearch_filter_t(nullptr), [&]() -> usearch_filter_t {
        unique_lock_t lock(index.mutex);
        snapshot_t* device_index = nullptr;
        if (const char* e = index.ready(&device_index))
            return fail(error, e), nullptr;
        std::unique_ptr<made_filter_t> made(new made_filter_t());
        made->index = &index;
        made->version = index.version;
        if (device_index)
            if (const char* e = make(index, *device_index, made->bitmap))
                return fail(error, e), nullptr;
        return made.release();
    });
}


/// An index without nodes still serializes to its headers (index_dense.hpp:995-1062 with zero rows).
void write_empty_image(const index_t& index, std::uint8_t* p) {
    std::memset(p, 0, 8 + 64 + 40);
    const std::uint32_t cols = (std::uint32_t)index.bpv();
    std::memcpy(p + 4, &cols, 4);
    p += 8;
    std::memcpy(p, "usearch", 7);
    const std::uint16_t version[3] = {2, 21, 0};
    std::memcpy(p + 7, version, 6);
    p[13] = (std::uint8_t)index.metric, p[14] = (std::uint8_t)index.scalar;
    p[15] = (std::uint8_t)scalar_u64_k, p[16] = (std::uint8_t)scalar_u32_k;
    const std::uint64_t dimensions = index.dimensions;
    std::memcpy(p + 33, &dimensions, 8);
    p[41] = index.multi ? 1 : 0;
    p += 64;
    const std::uint64_t header[5] = {0, index.connectivity, 2 * index.connectivity, 0, 0};
    std::memcpy(p, header, 40);
}

/// Serialized form of the current content: the image as it was loaded, or what the device build writes.
const char* serialize(index_t& index, std::vector<std::uint8_t>* into_vector, void* into_buffer, std::size_t buffer_length,
                      std::size_t* length_out) {
    if (!index.staged) {
        const std::size_t length = index.has_image ? index.image_length : 8 + 64 + 40;
        if (length_out)
            *length_out = length;
        std::uint8_t* target = nullptr;
        if (into_vector)
            into_vector->resize(length), target = into_vector->data();
        else if (into_buffer) {
            if (buffer_length < length)
                return "Buffer is too small";
            target = static_cast<std::uint8_t*>(into_buffer);
        }
        if (target) {
            if (index.has_image)
                std::memcpy(target, index.image_bytes, length);
            else
                write_empty_image(index, target);
        }
        return nullptr;
    }
    snapshot_t* device_index = nullptr;
    if (const char* e = index.ready(&device_index))
        return e;
    if (!index.builder) { // staged but empty
        if (length_out)
            *length_out = 8 + 64 + 40;
        std::uint8_t* target = nullptr;
        if (into_vector)
            into_vector->resize(8 + 64 + 40), target = into_vector->data();
        else if (into_buffer) {
            if (buffer_length < 8 + 64 + 40)
                return "Buffer is too small";
            target = static_cast<std::uint8_t*>(into_buffer);
        }
        if (target)
            write_empty_image(index, target);
        return nullptr;
    }
    const std::size_t length = index.builder->serialized_length();
    if (length_out)
        *length_out = length;
    if (into_vector) {
        into_vector->resize(length);
        return index.builder->save_buffer(into_vector->data(), length);
    }
    if (into_buffer) {
        if (buffer_length < length)
            return "Buffer is too small";
        return index.builder->save_buffer(into_buffer, length);
    }
    return nullptr;
}

void fill_options(const image_t& image, usearch_init_options_t* options) { // c/lib.cpp:224-242
    options->metric_kind = metric_to_c(image.metric);
    options->metric = nullptr;
    options->quantization = scalar_to_c(image.scalar);
    options->dimensions = (std::size_t)image.dimensions;
    options->connectivity = (std::size_t)image.connectivity;
    options->expansion_add = 0;
    options->expansion_search = 0;
    options->multi = image.multi;
}

/// `dump_to`-style padding for queries that cannot match anything.
void pad_results(usearch_key_t* keys, usearch_distance_t* distances, std::size_t count) {
    for (std::size_t i = 0; i < count; ++i) {
        if (keys)
            keys[i] = 0;
        if (distances)
            std::memcpy(distances + i, &signaling_nan_bits_k, 4);
    }
}

} // namespace

extern "C" {

char const* usearch_version(void) { return "2.21.0"; }

usearch_index_t usearch_init(usearch_init_options_t* options, usearch_error_t* error) {
    index_t* index = new (std::nothrow) index_t();
    if (!index) {
        fail(error, "Out of memory!");
        return nullptr;
    }
    index->combiner.window_limit(std::chrono::microseconds(env_size("USEARCH_AMD_COALESCE_WINDOW_US", 200)));
    if (!options) // an empty shell to `usearch_load` / `usearch_view` into, c/lib.cpp:142-147
        return index;
    if (options->metric) {
        fail(error, "User-defined metric functions cannot run on the device");
        delete index;
        return nullptr;
    }
    index->metric = metric_from_c(options->metric_kind);
    index->scalar = scalar_from_c(options->quantization);
    if (index->metric == metric_unknown_k || index->scalar == scalar_unknown_k) {
        fail(error, index->metric == metric_unknown_k ? "Unknown metric kind!" : "Unknown scalar kind!");
        delete index;
        return nullptr;
    }
    if (!kernel_available(index->metric, index->scalar)) {
        fail(error, "No MI355X kernel for this metric / scalar kind combination");
        delete index;
        return nullptr;
    }
    index->dimensions = options->dimensions;
    if (options->connectivity)
        index->connectivity = options->connectivity;
    if (options->expansion_add)
        index->expansion_add = options->expansion_add;
    if (options->expansion_search)
        index->expansion_search = options->expansion_search;
    index->multi = options->multi;
    // the device builder's limits, said here instead of at the first search (build.hip: a node's existing and incoming links
    // are ranked by one wave)
    if (2 * index->connectivity > builder_max_connectivity_base_k || index->connectivity < 2) {
        fail(error, "Connectivity must be between 2 and 64 for the device builder (base connectivity 2·M ≤ 128)");
        delete index;
        return nullptr;
    }
    if (index->expansion_add > builder_max_expansion_k) {
        fail(error, "Expansion (add) is too large for the device builder");
        delete index;
        return nullptr;
    }
    return index;
}

void usearch_free(usearch_index_t handle, usearch_error_t*) { delete as_index(handle); }

size_t usearch_memory_usage(usearch_index_t handle, usearch_error_t*) {
    index_t& index = *as_index(handle);
    unique_lock_t lock(index.mutex);
    // what the host side holds (the handle itself, the reserved staging arrays, an owned image) plus the arrays in HBM; never
    // zero for a live index (c/test.c:83 expects as much of a freshly reserved one)
    std::size_t bytes = sizeof(index_t) + index.image_owned.capacity() + index.vectors.capacity() + index.keys.capacity() * 8;
    if (index.snapshot)
        bytes += index.snapshot->device_bytes();
    if (index.builder)
        bytes += index.builder->snapshot().device_bytes();
    return bytes;
}

char const* usearch_hardware_acceleration(usearch_index_t, usearch_error_t* error) {
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices <= 0) {
        fail(error, "No HIP device: this library has no CPU path, every search will fail");
        return "none";
    }
    return "gfx950";
}

size_t usearch_serialized_length(usearch_index_t handle, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    return guarded(error, std::size_t(0), [&] {
        unique_lock_t lock(index.mutex);
        std::size_t length = 0;
        if (const char* e = serialize(index, nullptr, nullptr, 0, &length))
            fail(error, e);
        return length;
    });
}

void usearch_save_buffer(usearch_index_t handle, void* buffer, size_t length, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        if (const char* e = serialize(index, nullptr, buffer, length, nullptr))
            fail(error, e);
    });
}

void usearch_save(usearch_index_t handle, char const* path, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        std::vector<std::uint8_t> bytes;
        if (const char* e = serialize(index, &bytes, nullptr, 0, nullptr))
            return fail(error, e);
        std::FILE* file = std::fopen(path, "wb");
        if (!file)
            return fail(error, "Can't open file!");
        const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), file) == bytes.size();
        std::fclose(file);
        if (!ok)
            fail(error, "Failed to write to file");
    });
}

void usearch_load_buffer(usearch_index_t handle, void const* buffer, size_t length, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] { // the copy of a large file may not fit: an error string, not an exception across the C ABI
        unique_lock_t lock(index.mutex);
        index.drop_device();
        index.drop_image();
        index.image_owned.assign(static_cast<const std::uint8_t*>(buffer), static_cast<const std::uint8_t*>(buffer) + length);
        if (const char* e = index.open_image(index.image_owned.data(), length)) {
            index.drop_image();
            fail(error, e);
        }
    });
}

void usearch_view_buffer(usearch_index_t handle, void const* buffer, size_t length, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        index.drop_device();
        index.drop_image();
        if (const char* e = index.open_image(buffer, length)) // the caller's buffer is borrowed for the index lifetime
            fail(error, e);
    });
}

static const char* map_file(char const* path, void** mapped, std::size_t* length) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0)
        return "Can't open file!";
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size <= 0) {
        ::close(fd);
        return "Can't infer file size";
    }
    void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED)
        return "Can't memory-map the file";
    *mapped = m, *length = (std::size_t)st.st_size;
    return nullptr;
}

void usearch_view(usearch_index_t handle, char const* path, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        index.drop_device();
        index.drop_image();
        void* mapped = nullptr;
        std::size_t length = 0;
        if (const char* e = map_file(path, &mapped, &length))
            return fail(error, e);
        index.mapping = mapped, index.mapping_length = length;
        if (const char* e = index.open_image(mapped, length)) {
            index.drop_image();
            fail(error, e);
        }
    });
}

void usearch_load(usearch_index_t handle, char const* path, usearch_error_t* error) {
    void* mapped = nullptr;
    std::size_t length = 0;
    if (const char* e = map_file(path, &mapped, &length))
        return fail(error, e);
    usearch_load_buffer(handle, mapped, length, error); // copies: the file may change or vanish afterwards
    ::munmap(mapped, length);
}

void usearch_metadata_buffer(void const* buffer, size_t length, usearch_init_options_t* options, usearch_error_t* error) {
    guarded(error, [&] {
        image_t image;
        if (const char* e = image.open(buffer, length))
            return fail(error, e);
        fill_options(image, options);
    });
}

void usearch_metadata(char const* path, usearch_init_options_t* options, usearch_error_t* error) {
    void* mapped = nullptr;
    std::size_t length = 0;
    if (const char* e = map_file(path, &mapped, &length))
        return fail(error, e);
    usearch_metadata_buffer(mapped, length, options, error);
    ::munmap(mapped, length);
}

size_t usearch_size(usearch_index_t handle, usearch_error_t*) {
    index_t& index = *as_index(handle);
    unique_lock_t lock(index.mutex);
    if (!index.staged)
        return index.has_image ? (std::size_t)index.image.count_present : 0;
    std::size_t present = 0;
    for (std::uint64_t key : index.keys)
        present += key != free_key_k;
    return present;
}

size_t usearch_capacity(usearch_index_t handle, usearch_error_t*) {
    index_t& index = *as_index(handle);
    unique_lock_t lock(index.mutex);
    return std::max(index.capacity, index.size());
}

size_t usearch_dimensions(usearch_index_t handle, usearch_error_t*) { return as_index(handle)->dimensions; }
size_t usearch_connectivity(usearch_index_t handle, usearch_error_t*) { return as_index(handle)->connectivity; }

void usearch_reserve(usearch_index_t handle, size_t capacity, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        if (index.staged || !index.has_image) { // the staging arrays are where `usearch_add` puts members
            index.keys.reserve(capacity);
            index.vectors.reserve(capacity * index.bpv());
        }
        index.capacity = std::max(index.capacity, capacity); // only once the memory is there
    });
}

size_t usearch_expansion_add(usearch_index_t handle, usearch_error_t*) { return as_index(handle)->expansion_add; }
size_t usearch_expansion_search(usearch_index_t handle, usearch_error_t*) { return as_index(handle)->expansion_search; }
void usearch_change_expansion_add(usearch_index_t handle, size_t expansion, usearch_error_t*) {
    as_index(handle)->expansion_add = expansion;
}
void usearch_change_expansion_search(usearch_index_t handle, size_t expansion, usearch_error_t*) {
    as_index(handle)->expansion_search = expansion;
}
// Construction is batched on the device (build.hpp): there is no per-thread state to size, the call is accepted and ignored.
void usearch_change_threads_add(usearch_index_t, size_t, usearch_error_t*) {}
// Searches: the number of `usearch_search*` calls that may be in flight at once — the size of the engine's workspace pool,
// the counterpart of the reference's per-thread contexts (index_dense.hpp:931-936, 1984-2000). More callers than that wait.
void usearch_change_threads_search(usearch_index_t handle, size_t threads, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        index.threads_search = threads;
        if (snapshot_t* device_index = index.device_index())
            device_index->set_concurrency(threads ? threads : 16);
    });
}

void usearch_change_metric_kind(usearch_index_t handle, usearch_metric_kind_t kind, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        const metric_kind_t metric = metric_from_c(kind);
        if (!kernel_available(metric, index.scalar))
            return fail(error, "No MI355X kernel for this metric / scalar kind combination");
        if (metric == index.metric)
            return;
        index.materialize(); // the graph was linked under the old metric
        index.metric = metric;
        index.drop_device();
    });
}

void usearch_change_metric(usearch_index_t, usearch_metric_t, void*, usearch_metric_kind_t, usearch_error_t* error) {
    fail(error, "User-defined metric functions cannot run on the device");
}

void usearch_add(usearch_index_t handle, usearch_key_t key, void const* vector, usearch_scalar_kind_t vector_kind,
                 usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        const scalar_kind_t kind = scalar_from_c(vector_kind);
        if (kind == scalar_unknown_k)
            return fail(error, "Unknown scalar kind!");
        if (key == free_key_k)
            return fail(error, "Free key is reserved");
        if (!index.dimensions || !index.bpv())
            return fail(error, "Index is not initialized");
        // a loaded / viewed image at capacity: the slots its tombstones hold are room too — the reference fills `free_keys_` right
        // after a load (`reindex_keys_`, index_dense.hpp:2162-2200) — and here they are only listed once the image is staged
        if (index.size() >= index.capacity && !index.staged && index.has_image && index.key_lookup().size() < index.size())
            index.materialize();
        if (index.size() >= index.capacity && !(index.staged && index.free_head < index.free_slots.size()))
            return fail(error, "Reserve capacity ahead of insertions!"); // index.hpp:2812-2818: no growth on its own; a freed slot is room
        if (!index.multi && index.key_lookup().count(key))
            return fail(error, "Duplicate keys not allowed in high-level wrappers");
        index.materialize();
        const std::size_t bpv = index.bpv();
        // a slot that a removal freed is taken first, oldest first (index_dense.hpp:1479-1511: `free_keys_.try_pop`)
        const bool reuse = index.free_head < index.free_slots.size();
        const std::size_t slot = reuse ? index.free_slots[index.free_head++] : index.keys.size();
        if (slot + 1 >= none_slot_k)
            return fail(error, "Index is too large for 32-bit slots");
        if (!reuse)
            index.vectors.resize((slot + 1) * bpv);
        std::uint8_t* target = index.vectors.data() + slot * bpv;
        std::memset(target, 0, bpv);
        if (!cast_vector(kind, index.scalar, static_cast<const std::uint8_t*>(vector), index.dimensions, target))
            std::memcpy(target, vector, bpv);
        if (reuse) {
            index.keys[slot] = key;
            index.recycled.push_back((std::uint32_t)slot);
            if (index.free_head == index.free_slots.size())
                index.free_slots.clear(), index.free_head = 0;
        } else {
            index.keys.push_back(key);
        }
        ++index.version;
        if (index.lookup_valid)
            index.lookup.emplace(key, (std::uint32_t)slot);
        // the device index, if there is one, stays: the next search links the members added since (builder_t::extend) — or this
        // call does, for hosts that want the reference's visibility (a member is findable the moment `add` returns)
        const std::uint64_t arrived = index.searches_arrived.load(std::memory_order_acquire);
        const bool readers_about = arrived != index.searches_at_last_add;
        index.searches_at_last_add = arrived;
        if (index.immediate_add == 1 || (index.immediate_add == 2 && readers_about)) {
            snapshot_t* device_index = nullptr;
            if (const char* e = index.ready(&device_index))
                return fail(error, e);
        }
    });
}

bool usearch_contains(usearch_index_t handle, usearch_key_t key, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    return guarded(error, false, [&] { // the key table is filled on first use
        unique_lock_t lock(index.mutex);
        return index.key_lookup().count(key) != 0;
    });
}

size_t usearch_count(usearch_index_t handle, usearch_key_t key, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    return guarded(error, std::size_t(0), [&] {
        unique_lock_t lock(index.mutex);
        return (std::size_t)index.key_lookup().count(key);
    });
}

/// The search proper. Takes the index lock itself: shared while the device index is current (searches run side by side),
/// alone only to bring it up to date first (upload after load / view, build, or linking the members added since).
static size_t search_shared(index_t& index, void const* queries, scalar_kind_t kind, std::size_t queries_count,
                            std::size_t queries_stride, std::size_t count, usearch_key_t* keys, std::size_t keys_stride,
                            usearch_distance_t* distances, std::size_t distances_stride, std::size_t* counts,
                            std::size_t* visited_total, std::size_t* computed_total,
                            int (*filter)(usearch_key_t key, void* filter_state), void* filter_state,
                            usearch_error_t* error, const made_filter_t* made = nullptr) {
    if (!queries_count || !count)
        return 0;
    std::size_t result = 0;
    index.searches_arrived.fetch_add(1, std::memory_order_acq_rel); // readers are about: `usearch_add` links at once (index_t::immediate_add)
    guarded(error, [&] {
        for (;;) {
            {
                shared_lock_t shared(index.mutex);
                if (index.device_current()) {
                    snapshot_t* device_index = index.device_index();
                    // a single query that lands in the caller's dense buffers needs no staging copy at all
                    const bool direct = queries_count == 1 || (keys_stride == count * 8 && distances_stride == count * 4);
                    std::vector<std::uint64_t> found(queries_count, 0), visited(queries_count, 0), computed(queries_count, 0);
                    std::vector<std::uint64_t> dense_keys(direct ? 0 : queries_count * count);
                    std::vector<float> dense_distances(direct ? 0 : queries_count * count);
                    std::uint64_t* out_keys = direct ? reinterpret_cast<std::uint64_t*>(keys) : dense_keys.data();
                    float* out_distances = direct ? distances : dense_distances.data();
                    std::vector<std::uint64_t> spare_keys;
                    std::vector<float> spare_distances;
                    if (!out_keys)
                        spare_keys.resize(queries_count * count), out_keys = spare_keys.data();
                    if (!out_distances)
                        spare_distances.resize(queries_count * count), out_distances = spare_distances.data();
                    std::vector<std::uint32_t> bits;
                    search_extras_t extras;
                    std::shared_ptr<filter_t> remembered; // keeps a memoised bitmap alive while this call reads it
                    bool answered_lazily = false;
                    if (made) {
                        // the predicate is already a bitmap in HBM (usearch_filter_from_*): nothing per member happens here
                        if (made->index != &index || made->version != index.version)
                            return fail(error, "The index changed since the filter was made");
                        if (made->bitmap)
                            extras.allow_bits = made->bitmap->bits();
                    } else if (filter) {
                        // The callback is a host function: run it once per member and hand the device one bit per slot. The
                        // traversal then applies it where the reference does (index.hpp:4200-4205, 4236-4240), so results are
                        // the reference's as long as the predicate is a pure function of the key. O(members) callbacks PER
                        // CALL — the reference makes a few thousand; callers that search more than once under one predicate
                        // make a `usearch_filter_t` instead — or, if the predicate is pure for as long as its state pointer
                        // stays the same, switch the memo on (USEARCH_AMD_FILTER_MEMO=1): the binary stays as it is.
                        if (index.filter_memo && device_index) {
                            std::lock_guard<std::mutex> memo_lock(index.memo_mutex); // one maker at a time; searches overlap
                            for (const index_t::filter_memo_t& memo : index.memos)
                                if (memo.filter == filter && memo.state == filter_state && memo.version == index.version)
                                    remembered = memo.bitmap;
                            if (!remembered) {
                                callback_bits(index, filter, filter_state, bits);
                                std::unique_ptr<filter_t> fresh;
                                if (const char* e = filter_t::from_bits(*device_index, bits.data(), bits.size(), fresh))
                                    return fail(error, e);
                                remembered = std::move(fresh);
                                std::vector<index_t::filter_memo_t> kept;
                                kept.push_back({filter, filter_state, index.version, remembered});
                                for (index_t::filter_memo_t& memo : index.memos) // bitmaps of older versions describe nothing now
                                    if (memo.version == index.version && kept.size() < index_t::memo_limit_k)
                                        kept.push_back(std::move(memo));
                                index.memos.swap(kept);
                            }
                            extras.allow_bits = remembered->bits();
                        } else if (index.filter_lazy && device_index) {
                            // the callback only for the members the walk wants to admit (`lazy_predicate_t`): provisional runs until
                            // one asks nothing; that one is the answer
                            lazy_predicate_t lazy;
                            if (const char* e = lazy.open(index.size()))
                                return fail(error, e);
                            bool settled = false;
                            for (int round = 0; round < lazy_predicate_t::rounds_limit_k && !settled; ++round) {
                                search_extras_t lazy_extras;
                                lazy.fill(lazy_extras);
                                if (const char* e = lazy.before_run())
                                    return fail(error, e);
                                if (const char* e = device_index->search_host(queries, kind, queries_count, queries_stride, count,
                                                                              index.expansion_search, out_keys, out_distances, found.data(),
                                                                              visited.data(), computed.data(), search_tuning_t{}, nullptr,
                                                                              nullptr, &lazy_extras))
                                    return fail(error, e);
                                std::size_t asked = 0;
                                bool overflow = false;
                                if (const char* e = lazy.after_run(filter, filter_state, &asked, &overflow))
                                    return fail(error, e);
                                settled = asked == 0 && !overflow;
                                if (overflow)
                                    break; // the walk is visiting a large part of the index (a predicate that rejects nearly everything)
                            }
                            answered_lazily = settled;
                            if (!settled) // a predicate that keeps pushing the walk outward: every member, as before
                                callback_bits(index, filter, filter_state, bits);
                        } else {
                            callback_bits(index, filter, filter_state, bits);
                        }
                    }
                    if (answered_lazily) {
                        // the last provisional run asked nothing: its results are the answer
                    } else if (!device_index) { // nothing indexed yet: index.hpp:3034-3037
                        pad_results(reinterpret_cast<usearch_key_t*>(out_keys), out_distances, queries_count * count);
                    } else if (const char* e = device_index->search_host(
                                   queries, kind, queries_count, queries_stride, count, index.expansion_search, out_keys,
                                   out_distances, found.data(), visited.data(), computed.data(), search_tuning_t{}, nullptr,
                                   filter && !made && !remembered ? bits.data() : nullptr,
                                   made || remembered ? &extras : nullptr)) {
                        return fail(error, e);
                    }
                    std::size_t total_visited = 0, total_computed = 0;
                    for (std::size_t q = 0; q < queries_count; ++q) {
                        if (!direct) {
                            if (keys)
                                std::memcpy(reinterpret_cast<std::uint8_t*>(keys) + q * keys_stride, dense_keys.data() + q * count,
                                            count * 8);
                            if (distances)
                                std::memcpy(reinterpret_cast<std::uint8_t*>(distances) + q * distances_stride,
                                            dense_distances.data() + q * count, count * 4);
                        }
                        if (counts)
                            counts[q] = (std::size_t)found[q];
                        total_visited += (std::size_t)visited[q], total_computed += (std::size_t)computed[q];
                    }
                    if (visited_total)
                        *visited_total = total_visited;
                    if (computed_total)
                        *computed_total = total_computed;
                    result = (std::size_t)found[0];
                    return;
                }
            }
            unique_lock_t alone(index.mutex);
            snapshot_t* device_index = nullptr;
            if (const char* e = index.ready(&device_index))
                return fail(error, e);
        }
    });
    return result;
}

size_t usearch_search(usearch_index_t handle, void const* query, usearch_scalar_kind_t query_kind, size_t count,
                      usearch_key_t* keys, usearch_distance_t* distances, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k) {
        fail(error, "Unknown scalar kind!");
        return 0;
    }
    if (index.coalesce && query && keys && distances && count) {
        // callers that arrive while a launch is in flight go out together in the next one (csrc/combiner.hpp)
        combined_call_t call;
        call.query = query, call.query_bytes = bytes_per_vector(kind, index.dimensions), call.kind = (int)kind, call.wanted = count;
        call.keys = reinterpret_cast<std::uint64_t*>(keys), call.distances = distances;
        index.combiner.submit(call, [&index](std::vector<combined_call_t*>& batch) {
            const combined_call_t& first = *batch[0];
            if (batch.size() == 1) {
                usearch_error_t failure = nullptr;
                batch[0]->found = search_shared(index, first.query, (scalar_kind_t)first.kind, 1, first.query_bytes, first.wanted,
                                                reinterpret_cast<usearch_key_t*>(first.keys), first.wanted * 8, first.distances,
                                                first.wanted * 4, nullptr, nullptr, nullptr, nullptr, nullptr, &failure);
                batch[0]->error = failure;
                return;
            }
            std::vector<std::uint8_t> queries(batch.size() * first.query_bytes);
            std::vector<std::uint64_t> all_keys(batch.size() * first.wanted);
            std::vector<float> all_distances(batch.size() * first.wanted);
            std::vector<std::size_t> counts(batch.size(), 0);
            for (std::size_t i = 0; i < batch.size(); ++i)
                std::memcpy(queries.data() + i * first.query_bytes, batch[i]->query, first.query_bytes);
            usearch_error_t failure = nullptr;
            search_shared(index, queries.data(), (scalar_kind_t)first.kind, batch.size(), first.query_bytes, first.wanted,
                          reinterpret_cast<usearch_key_t*>(all_keys.data()), first.wanted * 8, all_distances.data(), first.wanted * 4,
                          counts.data(), nullptr, nullptr, nullptr, nullptr, &failure);
            for (std::size_t i = 0; i < batch.size(); ++i) {
                batch[i]->error = failure;
                if (failure)
                    continue;
                std::memcpy(batch[i]->keys, all_keys.data() + i * first.wanted, first.wanted * 8);
                std::memcpy(batch[i]->distances, all_distances.data() + i * first.wanted, first.wanted * 4);
                batch[i]->found = counts[i];
            }
        });
        if (call.error)
            fail(error, call.error);
        return call.error ? 0 : call.found;
    }
    return search_shared(index, query, kind, 1, bytes_per_vector(kind, index.dimensions), count, keys, count * 8, distances,
                         count * 4, nullptr, nullptr, nullptr, nullptr, nullptr, error);
}

void usearch_search_many(usearch_index_t handle, void const* queries, usearch_scalar_kind_t query_kind,
                         size_t queries_count, size_t queries_stride, size_t count, usearch_key_t* keys,
                         size_t keys_stride, usearch_distance_t* distances, size_t distances_stride, size_t* counts,
                         size_t* visited_members, size_t* computed_distances, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!");
    search_shared(index, queries, kind, queries_count, queries_stride, count, keys, keys_stride, distances,
                  distances_stride, counts, visited_members, computed_distances, nullptr, nullptr, error);
}

void usearch_cluster_many(usearch_index_t handle, void const* queries, usearch_scalar_kind_t query_kind,
                          size_t queries_count, size_t queries_stride, size_t level, usearch_key_t* keys,
                          usearch_distance_t* distances, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        const scalar_kind_t kind = scalar_from_c(query_kind);
        if (kind == scalar_unknown_k)
            return fail(error, "Unknown scalar kind!");
        if (!queries_count)
            return;
        if (!queries || !keys || !distances)
            return fail(error, "Cluster search needs the query, key and distance buffers");
        snapshot_t* device_index = nullptr;
        if (const char* e = index.ready(&device_index))
            return fail(error, e);
        if (!device_index) // index_gt::cluster on an empty index, index.hpp:3102-3103
            return fail(error, "No clusters to identify");
        std::vector<std::uint64_t> found(queries_count);
        if (const char* e = device_index->cluster_host(queries, kind, queries_count, queries_stride, level, found.data(),
                                                       distances, nullptr, nullptr))
            return fail(error, e);
        std::copy(found.begin(), found.end(), keys);
    });
}

size_t usearch_filtered_search(usearch_index_t handle, void const* query, usearch_scalar_kind_t query_kind, size_t count,
                               int (*filter)(usearch_key_t key, void* filter_state), void* filter_state,
                               usearch_key_t* keys, usearch_distance_t* distances, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k) {
        fail(error, "Unknown scalar kind!");
        return 0;
    }
    return search_shared(index, query, kind, 1, bytes_per_vector(kind, index.dimensions), count, keys, count * 8, distances,
                         count * 4, nullptr, nullptr, nullptr, filter, filter_state, error);
}

size_t usearch_get(usearch_index_t handle, usearch_key_t key, size_t count, void* vector, usearch_scalar_kind_t vector_kind,
                   usearch_error_t* error) {
    index_t& index = *as_index(handle);
    const scalar_kind_t kind = scalar_from_c(vector_kind);
    if (kind == scalar_unknown_k) {
        fail(error, "Unknown scalar kind!");
        return 0;
    }
    return guarded(error, std::size_t(0), [&] {
        unique_lock_t lock(index.mutex);
        const std::size_t out_bytes = bytes_per_vector(kind, index.dimensions);
        auto range = index.key_lookup().equal_range(key);
        std::vector<std::uint32_t> slots;
        for (auto it = range.first; it != range.second; ++it)
            slots.push_back(it->second);
        std::sort(slots.begin(), slots.end()); // insertion order: the hash table's own order is unspecified
        std::size_t exported = 0;
        for (; exported < slots.size() && exported < count; ++exported) {
            std::uint8_t* target = static_cast<std::uint8_t*>(vector) + exported * out_bytes;
            const std::uint8_t* stored = index.vector_of(slots[exported]);
            std::memset(target, 0, out_bytes);
            if (!cast_vector(index.scalar, kind, stored, index.dimensions, target))
                std::memcpy(target, stored, out_bytes);
        }
        return exported;
    });
}

size_t usearch_remove(usearch_index_t handle, usearch_key_t key, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    return guarded(error, std::size_t(0), [&] {
        unique_lock_t lock(index.mutex);
        if (!index.key_lookup().count(key))
            return std::size_t(0);
        index.materialize();
        auto range = index.key_lookup().equal_range(key);
        std::size_t removed = 0;
        for (auto it = range.first; it != range.second; ++it, ++removed) {
            // a tombstone: the member keeps routing, stops matching (index_dense.hpp:1479-1511)
            index.keys[it->second] = free_key_k;
            index.free_slots.push_back(it->second);
            ++index.version;
            if (index.builder && it->second < index.builder->size()) // in place on the device too: nothing is relinked
                if (const char* e = index.builder->set_key(it->second, free_key_k))
                    fail(error, e);
        }
        index.lookup.erase(key);
        return removed;
    });
}

size_t usearch_rename(usearch_index_t handle, usearch_key_t from, usearch_key_t to, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    if (to == free_key_k) {
        fail(error, "Free key is reserved");
        return 0;
    }
    return guarded(error, std::size_t(0), [&] {
        unique_lock_t lock(index.mutex);
        if (!index.key_lookup().count(from))
            return std::size_t(0);
        if (!index.multi && index.lookup.count(to)) {
            fail(error, "Renaming impossible, the key is already in use");
            return std::size_t(0);
        }
        index.materialize();
        std::vector<std::uint32_t> slots;
        auto range = index.key_lookup().equal_range(from);
        for (auto it = range.first; it != range.second; ++it)
            slots.push_back(it->second);
        index.lookup.erase(from);
        for (std::uint32_t slot : slots) {
            index.keys[slot] = to;
            ++index.version;
            index.lookup.emplace(to, slot);
            if (index.builder && slot < index.builder->size()) // keys live next to the graph in HBM: renamed in place
                if (const char* e = index.builder->set_key(slot, to))
                    fail(error, e);
        }
        return slots.size();
    });
}

usearch_distance_t usearch_distance(void const* first, void const* second, usearch_scalar_kind_t scalar_kind,
                                    size_t dimensions, usearch_metric_kind_t metric_kind, usearch_error_t* error) {
    return guarded(error, usearch_distance_t(0), [&] {
        std::uint64_t key = 0;
        float distance = 0.f;
        const scalar_kind_t scalar = scalar_from_c(scalar_kind);
        const std::size_t bpv = bytes_per_vector(scalar, dimensions);
        if (const char* e = exact_search_dataset_host(metric_from_c(metric_kind), scalar, dimensions, second, 1, bpv, first, 1,
                                                      bpv, 1, &key, 8, &distance, 4))
            fail(error, e);
        return usearch_distance_t(distance);
    });
}

void usearch_exact_search(void const* dataset, size_t dataset_size, size_t dataset_stride, void const* queries,
                          size_t queries_size, size_t queries_stride, usearch_scalar_kind_t scalar_kind, size_t dimensions,
                          usearch_metric_kind_t metric_kind, size_t count, size_t /*threads*/, usearch_key_t* keys,
                          size_t keys_stride, usearch_distance_t* distances, size_t distances_stride,
                          usearch_error_t* error) {
    guarded(error, [&] {
        if (const char* e = exact_search_dataset_host(metric_from_c(metric_kind), scalar_from_c(scalar_kind), dimensions, dataset,
                                                      dataset_size, dataset_stride, queries, queries_size, queries_stride, count,
                                                      keys, keys_stride, distances, distances_stride))
            fail(error, e);
    });
}

void usearch_clear(usearch_index_t handle, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        index.drop_device();
        index.drop_image();
        index.keys.clear(), index.vectors.clear();
        index.free_slots.clear(), index.free_head = 0, index.recycled.clear();
        index.staged = false;
        index.lookup.clear(), index.lookup_valid = false;
    });
}

static void search_exact_many(usearch_index_t handle, const made_filter_t* made, void const* queries,
                              usearch_scalar_kind_t query_kind, size_t queries_count, size_t queries_stride, size_t count,
                              usearch_key_t* keys, size_t keys_stride, usearch_distance_t* distances, size_t distances_stride,
                              size_t* counts, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        const scalar_kind_t kind = scalar_from_c(query_kind);
        if (kind == scalar_unknown_k)
            return fail(error, "Unknown scalar kind!");
        if (!queries_count || !count)
            return;
        unique_lock_t lock(index.mutex);
        snapshot_t* device_index = nullptr;
        const std::uint64_t version_before = index.version;
        if (const char* e = index.ready(&device_index))
            return fail(error, e);
        if (made && (made->index != &index || made->version != version_before || made->version != index.version))
            return fail(error, "The index changed since the filter was made");
        std::vector<std::uint64_t> dense_keys(queries_count * count), found(queries_count, 0);
        std::vector<float> dense_distances(queries_count * count);
        if (!device_index) // nothing indexed yet
            pad_results(reinterpret_cast<usearch_key_t*>(dense_keys.data()), dense_distances.data(), queries_count * count);
        else if (const char* e = device_index->exact_host(queries, kind, queries_count, queries_stride, count, dense_keys.data(),
                                                          dense_distances.data(), found.data(), nullptr, false,
                                                          made && made->bitmap ? made->bitmap->bits() : nullptr))
            return fail(error, e);
        for (std::size_t q = 0; q < queries_count; ++q) {
            if (keys)
                std::memcpy(reinterpret_cast<std::uint8_t*>(keys) + q * keys_stride, dense_keys.data() + q * count, count * 8);
            if (distances)
                std::memcpy(reinterpret_cast<std::uint8_t*>(distances) + q * distances_stride, dense_distances.data() + q * count,
                            count * 4);
            if (counts)
                counts[q] = (std::size_t)found[q];
        }
    });
}

void usearch_search_exact_many(usearch_index_t handle, void const* queries, usearch_scalar_kind_t query_kind, size_t queries_count,
                               size_t queries_stride, size_t count, usearch_key_t* keys, size_t keys_stride,
                               usearch_distance_t* distances, size_t distances_stride, size_t* counts, usearch_error_t* error) {
    search_exact_many(handle, nullptr, queries, query_kind, queries_count, queries_stride, count, keys, keys_stride, distances,
                      distances_stride, counts, error);
}

// ---- filters: a predicate made once, in HBM, for any number of searches (include/usearch_c_dropin.h)

usearch_filter_t usearch_filter_from_key_range(usearch_index_t handle, usearch_key_t first_key, usearch_key_t last_key,
                                               usearch_error_t* error) {
    return make_filter(handle, error, [&](index_t&, snapshot_t& device_index, std::unique_ptr<filter_t>& out) {
        return filter_t::from_key_range(device_index, first_key, last_key, out);
    });
}

usearch_filter_t usearch_filter_from_keys(usearch_index_t handle, usearch_key_t const* keys, size_t keys_count, bool allow,
                                          usearch_error_t* error) {
    return make_filter(handle, error, [&](index_t&, snapshot_t& device_index, std::unique_ptr<filter_t>& out) {
        return filter_t::from_keys(device_index, reinterpret_cast<const std::uint64_t*>(keys), keys_count, allow, out);
    });
}

usearch_filter_t usearch_filter_from_callback(usearch_index_t handle, int (*filter)(usearch_key_t key, void* filter_state),
                                              void* filter_state, usearch_error_t* error) {
    if (!filter)
        return fail(error, "No predicate given"), nullptr;
    return make_filter(handle, error, [&](index_t& index, snapshot_t& device_index, std::unique_ptr<filter_t>& out) {
        std::vector<std::uint32_t> bits;
        callback_bits(index, filter, filter_state, bits);
        return filter_t::from_bits(device_index, bits.data(), bits.size(), out);
    });
}

size_t usearch_filter_allowed(usearch_filter_t filter, usearch_error_t*) {
    const made_filter_t* made = static_cast<const made_filter_t*>(filter);
    return made && made->bitmap ? (std::size_t)made->bitmap->allowed() : 0;
}

void usearch_filter_free(usearch_filter_t filter, usearch_error_t*) { delete static_cast<made_filter_t*>(filter); }

void usearch_filtered_search_many(usearch_index_t handle, usearch_filter_t filter, void const* queries,
                                  usearch_scalar_kind_t query_kind, size_t queries_count, size_t queries_stride, size_t count,
                                  usearch_key_t* keys, size_t keys_stride, usearch_distance_t* distances, size_t distances_stride,
                                  size_t* counts, size_t* visited_members, size_t* computed_distances, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    const scalar_kind_t kind = scalar_from_c(query_kind);
    if (kind == scalar_unknown_k)
        return fail(error, "Unknown scalar kind!");
    if (!filter)
        return fail(error, "No filter given");
    search_shared(index, queries, kind, queries_count, queries_stride, count, keys, keys_stride, distances, distances_stride,
                  counts, visited_members, computed_distances, nullptr, nullptr, error, static_cast<const made_filter_t*>(filter));
}

void usearch_filtered_search_exact_many(usearch_index_t handle, usearch_filter_t filter, void const* queries,
                                        usearch_scalar_kind_t query_kind, size_t queries_count, size_t queries_stride, size_t count,
                                        usearch_key_t* keys, size_t keys_stride, usearch_distance_t* distances,
                                        size_t distances_stride, size_t* counts, usearch_error_t* error) {
    if (!filter)
        return fail(error, "No filter given");
    search_exact_many(handle, static_cast<const made_filter_t*>(filter), queries, query_kind, queries_count, queries_stride, count,
                      keys, keys_stride, distances, distances_stride, counts, error);
}

size_t usearch_threads_search(usearch_index_t handle, usearch_error_t*) { return as_index(handle)->threads_search; }

void usearch_gpu_sync(usearch_index_t handle, usearch_error_t* error) {
    index_t& index = *as_index(handle);
    guarded(error, [&] {
        unique_lock_t lock(index.mutex);
        snapshot_t* device_index = nullptr;
        if (const char* e = index.ready(&device_index))
            fail(error, e);
    });
}

void usearch_gpu_release(usearch_index_t handle, usearch_error_t*) {
    index_t& index = *as_index(handle);
    unique_lock_t lock(index.mutex);
    index.drop_device();
}

usearch_amd_c_api_t const* usearch_amd_c_api(void) {
    static const usearch_amd_c_api_t table = {
        51,
        &usearch_version, &usearch_init, &usearch_free, &usearch_memory_usage, &usearch_hardware_acceleration,
        &usearch_serialized_length, &usearch_save, &usearch_load, &usearch_view, &usearch_metadata, &usearch_save_buffer,
        &usearch_load_buffer, &usearch_view_buffer, &usearch_metadata_buffer, &usearch_size, &usearch_capacity,
        &usearch_dimensions, &usearch_connectivity, &usearch_reserve, &usearch_expansion_add, &usearch_expansion_search,
        &usearch_change_expansion_add, &usearch_change_expansion_search, &usearch_change_threads_add,
        &usearch_change_threads_search, &usearch_change_metric_kind, &usearch_change_metric, &usearch_add, &usearch_contains,
        &usearch_count, &usearch_search, &usearch_filtered_search, &usearch_get, &usearch_remove, &usearch_rename,
        &usearch_distance, &usearch_exact_search, &usearch_clear, &usearch_search_many, &usearch_cluster_many,
        &usearch_search_exact_many, &usearch_threads_search, &usearch_gpu_sync, &usearch_gpu_release,
        &usearch_filter_from_key_range, &usearch_filter_from_keys, &usearch_filter_from_callback, &usearch_filter_allowed,
        &usearch_filter_free, &usearch_filtered_search_many, &usearch_filtered_search_exact_many,
    };
    return &table;
}

} // extern "C"
