// tests/cpp/bench_loop.cpp — a C++ caller of the reference, unchanged but for its include path.
//
// The add and search loops of /root/reference/cpp/bench.cpp (index_many 318-349, search_many 352-377) in their own shape — one
// `index.add(key, vector, thread)` / `index.search(vector, wanted, thread).dump_to(ids, distances, wanted)` per row inside an
// OpenMP `parallel for schedule(static, 32)` — over `unum::usearch::index_dense_t` as include/usearch/index_dense.hpp provides it
// (the MI355X engine behind the class). `bench_loop link` proves that this compiles and links (no GPU); `bench_loop run` builds
// 4 000 x 96 f32 cosine vectors, searches them back and checks self-recall, ordering and padding, then runs the same batch through
// `search_many` (one launch) and expects the same rows.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#define USEARCH_USE_OPENMP 1
#else
#define USEARCH_USE_OPENMP 0
#endif

#include <usearch/index_dense.hpp>

using namespace unum::usearch;

template <typename index_at, typename vector_id_at, typename scalar_at>
void index_many(index_at& index, std::size_t n, vector_id_at const* ids, scalar_at const* vectors, std::size_t dims) {
#if USEARCH_USE_OPENMP
#pragma omp parallel for schedule(static, 32)
#endif
    for (std::size_t i = 0; i < n; ++i) {
        index_update_config_t config;
#if USEARCH_USE_OPENMP
        config.thread = omp_get_thread_num();
#endif
        index.add(ids[i], vectors + dims * i, config.thread);
    }
}

template <typename index_at, typename vector_id_at, typename scalar_at, typename distance_at>
void search_many(index_at& index, std::size_t n, scalar_at const* vectors, std::size_t dims, std::size_t wanted, vector_id_at* ids,
                 distance_at* distances) {
#if USEARCH_USE_OPENMP
#pragma omp parallel for schedule(static, 32)
#endif
    for (std::size_t i = 0; i < n; ++i) {
        index_search_config_t config;
#if USEARCH_USE_OPENMP
        config.thread = omp_get_thread_num();
#endif
        span_gt<scalar_at const> vector{vectors + dims * i, dims};
        index.search(vector, wanted, config.thread).dump_to(ids + wanted * i, distances + wanted * i, wanted);
    }
}

#define EXPECT(condition)                                                                                              \
    do {                                                                                                               \
        if (!(condition)) {                                                                                            \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #condition);                                         \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

int main(int argc, char** argv) {
    std::printf("usearch %d.%d.%d behind index_dense_t\n", USEARCH_VERSION_MAJOR, USEARCH_VERSION_MINOR, USEARCH_VERSION_PATCH);
    if (argc < 2 || std::strcmp(argv[1], "run") != 0)
        return 0;
    const std::size_t dims = 96, count = 4000, queries = 512, wanted = 10;
    std::mt19937 generator(11);
    std::normal_distribution<float> normal;
    std::vector<float> data(count * dims);
    for (float& x : data)
        x = normal(generator);
    std::vector<default_key_t> ids(count);
    for (std::size_t i = 0; i < count; ++i)
        ids[i] = 7000 + i;

    metric_punned_t metric(dims, metric_kind_t::cos_k, scalar_kind_t::f32_k);
    index_dense_config_t config(16, 128, 64);
    index_dense_t index = index_dense_t::make(metric, config);
    EXPECT(index);
    index.reserve(index_limits_t(count, 8));
    index_many(index, count, ids.data(), data.data(), dims);
    EXPECT(index.size() == count && index.dimensions() == dims && index.connectivity() == 16);
    EXPECT(index.limits().members >= count && index.limits().threads_search == 8);

    std::vector<default_key_t> found(queries * wanted);
    std::vector<float> distances(queries * wanted);
    search_many(index, queries, data.data(), dims, wanted, found.data(), distances.data());
    std::size_t self = 0;
    for (std::size_t i = 0; i < queries; ++i) {
        self += found[i * wanted] == ids[i];
        for (std::size_t j = 1; j < wanted; ++j)
            EXPECT(distances[i * wanted + j] >= distances[i * wanted + j - 1]);
        EXPECT(std::fabs(distances[i * wanted]) < 0.01f || found[i * wanted] != ids[i]);
    }
    std::printf("self-recall@1 %zu / %zu\n", self, queries);
    EXPECT(self >= queries * 99 / 100);

    // the same batch in one launch
    std::vector<default_key_t> batch_found(queries * wanted);
    std::vector<float> batch_distances(queries * wanted);
    std::vector<std::size_t> counts(queries);
    auto batch = index.search_many(data.data(), queries, dims * sizeof(float), wanted, batch_found.data(), batch_distances.data(),
                                   counts.data());
    EXPECT(batch);
    EXPECT(batch.computed_distances > 0 && batch.visited_members > 0);
    EXPECT(std::memcmp(batch_found.data(), found.data(), found.size() * sizeof(default_key_t)) == 0);
    EXPECT(std::memcmp(batch_distances.data(), distances.data(), distances.size() * sizeof(float)) == 0);

    // exact = true: brute force over every member; the first result of an in-sample query is the query itself
    auto exact = index.search(data.data() + 5 * dims, wanted, 0, true);
    EXPECT(exact && exact.size() == wanted && exact[0].member.key == ids[5]);
    // filtered: only even keys
    auto filtered = index.filtered_search(data.data() + 6 * dims, wanted, [](default_key_t key) { return key % 2 == 0; });
    EXPECT(filtered && filtered.size() == wanted);
    for (std::size_t j = 0; j < filtered.size(); ++j)
        EXPECT(filtered[j].member.key % 2 == 0);
    // filtered AND exact: the brute-force scan skips what the predicate rejects (index.hpp:4260-4263) — an in-sample query with an
    // odd key must not come back under "even keys only", and nothing odd may
    auto filtered_exact = index.filtered_search(data.data() + 7 * dims, wanted, [](default_key_t key) { return key % 2 == 0; }, 0, true);
    EXPECT(filtered_exact && filtered_exact.size() == wanted);
    for (std::size_t j = 0; j < filtered_exact.size(); ++j)
        EXPECT(filtered_exact[j].member.key % 2 == 0);
    auto unfiltered_exact = index.search(data.data() + 7 * dims, wanted, 0, true);
    EXPECT(unfiltered_exact[0].member.key == ids[7]);
    EXPECT((ids[7] % 2 == 0) == (filtered_exact[0].member.key == ids[7]));
    // dump_to with a capacity pads: key 0 and a (signalling) NaN
    default_key_t padded_keys[16];
    float padded_distances[16];
    EXPECT(filtered.dump_to(padded_keys, padded_distances, 16) == wanted);
    EXPECT(padded_keys[15] == 0 && std::isnan(padded_distances[15]));
    // get / contains / remove / rename, save / load into a fresh object
    std::vector<float> back(dims);
    EXPECT(index.get(ids[9], back.data()) == 1 && std::memcmp(back.data(), data.data() + 9 * dims, dims * 4) == 0);
    EXPECT(index.contains(ids[9]) && !index.contains(1));
    EXPECT(index.rename(ids[9], 99).completed == 1 && index.contains(99) && !index.contains(ids[9]));
    EXPECT(index.remove(99).completed == 1 && !index.contains(99) && index.size() == count - 1);
    EXPECT(index.save("/tmp/usearch_amd_bench_loop.usearch"));
    index_dense_t copy = index_dense_t::make("/tmp/usearch_amd_bench_loop.usearch");
    EXPECT(copy && copy.size() == count - 1 && copy.dimensions() == dims && copy.scalar_kind() == scalar_kind_t::f32_k);
    auto again = copy.search(data.data() + 5 * dims, wanted);
    EXPECT(again && again[0].member.key == ids[5]);
    std::printf("bench loop passed\n");
    return 0;
}
