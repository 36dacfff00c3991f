// tests/cpp/class_test.cpp — include/usearch_amd.hpp used the way C++ callers use `index_dense_gt`
// (/root/reference/cpp/test.cpp:200-260 `test_minimal_three_vectors`-style flow, 499-503 ordering, 1105-1145 filtered search).
// `class_test link` only proves that the header compiles and the drop-in resolves (no GPU); `class_test run` needs an MI355X.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "usearch_amd.hpp"

#define EXPECT(condition)                                                                                              \
    do {                                                                                                               \
        if (!(condition)) {                                                                                            \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #condition);                                         \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

int main(int argc, char** argv) {
    std::printf("usearch %s\n", usearch_version());
    if (argc < 2 || std::strcmp(argv[1], "run") != 0)
        return 0;
    using namespace usearch_amd;
    const std::size_t dimensions = 64, count = 1500;
    std::mt19937 generator(7);
    std::normal_distribution<float> normal;
    std::vector<float> data(count * dimensions);
    for (float& x : data)
        x = normal(generator);

    auto made = index_dense_t::make(dimensions, usearch_metric_cos_k, usearch_scalar_f32_k);
    EXPECT(made);
    index_dense_t index = std::move(made.index);
    EXPECT(index.try_reserve(count));
    for (std::size_t i = 0; i < count; ++i)
        EXPECT(index.add(1000 + i, data.data() + i * dimensions));
    EXPECT(index.size() == count && index.dimensions() == dimensions && index.contains(1000) && !index.contains(1));

    // every vector finds itself first; distances ascend
    for (std::size_t i = 0; i < count; i += 37) {
        auto result = index.search(data.data() + i * dimensions, 10);
        EXPECT(result && result.size() == 10);
        EXPECT(result[0].key == 1000 + i && std::fabs(result[0].distance) < 1e-3f);
        for (std::size_t j = 1; j < result.size(); ++j)
            EXPECT(result[j - 1].distance <= result[j].distance);
    }
    // the same through the batched call
    auto batch = index.search_many(data.data(), usearch_scalar_f32_k, 200, dimensions * sizeof(float), 5);
    EXPECT(batch);
    for (std::size_t i = 0; i < 200; ++i)
        EXPECT(batch.counts[i] == 5 && batch.keys[i * 5] == 1000 + i);
    EXPECT(batch.computed_distances > 0 && batch.visited_members > 0);
    // cluster(vector, level): level 0 and 1 both end on level 1's winner; far above the top level it is the entry point
    auto low = index.cluster(data.data(), 0), same = index.cluster(data.data(), 1), top_most = index.cluster(data.data(), 99);
    EXPECT(low && same && top_most && low.key == same.key && low.distance == same.distance);
    EXPECT(index.contains(low.key) && index.contains(top_most.key) && top_most.distance >= low.distance);
    // predicate
    auto odd = index.filtered_search(data.data(), 10, [](vector_key_t key) { return key % 2 == 1; });
    EXPECT(odd && odd.size() == 10);
    for (std::size_t j = 0; j < odd.size(); ++j)
        EXPECT(odd[j].key % 2 == 1);
    // get / rename / remove
    std::vector<float> back(dimensions);
    EXPECT(index.get(1005, back.data()) == 1 && std::memcmp(back.data(), data.data() + 5 * dimensions, dimensions * 4) == 0);
    EXPECT(index.rename(1005, 5) == 1 && index.contains(5) && !index.contains(1005));
    EXPECT(index.remove(1006) == 1 && index.size() == count - 1);
    auto after = index.search(data.data() + 6 * dimensions, 3);
    for (std::size_t j = 0; j < after.size(); ++j)
        EXPECT(after[j].key != 1006);
    // save → a second object loads it and answers alike
    EXPECT(index.save("/tmp/usearch_amd_class_test.usearch"));
    auto reopened = index_dense_t::make("/tmp/usearch_amd_class_test.usearch");
    EXPECT(reopened && reopened.index.size() == count - 1 && reopened.index.connectivity() == index.connectivity());
    auto a = index.search(data.data() + 99 * dimensions, 10), b = reopened.index.search(data.data() + 99 * dimensions, 10);
    EXPECT(a.size() == b.size());
    for (std::size_t j = 0; j < a.size(); ++j)
        EXPECT(a[j].key == b[j].key && a[j].distance == b[j].distance);
    std::remove("/tmp/usearch_amd_class_test.usearch");

    // the remaining scalar overloads of index_dense.hpp:760-772: doubles (pearson, as c/test.c:246-250 configures it) and
    // brain floats as bit patterns
    auto doubles = index_dense_t::make(dimensions, usearch_metric_pearson_k, usearch_scalar_f64_k);
    EXPECT(doubles);
    std::vector<double> wide(data.begin(), data.begin() + 300 * dimensions);
    EXPECT(!doubles.index.add(0, wide.data())); // index.hpp:2812-2818: "Reserve capacity ahead of insertions!"
    EXPECT(doubles.index.try_reserve(300));
    for (std::size_t i = 0; i < 300; ++i)
        EXPECT(doubles.index.add(i, wide.data() + i * dimensions));
    auto nearest = doubles.index.search(wide.data() + 17 * dimensions, 3);
    EXPECT(nearest && nearest.size() == 3 && nearest[0].key == 17 && std::fabs(nearest[0].distance) < 1e-6f);
    auto brains = index_dense_t::make(dimensions, usearch_metric_cos_k, usearch_scalar_bf16_k);
    EXPECT(brains);
    std::vector<bf16_bits_t> narrow(300 * dimensions);
    for (std::size_t i = 0; i < narrow.size(); ++i) {
        std::uint32_t bits;
        std::memcpy(&bits, &data[i], 4);
        narrow[i].bits = (std::uint16_t)(bits >> 16);
    }
    EXPECT(brains.index.try_reserve(300));
    for (std::size_t i = 0; i < 300; ++i)
        EXPECT(brains.index.add(i, narrow.data() + i * dimensions));
    auto brain_hit = brains.index.search(narrow.data() + 5 * dimensions, 3);
    EXPECT(brain_hit && brain_hit[0].key == 5 && std::fabs(brain_hit[0].distance) < 1e-3f);
    std::printf("class test passed\n");
    return 0;
}
