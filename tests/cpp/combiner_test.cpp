// tests/cpp/combiner_test.cpp — the call combiner of the drop-in (usearch_amd/csrc/combiner.hpp) under 32 threads with a mock launch:
// every call gets its own answer, calls that arrive during a launch go out together, groups never mix kinds or result counts.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../usearch_amd/csrc/combiner.hpp"

using namespace usearch_amd;

int main() {
    combiner_t combiner;
    std::atomic<int> mixed{0}, largest{0};
    const int threads = 32, rounds = 40;
    std::vector<std::thread> pool;
    std::atomic<long> wrong{0};
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            for (int i = 0; i < rounds; ++i) {
                float query[4] = {(float)t, (float)i, 0.f, 0.f};
                std::uint64_t keys[8] = {0};
                float distances[8] = {0};
                combined_call_t call;
                call.query = query, call.query_bytes = sizeof(query), call.kind = t % 2 ? 1 : 3, call.wanted = t % 3 == 0 ? 8 : 5;
                call.keys = keys, call.distances = distances;
                combiner.submit(call, [&](std::vector<combined_call_t*>& batch) {
                    if ((int)batch.size() > largest.load())
                        largest = (int)batch.size();
                    for (combined_call_t* other : batch) {
                        if (other->kind != batch[0]->kind || other->wanted != batch[0]->wanted)
                            ++mixed;
                        const float* q = static_cast<const float*>(other->query);
                        for (std::size_t j = 0; j < other->wanted; ++j) // the mock index answers with what identifies the query
                            other->keys[j] = (std::uint64_t)q[0] * 1000 + (std::uint64_t)q[1], other->distances[j] = q[0] + q[1];
                        other->found = other->wanted;
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(300)); // a launch takes its time: callers queue up meanwhile
                });
                if (!call.done || call.error || call.found != call.wanted || keys[0] != (std::uint64_t)t * 1000 + (std::uint64_t)i ||
                    keys[call.wanted - 1] != keys[0] || distances[0] != (float)(t + i))
                    ++wrong;
            }
        });
    for (std::thread& thread : pool)
        thread.join();
    std::uint64_t launches = 0, calls = 0;
    combiner.totals(launches, calls);
    std::printf("calls %llu launches %llu largest batch %d mixed %d wrong %ld\n", (unsigned long long)calls, (unsigned long long)launches,
                largest.load(), mixed.load(), wrong.load());
    const bool ok = calls == (std::uint64_t)threads * rounds && launches < calls && largest.load() > 1 && !mixed.load() && !wrong.load();
    // an exception inside a launch becomes an error string for every call of that batch
    combined_call_t failing;
    failing.wanted = 1;
    combiner.submit(failing, [](std::vector<combined_call_t*>&) { throw 1; });
    const bool reported = failing.done && failing.error && !failing.found;
    std::printf("%s\n", ok && reported ? "PASSED" : "FAILED");
    return ok && reported ? 0 : 1;
}
