// tests/cpp/combiner_test.cpp — the call combiner of the drop-in (usearch_amd/csrc/combiner.hpp) under 32 threads with a mock launch:
// every call gets its own answer, calls that arrive during a launch go out together, groups never mix kinds or result counts;
// sixteen looping callers end up in ONE launch each time when the launcher gives the returning callers a moment (and in two
// alternating groups of eight when it does not), and a lone caller is never made to wait.
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

    // one caller's trouble stays that caller's: a shared launch that reports an error goes out again call by call, and only the
    // call that is at fault comes back with it
    combiner_t shared;
    std::atomic<int> innocent_failures{0}, guilty_failures{0};
    {
        std::vector<std::thread> callers;
        for (int t = 0; t < 12; ++t)
            callers.emplace_back([&, t] {
                for (int i = 0; i < 10; ++i) {
                    float query[4] = {(float)t, 0.f, 0.f, 0.f};
                    std::uint64_t keys[4];
                    float distances[4];
                    combined_call_t call;
                    call.query = query, call.query_bytes = sizeof(query), call.kind = 1, call.wanted = 4, call.keys = keys,
                    call.distances = distances;
                    shared.submit(call, [](std::vector<combined_call_t*>& batch) {
                        bool poisoned = false;
                        for (combined_call_t* other : batch)
                            poisoned = poisoned || static_cast<const float*>(other->query)[0] == 5.f; // caller 5 breaks a launch
                        for (combined_call_t* other : batch)
                            other->found = poisoned ? 0 : other->wanted, other->error = poisoned ? "This launch failed" : nullptr;
                        std::this_thread::sleep_for(std::chrono::microseconds(500));
                    });
                    if (t == 5)
                        guilty_failures += call.error != nullptr;
                    else
                        innocent_failures += call.error != nullptr || call.found != call.wanted;
                }
            });
        for (std::thread& thread : callers)
            thread.join();
    }
    std::printf("a caller whose calls fail among eleven whose calls do not: %d of its 10 calls failed, %d of the others', %llu shared "
                "launches repeated call by call\n", guilty_failures.load(), innocent_failures.load(), (unsigned long long)shared.relaunched());
    const bool contained = guilty_failures.load() == 10 && innocent_failures.load() == 0;

    // looping callers: launches needed for the same calls with and without the launcher's wait for the callers just served
    std::uint64_t launches_for[2] = {0, 0}, expired_alone = 0;
    for (int with_window = 0; with_window < 2; ++with_window)
        for (int callers : {16, 1}) {
            combiner_t looped;
            looped.window_limit(std::chrono::microseconds(with_window ? 400 : 0));
            std::vector<std::thread> loopers;
            for (int t = 0; t < callers; ++t)
                loopers.emplace_back([&] {
                    for (int i = 0; i < 30; ++i) {
                        float query[4] = {0};
                        std::uint64_t keys[4];
                        float distances[4];
                        combined_call_t call;
                        call.query = query, call.query_bytes = sizeof(query), call.kind = 1, call.wanted = 4, call.keys = keys,
                        call.distances = distances;
                        looped.submit(call, [](std::vector<combined_call_t*>& batch) {
                            for (combined_call_t* other : batch)
                                other->found = other->wanted;
                            std::this_thread::sleep_for(std::chrono::microseconds(4000));
                        });
                    }
                });
            for (std::thread& thread : loopers)
                thread.join();
            std::uint64_t served = 0;
            if (callers == 16)
                looped.totals(launches_for[with_window], served);
            else if (with_window)
                expired_alone = looped.windows_expired();
        }
    std::printf("sixteen looping callers, thirty calls each: %llu launches without the wait, %llu with it; a lone caller waited %llu times\n",
                (unsigned long long)launches_for[0], (unsigned long long)launches_for[1], (unsigned long long)expired_alone);
    const bool gathered = launches_for[1] * 100 < launches_for[0] * 85 && !expired_alone;
    std::printf("%s\n", ok && reported && gathered && contained ? "PASSED" : "FAILED");
    return ok && reported && gathered && contained ? 0 : 1;
}
