/**
 *  usearch_amd/csrc/combiner.hpp — calls in flight share a launch.
 *
 *  The reference serves T concurrent `usearch_search` callers with T contexts on T cores (index_dense.hpp:1984-2000). On the
 *  device every call is a kernel launch, the runtime runs only a handful of launches side by side (its hardware queues) and
 *  each costs host API time, so T callers get far less than T × (1 / latency) (profiles/r03_team/threads_*.log) — while ONE
 *  launch walks hundreds of queries in the time it walks one. So callers that arrive while a launch is in flight wait for it
 *  and then go out TOGETHER: the first one to find nobody launching takes every waiting call that is compatible with its own
 *  (same query kind, same result count) and runs them as one batch through `usearch_search_many`'s path; the others sleep
 *  until their results are there. Pure host logic, no HIP: `tests/cpp/combiner_test.cpp` drives it with a mock launch.
 *
 *  Opt-in for now (`USEARCH_AMD_COALESCE=1`, read at `usearch_init`): the device side of it has not been measured yet.
 */
#pragma once
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

namespace usearch_amd {

/// One caller's `usearch_search`: inputs, where its results go, and what came back.
struct combined_call_t {
    const void* query = nullptr;
    std::size_t query_bytes = 0; ///< bytes of one query of `kind`
    int kind = 0;                ///< query scalar kind
    std::size_t wanted = 0;
    std::uint64_t* keys = nullptr; ///< [wanted], the caller's
    float* distances = nullptr;    ///< [wanted], the caller's
    std::size_t found = 0;
    const char* error = nullptr;   ///< static string or null, as the C ABI reports errors
    bool done = false;
};

class combiner_t {
  public:
    /**
     *  Blocks until `call` has its results. `run(batch)` is invoked by whichever caller finds nobody launching, with every waiting
     *  call that shares (kind, wanted, query_bytes) with its own, its own included; it must fill `found` / `error` of each.
     *  At most one `run` is in flight per combiner.
     */
    template <typename run_at> void submit(combined_call_t& call, run_at&& run) {
        std::unique_lock<std::mutex> lock(mutex_);
        waiting_.push_back(&call);
        while (!call.done && launching_)
            changed_.wait(lock);
        if (call.done)
            return;
        launching_ = true;
        std::vector<combined_call_t*> batch, rest;
        for (combined_call_t* other : waiting_)
            (other->kind == call.kind && other->wanted == call.wanted && other->query_bytes == call.query_bytes ? batch : rest)
                .push_back(other);
        waiting_.swap(rest);
        lock.unlock();
        try {
            run(batch);
        } catch (...) {
            for (combined_call_t* other : batch)
                other->found = 0, other->error = "Unexpected failure inside the index";
        }
        lock.lock();
        for (combined_call_t* other : batch)
            other->done = true;
        launching_ = false;
        ++launches_;
        calls_ += batch.size();
        changed_.notify_all();
    }

    /// How many launches served how many calls so far (telemetry / tests).
    void totals(std::uint64_t& launches, std::uint64_t& calls) {
        std::lock_guard<std::mutex> lock(mutex_);
        launches = launches_, calls = calls_;
    }

  private:
    std::mutex mutex_;
    std::condition_variable changed_;
    std::vector<combined_call_t*> waiting_;
    bool launching_ = false;
    std::uint64_t launches_ = 0, calls_ = 0;
};

} // namespace usearch_amd
