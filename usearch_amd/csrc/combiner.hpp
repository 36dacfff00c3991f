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
 *  Left at that, T looping callers settle into TWO groups of T/2 that alternate (one in flight, one waiting: each call then
 *  costs two launches of wall time, profiles/r04_single_query/: 5.5 ms per call for a 2.7 ms launch). The callers
 *  of the launch that just finished are microseconds away from calling again, so the one who launches next gives them a
 *  moment — at most an eighth of the last launch's duration and `window_limit` (200 us unless told otherwise) — and stops waiting the
 *  instant as many calls have arrived as the finished launch had served. A lone caller never waits (its own return is the one
 *  arrival expected), a caller that stops calling costs the others one window, once.
 *
 *  On by default in the drop-in (`USEARCH_AMD_COALESCE=0` turns it off, `USEARCH_AMD_COALESCE_WINDOW_US` sets the limit;
 *  both read at `usearch_init`).
 */
#pragma once
#include <chrono>
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
        using clock_t = std::chrono::steady_clock;
        std::unique_lock<std::mutex> lock(mutex_);
        waiting_.push_back(&call);
        if (expected_back_ && !--expected_back_)
            gathered_.notify_one();
        while (!call.done && launching_)
            changed_.wait(lock);
        if (call.done)
            return;
        launching_ = true;
        if (expected_back_ && window_limit_.count()) { // the callers the last launch served are on their way back
            const auto window = last_duration_ / 8 < window_limit_ ? last_duration_ / 8 : window_limit_;
            if (!gathered_.wait_for(lock, window, [this] { return expected_back_ == 0; }))
                ++windows_expired_;
        }
        expected_back_ = 0;
        std::vector<combined_call_t*> batch, rest;
        try { // allocation is the one thing that can fail between taking the launcher's seat and the launch
            batch.reserve(waiting_.size()), rest.reserve(waiting_.size());
        } catch (...) {
            // give the seat back, or every later caller sleeps for good: this call leaves the queue and reports the failure,
            // the others find nobody launching and one of them launches
            for (std::size_t i = 0; i < waiting_.size(); ++i)
                if (waiting_[i] == &call) {
                    waiting_.erase(waiting_.begin() + (std::ptrdiff_t)i);
                    break;
                }
            call.found = 0, call.error = "Out of memory", call.done = true;
            launching_ = false;
            changed_.notify_all();
            return;
        }
        for (combined_call_t* other : waiting_)
            (other->kind == call.kind && other->wanted == call.wanted && other->query_bytes == call.query_bytes ? batch : rest)
                .push_back(other);
        waiting_.swap(rest);
        lock.unlock();
        const clock_t::time_point started = clock_t::now();
        auto guarded_run = [&](std::vector<combined_call_t*>& calls) {
            try {
                run(calls);
            } catch (...) {
                for (combined_call_t* other : calls)
                    other->found = 0, other->error = "Unexpected failure inside the index";
            }
        };
        guarded_run(batch);
        // a launch that failed reports its error to every call it carried; so that one caller's trouble stays that caller's, the
        // calls of a failed shared launch go out again one by one (rare, and no slower than the uncombined path)
        bool failed = false;
        for (combined_call_t* other : batch)
            failed = failed || other->error != nullptr;
        if (failed && batch.size() > 1) {
            std::vector<combined_call_t*> single(1);
            for (combined_call_t* other : batch) {
                other->found = 0, other->error = nullptr;
                single[0] = other;
                guarded_run(single);
            }
            ++relaunched_;
        }
        const auto duration = std::chrono::duration_cast<std::chrono::nanoseconds>(clock_t::now() - started);
        lock.lock();
        for (combined_call_t* other : batch)
            other->done = true;
        launching_ = false;
        last_duration_ = duration;
        expected_back_ = batch.size();
        ++launches_;
        calls_ += batch.size();
        changed_.notify_all();
    }

    /// How many launches served how many calls so far (telemetry / tests).
    void totals(std::uint64_t& launches, std::uint64_t& calls) {
        std::lock_guard<std::mutex> lock(mutex_);
        launches = launches_, calls = calls_;
    }

    /// How many times a launcher's wait for returning callers ran its full length (telemetry / tests).
    std::uint64_t windows_expired() {
        std::lock_guard<std::mutex> lock(mutex_);
        return windows_expired_;
    }

    /// How many shared launches reported an error and were repeated call by call (telemetry / tests).
    std::uint64_t relaunched() {
        std::lock_guard<std::mutex> lock(mutex_);
        return relaunched_;
    }

    /// The longest a launcher waits for the callers of the launch before its own; zero: it never waits.
    void window_limit(std::chrono::nanoseconds limit) {
        std::lock_guard<std::mutex> lock(mutex_);
        window_limit_ = limit;
    }

  private:
    std::mutex mutex_;
    std::condition_variable changed_;  ///< a launch finished: its calls are done, somebody else may launch
    std::condition_variable gathered_; ///< everybody the launcher was waiting for has arrived
    std::vector<combined_call_t*> waiting_;
    bool launching_ = false;
    std::size_t expected_back_ = 0; ///< calls the last launch served whose callers have not called again yet
    std::chrono::nanoseconds last_duration_{0}, window_limit_{200000};
    std::uint64_t launches_ = 0, calls_ = 0, windows_expired_ = 0, relaunched_ = 0;
};

} // namespace usearch_amd
