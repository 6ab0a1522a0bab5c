// host_parallel.h -- a few host threads for the one-time weight repacking (iris_hifigan_create / _prepare).
//
// The reference's only real caller loads the model and vocodes ONE utterance (scripts/synthesize.py:197-198 ->
// hifigan_pretrained.py:250-283), so the cold start -- 13.9 M weights repacked into MFMA fragment order, twice for the
// fp32 path (32 x 32 and 16 x 16 fragments) -- is part of what that caller waits for.  The packers are pure
// gather loops over disjoint output ranges: jobs = (layer, tap) pieces drawn from an atomic counter.
#pragma once
#include <atomic>
#include <exception>
#include <functional>
#include <new>
#include <thread>
#include <vector>

namespace iris {

// Runs jobs[0 .. n) on up to `max_threads` host threads (the calling thread is one of them).  A job that throws
// (std::bad_alloc from a packer's temporary) stops the remaining ones; the first exception is rethrown here.
inline void run_host_jobs(const std::vector<std::function<void()>>& jobs, unsigned max_threads = 16) {
    const size_t n = jobs.size();
    if (n == 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    unsigned nt = hw < max_threads ? hw : max_threads;
    if (nt > n) nt = (unsigned)n;
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr err;
    std::atomic<bool> err_set{false};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n || failed.load(std::memory_order_relaxed)) return;
            try {
                jobs[i]();
            } catch (...) {
                failed.store(true);
                bool expected = false;
                if (err_set.compare_exchange_strong(expected, true)) err = std::current_exception();
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    if (nt > 1) {
        pool.reserve(nt - 1);
        for (unsigned t = 0; t + 1 < nt; ++t) {
            try {
                pool.emplace_back(worker);
            } catch (...) {
                break;                      // no more threads to be had: the ones that exist (and this one) do the work
            }
        }
    }
    worker();
    for (auto& th : pool) th.join();
    if (err_set.load()) std::rethrow_exception(err);
}

}  // namespace iris
