// par.h — the host mirror's worker pool: parallel_for over independent items (contigs of the split-read pass, regions of the
// copy-number pass, (contig, type) sets of the final merges). Persistent threads (a std::thread start costs tens of microseconds, a
// whole-genome step has a dozen parallel sections of a millisecond each); results never depend on the thread count, every item
// writes only its own slot. The reference's counterpart is its ThreadPool over chromosomes (include/ThreadPool.h, sv_caller.cpp:827-863).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace csvhost {

class HostPool {
public:
    static HostPool &instance()
    {
        static HostPool p;
        return p;
    }

    // f(i) for every i in [0, n), on up to `threads` threads (0: all of the pool's), the caller's included; returns when all are done.
    // Items are handed out one at a time in index order (put the heavy ones first). The first exception is rethrown here.
    template <class F>
    void parallel_for(size_t n, int threads, F &&f)
    {
        if (n == 0) return;
        size_t T = threads > 0 ? (size_t)threads : workers_.size() + 1;
        T = std::min(T, std::min(n, workers_.size() + 1));
        if (T <= 1 || busy_.exchange(true)) {                    // nested or concurrent use: run inline
            const bool mine = T > 1;
            for (size_t i = 0; i < n; i++) f(i);
            if (mine) busy_ = false;
            return;
        }
        Job job;
        job.n = n;
        job.fn = [&](size_t i) { f(i); };
        {
            std::lock_guard<std::mutex> l(mu_);
            job_ = &job;
            want_ = T - 1;
            generation_++;
        }
        cv_.notify_all();
        run(job);
        {
            std::unique_lock<std::mutex> l(mu_);
            done_cv_.wait(l, [&] { return job.active == 0 && job.entered == want_; });
            job_ = nullptr;
        }
        busy_ = false;
        if (job.err) std::rethrow_exception(job.err);
    }

    size_t size() const { return workers_.size() + 1; }

private:
    struct Job {
        size_t n = 0;
        std::function<void(size_t)> fn;
        std::atomic<size_t> next{0};
        size_t active = 0, entered = 0;                          // guarded by mu_
        std::exception_ptr err;
        std::mutex err_mu;
    };

    HostPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        size_t n = hw ? hw : 4;
        if (n > 32) n = 32;                                      // the CPU share of one GPU on the boxes this runs on is 16
        for (size_t t = 1; t < n; t++) workers_.emplace_back([this, t] { loop(t); });
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }

    void run(Job &job)
    {
        for (size_t i; (i = job.next.fetch_add(1, std::memory_order_relaxed)) < job.n;) {
            try { job.fn(i); } catch (...) {
                std::lock_guard<std::mutex> l(job.err_mu);
                if (!job.err) job.err = std::current_exception();
            }
        }
    }

    void loop(size_t id)
    {
        size_t seen = 0;
        for (;;) {
            Job *job = nullptr;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                if (!job_ || id > want_) continue;               // this section wants fewer threads
                job = job_;
                job->active++; job->entered++;
            }
            run(*job);
            {
                std::lock_guard<std::mutex> l(mu_);
                job->active--;
            }
            done_cv_.notify_all();
        }
    }

    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    Job *job_ = nullptr;
    size_t want_ = 0, generation_ = 0;
    bool stop_ = false;
    std::atomic<bool> busy_{false};
};

template <class F>
inline void parallel_for(size_t n, int threads, F &&f) { HostPool::instance().parallel_for(n, threads, std::forward<F>(f)); }

}  // namespace csvhost
