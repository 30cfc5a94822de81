// par.h — the host mirror's worker pool: parallel_for over independent items (contigs of the split-read pass, regions of the
// copy-number pass, (contig, type) sets of the final merges). Persistent threads (a std::thread start costs tens of microseconds, a
// whole-genome step has a dozen parallel sections of a millisecond each); results never depend on the thread count, every item
// writes only its own slot. The reference's counterpart is its ThreadPool over chromosomes (include/ThreadPool.h, sv_caller.cpp:827-863).
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstddef>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace csvhost {

// CSV_TRACE=1 in the environment: wall time of named sections to stderr (where a whole-genome step's host time goes)
struct TraceScope {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    static bool on() { static const bool v = [] { const char *e = getenv("CSV_TRACE"); return e && *e && *e != '0'; }(); return v; }
    explicit TraceScope(const char *n) : name(n) { if (on()) { (void)origin(); t0 = std::chrono::steady_clock::now(); } }
    // (the second figure: when the section began, in ms since the first traced section of the process — sections of different threads overlap)
    static std::chrono::steady_clock::time_point origin() { static const auto o = std::chrono::steady_clock::now(); return o; }
    ~TraceScope()
    {
        if (!on()) return;
        const auto o = origin();
        fprintf(stderr, "[csv trace] %-36s %8.3f ms  @ %10.3f\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                std::chrono::duration<double, std::milli>(t0 - o).count());
    }
};

class HostPool {
public:
    // Two pools: a thread that has called use_second_pool() gets the second one, so that two independent chains of a run (the CIGAR
    // copy-number pass and the split-read chain) can each keep their parallel sections parallel while they run side by side.
    static HostPool &instance()
    {
        static HostPool p[2];
        return p[second_pool_flag() ? 1 : 0];
    }
    static bool &second_pool_flag() { static thread_local bool f = false; return f; }

    // f(i) for every i in [0, n), on up to `threads` threads (0: all of the pool's), the caller's included; returns when all are done.
    // Items are handed out one at a time in index order (put the heavy ones first). The first exception is rethrown here.
    // A section is over when its last ITEM is done, not when every worker has woken up: a worker that arrives late finds nothing left.
    template <class F>
    void parallel_for(size_t n, int threads, F &&f)
    {
        if (n == 0) return;
        size_t T = threads > 0 ? (size_t)threads : workers_.size() + 1;
        T = std::min(T, std::min(n, workers_.size() + 1));
        if (T <= 1 || busy_.exchange(true)) {                    // one thread asked for, or nested / concurrent use (the flag stays its owner's): run inline
            for (size_t i = 0; i < n; i++) f(i);
            return;
        }
        std::shared_ptr<Job> job = std::make_shared<Job>();
        job->n = n;
        job->fn = [&](size_t i) { f(i); };
        {
            std::lock_guard<std::mutex> l(mu_);
            job_ = job;
            want_ = T - 1;
            generation_++;
        }
        if (T - 1 >= workers_.size()) cv_.notify_all();
        else for (size_t k = 0; k + 1 < T; k++) cv_.notify_one();
        run(*job);
        {
            std::unique_lock<std::mutex> l(job->done_mu);
            job->done_cv.wait(l, [&] { return job->done.load(std::memory_order_acquire) == n; });
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            job_.reset();
        }
        busy_ = false;
        if (job->err) std::rethrow_exception(job->err);
    }

    size_t size() const { return workers_.size() + 1; }

private:
    struct Job {
        size_t n = 0;
        std::function<void(size_t)> fn;                          // refers to the caller's frame: only called for items, and the caller waits for every item
        std::atomic<size_t> next{0}, done{0};
        std::exception_ptr err;
        std::mutex err_mu, done_mu;
        std::condition_variable done_cv;
    };

    HostPool()
    {
        // CSV_HOST_THREADS, else the hardware's count capped at 16: the CPU share of one GPU on the boxes this runs on
        size_t n = 0;
        if (const char *e = getenv("CSV_HOST_THREADS")) n = (size_t)atoi(e);
        if (n == 0) { const unsigned hw = std::thread::hardware_concurrency(); n = hw ? hw : 4; if (n > 16) n = 16; }
        if (n > 256) n = 256;
        for (size_t t = 1; t < n; t++) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }

    static void run(Job &job)
    {
        size_t mine = 0;
        for (size_t i; (i = job.next.fetch_add(1, std::memory_order_relaxed)) < job.n;) {
            try { job.fn(i); } catch (...) {
                std::lock_guard<std::mutex> l(job.err_mu);
                if (!job.err) job.err = std::current_exception();
            }
            mine++;
        }
        if (mine && job.done.fetch_add(mine, std::memory_order_acq_rel) + mine == job.n) {
            std::lock_guard<std::mutex> l(job.done_mu);
            job.done_cv.notify_all();
        }
    }

    void loop()
    {
        size_t seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;
            }
            if (job) run(*job);
        }
    }

    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::shared_ptr<Job> job_;
    size_t want_ = 0, generation_ = 0;
    bool stop_ = false;
    std::atomic<bool> busy_{false};
};

template <class F>
inline void parallel_for(size_t n, int threads, F &&f) { HostPool::instance().parallel_for(n, threads, std::forward<F>(f)); }

// A few long-running tasks side by side on persistent threads — the lane drivers and merge workers of the per-chromosome pipeline,
// which block on device events and on each other and so cannot be items of a parallel_for. start() hands the task to an idle thread
// (a new one when none is idle), wait() returns when it has finished. A whole-genome step needs nine such threads; creating and joining
// them per step cost 1.6 ms of a 25 ms pass.
class WorkerThreads {
public:
    struct Worker {
        std::mutex mu;
        std::condition_variable cv;
        std::function<void()> fn;
        bool has_task = false, done = false, stop = false;
        std::thread th;
    };
    using Ticket = Worker *;

    static WorkerThreads &instance()
    {
        static WorkerThreads w;
        return w;
    }

    Ticket start(std::function<void()> fn)
    {
        Worker *w = nullptr;
        {
            std::lock_guard<std::mutex> l(mu_);
            if (!idle_.empty()) { w = idle_.back(); idle_.pop_back(); }
        }
        if (!w) {
            w = new Worker();
            w->th = std::thread([w] { loop(w); });
            std::lock_guard<std::mutex> l(mu_);
            all_.push_back(w);
        }
        {
            std::lock_guard<std::mutex> l(w->mu);
            w->fn = std::move(fn); w->has_task = true; w->done = false;
        }
        w->cv.notify_all();
        return w;
    }

    void wait(Ticket w)
    {
        {
            std::unique_lock<std::mutex> l(w->mu);
            w->cv.wait(l, [&] { return w->done; });
            w->done = false;
        }
        std::lock_guard<std::mutex> l(mu_);
        idle_.push_back(w);
    }

private:
    WorkerThreads() = default;
    ~WorkerThreads()
    {
        for (Worker *w : all_) {
            { std::lock_guard<std::mutex> l(w->mu); w->stop = true; }
            w->cv.notify_all();
            w->th.join();
            delete w;
        }
    }
    static void loop(Worker *w)
    {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> l(w->mu);
                w->cv.wait(l, [&] { return w->has_task || w->stop; });
                if (w->stop) return;
                fn = std::move(w->fn); w->has_task = false;
            }
            fn();                                         // (tasks catch their own exceptions)
            { std::lock_guard<std::mutex> l(w->mu); w->done = true; }
            w->cv.notify_all();
        }
    }
    std::mutex mu_;
    std::vector<Worker *> idle_, all_;
};

}  // namespace csvhost
