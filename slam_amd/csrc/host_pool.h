// Host worker pool for the per-pair host steps of slam_track_pairs (the replay of the adaptive RANSAC loop).
//
// Persistent: creating and joining 16 threads per call costs several hundred microseconds, which sits on the critical
// path between a chunk's RANSAC download and its mask / refit launch.  One job at a time (calls from different host
// threads queue on `gate`); the calling thread works too.
//
// Every field a worker reads (fn, total, next, pending, generation) is read and written under `m`, and a chunk is
// claimed together with a check that the job it belongs to is still the current one: a worker that woke for job G and
// was preempted before claiming anything finds generation != G when it resumes and claims nothing, so a chunk can
// neither run twice nor run after run() has returned.  (Round 1 claimed chunks with a lock-free fetch_add on `next`
// against a plain `total`, which a stale worker could read while run() was setting up the next job.)
//
// Plain C++ (no HIP): tests/host_pool_stress.cpp builds this header with -fsanitize=thread.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class HostPool {
public:
    // The process-wide pool is created on first use and NEVER destroyed: a static destructor would join the workers
    // from inside library finalisation (__cxa_finalize), after other libraries -- the HIP runtime, a profiler's tool
    // library -- may already be gone and while their own exit handlers run.  Callers that want the threads stopped
    // before the process ends call shutdown() (slam_shutdown(); slam_amd/_lib.py does, from a Python atexit hook);
    // otherwise the workers stay parked on `cv_work`, which then outlives them, until the process exits.
    static HostPool &get() {
        static HostPool *pool = new HostPool(default_threads());
        return *pool;
    }
    explicit HostPool(int n_threads) {
        try {
            for (int t = 1; t < n_threads; ++t) threads.emplace_back([this] { loop(); });
        } catch (...) {
            // the system refused another thread: the pool works with those it has (the caller is always one of them)
        }
        n_workers = (int)threads.size() + 1;
    }
    ~HostPool() { shutdown(); }      // pools with automatic storage (tests); the process-wide one is never destroyed
    // Stops and joins the workers (waiting for a job in flight first).  Idempotent.  The pool stays usable: run() then
    // executes every chunk on the calling thread.
    void shutdown() {
        std::lock_guard<std::mutex> serial(gate);
        {
            std::lock_guard<std::mutex> lk(m);
            if (stop) return;
            stop = true;
            ++generation;
        }
        cv_work.notify_all();
        n_workers = 1;
        for (auto &t : threads) t.join();
        threads.clear();
    }
    HostPool(const HostPool &) = delete;
    HostPool &operator=(const HostPool &) = delete;

    // runs job(0) .. job(n_chunks - 1), each exactly once, on the pool's threads and the caller; returns when all are
    // done.  A job that throws does not take its thread (and the process) down: the first exception is kept and thrown
    // again here, on the caller, after every chunk has finished.
    void run(int n_chunks, const std::function<void(int)> &job) {
        if (n_chunks <= 0) return;
        std::lock_guard<std::mutex> serial(gate);
        unsigned long gen;
        {
            std::lock_guard<std::mutex> lk(m);
            fn = &job;
            total = n_chunks;
            next = 0;
            pending = n_chunks;
            failure = nullptr;
            gen = ++generation;
        }
        cv_work.notify_all();
        work(gen);
        std::exception_ptr failed;
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return pending == 0; });
            fn = nullptr;
            total = 0;
            failed = failure;
            failure = nullptr;
        }
        if (failed) std::rethrow_exception(failed);
    }
    int workers() const { return n_workers.load(std::memory_order_relaxed); }

    // Threads per process: the host's cores shared between the ranks of this node (torchrun exports LOCAL_WORLD_SIZE),
    // at most 16 (a one-GPU box's CPU share); SLAM_HOST_THREADS overrides.
    static int default_threads() {
        if (const char *e = std::getenv("SLAM_HOST_THREADS")) {
            const int v = std::atoi(e);
            if (v >= 1) return v > 64 ? 64 : v;
        }
        unsigned hw = std::thread::hardware_concurrency();
        int nt = (int)(hw ? hw : 4);
        int local_world = 1;
        if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) local_world = std::atoi(e) > 0 ? std::atoi(e) : 1;
        nt /= local_world;
        if (nt > 16) nt = 16;
        return nt < 1 ? 1 : nt;
    }

private:
    void work(unsigned long gen) {
        for (;;) {
            const std::function<void(int)> *f;
            int c;
            {
                std::lock_guard<std::mutex> lk(m);
                if (generation != gen || next >= total) return;     // not this job any more, or nothing left of it
                c = next++;
                f = fn;
            }
            std::exception_ptr thrown;
            try {
                (*f)(c);
            } catch (...) {
                thrown = std::current_exception();
            }
            std::lock_guard<std::mutex> lk(m);
            if (thrown && !failure) failure = thrown;
            if (--pending == 0) cv_done.notify_all();
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return generation != seen; });
                seen = generation;
                if (stop) return;
            }
            work(seen);
        }
    }
    std::vector<std::thread> threads;
    std::atomic<int> n_workers{1};
    std::mutex gate, m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *fn = nullptr;
    std::exception_ptr failure;
    int next = 0, total = 0, pending = 0;
    unsigned long generation = 0;
    bool stop = false;
};

// fn(a, e) over [0, n) cut into at most workers() ranges of at least `grain` items
template <typename F>
void parallel_for(HostPool &pool, int n, int grain, F fn) {
    int nt = pool.workers();
    if (nt > (n + grain - 1) / grain) nt = (n + grain - 1) / grain;
    if (nt <= 1) {
        if (n > 0) fn(0, n);
        return;
    }
    const int per = (n + nt - 1) / nt;
    const std::function<void(int)> job = [&](int c) {
        const int a = c * per, e = a + per < n ? a + per : n;
        if (a < e) fn(a, e);
    };
    pool.run(nt, job);
}
