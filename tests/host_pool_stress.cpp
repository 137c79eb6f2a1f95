// ThreadSanitizer stress of slam_amd/csrc/host_pool.h (built and run by tests/test_host_pool_cpu.py):
// jobs of alternating sizes back to back, as slam_track_pairs issues them (P/64-way replay, then rows/16-way later
// rounds); every chunk must run exactly once and no chunk may still be running when run() returns; a chunk that throws
// reaches the caller as an exception, not the process as std::terminate.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include "../slam_amd/csrc/host_pool.h"

int main() {
    HostPool pool(8);
    const int sizes[] = {8, 2, 7, 1, 8, 3, 5, 8, 2, 8};
    std::vector<int> hits(4096);
    long bad = 0;
    for (int it = 0; it < 20000; ++it) {
        const int n = 64 + (it % 7) * 37;
        const int grain = sizes[it % 10] * 4;
        for (int i = 0; i < n; ++i) hits[i] = 0;
        std::atomic<int> running{0};
        parallel_for(pool, n, grain, [&](int a, int e) {
            running.fetch_add(1);
            for (int i = a; i < e; ++i) ++hits[i];          // plain writes: a chunk handed out twice is a TSAN race
            running.fetch_sub(1);
        });
        if (running.load() != 0) ++bad;                      // a chunk outlived run()
        for (int i = 0; i < n; ++i) bad += hits[i] != 1;
    }
    // calls from several host threads queue on the pool
    std::vector<std::thread> callers;
    std::atomic<long> total{0};
    for (int t = 0; t < 4; ++t)
        callers.emplace_back([&] {
            for (int it = 0; it < 2000; ++it)
                parallel_for(pool, 100 + it % 13, 4, [&](int a, int e) { total.fetch_add(e - a); });
        });
    for (auto &t : callers) t.join();
    long want = 0;
    for (int it = 0; it < 2000; ++it) want += 4 * (100 + it % 13);
    if (total.load() != want) ++bad;
    // a job that throws: the exception comes back on the caller after every chunk has run, no thread is lost and the
    // pool serves the next job as before
    for (int it = 0; it < 200; ++it) {
        std::atomic<int> ran{0};
        bool caught = false;
        try {
            parallel_for(pool, 64, 4, [&](int a, int) {
                ran.fetch_add(1);
                if (a == 8 * (it % 8)) throw std::runtime_error("chunk failed");
            });
        } catch (const std::runtime_error &) {
            caught = true;
        }
        if (!caught || ran.load() != 8) ++bad;
        std::atomic<int> sum{0};
        parallel_for(pool, 64, 4, [&](int a, int e) { sum.fetch_add(e - a); });
        if (sum.load() != 64) ++bad;
    }
    std::printf("bad=%ld\n", bad);
    return bad ? 1 : 0;
}
