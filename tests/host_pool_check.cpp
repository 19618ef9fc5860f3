// Stress check of the host packing pool (hg_host_pack.hpp): three threads submit jobs at once (busy callers fall back to their
// own threads), every part of every job must run exactly once; a forked child must get a pool of its own; the pooled
// packing must equal the single-threaded one.  Built and run by tests/test_host_pool.py.
#include "hg_host_pack.hpp"
#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>
int main() {
    using hg::hostpack::pool;
    std::atomic<long long> bad{0}, busy{0}, jobs{0};
    auto hammer = [&](int seed) {
        unsigned x = seed * 2654435761u + 1;
        for (int it = 0; it < 20000; ++it) {
            x = x * 1664525u + 1013904223u;
            const int parts = 2 + (x >> 16) % 40;
            std::vector<std::atomic<int>> hit(parts);
            for (auto& h : hit) h = 0;
            auto f = [&](int i) { hit[i]++; if ((x >> 8) % 7 == 0) for (volatile int k = 0; k < 200; ++k) {} };
            if (!pool().run(parts, f)) { ++busy; continue; }
            ++jobs;
            for (int i = 0; i < parts; ++i) if (hit[i] != 1) ++bad;
        }
    };
    std::thread ta(hammer, 1), tb(hammer, 2), tc(hammer, 3);
    ta.join(); tb.join(); tc.join();
    printf("jobs %lld busy %lld bad %lld\n", (long long)jobs, (long long)busy, (long long)bad);
    // a forked child starts its own workers
    pid_t pid = fork();
    if (pid == 0) {
        std::vector<std::atomic<int>> hit(16);
        for (auto& h : hit) h = 0;
        const bool ok = pool().run(16, [&](int i) { hit[i]++; });
        int wrong = 0;
        for (int i = 0; i < 16; ++i) wrong += hit[i] != 1;
        printf("child: ran %d wrong %d\n", (int)ok, wrong);
        _exit(ok && !wrong ? 0 : 1);
    }
    int st = 0; waitpid(pid, &st, 0);
    printf("child exit %d\n", WEXITSTATUS(st));
    // host_pack itself against a scalar restatement
    const long long n = 300001; const int b = 64, C = 10;
    std::vector<float> x((size_t)n * b); std::vector<int64_t> lab((size_t)n * C);
    unsigned s = 7; for (auto& v : x) { s = s * 1664525u + 1013904223u; v = (s >> 20) & 1 ? 1.f : -1.f; }
    for (auto& v : lab) { s = s * 1664525u + 1013904223u; v = (s >> 21) & 1; }
    std::vector<uint32_t> c1((size_t)n * 2), c2((size_t)n * 2); std::vector<uint64_t> l1(n), l2(n);
    hg::HostPackCensus cs1, cs2;
    hg::host_pack(x.data(), lab.data(), n, b, C, c1.data(), l1.data(), &cs1, 24);
    hg::host_pack(x.data(), lab.data(), n, b, C, c2.data(), l2.data(), &cs2, 1);
    printf("pack equal %d census %lld %lld\n", (int)(c1 == c2 && l1 == l2), cs1.minus_ones, cs2.minus_ones);
    return bad != 0 || WEXITSTATUS(st) != 0 || !(c1 == c2 && l1 == l2) || cs1.minus_ones != cs2.minus_ones;
}
