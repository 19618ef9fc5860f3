// hashgan_amd -- host side of the hand-over of forward_all()'s arrays (main.py:151-158): float32 features [n][b] and
// int64 labels [n][C] become packed code / label words BEFORE they cross PCIe.  339 MB of raw arrays at C2 cost 17 ms
// of upload; their packed form is 16 MB.  A pool of host threads streams the arrays once (AVX-512 compares give 16
// sign bits per load), producing the device layout directly -- codes uint32 [n][ceil(b/32)], labels uint64
// [n][ceil(C/64)] -- plus the census the caller needs to tell +-1 codes, {0,1} bits and real-valued features apart.
// Same bits as k_pack_sign_f32 / k_pack_labels_i64 (bit = x > 0, bit = label != 0): tests/test_hip_parity.py.
#pragma once
#include <stdint.h>

namespace hg {

struct HostPackCensus {
    long long nonbinary = 0;     // feature entries outside {-1, 0, +1} (NaN included)
    long long zeros = 0;
    long long minus_ones = 0;
    long long bad_labels = 0;    // label entries outside {0, 1}
};

// x may be null (labels only) and lab may be null (codes only).  threads <= 0: pick from the hardware.
void host_pack(const float* x, const int64_t* lab, long long n, int b, int C, uint32_t* codes, uint64_t* labels,
               HostPackCensus* census, int threads);
// the same with the calling thread SHIPPING finished row prefixes instead of packing: (*on_rows)(rows_done), rows_done growing
void host_pack_ship(const float* x, const int64_t* lab, long long n, int b, int C, uint32_t* codes, uint64_t* labels,
                    HostPackCensus* census, int threads, void (*on_rows)(void*, long long), void* on_rows_arg);
// rows [r0, r1) of x [n][b] copied to dst [r1 - r0][bpad] (pad columns zeroed) by the pool's threads: how the float table
// reaches pinned staging memory at memory speed before it crosses PCIe (a pageable source is staged by the runtime at ~25 GB/s)
void host_copy_rows(const float* x, long long r0, long long r1, int b, int bpad, float* dst, int threads, int which_pool = 0);
void host_pack_warm();      // start the pools' threads now (hg_preload)

}  // namespace hg

#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <sys/syscall.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace hg {
namespace hostpack {

inline void rows_scalar(const float* x, const int64_t* lab, long long r0, long long r1, int b, int C, uint32_t* codes,
                        uint64_t* labels, HostPackCensus& cs) {
    const int NW = (b + 31) / 32, LW = (C + 63) / 64;
    for (long long r = r0; r < r1; ++r) {
        if (x) {
            const float* row = x + r * b;
            for (int w = 0; w < NW; ++w) {
                uint32_t v = 0;
                const int hi = b - w * 32 < 32 ? b - w * 32 : 32;
                for (int j = 0; j < hi; ++j) {
                    const float f = row[w * 32 + j];
                    v |= (uint32_t)(f > 0.0f) << j;
                    cs.zeros += f == 0.0f;
                    cs.minus_ones += f == -1.0f;
                    cs.nonbinary += !(f == 1.0f || f == -1.0f || f == 0.0f);
                }
                codes[r * NW + w] = v;
            }
        }
        if (lab) {
            const int64_t* row = lab + r * C;
            for (int w = 0; w < LW; ++w) {
                uint64_t v = 0;
                const int hi = C - w * 64 < 64 ? C - w * 64 : 64;
                for (int j = 0; j < hi; ++j) {
                    const int64_t l = row[w * 64 + j];
                    v |= (uint64_t)(l != 0) << j;
                    cs.bad_labels += !(l == 0 || l == 1);
                }
                labels[r * LW + w] = v;
            }
        }
    }
}

__attribute__((target("avx512f,avx512bw,avx512vl,popcnt")))
inline void rows_avx512(const float* x, const int64_t* lab, long long r0, long long r1, int b, int C, uint32_t* codes,
                        uint64_t* labels, HostPackCensus& cs) {
    const int NW = (b + 31) / 32, LW = (C + 63) / 64;
    const __m512 zero = _mm512_setzero_ps(), one = _mm512_set1_ps(1.0f), mone = _mm512_set1_ps(-1.0f);
    long long nz = 0, nm = 0, np = 0, nbad = 0;
    for (long long r = r0; r < r1; ++r) {
        if (x) {
            const float* row = x + r * b;
            for (int w = 0; w < NW; ++w) {
                uint32_t v = 0;
                for (int half = 0; half < 2; ++half) {
                    const int k = w * 32 + half * 16;
                    if (k >= b) break;
                    const int left = b - k;
                    const __mmask16 m = left >= 16 ? (__mmask16)0xFFFF : (__mmask16)((1u << left) - 1u);
                    const __m512 f = _mm512_maskz_loadu_ps(m, row + k);
                    v |= (uint32_t)_mm512_mask_cmp_ps_mask(m, f, zero, _CMP_GT_OQ) << (16 * half);
                    nz += _mm_popcnt_u32(_mm512_mask_cmp_ps_mask(m, f, zero, _CMP_EQ_OQ));
                    nm += _mm_popcnt_u32(_mm512_mask_cmp_ps_mask(m, f, mone, _CMP_EQ_OQ));
                    np += _mm_popcnt_u32(_mm512_mask_cmp_ps_mask(m, f, one, _CMP_EQ_OQ));
                }
                codes[r * NW + w] = v;
            }
        }
        if (lab) {
            const int64_t* row = lab + r * C;
            const __m512i z = _mm512_setzero_si512(), o = _mm512_set1_epi64(1);
            for (int w = 0; w < LW; ++w) {
                uint64_t v = 0;
                for (int e = 0; e < 8; ++e) {
                    const int k = w * 64 + e * 8;
                    if (k >= C) break;
                    const int left = C - k;
                    const __mmask8 m = left >= 8 ? (__mmask8)0xFF : (__mmask8)((1u << left) - 1u);
                    const __m512i l = _mm512_maskz_loadu_epi64(m, row + k);
                    const __mmask8 nzm = _mm512_mask_cmpneq_epi64_mask(m, l, z);
                    v |= (uint64_t)nzm << (8 * e);
                    nbad += _mm_popcnt_u32(_mm512_mask_cmpneq_epi64_mask(nzm, l, o));
                }
                labels[r * LW + w] = v;
            }
        }
    }
    cs.zeros += nz;
    cs.minus_ones += nm;
    cs.bad_labels += nbad;
    if (x) cs.nonbinary += (r1 - r0) * (long long)b - nz - nm - np;
}

}  // namespace hostpack

namespace hostpack {

inline long futex(std::atomic<uint32_t>* addr, int op, uint32_t val) {
    return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), op | FUTEX_PRIVATE_FLAG, val, nullptr, nullptr, 0);
}

// NUMA: which node holds a host array, and which processors belong to a node.  On the GPU box (two sockets) a packing pass whose threads
// sit on the OTHER node than the caller's arrays takes 3.4 - 3.6 ms instead of 1.5 - 1.7 at C2 (tools/numa_probe.py) -- and where the
// scheduler puts a pool's threads is luck: whole boxes measured 4.5 ms per literal call where others measured 2.6.  A pool's workers are
// therefore confined to the node of the array they are about to read (the process's own affinity mask permitting).
inline int node_of_address(const void* p) {
    int node = -1;
    const long r = syscall(SYS_get_mempolicy, &node, nullptr, 0ul, (unsigned long)(uintptr_t)p, 3ul /* MPOL_F_NODE | MPOL_F_ADDR */);
    return r == 0 ? node : -2;                                           // -2: the call is not available (seccomp): ask where the caller runs
}
inline int node_of_range(const void* base, size_t bytes, size_t min_bytes = (size_t)32 << 20) {   // -1: mixed or unknown (no confinement); -3: no opinion
    if (!base || bytes < min_bytes) return -3;                           // small arrays: not worth moving threads for -- the pool stays where it is
    int node = -1;
    for (int k = 0; k < 5; ++k) {
        const int n = node_of_address((const char*)base + (bytes - 1) / 4 * k);
        if (n == -2) {                                                   // the caller's own node: where its arrays were most likely first touched
            unsigned cpu = 0, nd = 0;
            return syscall(SYS_getcpu, &cpu, &nd, nullptr) == 0 ? (int)nd : -1;
        }
        if (n < 0) return -1;
        if (k == 0) node = n; else if (n != node) return -1;
    }
    return node;
}
inline bool node_cpus(int node, cpu_set_t* out) {                        // the node's processors that this process may use
    CPU_ZERO(out);
    char path[64];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[1024] = {0};
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int n = 0;
    char* save = nullptr;                                                // (two pools may ask at once: the packing pool and the staging thread's)
    for (char* tok = strtok_r(buf, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
        int a = 0, b = 0;
        const int got = sscanf(tok, "%d-%d", &a, &b);
        if (got < 1) continue;
        if (got == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) { CPU_SET(c, out); ++n; }
    }
    return n >= 8;                                                       // (a handful of processors would be worse than the wrong node)
}

// Workers that outlive a call: starting and joining 31 threads cost 0.5 ms of a 4 ms call at C2 (17 us each on the GPU
// box's EPYC).  One job at a time (a second caller in another thread finds the pool busy and starts threads of its own,
// as every call did before).
// Sleepers wait on a futex word (the job generation) and are woken by ONE system call; nobody takes a mutex on the way in
// or out: with a condition variable a hundred woken workers queue for its mutex one after the other, and the call
// started 0.3 - 0.5 ms late (round 3: hg_set_database_f32 at C2 2.0 -> 1.6 ms with 96 threads).  A job lives in one of two
// slots (generation parity); parts are claimed from {generation, next part} with compare-and-swap, so a worker that wakes
// up late can never run a part of a newer job with an older job's arguments.
class Pool {
public:
    // f(0) .. f(parts - 1); the caller takes parts too unless `caller_waits` hands it another duty: then it runs
    // on_progress(done_parts) in a loop (done_parts: how many of the LOWEST-numbered parts are complete, monotone) until all
    // are -- parts are claimed in ascending order, so a caller can ship finished prefixes while the rest is in the works.
    // max_helpers: worker threads that may take parts (0: one per part).  With a shipping caller the parts outnumber the threads
    // meant to work on them (four per thread, so that prefixes complete early): until round 6 every PART got a thread -- 256 of
    // them on the GPU box's 256 hardware threads, next to the caller, the staging thread and its pool -- and one call in fifteen
    // waited 20 - 70 ms for a worker that had claimed a part and lost its processor (tools/literal_outliers.py).
    // confine the workers to `node`'s processors (-1: wherever the process may run) from the next job on; a busy pool keeps what it has
    void set_node(int node) {
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock() || node == node_) return;
        cpu_set_t mask;
        if (node >= 0) { if (!node_cpus(node, &mask)) return; }
        else if (sched_getaffinity(0, sizeof mask, &mask) != 0) return;
        int failed = 0;
        for (auto& t : th_) failed += pthread_setaffinity_np(t.native_handle(), sizeof mask, &mask) != 0;
        static const bool trace = getenv("HG_PACK_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[hg pack] pool %p: %zu workers -> node %d (%d processors; %d refused)\n", (void*)this, th_.size(), node, CPU_COUNT(&mask), failed);
        mask_ = mask; node_ = node;
    }
    template <class F> bool run(int parts, const F& f, int max_helpers = 0) { return run_impl(parts, f, (void (*)(void*, int))nullptr, nullptr, max_helpers); }
    template <class F, class P> bool run_progress(int parts, const F& f, const P& on_progress, int max_helpers = 0) {
        struct PC { const P* p; } pc{&on_progress};
        return run_impl(parts, f, +[](void* c, int done) { (*static_cast<PC*>(c)->p)(done); }, &pc, max_helpers);
    }
private:
    struct Job { void (*call)(void*, int) = nullptr; void* arg = nullptr; int parts = 0; };
    template <class F> bool run_impl(int parts, const F& f, void (*progress)(void*, int), void* parg, int max_helpers) {
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock()) return false;
        int helpers = progress ? parts : parts - 1;                      // (a caller that ships prefixes packs nothing itself)
        if (max_helpers > 0 && helpers > max_helpers) helpers = max_helpers;
        if (progress && helpers < 1) helpers = 1;
        while ((int)th_.size() < helpers) {
            const uint32_t seen = gen_.load();
            th_.emplace_back([this, seen] { worker(seen); });
            if (node_ >= 0) (void)pthread_setaffinity_np(th_.back().native_handle(), sizeof mask_, &mask_);
        }
        struct Ctx { const F* f; } ctx{&f};
        if ((int)flags_.size() < parts) flags_ = std::vector<std::atomic<unsigned char>>((size_t)parts);
        for (int i = 0; i < parts; ++i) flags_[(size_t)i].store(0, std::memory_order_relaxed);
        const uint32_t g = gen_.load(std::memory_order_relaxed) + 1;
        Job& j = jobs_[g & 1];
        j.call = [](void* c, int i) { (*static_cast<Ctx*>(c)->f)(i); };
        j.arg = &ctx;
        j.parts = parts;
        left_.store(parts, std::memory_order_relaxed);
        next_.store(((unsigned long long)g << 32), std::memory_order_release);
        gen_.store(g, std::memory_order_release);
        futex(&gen_, FUTEX_WAKE, helpers < (int)th_.size() ? (uint32_t)helpers : 0x7FFFFFFFu);     // one system call wakes them
        if (!progress) {
            work_on(g, j);
        } else {
            int done = 0;
            while (done < parts) {
                int d = done;
                while (d < parts && flags_[(size_t)d].load(std::memory_order_acquire)) ++d;
                if (d != done) { done = d; progress(parg, done); }
                else __builtin_ia32_pause();
            }
        }
        // wait for the parts others still run (short: spin, then sleep on the counter)
        for (int spin = 0; left_.load(std::memory_order_acquire) != 0; ++spin) {
            if (spin < 4000) { __builtin_ia32_pause(); continue; }
            const uint32_t v = left_.load(std::memory_order_acquire);
            if (v) futex(&left_, FUTEX_WAIT, v);
        }
        return true;
    }
    void work_on(const uint32_t g, const Job& j) {
        for (;;) {
            unsigned long long v = next_.load(std::memory_order_acquire);
            if ((uint32_t)(v >> 32) != g || (int)(v & 0xFFFFFFFFull) >= j.parts) break;
            if (!next_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) continue;
            const int i = (int)(v & 0xFFFFFFFFull);
            j.call(j.arg, i);                                             // (the job is still g: its caller waits for this part)
            flags_[(size_t)i].store(1, std::memory_order_release);
            if (left_.fetch_sub(1, std::memory_order_acq_rel) == 1) futex(&left_, FUTEX_WAKE, 1);
        }
    }
    void worker(uint32_t seen) {
        for (;;) {
            uint32_t g;
            while ((g = gen_.load(std::memory_order_acquire)) == seen) futex(&gen_, FUTEX_WAIT, seen);
            seen = g;
            const Job j = jobs_[g & 1];                                   // (a copy that may be torn if the job is over already: then no claim succeeds)
            work_on(g, j);
        }
    }
    std::mutex job_mu_;
    int node_ = -1;                                                        // the node the workers are confined to (set_node)
    cpu_set_t mask_;
    std::vector<std::thread> th_;
    Job jobs_[2];
    std::vector<std::atomic<unsigned char>> flags_;                       // part i done (only resized under job_mu_, between jobs)
    std::atomic<unsigned long long> next_{0};                             // {generation, next part}
    std::atomic<uint32_t> gen_{0}, left_{0};
};
// The pool is never torn down (its workers sleep until the process ends).  A forked child has none of the workers and must
// not touch the parent's mutexes and condition variables (a broadcast on the copy of one with sleepers never returns):
// the fork handler drops the pointer, the child's first call builds a pool of its own.
// which = 0: the packing pool; 1: a second, small one that copies float chunks into pinned staging WHILE the first packs
inline std::atomic<Pool*>& pool_slot(int which) { static std::atomic<Pool*> slot[2]{{nullptr}, {nullptr}}; return slot[which]; }
inline Pool& pool(int which = 0) {
    static const int at_fork = pthread_atfork(nullptr, nullptr, +[] { pool_slot(0).store(nullptr); pool_slot(1).store(nullptr); });
    (void)at_fork;
    Pool* p = pool_slot(which).load(std::memory_order_acquire);
    if (!p) {
        Pool* fresh = new Pool;                                           // (no threads yet: they start with the first job)
        if (pool_slot(which).compare_exchange_strong(p, fresh)) p = fresh;
        else delete fresh;
    }
    return *p;
}

}  // namespace hostpack

// Threads a pool may put to work at once.  Two limits beyond the hardware's count: the process's affinity mask, and the cgroup's CPU
// quota -- the GPU box gives a 256-hardware-thread machine a quota of 16 CPUs (cpu.max "1600000 100000"): 64 packing threads in
// back-to-back calls ran it dry and one call in twenty stood still for 20 - 70 ms until the next period (tools/literal_outliers.py;
// with 32 threads none did, and the median call was faster too).  Twice the quota: a call's burst is a millisecond or two.
inline int default_pack_threads() {
    static const int cached = [] {
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0 && (unsigned)n < hw) hw = (unsigned)n; }
        int threads = hw <= 64 ? (int)std::min<unsigned>(hw ? hw : 1u, 32u) : (int)std::min<unsigned>(hw / 4, 64u);
        double quota = 0.0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            long long per = 0;
            if (fscanf(f, "%31s %lld", q, &per) == 2 && per > 0 && strcmp(q, "max") != 0) quota = (double)atoll(q) / (double)per;
            fclose(f);
        } else {                                                                  // cgroup v1
            long long q = -1, per = 0;
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
            if (q > 0 && per > 0) quota = (double)q / (double)per;
        }
        if (quota > 0.0) {
            const int lim = std::max(2, (int)(2.0 * quota + 0.5));
            if (threads > lim) threads = lim;
        }
        if (const char* e = getenv("HG_PACK_THREADS")) { const int v = atoi(e); if (v > 0) threads = v; }
        return threads;
    }();
    return cached;
}

// on_rows(rows_done): optional; called from the CALLING thread, with a growing count, whenever another prefix of the rows
// is packed (the caller then packs nothing itself: it ships those rows -- hipMemcpyAsync -- while the workers go on).
inline void host_pack_ship(const float* x, const int64_t* lab, long long n, int b, int C, uint32_t* codes, uint64_t* labels,
                           HostPackCensus* census, int threads, void (*on_rows)(void*, long long), void* on_rows_arg) {
    const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    if (threads <= 0) {
        // the pass is memory bound: on the GPU box (2 x 64 cores, 256 hardware threads) hg_set_database_f32 at C2 takes
        // 1.67 ms with 32 threads, 1.38 with 64, 1.39 with 96, 1.59 with 128, 1.86 with 192 (tools/h2d_split.py, round 3:
        // futex wake-up, four parts per thread claimed in order, prefixes shipped while the rest packs; with the
        // condition-variable pool and one part per thread it was 2.8 / 2.2 / 2.0 / 1.65 / 1.66)
        // (round 6: and never more than twice the cgroup's CPU quota: default_pack_threads)
        threads = default_pack_threads();
    }
    const long long bytes = n * ((long long)(x ? b * 4 : 0) + (lab ? C * 8 : 0));
    threads = (int)std::max<long long>(1, std::min<long long>(threads, bytes >> 20));    // at least ~1 MB of input per thread
    // with a shipping caller: eight parts per thread, claimed in ascending order, so prefixes complete early (round 6: 32 threads x 8 parts
    // measured 2.64 - 2.76 ms per C2 call against 2.88 - 3.30 with x 4: finer prefixes to ship, an evener finish)
    const int parts = on_rows && threads > 1 ? threads * 8 : threads;
    std::vector<HostPackCensus> part((size_t)parts);
    auto work = [&](int t) {
        const long long r0 = n * t / parts, r1 = n * (t + 1) / parts;
        if (wide) hostpack::rows_avx512(x, lab, r0, r1, b, C, codes, labels, part[(size_t)t]);
        else hostpack::rows_scalar(x, lab, r0, r1, b, C, codes, labels, part[(size_t)t]);
    };
    bool ran = false;
    if (parts > 1) {
        static const bool pin = !getenv("HG_PACK_NO_NUMA");
        if (pin) {
            const int node = x ? hostpack::node_of_range(x, (size_t)n * b * 4) : hostpack::node_of_range(lab, (size_t)n * C * 8);
            if (node != -3) hostpack::pool().set_node(node);
        }
    }
    if (parts == 1) {
        work(0);
        ran = true;
    } else if (on_rows) {
        static const int cap = getenv("HG_PACK_HELPERS") ? atoi(getenv("HG_PACK_HELPERS")) : 0;      // (experiments)
        ran = hostpack::pool().run_progress(parts, work, [&](int done) { on_rows(on_rows_arg, n * done / parts); }, cap > 0 ? cap : threads);
    } else {
        ran = hostpack::pool().run(parts, work);
    }
    if (!ran) {                                                           // the pool is busy with another caller's job
        std::vector<std::thread> own;
        const int nt = std::min(threads, parts);
        std::atomic<int> next{0};
        auto loop = [&] { for (int i; (i = next.fetch_add(1)) < parts;) work(i); };
        for (int t = 1; t < nt; ++t) own.emplace_back(loop);
        loop();
        for (auto& th : own) th.join();
    }
    if (on_rows) on_rows(on_rows_arg, n);
    HostPackCensus tot;
    for (const auto& p : part) {
        tot.nonbinary += p.nonbinary; tot.zeros += p.zeros; tot.minus_ones += p.minus_ones; tot.bad_labels += p.bad_labels;
    }
    if (census) *census = tot;
}

inline void host_copy_rows(const float* x, long long r0, long long r1, int b, int bpad, float* dst, int threads, int which_pool) {
    const long long rows = r1 - r0;
    if (rows <= 0) return;
    if (threads <= 0) threads = default_pack_threads();
    // a 16 MB chunk per call: sixteen threads copy it faster than PCIe takes it (the next chunk's copy runs under this one's
    // DMA); waking 64 for 256 KB each was slower -- 1M x 64 floats: 8.8 ms per hg_set_database_f32 against 7.0
    const long long bytes = rows * (long long)bpad * 4;
    threads = (int)std::max<long long>(1, std::min<long long>(std::min(threads, 16), bytes >> 20));
    auto work = [&](int t) {
        const long long a = rows * t / threads, e = rows * (t + 1) / threads;
        if (bpad == b) {
            memcpy(dst + a * bpad, x + (r0 + a) * b, (size_t)(e - a) * b * 4);
        } else {
            for (long long r = a; r < e; ++r) {
                memcpy(dst + r * bpad, x + (r0 + r) * b, (size_t)b * 4);
                memset(dst + r * bpad + b, 0, (size_t)(bpad - b) * 4);
            }
        }
    };
    if (threads > 1) {
        static const bool pin = !getenv("HG_PACK_NO_NUMA");
        if (pin) {
            const int node = hostpack::node_of_range(x + r0 * b, (size_t)rows * b * 4, (size_t)4 << 20);
            if (node != -3) hostpack::pool(which_pool).set_node(node);
        }
    }
    if (threads == 1 || !hostpack::pool(which_pool).run(threads, work))  // (a busy pool: this thread alone)
        for (int t = 0; t < threads; ++t) work(t);
}

// hg_preload: the pools' threads start with the first job that wants them (a few milliseconds for the packing pool's 4 x 64):
// an empty job of the largest shape starts them now
inline void host_pack_warm() {
    const int threads = default_pack_threads();
    auto nop = [](int) {};
    if (threads > 1) (void)hostpack::pool(0).run_progress(threads * 8, nop, [](int) {}, threads);
    (void)hostpack::pool(1).run(std::min(threads, 16), nop);
}

inline void host_pack(const float* x, const int64_t* lab, long long n, int b, int C, uint32_t* codes, uint64_t* labels,
                      HostPackCensus* census, int threads) {
    host_pack_ship(x, lab, n, b, C, codes, labels, census, threads, nullptr, nullptr);
}

}  // namespace hg
#endif
