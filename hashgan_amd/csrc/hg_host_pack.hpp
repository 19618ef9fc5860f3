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

}  // namespace hg

#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#include <pthread.h>
#include <algorithm>
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

// Workers that outlive a call: starting and joining 31 threads cost 0.5 ms of a 4 ms call at C2 (17 us each on the GPU
// box's EPYC).  One job at a time (a second caller in another thread finds the pool busy and starts threads of its own,
// as every call did before).
class Pool {
public:
    template <class F> bool run(int parts, const F& f) {                  // f(0) .. f(parts - 1), the caller takes part; false: busy
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock()) return false;
        while ((int)th_.size() < parts - 1) {
            const unsigned long long seen = gen_;                         // (gen_ only changes under job_mu_)
            th_.emplace_back([this, seen] { worker(seen); });
        }
        struct Ctx { const F* f; } ctx{&f};
        {
            std::lock_guard<std::mutex> lk(mu_);
            call_ = [](void* c, int i) { (*static_cast<Ctx*>(c)->f)(i); };
            arg_ = &ctx;
            parts_ = parts;
            left_ = parts - 1;
            ++gen_;
            next_.store((gen_ << 32) | 1ull);                             // {generation, next part}: a late worker cannot claim a part of a newer job
        }
        // wake as many workers as there are parts for them (a small job -- the queries of a resident database: 3 parts --
        // must not stampede a hundred sleepers through the mutex; those left asleep join a later job with a stale `seen`)
        if (2 * (parts - 1) >= (int)th_.size()) cv_work_.notify_all();
        else for (int i = 1; i < parts; ++i) cv_work_.notify_one();
        f(0);
        const unsigned long long mine = gen_ & 0xFFFFFFFFull;
        int done = 0;                                                     // the caller takes what the workers have not claimed yet
        for (;;) {
            unsigned long long v = next_.load();
            if ((v >> 32) != mine || (int)(v & 0xFFFFFFFFull) >= parts) break;
            if (next_.compare_exchange_weak(v, v + 1)) { f((int)(v & 0xFFFFFFFFull)); ++done; }
        }
        std::unique_lock<std::mutex> lk(mu_);
        left_ -= done;
        cv_done_.wait(lk, [this] { return left_ == 0; });
        return true;
    }
private:
    void worker(unsigned long long seen) {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            void (*call)(void*, int) = call_;
            void* arg = arg_;
            const int parts = parts_;
            lk.unlock();
            int done = 0;
            for (;;) {
                unsigned long long v = next_.load();
                if ((v >> 32) != (seen & 0xFFFFFFFFull) || (int)(v & 0xFFFFFFFFull) >= parts) break;
                if (next_.compare_exchange_weak(v, v + 1)) { call(arg, (int)(v & 0xFFFFFFFFull)); ++done; }
            }
            if (done) {                                                   // (then the job is still this one: its caller waits for these parts)
                lk.lock();
                left_ -= done;
                if (left_ == 0) cv_done_.notify_one();
            }
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> th_;
    void (*call_)(void*, int) = nullptr;
    void* arg_ = nullptr;
    int parts_ = 0, left_ = 0;
    std::atomic<unsigned long long> next_{0};
    unsigned long long gen_ = 0;
};
// The pool is never torn down (its workers sleep until the process ends).  A forked child has none of the workers and must
// not touch the parent's mutexes and condition variables (a broadcast on the copy of one with sleepers never returns):
// the fork handler drops the pointer, the child's first call builds a pool of its own.
inline std::atomic<Pool*>& pool_slot() { static std::atomic<Pool*> slot{nullptr}; return slot; }
inline Pool& pool() {
    static const int at_fork = pthread_atfork(nullptr, nullptr, +[] { pool_slot().store(nullptr); });
    (void)at_fork;
    Pool* p = pool_slot().load(std::memory_order_acquire);
    if (!p) {
        Pool* fresh = new Pool;                                           // (no threads yet: they start with the first job)
        if (pool_slot().compare_exchange_strong(p, fresh)) p = fresh;
        else delete fresh;
    }
    return *p;
}

}  // namespace hostpack

inline void host_pack(const float* x, const int64_t* lab, long long n, int b, int C, uint32_t* codes, uint64_t* labels,
                      HostPackCensus* census, int threads) {
    const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    if (threads <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        // the pass is memory bound: on the GPU box (2 x 64 cores, 256 hardware threads) the C2 call takes 4.3 ms with 32
        // threads, 3.3 with 64, 3.15 with 96 or 128 (tools/h2d_threads.py) -- now that the workers are not started per call
        threads = hw <= 64 ? (int)std::min<unsigned>(hw ? hw : 1u, 32u) : (int)std::min<unsigned>(hw / 2, 96u);
    }
    const long long bytes = n * ((long long)(x ? b * 4 : 0) + (lab ? C * 8 : 0));
    threads = (int)std::max<long long>(1, std::min<long long>(threads, bytes >> 20));    // at least ~1 MB of input per thread
    std::vector<HostPackCensus> part((size_t)threads);
    auto work = [&](int t) {
        const long long r0 = n * t / threads, r1 = n * (t + 1) / threads;
        if (wide) hostpack::rows_avx512(x, lab, r0, r1, b, C, codes, labels, part[(size_t)t]);
        else hostpack::rows_scalar(x, lab, r0, r1, b, C, codes, labels, part[(size_t)t]);
    };
    if (threads == 1) {
        work(0);
    } else if (!hostpack::pool().run(threads, work)) {
        std::vector<std::thread> own;
        for (int t = 1; t < threads; ++t) own.emplace_back(work, t);
        work(0);
        for (auto& th : own) th.join();
    }
    HostPackCensus tot;
    for (const auto& p : part) {
        tot.nonbinary += p.nonbinary; tot.zeros += p.zeros; tot.minus_ones += p.minus_ones; tot.bad_labels += p.bad_labels;
    }
    if (census) *census = tot;
}

}  // namespace hg
#endif
