// Host side of libhashgan_amd.so, shared by its translation units: error plumbing, device buffers, the context
// (struct hg_ctx, opaque in include/hashgan_amd.h) and the handful of host functions one unit calls in another.
//
//   hg_core.hip        context, tables (hg_set_*), options / statistics / timing, label match, AP, downloads
//   hg_seq.hip         the Hamming sequences: geometry, histogram -> plan -> select -> rank, staged (sharded) and one-shot forms
//   hg_pairs_valu.hip  launchers of the vector-ALU pair passes (k_hist, k_select, k_select_dense)
//   hg_pairs_mx.hip    launchers of the matrix-core pair passes (k_select_mx3 / mx4, k_hist_mx, k_hist_i8) and their images
//   hg_pairs_mx1.hip   launcher of k_select_mx (every code length: the longest compile)
//   hg_real.hip        real-valued (float32 inner product) ranking
//   hg_comm.hip        RCCL collectives (library dlopen'ed on first use)
//
// No torch, no CPU compute path: every entry point either runs HIP kernels or fails.
#pragma once
#include "hg_kernels.hpp"
#pragma GCC visibility push(default)     // the library is built with -fvisibility=hidden: only the C ABI is exported
#include "../../include/hashgan_amd.h"
#pragma GCC visibility pop

#include <rccl/rccl.h>     // types and enums only: the library itself is dlopen'ed by hg_comm_init (573 MB, not every process needs it)
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace hg;

extern thread_local std::string g_err;          // hg_last_error(): per thread
int fail(int code, const char* fmt, ...);       // sets g_err, returns code

#define HG_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? HG_ERR_NOMEM : HG_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                          \
    } while (0)

#define HG_TRY(expr)                \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != HG_OK) return rc_; \
    } while (0)


// Bumped whenever a device buffer moves: captured graphs hold raw addresses and die with the epoch they were built in.
extern std::atomic<unsigned long long> g_alloc_epoch;   // bumped by every context's buffers (one MAPs object per thread is supported): atomic

// Host-side cost of the runtime calls a context makes outside its kernels -- creating it, device and pinned allocations and
// their release -- accumulated process-wide (hg_get_stat "host_us_<phase>", "host_n_<phase>", "host_max_us_<phase>"): what a
// caller that builds a context per evaluation (main.py:164 builds a MAPs per evaluation) pays before any kernel runs.
enum HostPhase { HP_INIT = 0, HP_DEVMALLOC, HP_DEVFREE, HP_HOSTMALLOC, HP_HOSTFREE, HP_DESTROY, HP_STREAM, HP_EVENT, HP_SYNC, HP_PACK, HP_THREAD, HP_COUNT };
extern const char* const kHostPhaseNames[HP_COUNT];
extern std::atomic<long long> g_host_ns[HP_COUNT], g_host_calls[HP_COUNT], g_host_max_ns[HP_COUNT];
struct HostTimer {
    int id;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(int i) : id(i), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() {
        const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        g_host_ns[id] += ns;
        g_host_calls[id] += 1;
        long long m = g_host_max_ns[id].load();
        while (ns > m && !g_host_max_ns[id].compare_exchange_weak(m, ns)) {}
    }
};
template <class F> inline auto host_timed(int id, F&& f) { HostTimer t_(id); return f(); }

// HG_EFENCE=1 (debugging): every device buffer ends 64..127 bytes before an UNMAPPED 2 MiB page of its own virtual range
// (hipMemAddressReserve / hipMemMap), so a kernel reading or writing past a buffer -- beyond the 64 bytes of slack the
// kernels are allowed -- faults at once instead of only when hipMalloc happens to place the buffer at the end of a mapping;
// and every new buffer starts out filled with 0xCB, so nothing can rely on fresh memory being zero.  HG_EFENCE=2: the
// fill only, on plain allocations; HG_EFENCE=3: the unmapped page in FRONT of every buffer.  (tools/fuzz_*.py and the gpu
// tests run under all three.)
struct Fence { void* va = nullptr; void* map_at = nullptr; size_t va_size = 0, map_size = 0; hipMemGenericAllocationHandle_t h{}; };
inline int efence_mode() { static const int m = getenv("HG_EFENCE") ? atoi(getenv("HG_EFENCE")) : 0; return m; }
inline bool efence_on() { return efence_mode() != 0; }
inline hipError_t fence_alloc(Fence& f, void** out, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;
    f.map_size = (bytes + gran - 1) / gran * gran;
    f.va_size = f.map_size + gran;                                   // the last granule stays unmapped
    e = hipMemAddressReserve(&f.va, f.va_size, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&f.h, f.map_size, &prop, 0);
    if (e != hipSuccess) return e;
    // HG_EFENCE=3: the unmapped granule comes FIRST and the buffer starts right behind it (reads before a buffer)
    const bool front = efence_mode() == 3;
    f.map_at = (char*)f.va + (front ? gran : 0);
    e = hipMemMap(f.map_at, f.map_size, 0, f.h, 0);
    if (e != hipSuccess) return e;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(f.map_at, f.map_size, &acc, 1);
    if (e != hipSuccess) return e;
    *out = front ? f.map_at : (char*)f.va + ((f.map_size - bytes) & ~(size_t)63);       // 64-byte aligned, ends < 64 bytes before the fence
    return hipSuccess;
}
inline void fence_free(Fence& f) {
    if (!f.va) return;
    (void)hipDeviceSynchronize();                                    // hipFree waits for the device; unmapping does not
    (void)hipMemUnmap(f.map_at, f.map_size);
    (void)hipMemRelease(f.h);
    // (the virtual range is NOT returned: a later buffer at the same address could meet stale cache lines of this one)
    f = Fence{};
}

// Process-wide cache of what a context allocates and a destroyed (or trimmed) context gives back: device blocks, pinned host
// blocks, streams.  hipMalloc on this stack takes 0.03 - 0.4 ms -- and now and then SECONDS (3.5 s measured for one call among
// a few hundred: tools/new_context_probe.py, profiles/r06_new_context_probe.txt), hipStreamCreate 1.5 - 18 ms, a 64 MB
// hipHostMalloc 9 ms, hg_destroy's frees 3.7 ms: a caller that builds a context per evaluation paid all of that per call.  A
// block goes back to the cache instead of the runtime and the next request of a similar size takes it (best fit, at most twice
// the size asked for); the cache holds at most HG_CACHE_MB (default 49152 -- a class-sorted C2 database's widened record slices alone are 14.9 GB) of device and HG_PIN_CACHE_MB (default 512) of pinned
// memory -- the largest blocks go first when it is full -- and hg_release_cache() empties it.  HG_EFENCE builds bypass it.
void* cache_take_dev(int device, size_t want, size_t* got);           // nullptr: nothing suitable cached
void cache_give_dev(int device, void* p, size_t bytes);              // (frees it if the cache is full)
void* cache_take_pin(size_t want, size_t* got);
void cache_give_pin(void* p, size_t bytes);
hipStream_t cache_take_stream(int device);                           // nullptr: none cached
void cache_give_stream(int device, hipStream_t s);
void cache_release_all();
inline size_t cache_round(size_t bytes) {                            // sizes a later request can match: 4 KB up to 1 MB, then 1 MB
    const size_t g = bytes < ((size_t)1 << 20) ? 4096 : (size_t)1 << 20;
    return (bytes + g - 1) / g * g;
}
// pinned host block through the cache (*cap = what the block really holds)
inline hipError_t pin_alloc(void** p, size_t want, size_t* cap) {
    *p = cache_take_pin(want, cap);
    if (*p) return hipSuccess;
    const size_t sz = cache_round(want);
    const hipError_t e = host_timed(HP_HOSTMALLOC, [&] { return hipHostMalloc(p, sz, hipHostMallocDefault); });
    if (e == hipSuccess) *cap = sz;
    return e;
}
inline void pin_free(void* p, size_t cap) { if (p) cache_give_pin(p, cap); }

// A device buffer that only ever grows.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    size_t blk = 0;             // bytes of the allocation behind p (0: borrowed, or a fenced one)
    int dev = 0;
    bool borrowed = false;      // points into another context's allocation
    Fence fence;                // HG_EFENCE: the buffer's own virtual range
    void drop() {
        if (p && !borrowed) {
            if (fence.va) { HostTimer t_(HP_DEVFREE); fence_free(fence); }
            else if (blk) cache_give_dev(dev, p, blk);
            else { HostTimer t_(HP_DEVFREE); (void)hipFree(p); }
        }
        blk = 0;
    }
    int reserve(size_t bytes) {
        if (bytes <= cap) return HG_OK;
        drop();
        p = nullptr; cap = 0; borrowed = false;
        // slack: 16-byte wide copies may read past the last row of a table
        if (efence_mode() == 2) {                           // plain allocation, poisoned
            HG_HIP(hipMalloc(&p, bytes + 64));
            HG_HIP(hipMemset(p, 0xCB, bytes + 64));
            HG_HIP(hipDeviceSynchronize());                  // (the fill runs on the null stream; the context's stream does not wait for it)
        } else if (efence_on()) {
            HG_HIP(fence_alloc(fence, &p, bytes + 64));
            HG_HIP(hipMemset(p, 0xCB, bytes + 64));
            HG_HIP(hipDeviceSynchronize());
        } else {
            HG_HIP(hipGetDevice(&dev));
            p = cache_take_dev(dev, bytes + 64, &blk);
            if (!p) {
                const size_t sz = cache_round(bytes + 64);
                HG_HIP(host_timed(HP_DEVMALLOC, [&] { return hipMalloc(&p, sz); }));
                blk = sz;
            }
        }
        cap = bytes;
        ++g_alloc_epoch;
        return HG_OK;
    }
    void borrow(const DevBuf& o) {
        drop();
        if (p != o.p) ++g_alloc_epoch;
        p = o.p; cap = o.cap; borrowed = true;
    }
    // `bytes` at `off` inside another buffer (which must outlive the view)
    void view(const DevBuf& o, size_t off, size_t bytes) {
        drop();
        if (p != (char*)o.p + off) ++g_alloc_epoch;
        p = (char*)o.p + off; cap = bytes; borrowed = true;
    }
    void release() { drop(); if (p) ++g_alloc_epoch; p = nullptr; cap = 0; borrowed = false; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum KernelId { KI_HIST = 0, KI_HIST_REDUCE, KI_PLAN, KI_SEG_COUNTS, KI_SEG_LAYOUT, KI_GUESS, KI_SELECT, KI_CAND_HIST,
                KI_ORDER, KI_RANK_FUSED, KI_MATCH, KI_AP, KI_MERGE, KI_PACK, KI_REAL_SAMPLE, KI_REAL_GUESS, KI_REAL_SELECT,
                KI_RADIX, KI_REAL_FINISH, KI_SELECT_MX, KI_RANK_LDS, KI_COMM, KI_STEP, KI_REAL_RESCORE, KI_COUNT };
enum Stage { ST_NONE = 0, ST_DB = 1, ST_Q = 2, ST_HIST = 4, ST_PLAN = 8, ST_SELECT = 16, ST_MATCH = 32, ST_AP = 64 };
extern const char* const kKernelNames[KI_COUNT];

void build_shape(int n, ApShape& sh);           // NumPy's pairwise-summation tree of an n-element chunk, flattened (hg_core.hip)

struct Pending { int id; hipEvent_t a, b; };

struct hg_ctx {
    int device = 0;
    int n_cu = 256;            // compute units of the device
    hipStream_t stream = nullptr;
    bool own_stream = true;    // false: the stream belongs to the caller (hg_set_stream) or to the parent context
    bool stage_sync = true;    // staged calls synchronise the stream before returning
    unsigned stage = ST_NONE;

    // problem
    i64 N = 0, Q = 0, R = 0, n_total = 0;
    int b = 0, C = 0, NW = 0, NB = 0, LW = 0;
    u32 idx_base = 0;
    int G = 1, rank = 0;
    Geo geo{};
    i64 RW = 0;

    // options
    i64 target_units = 16384;
    i64 min_segment = 256;
    i64 opt_max_segments = 2048;   // "max_segments"
    i64 opt_enable = 1;        // one-shot calls may bet on a sampled threshold (verified, exact fallback)
    i64 opt_stride = 0;        // sampling stride in row batches, 0 = auto
    i64 opt_sigma = 5;         // safety margin of the guess, in standard deviations of the sample count (5: a query loses its bet
                               // about once in 3 million -- it is then rerun alone; 6 -> 5 keeps ~4 % fewer surplus records)
    i64 staged_lists = 1;      // staged hg_select materialises the idx/dist lists
    i64 cand_budget_x10 = 40;  // optimistic record budget per query, in tenths of R
    i64 opt_select_mfma = 1;   // optimistic select: 1 = matrix-core kernel (k_select_mx), 0 = vector-ALU k_select
    i64 opt_probe = 0;         // measurement probes of the matrix-core select kernels (SelArgs::probe)
    i64 opt_select_packed = 3; // several distances per MFMA accumulator: 3 = k_select_mx3 (<= 64 bits, three) / k_select_mx4 (65..128 bits, two) with
                               // the batched drain, for one-byte records; anything else = k_select_mx (one distance per accumulator)
    i64 opt_all_rows = 1;      // R = N: skip histogram and plan (every row is a member)
    i64 opt_rank_cnt = 1;      // the bet's rank stage keeps a query's records in LDS and ranks them with the per-thread counting sort (k_rank_cnt / k_rank_lean) where it applies
    i64 opt_rank_lean = 1;     // "rank_lean": ... in its lean form (k_rank_lean) for one-byte records without lists, <= 256 slices, <= 1024 pieces per query
    i64 real_grouped = 0;      // stat: the last real-valued ranking ordered its record lists group by group (k_real_group_*)
    i64 opt_real_groups = 1;   // "real_groups": record lists beyond the LDS are split by score range and ordered group by group in LDS (0: the four radix passes)
    i64 real_cap_boost = 1;    // the same for the real-valued ranking's slices (run_real)
    bool crowd_probed = false; // the first bet on this database has measured how its near rows crowd (k_guess_direct's probe)
    i64 crowd_x100 = 0;        // stat "crowding_x100": that measure, x 100 (~200: rows in random order; ~100 x classes: stored class by class)
    i64 opt_crowd_probe = 1;   // "crowd_probe"
    i64 cap_boost = 1;         // slice capacity multiplier a lost bet escalated to on this database (run_oneshot); 1 after every load
    i64 opt_rank_dense = 1;    // "rank_dense": N/8 < R <= N on one shard through the byte matrix (k_dense_bytes + k_rank_dense, hg_rank_dense.hpp; codes of <= 126 bits, <= 128 classes); 0: off
    bool leftovers_expected = false;   // the last fused step on this context left queries to the general kernel
    bool last_leftovers_inline = false;   // (the last finished step did: finish_leftovers)
    bool leftovers_inline = false;     // ... and this step ranked its own within the stream (launch_rank_slices with the flags)
    i64 opt_inline_leftovers = 1;      // "inline_leftovers"
    i64 opt_rank_slices = 7000;    // "rank_slices": a bet's one-byte records with R >= this are ranked by k_rank_dense<slices>; 0: off (k_rank_cnt's tiles).
                                   // Q = 10k, N = 1M: R = 5000 0.274 ms against k_rank_lean's 0.139 (fixed costs of the counter columns); R = 8000 0.318 / 0.372; R = 50 000 1.38 / 3.83
    i64 opt_rank_dense_gbm = -1;   // "rank_dense_gbm": k_rank_dense's bitmap in global memory (1) or LDS (0, where it fits); -1: by the blocks per CU
    i64 opt_dense_budget_mb = 16384;   // "dense_budget_mb": the byte matrix D holds at most this much (queries are chunked)
    bool dense_rank = false;   // run state of enqueue_all_rows: rank through the byte matrix

    // run state
    bool optimistic = false;   // records come from a guessed threshold (fixed-capacity slices)
    bool want_lists = true;
    bool lists_valid = false;
    u32 cap = 0;               // optimistic slice capacity
    i64 crow = 0;              // record-row stride
    i64 opt_runs = 0, opt_fallbacks = 0, opt_requeried = 0;
    int last_select = 0;       // stat "select_variant": 1 k_select, 2 k_select_dense, 3 k_select_mx, 5 k_select_mx3, 6 k_select_mx4
    int last_rank = 0;         // stat "rank_variant": 1 k_rank_fused, 3 k_rank_cnt, 6 k_rank_lean, 7 k_rank_dense, 8 k_rank_dense<slices>
    i64 opt_leftover = 0;      // stat "rank_leftovers": queries of fused steps that k_rank_cnt left to the general rank kernel
    int opt_consecutive_fail = 0;   // one-shot bets lost in a row (this context only)
    int shard_bet_fail = 0;         // sharded bets lost in a row: identical on every rank by construction
    hg_ctx* sub = nullptr;     // child context (shares the database) that reruns single lost queries exactly
    bool is_sub = false;

    // device state
    DevBuf db, dblab, qc, qlab;
    DevBuf beyond;             // one word: hg_guess_finish met a query whose cut lies beyond the planes its owner was sent
    DevBuf dbx, qx;            // fp4 images of db / qc in MFMA fragment order for k_select_mx (built on first use)
    bool dbx_valid = false, qx_valid = false;
    DevBuf dbx8;               // i8 image of the database codes in A-fragment order (k_hist_i8), built on first use
    bool dbx8_valid = false;
    DevBuf dbx3;               // fp4 image for k_select_mx3 (48-row supertiles, three rows per accumulator), built on first use
    bool dbx3_valid = false;
    DevBuf dbx4;               // fp4 image for k_select_mx4 (32-row supertiles, two rows per accumulator; codes of 65..128 bits), built on first use
    bool dbx4_valid = false;
    bool direct_rank = false;  // R = N: k_rank_fused computes distance and match bit per row itself (no records)
    i64 opt_hist_mfma = 2;     // "hist_mfma": histograms (sampled pass; full pass of the one-shot exact sequence) on the matrix cores -- 2: the integer instruction delivers the counter address (k_hist_i8, codes of <= 128 bits), 1: fp4 distances (k_hist_mx), 0: vector ALU
    bool hist_pairs = false;   // the last FULL histogram pass ran per segment pair (k_hist_mx)
    bool exact_mx = false;     // the matrix-core select runs with the EXACT threshold (hg_hist + k_plan) instead of a guess
    bool rec8 = false;         // the record rows hold one-byte compact records (matrix-core select, no lists wanted)
    i64 opt_compact = 1;       // "compact_records": allow them
    i64 opt_second_bet = 1;    // "second_bet": a lost one-shot bet is retried once with a wider margin before the exact sequence
    i64 opt_rebets = 0;
    bool err_zeroed = false;   // the guess kernel of a one-shot bet already cleared err
    // AP from the rank kernel's epilogue (k_rank_cnt: the bitmap is still in LDS) -- one launch less per step, and the general
    // rank kernel for the queries k_rank_cnt declines is launched only when the step's download says there are any
    i64 opt_fuse_ap = 1;       // "fuse_ap"
    bool fuse_ap = false;      // request of the current enqueue (hg_map's bet)
    bool ap_fused = false;     // the last launch_rank left the AP of every query it ranked in c->ap / c->rel, leftovers counted in err[1]
    i64 defer_verdict = 0;     // hg_rank does not wait for the bet's verdict; hg_bet_verdict reads it later
    bool verdict_pending = false, verdict_known = false;
    int verdict_flag = 0;
    // pinned landing zone for a one-shot call's results: AP, hit counts and the lost-bet flag come back with the
    // call's single synchronisation instead of three blocking copies into pageable memory afterwards
    void* pin = nullptr;
    size_t pin_cap = 0;
    // hg_map_begin / hg_map_end: up to two steps in flight, each with its own pinned landing block and event
    struct MapSlot {
        void* pin = nullptr; size_t cap = 0; hipEvent_t ev = nullptr;
        bool async = false;            // enqueued only (else: ran synchronously, results in ap / rel)
        bool inline_ok = false;        // the step ranked its fused kernel's leftovers itself
        i64 R = 0, Q = 0;
        unsigned long long q_gen = 0, db_gen = 0;   // the tables the step was enqueued on (hg_map_end redoes a lost step only on those)
        std::vector<double> ap; std::vector<int64_t> rel;
    } mslot[2];
    int ms_head = 0, ms_n = 0;
    unsigned long long map_warm_cfg = 0, map_warm_epoch = 0;   // configuration of the last synchronous hg_map that won its bet outright
    i64 map_warm_R = -1;
    i64 map_async_steps = 0, map_async_redone = 0;
    unsigned long long q_gen = 0, db_gen = 0;      // bumped by every (re)load of the query / database tables
    i64 handicap_next = 0;     // test hook "handicap_next_bet": the NEXT bet's guess sits this many deviations BELOW the expected count (it loses), once
    // hg_set_queries stages the packed tables through two alternating pinned blocks and does not wait for the stream: a new batch
    // can be handed over while a step on the previous one is still in flight (hg_map_begin / hg_map_end, batch after batch)
    struct QStage { void* pin = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; } qstage[2];
    int qstage_next = 0;
    bool ap_staged = false;
    DevBuf hist, hown, posbase, seglt, segtie;
    DevBuf t, tguess, sstar, cnt_lt, quota, tie_before, n_lt, err;
    DevBuf sl_start, sl_tie, sl_cnt, tot, failq;
    DevBuf mbits2;             // hg_merge_ranked's output (swapped with mbits)
    DevBuf part;               // hg_merge_ap_part's output: {AP, hits} of this rank's queries + its verdict
    DevBuf obuf[2];            // owner-routed exchanges: [0] the blocks this rank sends (hg_pack_*_by_owner), [1] its answers as an owner (hg_guess_owned)
    bool ranked_local = false; // mbits holds this shard's bitmap in LOCAL rank order (hg_select_ranked)
    DevBuf cand, out_idx, out_dist, mbits, shapes, ap_recip, ap, rel, stage_in, badcnt, qbad, flist, hwq, bigq;
    DevBuf outblk;             // [verdict 16 B][ap Q x 8][rel Q x 4]: err, ap and rel are views of it (ensure_out_block), so a call's results come home in ONE copy
    i64 outblk_q = -1;         // the Q those views were cut for
    DevBuf dbytes;             // the byte matrix D[q][Npad] of the dense regime (k_dense_bytes)
    DevBuf dbf, qf, samp, thr, sortA, sortB, scores, gtab;   // real-valued path
    DevBuf dbfx;               // float features of the database in MFMA A-fragment order (k_real_select_mx), built on first use
    bool dbfx_valid = false;
    DevBuf sampx;              // float features of the sampled rows in MFMA A-fragment order (k_real_sample_mx), rebuilt per call
    DevBuf cntq;               // real-valued path: the slices' record counts after the rescore, query-major [Q][S] (k_real_rank_lds reads a query's row in one piece)
    DevBuf krows;              // real-valued path: the rows the filter kept, 4-byte row numbers [Q][S][cap] (k_real_select_bf writes, k_real_rescore reads)
    DevBuf dbfb, thr2, xmax2;  // filter + rescore path (hg_real_bf.hpp): bf16 image of the database, lowered cuts, max row norm^2
    bool dbfb_valid = false;
    double real_expect = 0.0;  // rows per query the current real-valued attempt expects its cut to keep (real_attempt; picks the rescore's slices per wavefront)
    bool dbfb_half = false;    // ... in IEEE half instead of bfloat16 (no feature of the database can overflow it: real_launch_select_bf)
    DevBuf hist2;              // the second sample's counts [Q][RC_BINS] (k_real_sample_count)
    i64 opt_real_second = 1;   // "real_second_sample": a second, counting sample four times as large tightens the sampled cut
    i64 opt_real_rounds = 3;   // "real_whole_rounds": the no-cut float32 MFMA pass (k_real_select_mx) cuts the database so that its blocks fill whole rounds of this many per CU; 0: the plain geometry
    i64 opt_real_map_lists = 0;   // "real_map_lists": hg_map_real also writes the ranked idx / score lists (hg_get_topr_real after it); 0: match bits and APs only, like hg_map
    bool samp16 = false;       // the current attempt's sample scores are bfloat16 (k_real_sample_h -> k_real_guess_lds)
    i64 opt_real_sample_h = 1; // "real_sample_half": the sampled cut's scores in the filter's 16-bit arithmetic (k_real_sample_h) instead of exact float32 chains
    i64 opt_real_sort_lds = 1; // "real_sort_lds": sort + finish of the filter path in one LDS-resident kernel when the records fit
    bool real_no_cut = false;     // the current real_attempt takes every row (thr = -inf)
    bool real_filtered = false;   // the last real_select left unscored candidates that k_real_rescore completed
    i64 real_requeried = 0;       // queries that lost the first real-valued bet and were ranked again on their own (cumulative)
    i64 real_attempts = 0;        // statistics of the last real-valued ranking: attempts made (1 = the first bet held) ...
    i64 real_lds_ranked = 0;      // ... and whether the LDS-resident rank kernel produced its lists
    i64 opt_real_mfma = 2;     // "real_mfma": 2 = bf16 filter on the matrix cores + exact rescoring of the survivors, 1 = exact float32 MFMA pass, 0 = vector ALU
    int bpad = 0;              // feature count padded to a multiple of 16 (0: no float tables loaded)
    i64 census_db[3] = {0, 0, 0}, census_q[3] = {0, 0, 0};
    // hand-over of float32 / int64 arrays: packed on the host by a thread pool before the upload (hg_host_pack.hpp)
    i64 opt_host_pack = 1;     // "host_pack": 0 = upload the raw arrays and pack on the GPU (k_pack_*)
    i64 opt_keep_floats = 2;   // "keep_floats": database float table on the GPU -- 0 never, 1 always, 2 only if it is not a +-1 code
    hipStream_t stream2 = nullptr;   // the float table's uploads while the packing pool works (pack_on_host)
    hipEvent_t stream2_ev = nullptr;
    size_t fstage_cap = 0;
    void* fstage = nullptr;    // 4 x 16 MB of pinned staging for float tables on their way to the GPU (pack_on_host)
    hipEvent_t fstage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void* hpk = nullptr;       // pinned staging for the packed tables
    size_t hpk_cap = 0;
    bool dbf_resident = false, qf_resident = false;   // float tables as loaded: entries outside {-1,0,+1}, zeros, minus ones
    bool real_lists = false;
    bool real_lists_made = false;   // ... by the last attempt (hg_map_real skips them on the paths that rank in LDS)
    i64 shapes_for_R = -1;
    i64 recip_for_R = -1;      // ap_recip holds RN(1 / k) for k = 1 .. this
    i64 opt_ap_recip = 1;      // "ap_recip": k_ap divides through the table of reciprocals (bit for bit the division; 0: divide)
    i64 opt_ap_wide = 1;       // "ap_wide": k_ap with 512 threads per query when the queries are few and their lists long (0: always 128)

    // collectives (RCCL over xGMI), one communicator per context; gathered[] are the landing zones of hg_allgather
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    DevBuf gathered[4], scratch[4], comm_tmp, gath_idx, gath_dist;

    // one-shot step as a hipGraph: the bet's whole sequence (memsets, ~7 kernels, the result download) is captured the
    // second time hg_map sees the same problem and replayed afterwards -- one launch per step instead of ~15 enqueues,
    // so the step time no longer depends on how fast the host can feed the stream
    struct StepGraph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        unsigned long long epoch = 0, cfg = 0, seen_epoch = 0, seen_cfg = 0;   // key of exec / of the last eager step
        i64 R = -1, seen_R = -1;
        int timing = -1, seen_timing = -1;
        // host-side state the captured enqueue functions leave behind
        unsigned stage = 0; bool optimistic = false, lists_valid = false, ap_fused = false, rec8 = false; u32 cap = 0; i64 crow = 0, RW = 0; Geo geo{};
        std::vector<Pending> evs;          // event-record nodes inside the graph (kernel timing)
    } sg;
    i64 opt_graph = 0;         // "step_graph": 1 = hg_map captures and replays its step (see run_oneshot); off by default
    unsigned long long cfg_epoch = 1;      // bumped by everything that changes what a step enqueues (tables, options, stream)
    bool capturing = false;
    i64 graph_replays = 0, graph_captures = 0;

    // timing
    int timing = 0;            // 0 off, 1 the pair passes only (hist, select), 2 every kernel
    double t_ms[KI_COUNT] = {0};
    i64 t_n[KI_COUNT] = {0};
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;

    int use() { HG_HIP(hipSetDevice(device)); return HG_OK; }

    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)host_timed(HP_EVENT, [&] { return hipEventCreate(&e); });
        return e;
    }
    bool t_wanted(int id) const {
        // 1: the select pass (the roofline kernel) and the step's span only -- every event pair costs the stream ~2-4 us
        // and only on one step in "timing_every" (the averages are over the sampled launches)
        return timing >= 2 || (timing == 1 && (id == KI_SELECT || id == KI_SELECT_MX || id == KI_STEP) &&
                               (capturing || opt_timing_every <= 1 || t_seq % opt_timing_every == 0));
    }
    i64 opt_timing_every = 1;  // "timing_every": level-1 timing records its events on every n-th one-shot step only
    i64 t_seq = 0;             // one-shot steps since timing was enabled
    // while a step is being captured the events become event-record nodes of the graph and stay with it
    std::vector<Pending>& t_list() { return capturing ? sg.evs : pending; }
    bool t_open = false;
    void t_begin(int id) {
        t_open = t_wanted(id);
        if (!t_open) return;
        Pending p{id, get_event(), get_event()};
        (void)hipEventRecord(p.a, stream);
        t_list().push_back(p);
    }
    void t_end() {
        if (!t_open) return;
        t_open = false;
        (void)hipEventRecord(t_list().back().b, stream);
    }
    // the whole step's span on the GPU (first enqueue to the last byte of the download): nests around the kernels' pairs
    int step_slot = -1;
    void t_step_begin() {
        step_slot = -1;
        ++t_seq;
        if (!t_wanted(KI_STEP)) return;
        Pending p{KI_STEP, get_event(), get_event()};
        (void)hipEventRecord(p.a, stream);
        step_slot = (int)t_list().size();
        t_list().push_back(p);
    }
    void t_step_end() {
        if (step_slot < 0) return;
        (void)hipEventRecord(t_list()[step_slot].b, stream);
        step_slot = -1;
    }
    void t_collect_graph() {   // after a replay has completed
        for (auto& p : sg.evs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { t_ms[p.id] += ms; t_n[p.id] += 1; }
        }
    }
    void drop_graph() {
        if (sg.exec) (void)hipGraphExecDestroy(sg.exec);
        if (sg.graph) (void)hipGraphDestroy(sg.graph);
        sg.exec = nullptr; sg.graph = nullptr;
        for (auto& p : sg.evs) { pool.push_back(p.a); pool.push_back(p.b); }
        sg.evs.clear();
    }
    void t_collect() {   // after a stream sync
        for (auto& p : pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { t_ms[p.id] += ms; t_n[p.id] += 1; }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    int sync() {
        HG_HIP(host_timed(HP_SYNC, [&] { return hipStreamSynchronize(stream); }));     // ("sync": waiting for the GPU -- kernels and copies included)
        if (pending.size() > 4096) t_collect();       // otherwise the elapsed times are read when somebody asks for them
        return HG_OK;
    }
    // end of a staged call that only enqueued work: synchronise unless the caller orders everything on
    // one stream itself (hg_set_stream + stage_sync = 0, e.g. torch's current stream in sharded mode)
    int stage_end() { return stage_sync ? sync() : HG_OK; }
    int check_launch(const char* what) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(HG_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
        return HG_OK;
    }
};


inline int grid_for(i64 n, int per_block = 256) { return (int)((n + per_block - 1) / per_block); }
inline int padded_grid(int nBlk) { return (nBlk + 7) / 8 * 8; }

// ---- across translation units --------------------------------------------------------------------------------------
// hg_core.hip
int need(hg_ctx* c, unsigned st, const char* who, const char* what);     // stage check + hipSetDevice
int upload_codes(hg_ctx* c, DevBuf& dst, const uint64_t* host, i64 n, int W, int NW);
int ensure_pin(hg_ctx* c, size_t need_b);
int do_match(hg_ctx* c);                         // k_match through the ranked idx list (> 128 classes, real-valued lists)
int ensure_ap_tables(hg_ctx* c, bool* use_recip);   // summation trees + reciprocals for the current R, c->ap / c->rel sized
int do_ap_range(hg_ctx* c, i64 q0, i64 nq, const u32* only = nullptr);      // k_ap on queries [q0, q0 + nq) (only: device flags [Q], just the flagged ones)
inline int do_ap(hg_ctx* c) { return do_ap_range(c, 0, c->geo.Q); }
int ensure_out_block(hg_ctx* c);                   // err / ap / rel as views of one block (before anything of the call is enqueued)
int read_plan_flag(hg_ctx* c, int* flag);        // *err back to the host (synchronises)
int launch_min_topr(hg_ctx* c, const u32* idx_all, const u8* dist_all, i64 n, int G);
// hg_seq.hip
int stage_ap_download(hg_ctx* c, void* dst = nullptr);   // {verdict, AP, hit counts} into pinned host memory behind everything enqueued so far (no synchronisation)
void make_geometry(hg_ctx* c);
Geo hist_geometry(const hg_ctx* c);
int set_R(hg_ctx* c, int64_t R, int G, int rank);
// hg_pairs_valu.hip
int launch_hist(hg_ctx* c);                      // k_hist<NW>
int launch_select_valu(hg_ctx* c, int lw, bool optimistic);   // k_select<NW, LW, OPT>
int launch_select_dense(hg_ctx* c, int lw);      // k_select_dense<NW, LW>
// hg_pairs_mx.hip (k_select_mx: hg_pairs_mx1.hip)
int ensure_mx_images(hg_ctx* c, bool need_db);  // fp4 images of database / query codes, built on first use
int launch_hist_mx(hg_ctx* c);                   // k_hist_i8 / k_hist_mx
int launch_select_mx(hg_ctx* c, int lw);         // k_select_mx<NW, LW, QT, COMPACT>
int launch_select_mx3(hg_ctx* c, int lw);        // k_select_mx3 (codes of <= 64 bits, one-byte records)
int launch_select_mx4(hg_ctx* c, int lw);        // k_select_mx4 (codes of 65..128 bits, one-byte records)
int preload_valu(); int preload_mx(); int preload_mx1(); int preload_real(); int preload_seq();   // one per translation unit with kernels (hg_preload)
// hg_comm.hip
void comm_release(hg_ctx* c);                    // destroys the context's communicator, if any

#define HG_DISPATCH_NW(fn, c, ...)                              \
    switch ((c)->NW) {                                          \
        case 1: return fn<1>(c, ##__VA_ARGS__);                 \
        case 2: return fn<2>(c, ##__VA_ARGS__);                 \
        case 3: return fn<3>(c, ##__VA_ARGS__);                 \
        case 4: return fn<4>(c, ##__VA_ARGS__);                 \
        case 5: return fn<5>(c, ##__VA_ARGS__);                 \
        case 6: return fn<6>(c, ##__VA_ARGS__);                 \
        case 7: return fn<7>(c, ##__VA_ARGS__);                 \
        case 8: return fn<8>(c, ##__VA_ARGS__);                 \
        default: return fail(HG_ERR_ARG, "unsupported code length: %d words", (c)->NW); \
    }
