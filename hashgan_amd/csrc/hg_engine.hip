// Host side of libhashgan_amd.so: device context, buffers, launch geometry and
// the C ABI declared in include/hashgan_amd.h.  No torch, no CPU compute path:
// every entry point either runs the HIP kernels of hg_kernels.hpp or fails.
#include "hg_kernels.hpp"
#include "hg_real_kernels.hpp"
#include "hg_mx_drain.hpp"
#include "hg_select_mx.hpp"
#include "hg_rank_lds.hpp"
#include "hg_rank_cnt.hpp"
#include "hg_rank_wave.hpp"
#include "hg_rank_direct.hpp"
#include "hg_select_mx2.hpp"
#include "hg_select_mx3.hpp"
#include "hg_real_mx.hpp"
#include "hg_real_bf.hpp"
#include "hg_hist_i8.hpp"
#include "hg_hist_mx.hpp"
#include "hg_host_pack.hpp"
#include "../../include/hashgan_amd.h"

#include <rccl/rccl.h>     // types and enums only: the library itself is dlopen'ed by hg_comm_init (573 MB, not every process needs it)
#include <atomic>
#include <mutex>
#include <dlfcn.h>
#include <unistd.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

using namespace hg;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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
std::atomic<unsigned long long> g_alloc_epoch{1};   // bumped by every context's buffers (one MAPs object per thread is supported): atomic

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

// A device buffer that only ever grows.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool borrowed = false;      // points into another context's allocation
    Fence fence;                // HG_EFENCE: the buffer's own virtual range
    void drop() {
        if (p && !borrowed) { if (fence.va) fence_free(fence); else (void)hipFree(p); }
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
            HG_HIP(hipMalloc(&p, bytes + 64));
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
    void release() { drop(); if (p) ++g_alloc_epoch; p = nullptr; cap = 0; borrowed = false; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum KernelId { KI_HIST = 0, KI_HIST_REDUCE, KI_PLAN, KI_SEG_COUNTS, KI_SEG_LAYOUT, KI_GUESS, KI_SELECT, KI_CAND_HIST,
                KI_ORDER, KI_RANK_FUSED, KI_MATCH, KI_AP, KI_MERGE, KI_PACK, KI_REAL_SAMPLE, KI_REAL_GUESS, KI_REAL_SELECT,
                KI_RADIX, KI_REAL_FINISH, KI_SELECT_MX, KI_RANK_LDS, KI_COMM, KI_STEP, KI_REAL_RESCORE, KI_COUNT };
const char* const kKernelNames[KI_COUNT] = {"k_hist", "k_hist_reduce", "k_plan", "k_seg_counts", "k_seg_layout", "k_guess",
                                            "k_select", "k_rank_hist", "k_order", "k_rank_fused", "k_match", "k_ap", "k_merge", "k_pack",
                                            "k_real_sample", "k_real_guess", "k_real_select", "k_radix_pass", "k_real_finish", "k_select_mx", "k_rank_lds", "rccl_allgather", "step_gpu_span", "k_real_rescore"};

enum Stage { ST_NONE = 0, ST_DB = 1, ST_Q = 2, ST_HIST = 4, ST_PLAN = 8, ST_SELECT = 16, ST_MATCH = 32, ST_AP = 64 };

// Flatten NumPy's pairwise-summation tree for a chunk of n elements (n <= 8192):
// numpy/_core/src/umath/loops_utils.h.src, pairwise_sum: n <= 128 is a leaf,
// otherwise split at n/2 rounded down to a multiple of 8.
void build_shape(int n, ApShape& sh) {
    memset(&sh, 0, sizeof sh);
    sh.n = n;
    struct Rec {
        ApShape& s;
        void go(int off, int len) {
            if (len <= AP_LEAF) {
                s.leaf_start[s.n_leaves] = (unsigned short)off;
                s.leaf_len[s.n_leaves] = (unsigned short)len;
                s.prog[s.n_prog++] = (short)s.n_leaves++;
            } else {
                int n2 = len / 2;
                n2 -= n2 % 8;
                go(off, n2);
                go(off + n2, len - n2);
                s.prog[s.n_prog++] = -1;
            }
        }
    } rec{sh};
    if (n > 0) rec.go(0, n);
    // the same tree as a node table (k_ap evaluates it level by level): replay the postfix program on a stack of ids
    int stack[64], sp = 0, height[2 * AP_LEAF] = {0};
    for (int i = 0; i < sh.n_prog; ++i) {
        const int op = sh.prog[i];
        if (op >= 0) { stack[sp++] = op; continue; }
        const int r = stack[--sp], l = stack[--sp];
        const int k = sh.n_nodes++, id = sh.n_leaves + k;
        sh.nl[k] = (short)l;
        sh.nr[k] = (short)r;
        const int hgt = 1 + (height[l] > height[r] ? height[l] : height[r]);
        sh.nh[k] = (unsigned char)hgt;
        height[id] = hgt;
        if (hgt > sh.max_h) sh.max_h = hgt;
        stack[sp++] = id;
    }
}


// ---- RCCL, loaded on first use.  The product links no collective library: a single-GPU process never pays for it. ----
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

std::mutex g_rccl_mu;                              // contexts of several threads may initialise communicators at once

int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return HG_OK;
    const char* cands[] = {getenv("HG_RCCL_LIBRARY"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    std::string tried;
    for (const char* name : cands) {
        if (!name || !*name) continue;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        const char* why = dlerror();
        tried += std::string(name) + ": " + (why ? why : "?") + "; ";
    }
    if (!h) return fail(HG_ERR_STATE, "RCCL not found (%s)", tried.c_str());
    RcclApi a;
    a.handle = h;
#define HG_SYM(field, name)                                                             \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                      \
    if (!a.field) { dlclose(h); return fail(HG_ERR_STATE, "RCCL: symbol %s missing", name); }
    HG_SYM(GetUniqueId, "ncclGetUniqueId")
    HG_SYM(CommInitRank, "ncclCommInitRank")
    HG_SYM(CommDestroy, "ncclCommDestroy")
    HG_SYM(AllGather, "ncclAllGather")
    HG_SYM(AllReduce, "ncclAllReduce")
    HG_SYM(GetErrorString, "ncclGetErrorString")
#undef HG_SYM
    g_rccl = a;
    return HG_OK;
}

#define HG_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return fail(HG_ERR_HIP, "%s: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
}  // namespace

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
    i64 opt_real_seg_bytes = 512 * 1024;   // real-valued path: bytes of feature rows per segment
    i64 opt_real_qpl = 1;      // real-valued path: queries per lane (1 or 2)
    i64 opt_rank_waves = 0;    // k_rank_fused wavefronts per query: 0 = by list length, else 4 or 16
    i64 opt_select_mfma = 1;   // optimistic select: 1 = matrix-core kernel (k_select_mx), 0 = vector-ALU k_select
    i64 opt_probe = 0;         // measurement probes of the matrix-core select kernels (SelArgs::probe)
    i64 opt_select_packed = 3; // codes of <= 64 bits, several distances per MFMA accumulator: 1 = k_select_mx2 (two) for <= 32 bits,
                               // 2 = k_select_mx2 up to 64 bits, 3 = k_select_mx3 (three, batched drain) for compact records, else like 1
    i64 opt_sample_ratio = 2;  // the sampled pass works on segments this many times longer than the select pass's
    i64 opt_all_rows = 1;      // R = N: skip histogram and plan (every row is a member)
    i64 opt_rank_lds = 1;      // the bet's rank stage keeps a query's records in LDS when they fit (k_rank_lds)
    i64 opt_rank_cnt = 1;      // ... and ranks them with the per-thread counting sort (k_rank_cnt) where it applies
    i64 real_grouped = 0;      // stat: the last real-valued ranking ordered its record lists group by group (k_real_group_*)
    i64 opt_real_groups = 1;   // "real_groups": record lists beyond the LDS are split by score range and ordered group by group in LDS (0: the four radix passes)
    i64 real_cap_boost = 1;    // the same for the real-valued ranking's slices (run_real)
    i64 cap_boost = 1;         // slice capacity multiplier a lost bet escalated to on this database (run_oneshot); 1 after every load
    i64 opt_rank_direct_lds = 80;    // "rank_direct_lds": KB of LDS a k_rank_direct block may take (80: two blocks per CU -- C1 0.25 ms vs 0.31 with 160 and one)
    i64 opt_rank_direct = 1;   // "rank_direct": R = N on one shard in one counting-sort kernel, k_rank_direct, when its LDS fits (2: also N/8 < R < N)
    i64 opt_rank_wave = 40;    // "rank_wave": one wavefront per query (k_rank_wave) for SHORT lists of one-byte records; the value is the
                               // record capacity of a query's LDS share in tenths of the shard's share of R (+ 256; lists beyond it
                               // go to k_rank_fused); 0 = off
    i64 opt_rank_wave_max = 4608;   // "rank_wave_max": ... used when that capacity is at most this many records (<= 16128)
    i64 opt_select_qt = 2;     // k_select_mx query tiles per wavefront (2: 4 wavefronts per SIMD, 4: 2)

    // run state
    bool optimistic = false;   // records come from a guessed threshold (fixed-capacity slices)
    bool want_lists = true;
    bool lists_valid = false;
    u32 cap = 0;               // optimistic slice capacity
    i64 crow = 0;              // record-row stride
    i64 opt_runs = 0, opt_fallbacks = 0, opt_requeried = 0;
    int opt_consecutive_fail = 0;   // one-shot bets lost in a row (this context only)
    int shard_bet_fail = 0;         // sharded bets lost in a row: identical on every rank by construction
    hg_ctx* sub = nullptr;     // child context (shares the database) that reruns single lost queries exactly
    bool is_sub = false;

    // device state
    DevBuf db, dblab, qc, qlab;
    DevBuf dbx, qx;            // fp4 images of db / qc in MFMA fragment order for k_select_mx (built on first use)
    bool dbx_valid = false, qx_valid = false;
    DevBuf dbx2, qx2;          // the same for k_select_mx2 (two rows per accumulator, codes of <= 64 bits)
    bool dbx2_valid = false, qx2_valid = false;
    DevBuf dbx8;               // i8 image of the database codes in A-fragment order (k_hist_i8), built on first use
    bool dbx8_valid = false;
    DevBuf dbx3;               // fp4 image for k_select_mx3 (48-row supertiles, three rows per accumulator), built on first use
    bool dbx3_valid = false;
    bool direct_rank = false;  // R = N: k_rank_fused computes distance and match bit per row itself (no records)
    i64 opt_hist_mfma = 2;     // "hist_mfma": histograms (sampled pass; full pass of the one-shot exact sequence) on the matrix cores -- 2: the integer instruction delivers the counter address (k_hist_i8, codes of <= 128 bits), 1: fp4 distances (k_hist_mx), 0: vector ALU
    bool hist_pairs = false;   // the last FULL histogram pass ran per segment pair (k_hist_mx)
    bool exact_mx = false;     // the matrix-core select runs with the EXACT threshold (hg_hist + k_plan) instead of a guess
    i64 opt_exact_mfma = 1;    // "exact_mfma": the one-shot exact sequence selects on the matrix cores when R << N
    bool rec8 = false;         // the record rows hold one-byte compact records (matrix-core select, no lists wanted)
    i64 opt_compact = 1;       // "compact_records": allow them
    i64 opt_second_bet = 1;    // "second_bet": a lost one-shot bet is retried once with a wider margin before the exact sequence
    i64 opt_rebets = 0;
    i64 opt_lds_pad = 0;       // "lds_pad": extra dynamic LDS per block of the matrix-core select (occupancy experiments)
    bool err_zeroed = false;   // the guess kernel of a one-shot bet already cleared err
    i64 defer_verdict = 0;     // hg_rank does not wait for the bet's verdict; hg_bet_verdict reads it later
    bool verdict_pending = false, verdict_known = false;
    int verdict_flag = 0;
    // pinned landing zone for a one-shot call's results: AP, hit counts and the lost-bet flag come back with the
    // call's single synchronisation instead of three blocking copies into pageable memory afterwards
    void* pin = nullptr;
    size_t pin_cap = 0;
    bool ap_staged = false;
    DevBuf hist, hown, posbase, seglt, segtie;
    DevBuf t, tguess, sstar, cnt_lt, quota, tie_before, n_lt, err;
    DevBuf sl_start, sl_tie, sl_cnt, tot, failq;
    DevBuf mbits2;             // hg_merge_ranked's output (swapped with mbits)
    DevBuf part;               // hg_merge_ap_part's output: {AP, hits} of this rank's queries + its verdict
    bool ranked_local = false; // mbits holds this shard's bitmap in LOCAL rank order (hg_select_ranked)
    DevBuf cand, out_idx, out_dist, mbits, shapes, ap_recip, ap, rel, stage_in, badcnt, qbad, flist, hwq, bigq;
    DevBuf dbf, qf, samp, thr, sortA, sortB, scores, gtab;   // real-valued path
    DevBuf dbfx;               // float features of the database in MFMA A-fragment order (k_real_select_mx), built on first use
    bool dbfx_valid = false;
    DevBuf sampx;              // float features of the sampled rows in MFMA A-fragment order (k_real_sample_mx), rebuilt per call
    DevBuf dbfb, thr2, xmax2;  // filter + rescore path (hg_real_bf.hpp): bf16 image of the database, lowered cuts, max row norm^2
    bool dbfb_valid = false;
    i64 opt_real_sample_hits = 64;   // "real_sample_hits": the real-valued bet samples so that this many of a query's top R rows are in the sample
    i64 opt_real_sort_lds = 1; // "real_sort_lds": sort + finish of the filter path in one LDS-resident kernel when the records fit
    bool real_no_cut = false;     // the current real_attempt takes every row (thr = -inf)
    bool real_filtered = false;   // the last real_select left unscored candidates that k_real_rescore completed
    i64 real_attempts = 0;        // statistics of the last real-valued ranking: attempts made (1 = the first bet held) ...
    i64 real_lds_ranked = 0;      // ... and whether the LDS-resident rank kernel produced its lists
    i64 opt_real_mfma = 2;     // "real_mfma": 2 = bf16 filter on the matrix cores + exact rescoring of the survivors, 1 = exact float32 MFMA pass, 0 = vector ALU
    int bpad = 0;              // feature count padded to a multiple of 16 (0: no float tables loaded)
    i64 census_db[3] = {0, 0, 0}, census_q[3] = {0, 0, 0};
    // hand-over of float32 / int64 arrays: packed on the host by a thread pool before the upload (hg_host_pack.hpp)
    i64 opt_host_pack = 1;     // "host_pack": 0 = upload the raw arrays and pack on the GPU (k_pack_*)
    i64 opt_keep_floats = 2;   // "keep_floats": database float table on the GPU -- 0 never, 1 always, 2 only if it is not a +-1 code
    i64 opt_pack_threads = 0;  // "pack_threads": 0 = from the hardware (up to 96)
    hipStream_t stream2 = nullptr;   // the float table's uploads while the packing pool works (pack_on_host)
    hipEvent_t stream2_ev = nullptr;
    void* fstage = nullptr;    // 4 x 16 MB of pinned staging for float tables on their way to the GPU (pack_on_host)
    hipEvent_t fstage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void* hpk = nullptr;       // pinned staging for the packed tables
    size_t hpk_cap = 0;
    bool dbf_resident = false, qf_resident = false;   // float tables as loaded: entries outside {-1,0,+1}, zeros, minus ones
    bool real_lists = false;
    i64 shapes_for_R = -1;
    i64 recip_for_R = -1;      // ap_recip holds RN(1 / k) for k = 1 .. this
    i64 opt_ap_recip = 1;      // "ap_recip": k_ap divides through the table of reciprocals (bit for bit the division; 0: divide)

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
        unsigned stage = 0; bool optimistic = false, lists_valid = false; u32 cap = 0; i64 crow = 0, RW = 0; Geo geo{};
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
        (void)hipEventCreate(&e);
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
        HG_HIP(hipStreamSynchronize(stream));
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

namespace {

int grid_for(i64 n, int per_block = 256) { return (int)((n + per_block - 1) / per_block); }

// Segment geometry of the pair passes: ~target_units wavefront-sized units.
void make_geometry(hg_ctx* c) {
    Geo& g = c->geo;
    g.Q = (int)c->Q;
    g.nQT = (int)((c->Q + 63) / 64);
    g.Qpad = g.nQT * 64;
    g.NW = c->NW; g.NB = c->NB; g.LW = c->LW;
    g.N = c->N; g.R = c->R; g.idx_base = c->idx_base;
    i64 S = (c->target_units + g.nQT - 1) / g.nQT;
    // few queries: more segments than fill the GPU twice only make every query's record row longer to walk
    // (Q = 64, N = 10M: 12500 slices per query cost the rank stage 3.4 ms; 2048 cost 0.1)
    if (S > c->opt_max_segments) S = c->opt_max_segments;
    const i64 maxS = (c->N + c->min_segment - 1) / c->min_segment;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    i64 L = (c->N + S - 1) / S;
    const i64 lq = (c->opt_select_packed >= 3 && c->NW <= 2) ? 96 : 32;   // k_select_mx2 walks segments in 32-row tiles, k_select_mx3 in 48-row supertiles
    L = (L + lq - 1) / lq * lq;
    if (L < lq) L = lq;
    S = (c->N + L - 1) / L;
    if (S < 1) S = 1;
    if (c->opt_enable && c->opt_select_mfma && S >= 4) {
        // k_select_mx runs (S / 2) x ceil(Q / 256 or 512) equal blocks, 4 or 2 resident per CU: pick the S near the
        // target that fills a whole number of such rounds, so the last round is not a nearly empty one
        const bool qt2 = c->NW <= 4;                                // mirrors launch_select_mx_t
        const bool mx3 = c->opt_select_packed == 3 && c->NW <= 2;   // k_select_mx3: blocks of M3_WPB wavefronts x 64 queries
        const i64 qblk = mx3 ? 64 * M3_WPB : qt2 ? 256 : 512;
        const i64 nQB = (c->Q + qblk - 1) / qblk;
        const i64 slots = (i64)c->n_cu * (mx3 ? 16 / M3_WPB : qt2 ? 4 : 2);
        i64 k = (S / 2 * nQB + slots / 2) / slots;
        if (k < 1) k = 1;
        i64 S2 = 2 * (slots * k / nQB);
        if (S2 > maxS) S2 = maxS / 2 * 2;
        for (; S2 >= 4; S2 -= 2) {                 // rounding L up to 16 rows can drop segments: land on an even count
            i64 L2 = (c->N + S2 - 1) / S2;
            L2 = (L2 + lq - 1) / lq * lq;
            const i64 Sr = (c->N + L2 - 1) / L2;
            if (Sr * 4 < S * 3) break;             // too far from the target: keep the plain choice
            if (Sr % 2 == 0 && Sr * 4 <= S * 5 && (Sr / 2) * nQB <= slots * k) { S = Sr; L = L2; break; }
        }
    }
    g.S = (int)S; g.L = L;
    g.nUnits = (i64)g.S * g.nQT;
    g.hist_stride = 1;
    g.hcap = 0;
    g.wpb = WPB;
    g.nBlk = (int)((g.nUnits + WPB - 1) / WPB);
}

int padded_grid(int nBlk) { return (nBlk + 7) / 8 * 8; }

// The sampled pass only needs the shard total: use 2x longer segments so the per-segment
// histogram array (and its reduction) shrinks with the work.
Geo hist_geometry(const hg_ctx* c) {
    Geo g = c->geo;
    if (g.hist_stride > 1 || c->hist_pairs) {
        const i64 L = g.L * (c->hist_pairs ? 2 : c->opt_sample_ratio);
        g.L = L;
        g.S = (int)((g.N + L - 1) / L);
        g.nUnits = (i64)g.S * g.nQT;
    }
    return g;
}

template <int NW> int launch_hist_t(hg_ctx* c) {
    Geo g = hist_geometry(c);
    // LDS: one u32 histogram column per lane: wpb * NB * 64 * 4 bytes (<= 160 KiB per workgroup)
    int wpb = WPB;
    while (wpb > 1 && (size_t)wpb * g.NB * 256 > 160u * 1024u) wpb >>= 1;
    g.wpb = wpb;
    g.nBlk = (int)((g.nUnits + wpb - 1) / wpb);
    const size_t lds = (size_t)wpb * g.NB * 256;
    if (lds > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    c->t_begin(KI_HIST);
    hipLaunchKernelGGL(k_hist<NW>, dim3(padded_grid(g.nBlk)), dim3(64 * wpb), lds, c->stream,
                       c->qc.as<u32>(), c->db.as<u32>(), c->hist.as<u32>(), g);
    c->t_end();
    return c->check_launch("k_hist");
}

template <int NW, int LW, bool OPT> int launch_select_t(hg_ctx* c) {
    const Geo& g = c->geo;
    SelArgs a{c->optimistic ? c->tguess.as<int>() : c->t.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(),
              c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow, c->optimistic ? 1 : 0, c->sstar.as<int>(), 0};
    c->t_begin(KI_SELECT);
    hipLaunchKernelGGL((k_select<NW, LW, OPT>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->db.as<u32>(), c->dblab.as<u64>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_select");
}

// matrix-core optimistic select: units = (pair of segments) x (group of 32 QT queries)
// matrix-core optimistic select: blocks = (pair of segments) x (block of 512 queries)
// the fp4 images of the codes in MFMA fragment order (k_select_mx, k_hist_mx), built on first use
template <int NW> int ensure_mx_images(hg_ctx* c, const bool need_db = true) {     // need_db = false: the query image only (k_select_mx3 has its own database image)
    constexpr int NM = (NW + 1) / 2;
    if (need_db && !c->dbx_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbx.reserve((size_t)(n16 > 0 ? n16 : 16) * NM * 32));
        const i64 items = n16 * 2 * NM;
        c->t_begin(KI_PACK);
        if (items) hipLaunchKernelGGL(k_expand_db, dim3(grid_for(items)), dim3(256), 0, c->stream, c->db.as<u32>(),
                                      c->dbx.as<uint4>(), (i64)c->N, n16, NW, NM);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db"));
        c->dbx_valid = true;
    }
    if (!c->qx_valid) {
        const i64 qpad = ((i64)c->Q + 511) / 512 * 512;
        HG_TRY(c->qx.reserve((size_t)qpad * NM * 32));
        const i64 items = qpad * 2 * NM;
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_queries, dim3(grid_for(items)), dim3(256), 0, c->stream, c->qc.as<u32>(), c->qx.as<uint4>(),
                           (i64)c->Q, qpad, NW, NM);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_queries"));
        c->qx_valid = true;
    }
    return HG_OK;
}

// histogram on the matrix cores: blocks = (pair of segments) x (256 queries); stride in tiles of 16 rows
bool hist_mx_applies(const hg_ctx* c, int stride, bool pairs_ok) {
    if (!c->opt_hist_mfma || !c->opt_select_mfma || c->NW > 8 || c->is_sub) return false;
    {   // long segments (>= 65536 visited rows per pair) need one dword counter per query tile: with long codes the four
        // wavefronts' columns then exceed the CU's LDS -- the vector kernel, which shrinks its block, takes those
        const Geo& g = c->geo;
        const i64 tiles_per_half = ((g.L + 15) / 16 + stride - 1) / stride;
        const bool pack16 = 2 * tiles_per_half * 16 < 65536;
        if ((size_t)WPB * (pack16 ? 1 : 2) * g.NB * 32 * 4 > 160u * 1024u) return false;
    }
    return stride > 1 ? c->opt_sample_ratio == 2 : pairs_ok;
}
template <int NW> int launch_hist_mx_t(hg_ctx* c) {
    HG_TRY(ensure_mx_images<NW>(c));
    Geo g = c->geo;                                    // fine geometry; the kernel pairs the segments itself
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 255) / 256;
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    const i64 tiles_per_half = ((g.L + 15) / 16 + g.hist_stride - 1) / g.hist_stride;      // visited by one lane-half
    const i64 visited_per_pair = 2 * tiles_per_half * 16;
    const bool pack16 = visited_per_pair < 65536;
    const size_t lds = (size_t)WPB * (pack16 ? 1 : 2) * g.NB * 32 * 4;
    c->t_begin(KI_HIST);
    if (pack16) {
        hipLaunchKernelGGL((k_hist_mx<NW, true>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->qx.as<u8>(),
                           c->dbx.as<u8>(), c->hist.as<u32>(), g);
    } else {
        if (lds > 64 * 1024)
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist_mx<NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_hist_mx<NW, false>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->qx.as<u8>(),
                           c->dbx.as<u8>(), c->hist.as<u32>(), g);
    }
    c->t_end();
    return c->check_launch("k_hist_mx");
}

template <int NW, int LW, int QT, bool COMPACT> int launch_select_mx_q(hg_ctx* c);
template <int NW, int LW> int launch_select_mx_t(hg_ctx* c) {
    // two query tiles per wavefront; codes of up to 128 bits run 4 wavefronts per SIMD, longer codes need the registers of
    // the 2-waves-per-SIMD variant (B fragments: 4 per query tile and 64 bits) -- window lengths: mx_wt()
    constexpr int QT = mx_qt(NW);
    return c->rec8 ? launch_select_mx_q<NW, LW, QT, true>(c) : launch_select_mx_q<NW, LW, QT, false>(c);
}
template <int NW, int LW, int QT, bool COMPACT> int launch_select_mx_q(hg_ctx* c) {
    constexpr int QBLK = WPB * 32 * QT;                // queries per block
    HG_TRY(ensure_mx_images<NW>(c));
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + QBLK - 1) / QBLK;           // query blocks
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    const MxLds L = mx_lds_layout(NW, LW, QT, COMPACT);
    static int lds_set = 0;                            // per instantiation
    if (L.total > 64 * 1024 && lds_set != L.total) {
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_mx<NW, LW, QT, COMPACT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
        lds_set = L.total;
    }
    SelArgs a{c->exact_mx ? c->t.as<int>() : c->tguess.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(),
              c->failq.as<u32>(), c->cap, c->crow, 1, c->sstar.as<int>(), (int)c->opt_probe};
    c->t_begin(KI_SELECT_MX);
    hipLaunchKernelGGL((k_select_mx<NW, LW, QT, COMPACT>), dim3(padded_grid(g.nBlk)), dim3(256), (size_t)L.total + (size_t)c->opt_lds_pad, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->qx.as<u8>(), c->db.as<u32>(), c->dbx.as<u8>(), c->dblab.as<u64>(), a,
                       c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_select_mx");
}

// codes of <= 64 bits: two rows per accumulator (k_select_mx2); blocks = (pair of segments) x (256 queries)
template <int NW, int LW, bool COMPACT> int launch_select_mx2_c(hg_ctx* c);
template <int NW, int LW> int launch_select_mx2_t(hg_ctx* c) {
    return c->rec8 ? launch_select_mx2_c<NW, LW, true>(c) : launch_select_mx2_c<NW, LW, false>(c);
}
template <int NW, int LW, bool COMPACT> int launch_select_mx2_c(hg_ctx* c) {
    if (!c->dbx2_valid) {
        const i64 n32 = (c->N + 31) / 32 * 32;
        HG_TRY(c->dbx2.reserve((size_t)(n32 > 0 ? n32 : 32) * NW * 16));
        const i64 items = n32 * NW;
        c->t_begin(KI_PACK);
        if (items) hipLaunchKernelGGL(k_expand_db2, dim3(grid_for(items)), dim3(256), 0, c->stream, c->db.as<u32>(),
                                      c->dbx2.as<uint4>(), (i64)c->N, n32, NW);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db2"));
        c->dbx2_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 255) / 256;
    if (!c->qx2_valid) {
        const i64 qpad = (i64)nQB * 256;
        HG_TRY(c->qx2.reserve((size_t)qpad * NW * 32));
        const i64 items = qpad * NW;
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_queries2, dim3(grid_for(items)), dim3(256), 0, c->stream, c->qc.as<u32>(), c->qx2.as<uint4>(),
                           (i64)c->Q, qpad, NW);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_queries2"));
        c->qx2_valid = true;
    }
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    const Mx2Lds L = mx2_lds_layout(NW, LW, COMPACT);
    if (L.total > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_mx2<NW, LW, COMPACT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   L.total));
    SelArgs a{c->exact_mx ? c->t.as<int>() : c->tguess.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(),
              c->failq.as<u32>(), c->cap, c->crow, 1, c->sstar.as<int>(), (int)c->opt_probe};
    c->t_begin(KI_SELECT_MX);
    hipLaunchKernelGGL((k_select_mx2<NW, LW, COMPACT>), dim3(padded_grid(g.nBlk)), dim3(256), (size_t)L.total, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->qx2.as<u8>(), c->db.as<u32>(), c->dbx2.as<u8>(), c->dblab.as<u64>(), a,
                       c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_select_mx2");
}

// codes of 33..64 bits, compact records: three rows per accumulator and the batched drain (k_select_mx3);
// blocks = (pair of segments) x (256 queries); the query image is k_select_mx's
template <int NW, int LW> int launch_select_mx3_t(hg_ctx* c) {
    HG_TRY(ensure_mx_images<NW>(c, false));
    if (!c->dbx3_valid) {
        const i64 n48 = (c->N + M3_ROWS - 1) / M3_ROWS * M3_ROWS + M3_WS_MAX * M3_ROWS;     // + one window of zero rows: the last segment's last window may run past the end
        HG_TRY(c->dbx3.reserve((size_t)(n48 > 0 ? n48 : M3_ROWS) * 32));
        const i64 items = n48 * 2;
        c->t_begin(KI_PACK);
        if (items) hipLaunchKernelGGL(k_expand_db3, dim3(grid_for(items)), dim3(256), 0, c->stream, c->db.as<u32>(),
                                      c->dbx3.as<uint4>(), (i64)c->N, n48, NW);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db3"));
        c->dbx3_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 64 * M3_WPB - 1) / (64 * M3_WPB);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = M3_WPB;
    g.nBlk = (int)g.nUnits;
    const Mx3Lds L = mx3_lds_layout(NW, LW);
    if (L.total > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_mx3<NW, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    SelArgs a{c->exact_mx ? c->t.as<int>() : c->tguess.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(),
              c->failq.as<u32>(), c->cap, c->crow, 1, c->sstar.as<int>(), (int)c->opt_probe};
    c->t_begin(KI_SELECT_MX);
    hipLaunchKernelGGL((k_select_mx3<NW, LW>), dim3(padded_grid(g.nBlk)), dim3(64 * M3_WPB), (size_t)L.total + (size_t)c->opt_lds_pad, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->qx.as<u8>(), c->db.as<u32>(), c->dbx3.as<u8>(), c->dblab.as<u64>(), a,
                       c->cand.as<u8>(), g);
    c->t_end();
    return c->check_launch("k_select_mx3");
}

template <int NW, int LW> int launch_select_dense_t(hg_ctx* c) {
    const Geo& g = c->geo;
    SelArgs a{c->t.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(),
              c->cap, c->crow, 0, nullptr, 0};
    c->t_begin(KI_SELECT);
    hipLaunchKernelGGL((k_select_dense<NW, LW>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->db.as<u32>(), c->dblab.as<u64>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_select_dense");
}

template <int NW> int launch_select_nw(hg_ctx* c) {
    const int lw = c->LW <= 2 ? c->LW : 0;           // > 128 classes: match bits come from k_match
    // one-byte records (no index): only the matrix-core kernels of the bet produce them, and only when nobody wants the lists
    const bool mx = c->optimistic && c->opt_select_mfma && c->cap < (1u << MX_POS_BITS);
    c->rec8 = mx && c->opt_compact && !c->want_lists && c->LW <= 2 && c->cap % 16 == 0 && c->crow * 64 < (1ll << 31);
    if (!c->optimistic && c->R * 4 >= c->n_total) {  // dense regime: most pairs are selected
        switch (lw) {
            case 1: return launch_select_dense_t<NW, 1>(c);
            case 2: return launch_select_dense_t<NW, 2>(c);
            default: return launch_select_dense_t<NW, 0>(c);
        }
    }
    // three rows per accumulator + batched drain: codes of <= 64 bits, one-byte records (<= 128 classes).  (For <= 32 bits the
    // second k-half of every MFMA is empty, and it still beats k_select_mx2's two rows per accumulator: 0.69 vs 0.85 ms at b = 32.)
    if (NW <= 2 && c->opt_select_packed == 3 && c->rec8 && c->geo.L % M3_ROWS == 0 && (lw == 1 || lw == 2)) {
        if (lw == 1) return launch_select_mx3_t<(NW == 2 ? 2 : 1), 1>(c);
        return launch_select_mx3_t<(NW == 2 ? 2 : 1), 2>(c);
    }
    // two rows per accumulator: wins for one-word codes (half the MFMAs: 0.92 vs 1.02 ms at b = 32); for 33-64 bits
    // its cheaper harvest (0.36 vs 0.44 ms) is eaten by the wider queue entries (select_packed = 2 forces it)
    if (c->optimistic && c->opt_select_mfma && c->cap < (1u << MX_POS_BITS) && c->geo.L % 32 == 0 &&
        (((c->opt_select_packed == 1 || c->opt_select_packed >= 3) && NW == 1) || (c->opt_select_packed == 2 && NW <= 2))) {
        switch (lw) {
            case 1: return launch_select_mx2_t<(NW <= 2 ? NW : 1), 1>(c);
            case 2: return launch_select_mx2_t<(NW <= 2 ? NW : 1), 2>(c);
            default: return launch_select_mx2_t<(NW <= 2 ? NW : 1), 0>(c);
        }
    }
    if (c->optimistic && c->opt_select_mfma && c->cap < (1u << MX_POS_BITS)) {
        switch (lw) {
            case 1: return launch_select_mx_t<NW, 1>(c);
            case 2: return launch_select_mx_t<NW, 2>(c);
            default: return launch_select_mx_t<NW, 0>(c);
        }
    }
    if (c->optimistic) {
        switch (lw) {
            case 1: return launch_select_t<NW, 1, true>(c);
            case 2: return launch_select_t<NW, 2, true>(c);
            default: return launch_select_t<NW, 0, true>(c);
        }
    }
    switch (lw) {
        case 1: return launch_select_t<NW, 1, false>(c);
        case 2: return launch_select_t<NW, 2, false>(c);
        default: return launch_select_t<NW, 0, false>(c);
    }
}

#define HG_DISPATCH_NW(fn, c)                                   \
    switch ((c)->NW) {                                          \
        case 1: return fn<1>(c);                                \
        case 2: return fn<2>(c);                                \
        case 3: return fn<3>(c);                                \
        case 4: return fn<4>(c);                                \
        case 5: return fn<5>(c);                                \
        case 6: return fn<6>(c);                                \
        case 7: return fn<7>(c);                                \
        case 8: return fn<8>(c);                                \
        default: return fail(HG_ERR_ARG, "unsupported code length: %d words", (c)->NW); \
    }

int launch_hist(hg_ctx* c) { HG_DISPATCH_NW(launch_hist_t, c) }
// the same pass with the integer matrix instruction delivering the counter addresses (hg_hist_i8.hpp); codes of <= 128 bits
template <int NW> int launch_hist_i8_t(hg_ctx* c) {
    if (!c->dbx8_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbx8.reserve((size_t)n16 * NW * 32));
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_db_i8, dim3(grid_for(n16 * NW * 2)), dim3(256), 0, c->stream, c->db.as<u32>(), c->dbx8.as<uint4>(),
                           (i64)c->N, n16, NW);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db_i8"));
        c->dbx8_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 255) / 256;
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    const i64 tiles_per_half = ((g.L + 15) / 16 + g.hist_stride - 1) / g.hist_stride;
    const bool pack16 = 2 * tiles_per_half * 16 < 65536;
    const size_t lds = (size_t)WPB * hist_i8_cols(pack16) * g.NB * 32 * 4;
    c->t_begin(KI_HIST);
    if (pack16) {
        if (lds > 64 * 1024)
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist_i8<NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_hist_i8<NW, true>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->dbx8.as<u8>(),
                           c->hist.as<u32>(), g);
    } else {
        if (lds > 64 * 1024)
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist_i8<NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_hist_i8<NW, false>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->dbx8.as<u8>(),
                           c->hist.as<u32>(), g);
    }
    c->t_end();
    return c->check_launch("k_hist_i8");
}
int launch_hist_mx(hg_ctx* c) {
    if (c->opt_hist_mfma == 2 && c->NW <= 4) {
        switch (c->NW) {
            case 1: return launch_hist_i8_t<1>(c);
            case 2: return launch_hist_i8_t<2>(c);
            case 3: return launch_hist_i8_t<3>(c);
            default: return launch_hist_i8_t<4>(c);
        }
    }
    HG_DISPATCH_NW(launch_hist_mx_t, c)
}
int launch_select(hg_ctx* c) { HG_DISPATCH_NW(launch_select_nw, c) }

// rows k_hist visits with batch stride `stride` (mirrors its loop)
template <int NW> i64 sampled_rows_t(Geo g, int stride, int ratio) {
    g.hist_stride = stride;
    {
        const i64 L = g.L * ratio;                  // mirrors hist_geometry()
        g.L = L;
        g.S = (int)((g.N + L - 1) / L);
    }
    constexpr int B = Batch<NW>::rows;
    i64 total = 0;
    for (int s = 0; s < g.S; ++s) {
        const i64 lo = (i64)s * g.L, hi = lo + g.L < g.N ? lo + g.L : g.N;
        const i64 nb = (hi - lo) / B;
        total += (nb + stride - 1) / stride * B;
    }
    return total;
}
i64 sampled_rows(hg_ctx* c, int stride) {
    if (hist_mx_applies(c, stride, false)) return hist_mx_sampled_rows(c->geo, stride);
    switch (c->NW) {
        case 1: return sampled_rows_t<1>(c->geo, stride, (int)c->opt_sample_ratio);
        case 2: return sampled_rows_t<2>(c->geo, stride, (int)c->opt_sample_ratio);
        case 3: return sampled_rows_t<3>(c->geo, stride, (int)c->opt_sample_ratio);
        case 4: return sampled_rows_t<4>(c->geo, stride, (int)c->opt_sample_ratio);
        case 5: return sampled_rows_t<5>(c->geo, stride, (int)c->opt_sample_ratio);
        case 6: return sampled_rows_t<6>(c->geo, stride, (int)c->opt_sample_ratio);
        case 7: return sampled_rows_t<7>(c->geo, stride, (int)c->opt_sample_ratio);
        default: return sampled_rows_t<8>(c->geo, stride, (int)c->opt_sample_ratio);
    }
}

int need(hg_ctx* c, unsigned st, const char* who, const char* what) {
    if (!c) return fail(HG_ERR_ARG, "%s: null context", who);
    if ((c->stage & st) != st) return fail(HG_ERR_STATE, "%s called before %s", who, what);
    return c->use();
}

// Upload packed uint64 codes as dense uint32 [n][NW] (NW = ceil(b/32)): when NW is
// odd the unused high half of the last uint64 word is dropped by a strided copy.
int upload_codes(hg_ctx* c, DevBuf& dst, const uint64_t* host, i64 n, int W, int NW) {
    HG_TRY(dst.reserve((size_t)(n > 0 ? n : 1) * NW * 4 + 64 * 4));  // +64 words: scalar loads may read past a ragged tail
    if (n == 0) return HG_OK;
    if (NW == 2 * W) {
        HG_HIP(hipMemcpyAsync(dst.p, host, (size_t)n * NW * 4, hipMemcpyHostToDevice, c->stream));
    } else {
        HG_HIP(hipMemcpy2DAsync(dst.p, (size_t)NW * 4, host, (size_t)W * 8, (size_t)NW * 4, (size_t)n,
                                hipMemcpyHostToDevice, c->stream));
    }
    return HG_OK;
}

template <int BP> int real_launch_sample(hg_ctx* c, i64 M, i64 stride) {
    const Geo& g = c->geo;
    const i64 units = (M + 63) / 64 * g.nQT;
    c->t_begin(KI_REAL_SAMPLE);
    hipLaunchKernelGGL(k_real_sample<BP>, dim3(grid_for(units, WPB)), dim3(256), (size_t)WPB * 64 * 65 * 4, c->stream,
                       c->qf.as<float>(), c->dbf.as<float>(), c->samp.as<float>(), M, stride, g);
    c->t_end();
    return c->check_launch("k_real_sample");
}
template <int BP, int QPL> int real_launch_select_q(hg_ctx* c) {
    Geo g = c->geo;
    g.nQT = (g.Q + 64 * QPL - 1) / (64 * QPL);
    g.nUnits = (i64)g.S * g.nQT;
    g.wpb = WPB;
    g.nBlk = (int)((g.nUnits + WPB - 1) / WPB);
    RealSelArgs a{c->thr.as<float>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow};
    c->t_begin(KI_REAL_SELECT);
    hipLaunchKernelGGL((k_real_select<BP, QPL>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qf.as<float>(),
                       c->dbf.as<float>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_real_select");
}
template <int BP> int real_launch_select(hg_ctx* c) {
    // queries per lane (see k_real_select): two while their features fit the register file comfortably
    if (BP <= 32 && c->opt_real_qpl == 2) return real_launch_select_q<BP, (BP <= 32 ? 2 : 1)>(c);
    return real_launch_select_q<BP, 1>(c);
}
// real-valued select on the matrix cores: blocks = (pair of segments) x (256 queries)
template <int KP> int real_launch_select_mx(hg_ctx* c) {
    if (!c->dbfx_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbfx.reserve((size_t)n16 * KP * 4));
        const i64 items = n16 * (KP / 4);
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_dbf, dim3(grid_for(items)), dim3(256), 0, c->stream, c->dbf.as<float>(), c->dbfx.as<float4>(),
                           (i64)c->N, n16, KP, (i64)1);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_dbf"));
        c->dbfx_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 32 * RMX_QT - 1) / (WPB * 32 * RMX_QT);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    RealSelArgs a{c->thr.as<float>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow};
    c->t_begin(KI_REAL_SELECT);
    hipLaunchKernelGGL((k_real_select_mx<KP>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream,
                       c->qf.as<float>(), c->dbfx.as<u8>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_real_select_mx");
}
// filter + rescore (hg_real_bf.hpp): bf16 pair pass with a rigorous margin, then the exact chain for the survivors
template <int KP> int real_launch_select_bf(hg_ctx* c) {
    constexpr int QT = KP <= 128 ? 2 : 1;
    if (!c->dbfb_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbfb.reserve((size_t)n16 * KP * 2));
        HG_TRY(c->xmax2.reserve(4));
        HG_HIP(hipMemsetAsync(c->xmax2.p, 0, 4, c->stream));
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_dbf_bf16, dim3(grid_for(n16 * (KP / 8))), dim3(256), 0, c->stream, c->dbf.as<float>(),
                           c->dbfb.as<uint4>(), (i64)c->N, n16, KP);
        hipLaunchKernelGGL(k_row_norm_max, dim3(grid_for(c->N)), dim3(256), 0, c->stream, c->dbf.as<float>(), (i64)c->N, KP, c->xmax2.as<u32>());
        c->t_end();
        HG_TRY(c->check_launch("k_expand_dbf_bf16"));
        c->dbfb_valid = true;
    }
    Geo g = c->geo;
    HG_TRY(c->thr2.reserve((size_t)g.Qpad * 4));
    c->t_begin(KI_REAL_GUESS);
    hipLaunchKernelGGL(k_real_thr2, dim3(grid_for(g.Q)), dim3(256), 0, c->stream, c->qf.as<float>(), c->thr.as<float>(),
                       c->xmax2.as<u32>(), c->thr2.as<float>(), g.Q, KP);
    c->t_end();
    HG_TRY(c->check_launch("k_real_thr2"));
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 32 * QT - 1) / (WPB * 32 * QT);
    Geo gs = g;
    gs.nQT = nQB;
    gs.nUnits = (i64)nSP * nQB;
    gs.wpb = WPB;
    gs.nBlk = (int)gs.nUnits;
    RealSelArgs a{c->thr.as<float>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow};
    c->t_begin(KI_REAL_SELECT);
    if (real_bf_lds_bytes(KP) > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_select_bf<KP, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, real_bf_lds_bytes(KP)));
    hipLaunchKernelGGL((k_real_select_bf<KP, QT>), dim3(padded_grid(gs.nBlk)), dim3(256), real_bf_lds_bytes(KP), c->stream,
                       c->qf.as<float>(), c->dbfb.as<u8>(), c->thr2.as<float>(), a, c->cand.as<u64>(), gs);
    c->t_end();
    HG_TRY(c->check_launch("k_real_select_bf"));
    // slices per wavefront of the rescoring pass: about one round of 64 kept rows (the filter keeps ~2 R per query)
    const double per_slice = 2.0 * (double)c->R / (double)g.S;
    const int SG = per_slice * 8 <= 60 ? 8 : per_slice * 4 <= 60 ? 4 : per_slice * 3 <= 60 ? 3 : per_slice * 2 <= 60 ? 2 : 1;
    const i64 waves = (i64)((g.S + SG - 1) / SG) * g.Q;
    c->t_begin(KI_REAL_RESCORE);
#define HG_RESCORE(sg)                                                                                                                   \
    case sg:                                                                                                                             \
        hipLaunchKernelGGL((k_real_rescore<(KP <= 128 ? KP : 0), sg>), dim3(grid_for(waves, WPB)), dim3(256), rescore_lds_bytes(), c->stream, c->qf.as<float>(),  \
                           c->dbf.as<float>(), c->sl_cnt.as<u32>(), c->cand.as<u64>(), c->cap, c->crow, c->thr.as<float>(),               \
                           c->sl_cnt.as<u32>(), KP, g);                                                                                  \
        break;
    switch (SG) { HG_RESCORE(8) HG_RESCORE(4) HG_RESCORE(3) HG_RESCORE(2) HG_RESCORE(1) }
#undef HG_RESCORE
    c->t_end();
    c->real_filtered = true;
    return c->check_launch("k_real_rescore");
}
int real_select_bf(hg_ctx* c) {
    switch (c->bpad) {
        case 16: return real_launch_select_bf<16>(c);
        case 32: return real_launch_select_bf<32>(c);
        case 48: return real_launch_select_bf<48>(c);
        case 64: return real_launch_select_bf<64>(c);
        case 80: return real_launch_select_bf<80>(c);
        case 96: return real_launch_select_bf<96>(c);
        case 112: return real_launch_select_bf<112>(c);
        case 128: return real_launch_select_bf<128>(c);
        case 144: return real_launch_select_bf<144>(c);
        case 160: return real_launch_select_bf<160>(c);
        case 176: return real_launch_select_bf<176>(c);
        case 192: return real_launch_select_bf<192>(c);
        case 208: return real_launch_select_bf<208>(c);
        case 224: return real_launch_select_bf<224>(c);
        case 240: return real_launch_select_bf<240>(c);
        case 256: return real_launch_select_bf<256>(c);
        default: return fail(HG_ERR_ARG, "real-valued ranking supports up to 256 features (have %d)", c->b);
    }
}
int real_select_mx(hg_ctx* c) {
    switch (c->bpad) {
        case 16: return real_launch_select_mx<16>(c);
        case 32: return real_launch_select_mx<32>(c);
        case 48: return real_launch_select_mx<48>(c);
        case 64: return real_launch_select_mx<64>(c);
        case 80: return real_launch_select_mx<80>(c);
        case 96: return real_launch_select_mx<96>(c);
        case 112: return real_launch_select_mx<112>(c);
        case 128: return real_launch_select_mx<128>(c);
        default: return fail(HG_ERR_ARG, "real-valued ranking supports up to 128 features (have %d)", c->b);
    }
}

#define HG_DISPATCH_BP(fn, c, ...)                                  \
    switch ((c)->bpad / 2) {                                        \
        case 8: return fn<8>(c, ##__VA_ARGS__);                     \
        case 16: return fn<16>(c, ##__VA_ARGS__);                   \
        case 24: return fn<24>(c, ##__VA_ARGS__);                   \
        case 32: return fn<32>(c, ##__VA_ARGS__);                   \
        case 40: return fn<40>(c, ##__VA_ARGS__);                   \
        case 48: return fn<48>(c, ##__VA_ARGS__);                   \
        case 56: return fn<56>(c, ##__VA_ARGS__);                   \
        case 64: return fn<64>(c, ##__VA_ARGS__);                   \
        default: return fail(HG_ERR_ARG, "real-valued ranking supports up to 128 features (have %d)", (c)->b); \
    }
// sample pass on the float32 MFMA: image of the M sampled rows (rebuilt per call: a few MB), 16 segments
template <int KP> int real_launch_sample_mx(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    const i64 m16 = (M + 15) / 16 * 16;
    HG_TRY(c->sampx.reserve((size_t)m16 * KP * 4));
    c->t_begin(KI_REAL_SAMPLE);
    hipLaunchKernelGGL(k_expand_dbf, dim3(grid_for(m16 * (KP / 4))), dim3(256), 0, c->stream, c->dbf.as<float>(), c->sampx.as<float4>(),
                       M, m16, KP, stride);
    Geo g = c->geo;
    g.N = M;
    i64 L = (M + 31) / 32;                               // ~32 segments (16 pairs) of a multiple of 16 rows
    L = (L + 15) / 16 * 16;
    g.L = L;
    g.S = (int)((M + L - 1) / L);
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 32 * RMX_QT - 1) / (WPB * 32 * RMX_QT);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    hipLaunchKernelGGL((k_real_sample_mx<KP>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qf.as<float>(), c->sampx.as<u8>(),
                       c->samp.as<float>(), mstride, g);
    c->t_end();
    return c->check_launch("k_real_sample_mx");
}
int real_sample_mx(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    switch (c->bpad) {
        case 16: return real_launch_sample_mx<16>(c, M, stride, mstride);
        case 32: return real_launch_sample_mx<32>(c, M, stride, mstride);
        case 48: return real_launch_sample_mx<48>(c, M, stride, mstride);
        case 64: return real_launch_sample_mx<64>(c, M, stride, mstride);
        case 80: return real_launch_sample_mx<80>(c, M, stride, mstride);
        case 96: return real_launch_sample_mx<96>(c, M, stride, mstride);
        case 112: return real_launch_sample_mx<112>(c, M, stride, mstride);
        default: return real_launch_sample_mx<128>(c, M, stride, mstride);
    }
}
int real_sample(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    if (c->bpad <= 128 && c->opt_real_mfma) return real_sample_mx(c, M, stride, mstride);
    if (c->bpad > 128) {                                 // k_real_sample keeps the query in registers: the staged form beyond
        const Geo& g = c->geo;
        const i64 units = (M + 63) / 64 * g.nQT;
        const size_t lds = (size_t)WPB * (64 * 65 * 4 + 16 * 128);
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_sample_any), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->t_begin(KI_REAL_SAMPLE);
        hipLaunchKernelGGL(k_real_sample_any, dim3(grid_for(units, WPB)), dim3(256), lds, c->stream, c->qf.as<float>(), c->dbf.as<float>(),
                           c->samp.as<float>(), M, stride, c->bpad, g);
        c->t_end();
        return c->check_launch("k_real_sample_any");
    }
    (void)mstride;                                       // (the vector kernels write samp[q][M] densely: the caller passes mstride = M)
    HG_DISPATCH_BP(real_launch_sample, c, M, stride)
}
int real_select(hg_ctx* c) {
    c->real_filtered = false;
    // without a cut (every row a record: R = N, or after lost bets) a filter filters nothing and every pair would be rescored:
    // the exact float32 MFMA pass gives the scores at once (C1: 4.6 -> 4.0 ms per call)
    const bool filter = c->opt_real_mfma == 2 && !(c->real_no_cut && c->bpad <= 128);
    if ((filter || c->bpad > 128) && c->geo.L % 16 == 0) return real_select_bf(c);   // (the only pass for > 128 features)
    if (c->opt_real_mfma && c->geo.L % 16 == 0) return real_select_mx(c);
    HG_DISPATCH_BP(real_launch_select, c)
}


}  // namespace

// =============================================================================
extern "C" {

const char* hg_last_error(void) { return g_err.c_str(); }
int hg_version(void) { return 100; }

int hg_device_count(int* count) {
    if (!count) return fail(HG_ERR_ARG, "hg_device_count: null pointer");
    HG_HIP(hipGetDeviceCount(count));
    return HG_OK;
}

int hg_init(int device, hg_ctx** out) {
    if (!out) return fail(HG_ERR_ARG, "hg_init: null pointer");
    *out = nullptr;
    int n = 0;
    HG_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(HG_ERR_ARG, "hg_init: device %d out of range (%d visible)", device, n);
    HG_HIP(hipSetDevice(device));
    // A step is a few milliseconds and ends in one stream synchronisation: spin instead of sleeping on it.
    // (Refused when the device is already initialised, e.g. by torch -- harmless.)
    if (hipSetDeviceFlags(hipDeviceScheduleSpin) != hipSuccess) (void)hipGetLastError();
    hg_ctx* c = new hg_ctx();
    c->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cu = cus;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(HG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    *out = c;
    return HG_OK;
}

int hg_destroy(hg_ctx* c) {
    if (!c) return HG_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->t_collect();
    c->drop_graph();
    for (auto e : c->pool) (void)hipEventDestroy(e);
    DevBuf* all[] = {&c->db, &c->dblab, &c->qc, &c->qlab, &c->hist, &c->hown, &c->posbase, &c->seglt, &c->segtie,
                     &c->t, &c->tguess, &c->sstar, &c->cnt_lt, &c->quota, &c->tie_before, &c->n_lt, &c->err, &c->sl_start,
                     &c->sl_tie, &c->sl_cnt, &c->tot, &c->failq, &c->cand, &c->out_idx, &c->out_dist, &c->mbits,
                     &c->shapes, &c->ap, &c->rel, &c->stage_in, &c->badcnt, &c->qbad, &c->flist, &c->hwq, &c->dbf, &c->qf, &c->samp, &c->thr,
                     &c->sortA, &c->sortB, &c->gtab, &c->scores, &c->dbx, &c->qx, &c->bigq, &c->dbx2, &c->qx2, &c->mbits2, &c->dbfx, &c->dbfb, &c->thr2, &c->xmax2, &c->dbx8, &c->dbx3, &c->sampx, &c->ap_recip, &c->part};
    for (auto* d : all) d->release();
    for (auto& d : c->gathered) d.release();
    for (auto& d : c->scratch) d.release();
    c->comm_tmp.release(); c->gath_idx.release(); c->gath_dist.release();
    if (c->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    if (c->sub) { hg_ctx* s = c->sub; c->sub = nullptr; (void)hg_destroy(s); }
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->hpk) (void)hipHostFree(c->hpk);
    if (c->fstage) (void)hipHostFree(c->fstage);
    if (c->stream2_ev) (void)hipEventDestroy(c->stream2_ev);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    for (auto& e : c->fstage_ev) if (e) (void)hipEventDestroy(e);
    if (c->stream && c->own_stream && !c->is_sub) (void)hipStreamDestroy(c->stream);
    delete c;
    return HG_OK;
}

int hg_pack_sign_f32(const float* x, int64_t n, int b, uint64_t* out) {
    if (!x || !out || n < 0 || b < 1) return fail(HG_ERR_ARG, "hg_pack_sign_f32: bad argument");
    const int W = (b + 63) / 64;
    for (int64_t i = 0; i < n; ++i) {
        const float* row = x + i * b;
        for (int w = 0; w < W; ++w) {
            uint64_t v = 0;
            const int hi = b - w * 64 < 64 ? b - w * 64 : 64;
            for (int j = 0; j < hi; ++j) v |= (uint64_t)(row[w * 64 + j] > 0.0f) << j;
            out[i * W + w] = v;
        }
    }
    return HG_OK;
}

int hg_set_database(hg_ctx* c, const uint64_t* codes, const uint64_t* labels, int64_t N, int b, int C,
                    int64_t idx_base, int64_t n_total) {
    if (!c) return fail(HG_ERR_ARG, "hg_set_database: null context");
    if (N < 1 || !codes || !labels) return fail(HG_ERR_ARG, "hg_set_database: need N >= 1 and data");
    if (b < 1 || b > HG_MAX_BITS) return fail(HG_ERR_ARG, "hg_set_database: b=%d outside 1..%d", b, HG_MAX_BITS);
    if (C < 1) return fail(HG_ERR_ARG, "hg_set_database: C=%d", C);
    if (idx_base < 0 || n_total < N || idx_base + N > n_total || n_total >= 0xFFFFFFFFll)
        return fail(HG_ERR_ARG, "hg_set_database: shard [%lld, %lld) does not fit a database of %lld rows (< 2^32 - 1)",
                    (long long)idx_base, (long long)(idx_base + N), (long long)n_total);
    HG_TRY(c->use());
    c->N = N; c->b = b; c->C = C; c->n_total = n_total;
    c->NW = (b + 31) / 32; c->NB = b + 1; c->LW = (C + 63) / 64;
    c->idx_base = (u32)idx_base;
    c->bpad = 0;                                       // packed input: no float tables for the real-valued path
    c->dbf_resident = false;
    HG_TRY(upload_codes(c, c->db, codes, N, (b + 63) / 64, c->NW));
    HG_TRY(c->dblab.reserve((size_t)(N > 0 ? N : 1) * c->LW * 8));
    if (N) HG_HIP(hipMemcpyAsync(c->dblab.p, labels, (size_t)N * c->LW * 8, hipMemcpyHostToDevice, c->stream));
    HG_TRY(c->sync());
    c->stage = ST_DB;   // queries must be (re)set after the database: b, C may have changed
    c->dbx_valid = false;
    c->dbx2_valid = false;
    c->dbx3_valid = false;
    c->dbx8_valid = false;
    c->opt_consecutive_fail = c->shard_bet_fail = 0;    // a new database: earlier lost bets say nothing about it
    c->cap_boost = c->real_cap_boost = 1;
    c->cfg_epoch++;
    return HG_OK;
}

// float32 features + int64 labels -> packed device tables, packed ON THE GPU (k_pack_sign_f32 / k_pack_labels_i64)
static int pack_on_device(hg_ctx* c, const float* x, const int64_t* lab, i64 n, DevBuf& codes, DevBuf& labels,
                          DevBuf& feats, int64_t* bad_codes, int64_t* bad_labels, i64 (&census)[3]) {
    const int b = c->b, C = c->C, NW = c->NW, LW = c->LW;
    // the float table stays resident, zero-padded to a multiple of 16 features: the real-valued
    // ranking (hg_map_real) streams it, and padding keeps its rows 64-byte aligned
    const int bpad = (b + 15) / 16 * 16;
    c->bpad = bpad;
    const size_t fb = (size_t)n * bpad * 4, lb = (size_t)n * C * 8;
    HG_TRY(feats.reserve(fb + 256));
    HG_TRY(c->stage_in.reserve(lb));
    HG_TRY(c->badcnt.reserve(32));
    HG_TRY(codes.reserve((size_t)n * NW * 4 + 64 * 4));
    HG_TRY(labels.reserve((size_t)n * LW * 8));
    HG_HIP(hipMemsetAsync(c->badcnt.p, 0, 32, c->stream));
    if (bpad != b) HG_HIP(hipMemsetAsync(feats.p, 0, fb, c->stream));
    HG_HIP(hipMemcpy2DAsync(feats.p, (size_t)bpad * 4, x, (size_t)b * 4, (size_t)b * 4, (size_t)n, hipMemcpyHostToDevice, c->stream));
    c->t_begin(KI_PACK);
    hipLaunchKernelGGL(k_pack_sign_f32, dim3(grid_for(n, WPB)), dim3(256), 0, c->stream, feats.as<float>(), (i64)bpad,
                       codes.as<u32>(), n, b, NW, c->badcnt.as<unsigned long long>());
    c->t_end();
    HG_TRY(c->check_launch("k_pack_sign_f32"));
    HG_HIP(hipMemcpyAsync(c->stage_in.p, lab, lb, hipMemcpyHostToDevice, c->stream));
    c->t_begin(KI_PACK);
    hipLaunchKernelGGL(k_pack_labels_i64, dim3(grid_for(n, WPB)), dim3(256), 0, c->stream, c->stage_in.as<long long>(),
                       labels.as<u64>(), n, C, LW, c->badcnt.as<unsigned long long>());
    c->t_end();
    HG_TRY(c->check_launch("k_pack_labels_i64"));
    unsigned long long bad[4] = {0, 0, 0, 0};
    HG_HIP(hipMemcpyAsync(bad, c->badcnt.p, 32, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    if (bad_codes) *bad_codes = (int64_t)bad[0];
    if (bad_labels) *bad_labels = (int64_t)bad[1];
    census[0] = (i64)bad[0]; census[1] = (i64)bad[2]; census[2] = (i64)bad[3];
    return HG_OK;
}

// The same hand-over with the packing done by host threads BEFORE the upload (hg_host_pack.hpp): 16 MB instead of 339 MB
// cross PCIe at C2.  The float table follows only when somebody will rank by inner product (`floats`: 0 no, 1 yes,
// 2 = iff the table is not a +-1 code).  *has_floats tells what happened.
// A big float table (256 MB at 1M x 64) on its way to the GPU: the runtime stages a pageable source at ~25 GB/s.  Host
// threads copy 16 MB chunks (rows padded on the way) into four pinned buffers instead, each chunk's DMA runs while the next
// is copied.  Enqueues on `stream`; which_pool: the host pool that copies (1 while pool 0 packs).
static int stage_floats(hg_ctx* c, const float* x, i64 n, int b, int bpad, DevBuf& feats, hipStream_t stream, int which_pool) {
    constexpr int NSL = 4;
    const size_t CH = (size_t)16 << 20;
    if (!c->fstage) {
        HG_HIP(hipHostMalloc(&c->fstage, CH * NSL, hipHostMallocDefault));
        for (int k = 0; k < NSL; ++k) HG_HIP(hipEventCreateWithFlags(&c->fstage_ev[k], hipEventDisableTiming));
    }
    const i64 rows_per = (i64)(CH / ((size_t)bpad * 4));
    int slot = 0;
    bool used[NSL] = {false, false, false, false};
    try {
        for (i64 r0 = 0; r0 < n; r0 += rows_per, slot = (slot + 1) % NSL) {
            const i64 r1 = r0 + rows_per < n ? r0 + rows_per : n;
            if (used[slot]) HG_HIP(hipEventSynchronize(c->fstage_ev[slot]));
            float* st = (float*)((char*)c->fstage + (size_t)slot * CH);
            host_copy_rows(x, r0, r1, b, bpad, st, (int)c->opt_pack_threads, which_pool);
            HG_HIP(hipMemcpyAsync((char*)feats.p + (size_t)r0 * bpad * 4, st, (size_t)(r1 - r0) * bpad * 4, hipMemcpyHostToDevice, stream));
            HG_HIP(hipEventRecord(c->fstage_ev[slot], stream));
            used[slot] = true;
        }
    } catch (const std::exception& e) {
        return fail(HG_ERR_NOMEM, "host-side staging of the float table failed: %s", e.what());
    }
    return HG_OK;
}

static int pack_on_host(hg_ctx* c, const float* x, const int64_t* lab, i64 n, DevBuf& codes, DevBuf& labels,
                        DevBuf& feats, int floats, bool* has_floats, int64_t* bad_codes, int64_t* bad_labels, i64 (&census)[3]) {
    const int b = c->b, C = c->C, NW = c->NW, LW = c->LW;
    const size_t cb = (size_t)n * NW * 4, lbytes = (size_t)n * LW * 8;
    const size_t need_b = ((cb + 63) & ~(size_t)63) + lbytes;
    if (c->hpk_cap < need_b) {
        HG_TRY(c->sync());
        if (c->hpk) (void)hipHostFree(c->hpk);
        c->hpk = nullptr; c->hpk_cap = 0;
        HG_HIP(hipHostMalloc(&c->hpk, need_b, hipHostMallocDefault));
        c->hpk_cap = need_b;
    }
    u32* hc = (u32*)c->hpk;
    u64* hl = (u64*)((char*)c->hpk + ((cb + 63) & ~(size_t)63));
    HostPackCensus cs;
    HG_TRY(codes.reserve(cb + 64 * 4));
    HG_TRY(labels.reserve(lbytes));
    // the packed rows cross PCIe while the host threads still pack the rest: the calling thread ships every finished
    // prefix (a quarter of a big table at a time), the workers claim row ranges in ascending order
    i64 shipped = 0;
    hipError_t ship_err = hipSuccess;
    const i64 piece = n >= (1 << 18) ? (n + 3) / 4 : n;
    auto ship = [&](long long rows_done) {
        if (rows_done < n && rows_done - shipped < piece) return;
        if (rows_done <= shipped || ship_err != hipSuccess) return;
        hipError_t e = hipMemcpyAsync((char*)codes.p + (size_t)shipped * NW * 4, (const char*)hc + (size_t)shipped * NW * 4,
                                      (size_t)(rows_done - shipped) * NW * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync((char*)labels.p + (size_t)shipped * LW * 8, (const char*)hl + (size_t)shipped * LW * 8,
                               (size_t)(rows_done - shipped) * LW * 8, hipMemcpyHostToDevice, c->stream);
        ship_err = e;
        shipped = rows_done;
    };
    // Will the float table follow?  keep_floats = 1: yes; = 2: only if the features are no +-1 code -- and a single entry
    // that is neither -1 nor +1 among the first rows settles that before the census is in (tanh outputs: the first entry).
    // Then a second thread stages and ships the floats (its own small pool, its own stream) WHILE the packing pool works.
    const int bpad_f = (b + 15) / 16 * 16;
    const size_t fb_f = (size_t)n * bpad_f * 4;
    bool early = false;
    if (x && floats >= 1 && fb_f >= ((size_t)8 << 20)) {
        early = floats == 1;
        const i64 probe = (i64)std::min<i64>(n, 64) * b;
        for (i64 k = 0; k < probe && !early; ++k) early = !(x[k] == 1.0f || x[k] == -1.0f);
    }
    int stage_rc = HG_OK;
    std::string stage_msg;                               // (the error text is thread-local: carried over by hand)
    std::thread stager;
    if (early) {
        if (!c->stream2) HG_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        if (!c->stream2_ev) HG_HIP(hipEventCreateWithFlags(&c->stream2_ev, hipEventDisableTiming));
        HG_TRY(feats.reserve(fb_f + 256));
        try {
            stager = std::thread([&] {
                if (hipSetDevice(c->device) != hipSuccess) { stage_rc = HG_ERR_HIP; stage_msg = "hipSetDevice failed in the staging thread"; return; }
                stage_rc = stage_floats(c, x, n, b, bpad_f, feats, c->stream2, 1);
                if (stage_rc != HG_OK) stage_msg = g_err;
            });
        } catch (const std::exception&) {
            early = false;                               // no second thread to be had: the floats follow the packing
        }
    }
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{stager};
    try {
        host_pack_ship(x, lab, n, b, C, hc, hl, &cs, (int)c->opt_pack_threads,
                       +[](void* f, long long rows) { (*static_cast<decltype(ship)*>(f))(rows); }, &ship);
    } catch (const std::exception& e) {               // no exception crosses the C ABI (thread creation can fail)
        return fail(HG_ERR_NOMEM, "host-side packing failed: %s", e.what());
    }
    if (ship_err != hipSuccess) return fail(HG_ERR_HIP, "upload of the packed tables failed: %s", hipGetErrorString(ship_err));
    const bool pm1 = cs.nonbinary == 0 && cs.zeros == 0;
    const bool up = floats == 1 || (floats == 2 && !pm1);
    if (stager.joinable()) stager.join();
    if (early) {
        if (stage_rc != HG_OK) return fail(stage_rc, "%s", stage_msg.c_str());
        HG_HIP(hipEventRecord(c->stream2_ev, c->stream2));
        HG_HIP(hipStreamWaitEvent(c->stream, c->stream2_ev, 0));
    }
    if (up) {
        const int bpad = (b + 15) / 16 * 16;
        c->bpad = bpad;
        const size_t fb = (size_t)n * bpad * 4;
        HG_TRY(feats.reserve(fb + 256));
        if (!early) {
            if (fb < ((size_t)8 << 20)) {
                if (bpad != b) HG_HIP(hipMemsetAsync(feats.p, 0, fb, c->stream));
                HG_HIP(hipMemcpy2DAsync(feats.p, (size_t)bpad * 4, x, (size_t)b * 4, (size_t)b * 4, (size_t)n, hipMemcpyHostToDevice, c->stream));
            } else {
                HG_TRY(stage_floats(c, x, n, b, bpad, feats, c->stream, 0));
            }
        }
    }
    *has_floats = up;
    HG_TRY(c->sync());                                 // the pinned staging is reused by the next call
    if (bad_codes) *bad_codes = (int64_t)cs.nonbinary;
    if (bad_labels) *bad_labels = (int64_t)cs.bad_labels;
    census[0] = cs.nonbinary; census[1] = cs.zeros; census[2] = cs.minus_ones;
    return HG_OK;
}

int hg_set_database_f32(hg_ctx* c, const float* host_x, const int64_t* host_labels, int64_t N, int b, int C,
                        int64_t idx_base, int64_t n_total, int64_t* bad_codes, int64_t* bad_labels) {
    if (!c) return fail(HG_ERR_ARG, "hg_set_database_f32: null context");
    if (N < 1 || !host_x || !host_labels) return fail(HG_ERR_ARG, "hg_set_database_f32: need N >= 1 and data");
    if (b < 1 || b > HG_MAX_BITS) return fail(HG_ERR_ARG, "hg_set_database_f32: b=%d outside 1..%d", b, HG_MAX_BITS);
    if (C < 1) return fail(HG_ERR_ARG, "hg_set_database_f32: C=%d", C);
    if (idx_base < 0 || n_total < N || idx_base + N > n_total || n_total >= 0xFFFFFFFFll)
        return fail(HG_ERR_ARG, "hg_set_database_f32: shard [%lld, %lld) does not fit a database of %lld rows (< 2^32 - 1)",
                    (long long)idx_base, (long long)(idx_base + N), (long long)n_total);
    HG_TRY(c->use());
    c->N = N; c->b = b; c->C = C; c->n_total = n_total;
    c->NW = (b + 31) / 32; c->NB = b + 1; c->LW = (C + 63) / 64;
    c->idx_base = (u32)idx_base;
    if (c->opt_host_pack) {
        HG_TRY(pack_on_host(c, host_x, host_labels, N, c->db, c->dblab, c->dbf, (int)c->opt_keep_floats, &c->dbf_resident,
                            bad_codes, bad_labels, c->census_db));
    } else {
        HG_TRY(pack_on_device(c, host_x, host_labels, N, c->db, c->dblab, c->dbf, bad_codes, bad_labels, c->census_db));
        c->dbf_resident = true;
    }
    if (!c->dbf_resident) c->bpad = 0;
    c->stage = ST_DB;
    c->dbx_valid = false;
    c->dbx2_valid = false;
    c->dbx3_valid = false;
    c->dbx8_valid = false;
    c->dbfx_valid = false;
    c->dbfb_valid = false;
    c->opt_consecutive_fail = c->shard_bet_fail = 0;
    c->cap_boost = c->real_cap_boost = 1;
    c->cfg_epoch++;
    return HG_OK;
}

int hg_set_queries_f32(hg_ctx* c, const float* host_x, const int64_t* host_labels, int64_t Q, int64_t* bad_codes,
                       int64_t* bad_labels) {
    HG_TRY(need(c, ST_DB, "hg_set_queries_f32", "hg_set_database"));
    if (Q < 1 || !host_x || !host_labels) return fail(HG_ERR_ARG, "hg_set_queries_f32: need Q >= 1 and data");
    if (Q > 0x7FFFFFC0ll) return fail(HG_ERR_ARG, "hg_set_queries_f32: Q too large");
    c->Q = Q;
    if (c->opt_host_pack) {
        // the query table is small: its floats follow whenever the database's are there (the inner-product ranking needs both)
        const int saved_bpad = c->bpad;
        HG_TRY(pack_on_host(c, host_x, host_labels, Q, c->qc, c->qlab, c->qf, c->dbf_resident ? 1 : 0, &c->qf_resident,
                            bad_codes, bad_labels, c->census_q));
        if (!c->qf_resident) c->bpad = saved_bpad;
    } else {
        HG_TRY(pack_on_device(c, host_x, host_labels, Q, c->qc, c->qlab, c->qf, bad_codes, bad_labels, c->census_q));
        c->qf_resident = true;
    }
    c->stage = ST_DB | ST_Q;
    c->qx_valid = false;
    c->qx2_valid = false;
    c->cfg_epoch++;
    return HG_OK;
}

// packed tables back to the host (tests; also lets a caller keep the packed form)
int hg_get_packed(hg_ctx* c, int which, uint32_t* host_codes, uint64_t* host_labels) {
    HG_TRY(need(c, which ? (ST_DB | ST_Q) : ST_DB, "hg_get_packed", "hg_set_database / hg_set_queries"));
    const i64 n = which ? c->Q : c->N;
    DevBuf& cd = which ? c->qc : c->db;
    DevBuf& lb = which ? c->qlab : c->dblab;
    if (host_codes) HG_HIP(hipMemcpyAsync(host_codes, cd.p, (size_t)n * c->NW * 4, hipMemcpyDeviceToHost, c->stream));
    if (host_labels) HG_HIP(hipMemcpyAsync(host_labels, lb.p, (size_t)n * c->LW * 8, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

int hg_set_queries(hg_ctx* c, const uint64_t* codes, const uint64_t* labels, int64_t Q) {
    HG_TRY(need(c, ST_DB, "hg_set_queries", "hg_set_database"));
    if (Q < 1 || !codes || !labels) return fail(HG_ERR_ARG, "hg_set_queries: need Q >= 1 and data");
    if (Q > 0x7FFFFFC0ll) return fail(HG_ERR_ARG, "hg_set_queries: Q too large");
    c->Q = Q;
    HG_TRY(upload_codes(c, c->qc, codes, Q, (c->b + 63) / 64, c->NW));
    HG_TRY(c->qlab.reserve((size_t)Q * c->LW * 8));
    HG_HIP(hipMemcpyAsync(c->qlab.p, labels, (size_t)Q * c->LW * 8, hipMemcpyHostToDevice, c->stream));
    HG_TRY(c->sync());
    c->stage = ST_DB | ST_Q;
    c->qx_valid = false;
    c->qx2_valid = false;
    c->qf_resident = false;                            // packed input: no float table
    c->cfg_epoch++;
    return HG_OK;
}

static int do_hist(hg_ctx* c, int stride, bool reduce = true, bool pairs_ok = false, int hcap = 0) {
    make_geometry(c);
    c->geo.hist_stride = stride;
    const bool mx = hist_mx_applies(c, stride, pairs_ok);
    c->geo.hcap = mx && !reduce && stride > 1 ? hcap : 0;
    c->hist_pairs = mx && stride == 1;                 // full pass per segment pair: the plan's per-segment steps follow suit
    const Geo& g = c->geo;
    const size_t plane = (size_t)g.NB * g.Qpad * 4;
    HG_TRY(c->hist.reserve(plane * g.S));
    HG_TRY(c->hown.reserve(plane + TAIL_WORDS * 4));
    if (reduce) {   // tail of the exported histogram: [0] overflow flag, [1] rows this pass visited
        const u32 visited = (u32)(stride == 1 ? g.N : sampled_rows(c, stride));
        // written by a kernel: ordered with the kernels that read it, no pageable staging memory to keep alive
        hipLaunchKernelGGL(k_set_tail, dim3(1), dim3(64), 0, c->stream, (u32*)(c->hown.as<char>() + plane), visited);
        HG_TRY(c->check_launch("k_set_tail"));
    }
    HG_TRY(mx ? launch_hist_mx(c) : launch_hist(c));
    if (!reduce) { c->stage = ST_DB | ST_Q; return HG_OK; }      // the caller reads the per-segment histograms itself
    const Geo gh = hist_geometry(c);
    c->t_begin(KI_HIST_REDUCE);
    hipLaunchKernelGGL(k_hist_reduce, dim3(grid_for((i64)g.NB * g.Qpad)), dim3(256), 0, c->stream,
                       c->hist.as<u32>(), c->hown.as<u32>(), gh);
    c->t_end();
    HG_TRY(c->check_launch("k_hist_reduce"));
    c->stage = ST_DB | ST_Q | (stride == 1 ? ST_HIST : 0);
    return HG_OK;
}

int hg_hist(hg_ctx* c) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_hist", "hg_set_database + hg_set_queries"));
    HG_TRY(do_hist(c, 1));
    return c->stage_end();
}

int hg_hist_buffer(hg_ctx* c, void** dev_ptr, int64_t* nbytes) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_hist_buffer", "hg_hist / hg_sample_hist / hg_select_candidates"));
    if (!c->hown.p) return fail(HG_ERR_STATE, "hg_hist_buffer: no histogram computed yet");
    if (dev_ptr) *dev_ptr = c->hown.p;
    if (nbytes) *nbytes = ((int64_t)c->geo.NB * c->geo.Qpad + TAIL_WORDS) * 4;
    return HG_OK;
}

static int set_R(hg_ctx* c, int64_t R, int G, int rank) {
    if (G < 1 || rank < 0 || rank >= G) return fail(HG_ERR_ARG, "rank %d of %d", rank, G);
    if (R < 1 || R > c->n_total)
        return fail(HG_ERR_ARG, "R=%lld outside 1..N (N=%lld rows in the database)", (long long)R, (long long)c->n_total);
    c->R = R; c->G = G; c->rank = rank;
    c->geo.R = R;
    c->RW = (R + 63) / 64;
    return HG_OK;
}

// k_plan on c->hown (full histogram, or the records' histogram in optimistic mode)
static int launch_plan(hg_ctx* c, const uint32_t* dev_hist_all) {
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->posbase.reserve((size_t)g.NB * qb));
    HG_TRY(c->t.reserve(qb)); HG_TRY(c->cnt_lt.reserve(qb)); HG_TRY(c->quota.reserve(qb));
    HG_TRY(c->tie_before.reserve(qb)); HG_TRY(c->n_lt.reserve(qb)); HG_TRY(c->err.reserve(4));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    Plan pl{c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->n_lt.as<u32>(),
            c->posbase.as<u32>(), c->err.as<int>()};
    c->t_begin(KI_PLAN);
    hipLaunchKernelGGL(k_plan, dim3(grid_for(g.Q)), dim3(256), 0, c->stream, c->hown.as<u32>(),
                       (const u32*)dev_hist_all, c->G, c->rank, pl, g);
    c->t_end();
    return c->check_launch("k_plan");
}

// exact plan: threshold from the full histogram, then the exact record-row layout
static int do_plan(hg_ctx* c, int64_t R, const uint32_t* dev_hist_all, int G, int rank) {
    if (G > 1 && !dev_hist_all) return fail(HG_ERR_ARG, "hg_plan: G > 1 needs the gathered histograms");
    HG_TRY(set_R(c, R, G, rank));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(launch_plan(c, dev_hist_all));
    HG_TRY(c->seglt.reserve((size_t)g.S * qb)); HG_TRY(c->segtie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_start.reserve((size_t)g.S * qb)); HG_TRY(c->sl_tie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->failq.reserve(qb));
    HG_TRY(c->sstar.reserve(qb));
    const Geo gp = hist_geometry(c);                   // per segment -- or per segment pair after k_hist_mx (then only sstar is used)
    const int ratio = (int)(gp.L / g.L);
    c->t_begin(KI_SEG_COUNTS);
    hipLaunchKernelGGL(k_seg_counts, dim3(grid_for((i64)gp.S * g.Qpad)), dim3(256), 0, c->stream, c->hist.as<u32>(),
                       c->t.as<int>(), c->seglt.as<u32>(), c->segtie.as<u32>(), gp);
    c->t_end();
    HG_TRY(c->check_launch("k_seg_counts"));
    c->t_begin(KI_SEG_LAYOUT);
    hipLaunchKernelGGL(k_seg_layout, dim3(grid_for(g.Qpad)), dim3(256), 0, c->stream, c->seglt.as<u32>(),
                       c->segtie.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->sl_start.as<u32>(),
                       c->sl_tie.as<u32>(), c->tot.as<u32>(), c->sstar.as<int>(), ratio, gp);
    c->t_end();
    HG_TRY(c->check_launch("k_seg_layout"));
    c->optimistic = false;
    c->crow = R;
    c->cap = 0;
    c->stage = ST_DB | ST_Q | ST_HIST | ST_PLAN;
    return HG_OK;
}

// The plan kernel flags R > (rows in the histograms it saw); read it back with the results.
static int read_plan_flag(hg_ctx* c, int* flag) {
    HG_HIP(hipMemcpyAsync(flag, c->err.p, 4, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

static int check_plan_flag(hg_ctx* c) {
    int err = 0;
    HG_TRY(read_plan_flag(c, &err));
    if (err) {
        c->stage = ST_DB | ST_Q | ST_HIST;
        return fail(HG_ERR_ARG, "R=%lld exceeds the rows present in the gathered histograms", (long long)c->R);
    }
    return HG_OK;
}

int hg_plan(hg_ctx* c, int64_t R, const uint32_t* dev_hist_all, int G, int rank) {
    HG_TRY(need(c, ST_HIST, "hg_plan", "hg_hist"));
    // a one-shot exact call may have left histograms per segment PAIR (k_hist_mx); the staged select lays its slices out
    // per segment
    if (c->hist_pairs) return fail(HG_ERR_STATE, "hg_plan called before hg_hist (the last histogram pass belonged to a one-shot call)");
    HG_TRY(do_plan(c, R, dev_hist_all, G, rank));
    // the device flag says "R exceeds the rows in the gathered histograms"; R <= n_total was checked on
    // the host already, so an unsynchronised caller loses nothing by skipping the read-back
    return c->stage_sync ? check_plan_flag(c) : HG_OK;
}

// k_rank_fused in one of its modes: 0 = histogram + plan + placement in one launch (single shard),
// 1 = histogram phase (several shards, before the exchange), 2 = placement phase (after k_plan).
// the dense regime in one kernel (k_rank_direct): fits when a block's LDS holds the counters, the R-bit bitmap and a tile of rows
static i64 rank_direct_tile(const hg_ctx* c, int64_t R) {
    if (!c->opt_rank_direct || c->LW > 2 || c->NW > 8 || c->b > 127) return 0;
    const i64 RW = (R + 63) / 64;
    const i64 fixed = rank_direct_layout(c->b + 1, RW, 0).total;
    i64 tile = (c->opt_rank_direct_lds * 1024 - fixed) & ~(i64)15;
    if (tile > 252 * 256) tile = 252 * 256;             // a thread's chunk must fit its byte counters
    const i64 n8 = (c->N + 7) / 8 * 8;
    if (tile > n8) tile = n8;
    return tile >= 8192 || tile >= n8 ? tile : 0;
}

static int launch_rank(hg_ctx* c, int mode, int nbits) {
    const Geo& g = c->geo;
    if (c->direct_rank && mode == 0) {
        const i64 tile = rank_direct_tile(c, g.R);
        if (tile > 0) {
            HG_TRY(c->err.reserve(4));
            HG_TRY(c->qbad.reserve((size_t)g.Qpad * 4));
            const RankDirectLds L = rank_direct_layout(g.NB, c->RW, (int)tile);
            if (L.total > 64 * 1024)
                HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rank_direct), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
            RankDirectArgs da{c->qc.as<u32>(), c->qlab.as<u64>(), c->db.as<u32>(), c->dblab.as<u64>(), c->err.as<int>(), c->qbad.as<u32>(),
                              c->RW, (int)tile, c->want_lists ? 1 : 0};
            c->t_begin(KI_RANK_FUSED);
            hipLaunchKernelGGL(k_rank_direct, dim3(g.Q), dim3(256), (size_t)L.total, c->stream, da, c->out_idx.as<u32>(), c->out_dist.as<u8>(),
                               c->mbits.as<u32>(), g);
            c->t_end();
            return c->check_launch("k_rank_direct");
        }
    }
    int nwav = c->opt_rank_waves ? (int)c->opt_rank_waves
                                 : ((c->optimistic ? 3 * c->R : c->R) >= 16384 ? 16 : 4);   // records per query ~ 3R / R
    // k_rank_lds: the query's records resident in LDS -- room for ~2.5 R per query (the bet keeps 1.3-2 R),
    // at most 64 KiB per block; queries with more are left to k_rank_fused (flagged in bigq)
    bool use_lds = false;
    i64 recs = 0;
    const size_t fixed = ((size_t)5 * g.NB + 8 + 4 + 8 + 2 * (size_t)c->RW + (size_t)g.S + 2) * 4;
    const size_t per_rec = c->want_lists ? 6 : 2;
    if (c->optimistic && c->opt_rank_lds) {
        const double share = (double)c->N / (double)(c->n_total > 0 ? c->n_total : 1);    // this shard's part of the list
        recs = (i64)(2.5 * (double)c->R * share) + 2048;   // small R: the guess's safety margin is relatively larger (R = 100 keeps ~4 R)
        const i64 fit = fixed < 64 * 1024 ? (i64)((64 * 1024 - fixed) / per_rec) : 0;
        if (recs > fit) recs = fit;
        recs = recs / 64 * 64;
        use_lds = (double)recs >= 2.0 * (double)c->R * share && recs >= 64;
        if (use_lds) nwav = 4;                        // the two kernels share hwq's [Q][4][NB] layout
    }
    const size_t fixed_words = (size_t)(nwav + 1) * g.NB + 8;
    const int bits_lds = (fixed_words + 2 * (size_t)c->RW) * 4 <= 64 * 1024;
    if (mode != 1 && !bits_lds) HG_HIP(hipMemsetAsync(c->mbits.p, 0, (size_t)g.Q * c->RW * 8, c->stream));
    HG_TRY(c->err.reserve(4));
    HG_TRY(c->qbad.reserve((size_t)g.Qpad * 4));
    if (mode != 0) HG_TRY(c->hwq.reserve((size_t)g.Q * nwav * g.NB * 4));
#ifdef HG_RANK_PROFILE
    HG_TRY(c->hwq.reserve((size_t)4096 * 16 * 4 + (size_t)g.Q * nwav * g.NB * 4));
#endif
    if (mode == 0) {
        if (c->optimistic) { if (!c->err_zeroed) HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream)); }
        else HG_HIP(hipMemsetAsync(c->failq.p, 0, (size_t)g.Qpad * 4, c->stream));
        c->err_zeroed = false;
    }
    const u32* only = nullptr;
    bool counted = false;
    if (c->optimistic && c->opt_rank_lds && c->opt_rank_cnt && c->opt_rank_wave > 0 && (mode == 0 || mode == 3) && c->rec8 && !c->want_lists &&
        g.S <= RW_SMAX) {
        // one wavefront per query (k_rank_wave): no block barriers, 5 KB + the records of LDS per query in flight
        // ... which pays for SHORT lists only (a sharded rank's share of R, a small R): a wavefront walks its query's records
        // with 64 lanes where k_rank_cnt has 256, and at C2's 6500 records (16 KB of LDS per query, 10 in flight per CU) it
        // is slower, 0.23 vs 0.19 ms; at 800 records (7 KB, 22 in flight) it wins, 0.105 vs 0.134 ms.
        const double share = (double)c->N / (double)(c->n_total > 0 ? c->n_total : 1);
        i64 r2 = (i64)(0.1 * (double)c->opt_rank_wave * (double)c->R * share) + 256;
        if (r2 < 1024) r2 = 1024;
        r2 = (r2 + 63) / 64 * 64;
        if (r2 > (i64)c->opt_rank_wave_max) r2 = 0;                 // long lists: k_rank_cnt below
        const int nbc = (mode == 0 && !c->exact_mx && c->G == 1) ? g.NB / 2 + 2 : 0;
        const RankWaveLds L = rank_wave_layout(g.NB, c->RW, g.S, (int)r2, nbc);
        // wavefronts per block: the split that wastes the least of a CU's 160 KB
        const int fit1 = (int)(160 * 1024 / L.per_wave), fit2 = 2 * (int)(160 * 1024 / (2 * L.per_wave));
        const int wpb = fit2 >= fit1 ? 2 : 1;
        if (r2 > 0 && fit1 >= 4) {
            HG_TRY(c->bigq.reserve((size_t)g.Qpad * 4));
            RankLdsArgs la{c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->err.as<int>(), c->qbad.as<u32>(), c->bigq.as<u32>(),
                           c->cap, c->crow, 0, 1, c->RW, (int)r2, mode, c->hwq.as<u32>(), c->hown.as<u32>(),
                           c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->posbase.as<u32>(), nbc};
            const size_t lds = (size_t)wpb * L.per_wave;
            if (lds > 64 * 1024)
                HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rank_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            c->t_begin(KI_RANK_LDS);
            hipLaunchKernelGGL(k_rank_wave, dim3(grid_for(g.Q, wpb)), dim3(64 * wpb), lds, c->stream, c->cand.as<u8>(), la, c->mbits.as<u32>(), g);
            c->t_end();
            HG_TRY(c->check_launch("k_rank_wave"));
            only = c->bigq.as<u32>();                // k_rank_fused below ranks what this path declined
            counted = true;
        }
    }
    if (!counted && c->optimistic && c->opt_rank_lds && c->opt_rank_cnt && (mode == 0 || mode == 3)) {
        // per-thread counting sort (k_rank_cnt): byte counters for every distance + a tile of the records, <= 64 KiB per
        // block; lists longer than a tile are ranked tile by tile
        const double share = (double)c->N / (double)(c->n_total > 0 ? c->n_total : 1);
        i64 r2 = (i64)(2.5 * (double)c->R * share) + 256;         // a tile of the records: the usual list (1.3 - 2 R) in one
        if (r2 < 4096) r2 = 4096;                                  // (small R: the guess's margin is relatively larger)
        r2 = r2 / 64 * 64;
        // a one-shot bet's guess stops below b/2 + 2 (enqueue_optimistic), so its records need no counters beyond: the
        // block fits 32 KB and five of them a CU; a query with farther records (thin sample: everything taken) goes to k_rank_fused
        const int nbc = (mode == 0 && !c->exact_mx && c->G == 1) ? g.NB / 2 + 2 : 0;
        RankCntLds L = rank_cnt_layout(g.NB, c->RW, g.S, (int)r2, c->want_lists ? 1 : 0, nbc);
        while (L.total > 64 * 1024 && r2 > 64) { r2 -= 64; L = rank_cnt_layout(g.NB, c->RW, g.S, (int)r2, c->want_lists ? 1 : 0, nbc); }
        if (L.total <= 64 * 1024 && r2 >= 4096) {
            HG_TRY(c->bigq.reserve((size_t)g.Qpad * 4));
            RankLdsArgs la{c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->err.as<int>(), c->qbad.as<u32>(), c->bigq.as<u32>(),
                           c->cap, c->crow, c->want_lists ? 1 : 0, c->rec8 ? 1 : 0, c->RW, (int)r2, mode, c->hwq.as<u32>(), c->hown.as<u32>(),
                           c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->posbase.as<u32>(), nbc};
            c->t_begin(KI_RANK_LDS);
            hipLaunchKernelGGL(k_rank_cnt, dim3(g.Q), dim3(256), (size_t)L.total, c->stream, c->cand.as<u64>(), la, c->out_idx.as<u32>(),
                               c->out_dist.as<u8>(), c->mbits.as<u32>(), g);
            c->t_end();
            HG_TRY(c->check_launch("k_rank_cnt"));
            only = c->bigq.as<u32>();                // k_rank_fused below ranks what this path declined
            counted = true;
        }
    }
    if (use_lds && !counted) {
        {
            HG_TRY(c->bigq.reserve((size_t)g.Qpad * 4));
            if (mode != 0) HG_TRY(c->hwq.reserve((size_t)g.Q * nwav * g.NB * 4));
            RankLdsArgs la{c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->err.as<int>(), c->qbad.as<u32>(), c->bigq.as<u32>(),
                           c->cap, c->crow, c->want_lists ? 1 : 0, c->rec8 ? 1 : 0, c->RW, (int)recs, mode, c->hwq.as<u32>(), c->hown.as<u32>(),
                           c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->posbase.as<u32>(), 0};
            const size_t lb = fixed + (size_t)recs * per_rec;
            c->t_begin(KI_RANK_LDS);
            hipLaunchKernelGGL(k_rank_lds<4>, dim3(g.Q), dim3(256), lb, c->stream, c->cand.as<u64>(), la, c->out_idx.as<u32>(),
                               c->out_dist.as<u8>(), c->mbits.as<u32>(), nbits, g);
            c->t_end();
            HG_TRY(c->check_launch("k_rank_lds"));
            only = c->bigq.as<u32>();                // k_rank_fused below only ranks what did not fit
        }
    }
    RankArgs ra{c->sl_cnt.as<u32>(), c->tot.as<u32>(), c->failq.as<u32>(), c->err.as<int>(), c->qbad.as<u32>(),
                mode, c->hwq.as<u32>(), c->hown.as<u32>(), c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(),
                c->tie_before.as<u32>(), c->posbase.as<u32>(),
                c->optimistic ? c->cap : 256u, c->crow, c->optimistic ? 0 : 1, c->want_lists ? 1 : 0, bits_lds, c->RW, only,
                c->direct_rank ? 1 : 0, c->rec8 ? 1 : 0, c->db.as<u32>(), c->dblab.as<u64>(), c->qc.as<u32>(), c->qlab.as<u64>()};
    const size_t lds_bytes = (fixed_words + (bits_lds ? 2 * (size_t)c->RW : 0)) * 4;
    c->t_begin(mode == 1 ? KI_CAND_HIST : KI_RANK_FUSED);
    if (nwav == 16)
        hipLaunchKernelGGL(k_rank_fused<16>, dim3(g.Q), dim3(1024), lds_bytes, c->stream, c->cand.as<u64>(), ra,
                           c->out_idx.as<u32>(), c->out_dist.as<u8>(), c->mbits.as<u32>(), nbits, g);
    else
        hipLaunchKernelGGL(k_rank_fused<4>, dim3(g.Q), dim3(256), lds_bytes, c->stream, c->cand.as<u64>(), ra,
                           c->out_idx.as<u32>(), c->out_dist.as<u8>(), c->mbits.as<u32>(), nbits, g);
    c->t_end();
    return c->check_launch("k_rank_fused");
}

// record pass + ordering (+ gather-based label match when labels are too wide for the record pass)
static int do_match(hg_ctx* c);
static int do_select(hg_ctx* c) {
    const Geo& g = c->geo;
    const size_t slots = (size_t)g.Q * g.R;
    if (c->LW > 2) c->want_lists = true;                  // k_match gathers through the idx list
    HG_TRY(c->cand.reserve((size_t)g.Q * c->crow * 8));
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(c->want_lists ? slots * 4 : 16));
    HG_TRY(c->out_dist.reserve(c->want_lists ? slots : 16));
    if (c->want_lists && c->G > 1) {  // slots of other shards stay IDX_NONE / 0xFF
        HG_HIP(hipMemsetAsync(c->out_idx.p, 0xFF, slots * 4, c->stream));
        HG_HIP(hipMemsetAsync(c->out_dist.p, 0xFF, slots, c->stream));
    }
    HG_TRY(launch_select(c));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    if (c->optimistic || c->G == 1) {
        // one block per query: verify (optimistic) + plan + order.  Exact single-shard rows hold
        // precisely the top R, so the same counting plan reproduces t and the bucket starts.
        HG_TRY(launch_rank(c, 0, nbits));
    } else {
        const size_t lds_words = (size_t)g.NB + 2 * (size_t)c->RW;
        const int bits_lds = WPB * lds_words * 4 <= 64 * 1024;
        if (!bits_lds) HG_HIP(hipMemsetAsync(c->mbits.p, 0, (size_t)g.Q * c->RW * 8, c->stream));
        OrdArgs oa{c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->posbase.as<u32>(),
                   c->tot.as<u32>(), c->crow, c->want_lists ? 1 : 0, bits_lds, c->RW};
        c->t_begin(KI_ORDER);
        hipLaunchKernelGGL(k_order, dim3(grid_for(g.Q, WPB)), dim3(256),
                           (size_t)WPB * (g.NB + (bits_lds ? 2 * (size_t)c->RW : 0)) * 4, c->stream, c->cand.as<u64>(), oa,
                           c->out_idx.as<u32>(), c->out_dist.as<u8>(), c->mbits.as<u32>(), nbits, g);
        c->t_end();
        HG_TRY(c->check_launch("k_order"));
    }
    c->lists_valid = c->want_lists;
    c->stage = (c->stage & (ST_DB | ST_Q | ST_HIST | ST_PLAN)) | ST_PLAN | ST_SELECT;
    if (c->LW <= 2) c->stage |= ST_MATCH;                 // match bits came with the records
    else HG_TRY(do_match(c));
    return HG_OK;
}

int hg_select(hg_ctx* c) {
    HG_TRY(need(c, ST_PLAN, "hg_select", "hg_plan"));
    c->want_lists = c->staged_lists != 0;
    HG_TRY(do_select(c));
    return c->stage_end();
}

static int do_match(hg_ctx* c) {
    const Geo& g = c->geo;
    const i64 nKB = (g.R + 255) / 256;
    const i64 blocks = nKB * g.Q;
    if (blocks > 0x7FFFFFFFll) return fail(HG_ERR_ARG, "hg_match: Q*R too large for one launch");
    c->t_begin(KI_MATCH);
    hipLaunchKernelGGL(k_match, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->out_idx.as<u32>(),
                       c->dblab.as<u64>(), c->qlab.as<u64>(), c->mbits.as<u64>(), c->RW, (int)nKB, g);
    c->t_end();
    HG_TRY(c->check_launch("k_match"));
    c->stage |= ST_MATCH;
    c->stage &= ~(unsigned)ST_AP;
    return HG_OK;
}

// Label-match bits are produced together with the ranking (k_select/k_order) for
// up to 128 classes; this stage exists for wider label sets and for API symmetry.
int hg_match(hg_ctx* c) {
    HG_TRY(need(c, ST_SELECT, "hg_match", "hg_select"));
    if (c->stage & ST_MATCH) return HG_OK;
    HG_TRY(do_match(c));
    return c->stage_end();
}

int hg_match_buffer(hg_ctx* c, void** dev_ptr, int64_t* nbytes) {
    HG_TRY(need(c, ST_MATCH, "hg_match_buffer", "hg_match"));
    if (dev_ptr) *dev_ptr = c->mbits.p;
    if (nbytes) *nbytes = (int64_t)c->geo.Q * c->RW * 8;
    return HG_OK;
}

int hg_merge_match(hg_ctx* c, const uint64_t* dev_bits_all, int G) {
    HG_TRY(need(c, ST_MATCH, "hg_merge_match", "hg_match"));
    if (!dev_bits_all || G < 1) return fail(HG_ERR_ARG, "hg_merge_match: bad argument");
    const i64 n = (i64)c->geo.Q * c->RW;
    c->t_begin(KI_MERGE);
    hipLaunchKernelGGL(k_or_bits, dim3(grid_for(n)), dim3(256), 0, c->stream, (const u64*)dev_bits_all, c->mbits.as<u64>(), n, G);
    c->t_end();
    HG_TRY(c->check_launch("k_or_bits"));
    return c->stage_end();
}

static int do_ap_range(hg_ctx* c, i64 q0, i64 nq) {      // k_ap's block index is the query: a range is a pointer offset
    c->ap_staged = false;
    const Geo& g = c->geo;
    if (c->shapes_for_R != g.R) {
        std::vector<ApShape> sh(2);
        build_shape(g.R >= AP_CHUNK ? AP_CHUNK : (int)g.R, sh[0]);
        build_shape((int)(g.R % AP_CHUNK), sh[1]);
        HG_TRY(c->shapes.reserve(sizeof(ApShape) * 2));
        HG_HIP(hipMemcpyAsync(c->shapes.p, sh.data(), sizeof(ApShape) * 2, hipMemcpyHostToDevice, c->stream));
        HG_HIP(hipStreamSynchronize(c->stream));   // sh goes out of scope
        c->shapes_for_R = g.R;
    }
    // reciprocals of the ranks 1 .. R (k_ap's division in three multiply-adds); lists beyond 2^20 divide
    const bool use_recip = c->opt_ap_recip && g.R <= (1ll << 20);
    if (use_recip && c->recip_for_R != g.R) {
        HG_TRY(c->ap_recip.reserve((size_t)(g.R + 1) * 8));
        hipLaunchKernelGGL(k_recip_table, dim3(grid_for(g.R + 1)), dim3(256), 0, c->stream, c->ap_recip.as<double>(), (i64)g.R);
        HG_TRY(c->check_launch("k_recip_table"));
        c->recip_for_R = g.R;
    }
    HG_TRY(c->ap.reserve((size_t)g.Q * 8));
    HG_TRY(c->rel.reserve((size_t)g.Q * 4));
    c->t_begin(KI_AP);
    if (nq > 0)
        hipLaunchKernelGGL(k_ap, dim3((unsigned)nq), dim3(AP_THREADS), 0, c->stream, c->mbits.as<u64>() + (size_t)q0 * c->RW, c->RW, g.R,
                           c->shapes.as<ApShape>(), use_recip ? c->ap_recip.as<double>() : (const double*)nullptr,
                           c->ap.as<double>() + q0, c->rel.as<u32>() + q0);
    c->t_end();
    HG_TRY(c->check_launch("k_ap"));
    c->stage |= ST_AP;
    return HG_OK;
}
static int do_ap(hg_ctx* c) { return do_ap_range(c, 0, c->geo.Q); }

int hg_ap(hg_ctx* c) {
    HG_TRY(need(c, ST_MATCH, "hg_ap", "hg_match"));
    HG_TRY(do_ap(c));
    return c->stage_end();
}

int hg_topr_buffers(hg_ctx* c, void** dev_idx, void** dev_dist, int64_t* n_slots) {
    HG_TRY(need(c, ST_SELECT, "hg_topr_buffers", "hg_select"));
    if (!c->lists_valid) return fail(HG_ERR_STATE, "ranked lists were not materialised by the last call (use hg_topr / staged_lists)");
    if (dev_idx) *dev_idx = c->out_idx.p;
    if (dev_dist) *dev_dist = c->out_dist.p;
    if (n_slots) *n_slots = (int64_t)c->geo.Q * c->geo.R;
    return HG_OK;
}

int hg_merge_topr(hg_ctx* c, const uint32_t* dev_idx_all, const uint8_t* dev_dist_all, int G) {
    HG_TRY(need(c, ST_SELECT, "hg_merge_topr", "hg_select"));
    if (!c->lists_valid) return fail(HG_ERR_STATE, "ranked lists were not materialised by the last call");
    if (!dev_idx_all || !dev_dist_all || G < 1) return fail(HG_ERR_ARG, "hg_merge_topr: bad argument");
    const i64 n = (i64)c->geo.Q * c->geo.R;
    c->t_begin(KI_MERGE);
    hipLaunchKernelGGL(k_min_topr, dim3(grid_for(n)), dim3(256), 0, c->stream, (const u32*)dev_idx_all,
                       (const u8*)dev_dist_all, c->out_idx.as<u32>(), c->out_dist.as<u8>(), n, G);
    c->t_end();
    HG_TRY(c->check_launch("k_min_topr"));
    return c->stage_end();
}

// ---- staged optimistic sequence (multi-shard): sample -> [gather] -> guess -> candidates ->
// [gather] -> rank.  Mirrors the one-shot bet, with the two histogram exchanges made explicit.
// Sampling stride of the bet, in row batches.  One row batch in 24: a fixed 4 % of a pass (with the
// matrix-core select the sampling pass is a visible share of the step; 16 -> 24 trades 0.05 ms of it for
// ~3 % more surplus records).  The
// guess's safety margin is relative to sqrt(sampled hits), so a small R only means relatively more
// surplus records (R = 100: ~3.5 R of them) -- still far cheaper than a full histogram pass.
// Capacity of a (segment, query) slice of the bet: the budgeted mean + 6 sigma, never more than the segment's rows, and
// the record rows of all queries together stay below 64 GB ("cap_boost" may ask for more than is sensible).
static u32 slice_capacity(const hg_ctx* c, double mean) {
    const Geo& g = c->geo;
    u32 cap = (u32)std::ceil(mean + 6.0 * std::sqrt(mean) + 16.0);
    cap = (cap + 15u) & ~15u;                      // a multiple of the compact records' ring (16) and flush piece (8)
    const u32 whole = (u32)((g.L + 15) & ~15ll);
    if (cap > whole) cap = whole;
    while (cap > 64u && (double)g.Q * (double)g.S * (double)cap * 8.0 > 64e9) cap = (cap / 2u + 15u) & ~15u;
    return cap;
}

static int auto_stride(hg_ctx* c, int64_t R) {
    (void)R;
    return c->opt_stride > 0 ? (int)c->opt_stride : 24;
}

int hg_bet_eligible(hg_ctx* c, int64_t R, int world, int* eligible) {
    if (!c || !eligible || world < 1) return fail(HG_ERR_ARG, "hg_bet_eligible: bad argument");
    // only quantities every rank shares: options, R, the size of the whole database, the world size -- and the count of
    // consecutive SHARDED bets lost, which only hg_merge_ranked / hg_rank / hg_bet_verdict touch, with a verdict that is
    // computed from gathered data and therefore the same on every rank (one-shot calls keep their own counter)
    const int stride = auto_stride(c, R);
    const i64 per_shard = c->n_total / world;
    *eligible = c->opt_enable && c->shard_bet_fail < 2 && stride >= 2 && R * 8 <= c->n_total && per_shard >= 65536;
    return HG_OK;
}

int hg_sample_hist(hg_ctx* c, int64_t R) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_sample_hist", "hg_set_database + hg_set_queries"));
    const int stride = auto_stride(c, R);
    if (stride < 2) return fail(HG_ERR_ARG, "hg_sample_hist: R=%lld is too small to sample for", (long long)R);
    HG_TRY(do_hist(c, stride));
    return c->stage_end();
}

int hg_guess(hg_ctx* c, int64_t R, const uint32_t* dev_hist_all, int G, int rank) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_guess", "hg_sample_hist"));
    if (G > 1 && !dev_hist_all) return fail(HG_ERR_ARG, "hg_guess: G > 1 needs the gathered sample histograms");
    HG_TRY(set_R(c, R, G, rank));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->tguess.reserve(qb));
    HG_TRY(c->sl_start.reserve((size_t)g.S * qb)); HG_TRY(c->sl_tie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->failq.reserve(qb));
    HG_HIP(hipMemsetAsync(c->failq.p, 0, qb, c->stream));
    c->t_begin(KI_GUESS);
    const Geo gh = hist_geometry(c);                   // the sampled pass ran on coarser segments
    HG_TRY(c->sstar.reserve(qb));
    hipLaunchKernelGGL(k_guess, dim3(grid_for(g.Q)), dim3(256), 0, c->stream, c->hown.as<u32>(), (const u32*)dev_hist_all, G,
                       rank, c->hist.as<u32>(), gh.S, (int)(gh.L / g.L), (double)c->opt_sigma, (i64)c->n_total,
                       c->tguess.as<int>(), c->sstar.as<int>(), g);
    c->t_end();
    HG_TRY(c->check_launch("k_guess"));
    // a guessed cut keeps at most ~2.6 R rows over ALL shards; a shard's share is proportional to its size,
    // with the same 6-sigma headroom per slice as the one-shot bet
    const double share = (double)c->N / (double)c->n_total;
    const double mean = 0.1 * (double)c->cand_budget_x10 * (double)c->cap_boost * (double)R * share / (double)g.S;
    u32 cap = slice_capacity(c, mean);
    c->optimistic = true;
    c->cap = cap;
    c->crow = (i64)g.S * cap;
    c->stage = ST_DB | ST_Q | ST_PLAN;
    return c->stage_end();
}

int hg_select_candidates(hg_ctx* c) {
    HG_TRY(need(c, ST_PLAN, "hg_select_candidates", "hg_guess"));
    if (!c->optimistic) return fail(HG_ERR_STATE, "hg_select_candidates: no guess in force (use hg_select after hg_plan)");
    const Geo& g = c->geo;
    c->want_lists = c->staged_lists != 0 || c->LW > 2;  // decides the record format (hg_rank places them)
    HG_TRY(c->cand.reserve((size_t)g.Q * c->crow * 8));
    HG_TRY(launch_select(c));
    const size_t plane = (size_t)g.NB * g.Qpad * 4;
    HG_HIP(hipMemsetAsync(c->hown.as<char>() + plane, 0, TAIL_WORDS * 4, c->stream));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    HG_TRY(launch_rank(c, 1, nbits));                // per-wave and shard histograms of the records
    return c->stage_end();
}

// Sharded bet, AP only: select, then rank THIS shard's records locally (rank kernels' mode 3).  What leaves the
// shard is its per-distance record counts (hg_hist_buffer) and its match bitmap in local rank order
// (hg_match_buffer); hg_merge_ranked stitches the global bitmap from the gathered pairs.  One exchange and one
// pass over the records fewer than hg_select_candidates + hg_rank.
int hg_select_ranked(hg_ctx* c) {
    HG_TRY(need(c, ST_PLAN, "hg_select_ranked", "hg_guess"));
    if (!c->optimistic) return fail(HG_ERR_STATE, "hg_select_ranked: no guess in force");
    const Geo& g = c->geo;
    // > 128 classes: the record pass leaves the match bit 0 (launch_select_nw instantiates LW = 0); the local bitmap
    // then comes from k_match gathering the labels through the LOCAL ranked index list, so that list is kept
    const bool wide = c->LW > 2;
    c->want_lists = wide;
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(c->cand.reserve((size_t)g.Q * c->crow * 8));
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(wide ? slots * 4 : 16)); HG_TRY(c->out_dist.reserve(wide ? slots : 16));
    if (wide) HG_HIP(hipMemsetAsync(c->out_idx.p, 0xFF, slots * 4, c->stream));    // slots past the shard's own records: IDX_NONE
    HG_TRY(c->err.reserve(4));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    HG_TRY(launch_select(c));
    const size_t plane = (size_t)g.NB * g.Qpad * 4;
    HG_HIP(hipMemsetAsync(c->hown.as<char>() + plane, 0, TAIL_WORDS * 4, c->stream));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    HG_TRY(launch_rank(c, 3, nbits));
    c->lists_valid = false;
    c->ranked_local = true;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT;
    if (wide) HG_TRY(do_match(c));                    // metric.py:17-19 through the local list
    c->want_lists = false;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_MATCH;     // hg_match_buffer hands out the LOCAL bitmap until the merge
    return c->stage_end();
}

static int merge_ranked_range(hg_ctx* c, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, i64 q0, i64 nq, const char* who) {
    HG_TRY(need(c, ST_MATCH, who, "hg_select_ranked"));
    if (!c->ranked_local) return fail(HG_ERR_STATE, "%s: hg_select_ranked has not run", who);
    if (G < 1 || G > 64 || (G > 1 && (!dev_hist_all || !dev_bits_all)))
        return fail(HG_ERR_ARG, "%s: bad argument (1 <= G <= 64, gathered buffers for G > 1)", who);
    const Geo& g = c->geo;
    if (q0 < 0 || nq < 0 || q0 + nq > g.Q) return fail(HG_ERR_ARG, "%s: queries [%lld, %lld) of %d", who, (long long)q0, (long long)(q0 + nq), g.Q);
    HG_TRY(c->mbits2.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->qbad.reserve((size_t)g.Qpad * 4));
    const u32* hall = G > 1 ? (const u32*)dev_hist_all : c->hown.as<u32>();
    const u64* ball = G > 1 ? (const u64*)dev_bits_all : c->mbits.as<u64>();
    const size_t rows_lds = (size_t)WPB * G * c->RW * 8;       // the G local bitmap rows of a block's four queries
    const size_t cnt_lds = (size_t)WPB * G * g.NB * 4;         // their per-distance counts: at most 4 * 64 * 256 * 4 = 256 KiB ...
    const int use_lds = rows_lds + cnt_lds <= 64 * 1024;
    const size_t merge_lds = (use_lds ? rows_lds : 0) + cnt_lds;
    if (merge_lds > 160 * 1024)                                // ... which only many shards of long codes reach
        return fail(HG_ERR_ARG, "hg_merge_ranked: G=%d shards of %d-bit codes need %zu bytes of LDS per block (160 KiB available): "
                                "use the staged sequence (hg_select_candidates / hg_rank)", G, c->b, merge_lds);
    if (merge_lds > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_ranked), hipFuncAttributeMaxDynamicSharedMemorySize, (int)merge_lds));
    c->t_begin(KI_MERGE);
    if (nq > 0)
        hipLaunchKernelGGL(k_merge_ranked, dim3(grid_for(nq, WPB)), dim3(256), merge_lds, c->stream, hall, ball, G,
                           c->RW, c->mbits2.as<u64>(), c->err.as<int>(), c->qbad.as<u32>(), use_lds, g, (int)q0, (int)(q0 + nq));
    c->t_end();
    HG_TRY(c->check_launch("k_merge_ranked"));
    std::swap(c->mbits, c->mbits2);                    // the global bitmap is what hg_ap and hg_get_match see
    c->ranked_local = false;
    c->G = G;
    return HG_OK;
}

int hg_merge_ranked(hg_ctx* c, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, int* bet_lost) {
    if (!bet_lost) return fail(HG_ERR_ARG, "hg_merge_ranked: null argument");
    HG_TRY(merge_ranked_range(c, dev_hist_all, dev_bits_all, G, 0, c ? c->geo.Q : 0, "hg_merge_ranked"));
    if (c->defer_verdict) {
        *bet_lost = -1;
        c->verdict_pending = true;
        c->verdict_known = false;
        c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
        return c->stage_end();
    }
    int flag = 0;
    HG_TRY(read_plan_flag(c, &flag));
    *bet_lost = flag;
    c->opt_runs++;
    if (flag) {
        c->opt_fallbacks++;
        c->shard_bet_fail++;
        c->stage = ST_DB | ST_Q;
        return HG_OK;
    }
    c->shard_bet_fail = 0;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
    return HG_OK;
}

static int ensure_pin(hg_ctx* c, size_t need_b) {
    if (c->pin_cap >= need_b) return HG_OK;
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr; c->pin_cap = 0;
    HG_HIP(hipHostMalloc(&c->pin, need_b, hipHostMallocDefault));
    c->pin_cap = need_b;
    ++g_alloc_epoch;                                   // captured downloads point into the old block
    return HG_OK;
}

// bookkeeping after the verdict of a staged bet is known
static int settle_bet(hg_ctx* c, int flag) {
    c->opt_runs++;
    if (flag) {
        c->opt_fallbacks++;
        c->shard_bet_fail++;
        c->stage = ST_DB | ST_Q;
        return HG_OK;
    }
    c->shard_bet_fail = 0;
    c->lists_valid = c->want_lists;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT;
    if (c->LW <= 2) c->stage |= ST_MATCH;
    else HG_TRY(do_match(c));
    return c->sync();
}

int hg_rank(hg_ctx* c, const uint32_t* dev_hist_all, int G, int rank, int* bet_lost) {
    HG_TRY(need(c, ST_PLAN, "hg_rank", "hg_select_candidates"));
    if (!c->optimistic) return fail(HG_ERR_STATE, "hg_rank: no guess in force");
    if (!bet_lost || G < 1 || (G > 1 && !dev_hist_all)) return fail(HG_ERR_ARG, "hg_rank: bad argument");
    const Geo& g = c->geo;
    c->G = G; c->rank = rank;
    c->want_lists = c->staged_lists != 0 || c->LW > 2;
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(c->want_lists ? slots * 4 : 16));
    HG_TRY(c->out_dist.reserve(c->want_lists ? slots : 16));
    if (c->want_lists && G > 1) {
        HG_HIP(hipMemsetAsync(c->out_idx.p, 0xFF, slots * 4, c->stream));
        HG_HIP(hipMemsetAsync(c->out_dist.p, 0xFF, slots, c->stream));
    }
    HG_TRY(launch_plan(c, dev_hist_all));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    HG_TRY(launch_rank(c, 2, nbits));                // placement with the shared plan
    if (c->defer_verdict) {
        // the caller goes on as if the bet held (match bits, exchange, AP) and asks hg_bet_verdict at the end,
        // together with its final download: no host round trip in the middle of the step
        *bet_lost = -1;
        c->verdict_pending = true;
        c->verdict_known = false;
        c->lists_valid = c->want_lists;
        c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT;
        if (c->LW <= 2) c->stage |= ST_MATCH;
        else HG_TRY(do_match(c));
        return c->stage_end();
    }
    int flag = 0;
    HG_TRY(read_plan_flag(c, &flag));
    *bet_lost = flag;
    return settle_bet(c, flag);
}

int hg_bet_verdict(hg_ctx* c, int* bet_lost) {
    if (!c || !bet_lost) return fail(HG_ERR_ARG, "hg_bet_verdict: null argument");
    if (!c->verdict_pending) return fail(HG_ERR_STATE, "hg_bet_verdict: no deferred hg_rank outstanding");
    c->verdict_pending = false;
    int flag = 0;
    if (c->verdict_known) flag = c->verdict_flag;      // came over with hg_get_ap's download
    else HG_TRY(read_plan_flag(c, &flag));
    c->verdict_known = false;
    *bet_lost = flag;
    if (!flag) { c->opt_runs++; c->shard_bet_fail = 0; return HG_OK; }
    c->opt_runs++;
    c->opt_fallbacks++;
    c->shard_bet_fail++;
    c->lists_valid = false;
    c->stage = ST_DB | ST_Q;
    return HG_OK;
}

// The sharded bet with the per-query stages SPLIT over the ranks: after the all-gather of record counts and local bitmaps
// every rank merges and evaluates only its own queries [q0, q0 + nq) (instead of all Q on every rank), packs {AP, hits}
// and its verdict into `part`; one small all-gather later hg_unpack_parts gives every rank all of it.
int hg_merge_ap_part(hg_ctx* c, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, int64_t q0, int64_t nq,
                     int64_t width, void** dev_part, int64_t* nbytes) {
    if (!c || !dev_part || !nbytes || width < nq || width < 1) return fail(HG_ERR_ARG, "hg_merge_ap_part: bad argument");
    HG_TRY(merge_ranked_range(c, dev_hist_all, dev_bits_all, G, q0, nq, "hg_merge_ap_part"));
    const Geo& g = c->geo;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
    // AP of the merged rows only: k_ap's block index is the query
    {
        const i64 Qsave = c->geo.Q;
        HG_TRY(c->ap.reserve((size_t)Qsave * 8));
        HG_TRY(c->rel.reserve((size_t)Qsave * 4));
        HG_TRY(do_ap_range(c, q0, nq));
    }
    const size_t pb = (size_t)(width + 1) * 16;
    HG_TRY(c->part.reserve(pb));
    hipLaunchKernelGGL(k_pack_part, dim3(grid_for(width + 1)), dim3(256), 0, c->stream, c->ap.as<double>(), c->rel.as<u32>(),
                       c->err.as<int>(), (i64)q0, (i64)nq, (i64)width, c->part.as<double>());
    HG_TRY(c->check_launch("k_pack_part"));
    (void)g;
    *dev_part = c->part.p;
    *nbytes = (int64_t)pb;
    return c->stage_end();
}

int hg_unpack_parts(hg_ctx* c, const void* dev_parts_all, int G, int64_t width, double* host_ap, int64_t* host_rel, int* bet_lost) {
    if (!c || !dev_parts_all || G < 1 || width < 1 || !host_ap || !host_rel || !bet_lost) return fail(HG_ERR_ARG, "hg_unpack_parts: bad argument");
    HG_TRY(c->use());
    const i64 Q = c->geo.Q;
    const size_t pb = (size_t)(width + 1) * 16, total = pb * (size_t)G;
    HG_TRY(ensure_pin(c, total));
    HG_HIP(hipMemcpyAsync(c->pin, dev_parts_all, total, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    int flag = 0;
    i64 done = 0;
    for (int r = 0; r < G; ++r) {
        const double* p = (const double*)((const char*)c->pin + (size_t)r * pb);
        const i64 n = (i64)p[2 * width + 1];
        if (n < 0 || n > width || done + n > Q) return fail(HG_ERR_ARG, "hg_unpack_parts: rank %d reports %lld queries (width %lld, %lld of %lld placed)",
                                                            r, (long long)n, (long long)width, (long long)done, (long long)Q);
        flag |= p[2 * width] != 0.0;
        for (i64 i = 0; i < n; ++i) { host_ap[done + i] = p[2 * i]; host_rel[done + i] = (int64_t)p[2 * i + 1]; }
        done += n;
    }
    if (done != Q) return fail(HG_ERR_ARG, "hg_unpack_parts: the parts cover %lld of %lld queries", (long long)done, (long long)Q);
    *bet_lost = flag;
    c->verdict_pending = false;
    c->verdict_known = false;
    c->opt_runs++;                                     // the verdict comes from gathered data: the same on every rank
    if (flag) {
        c->opt_fallbacks++;
        c->shard_bet_fail++;
        c->lists_valid = false;
        c->stage = ST_DB | ST_Q;
    } else {
        c->shard_bet_fail = 0;
    }
    return HG_OK;
}

// ---- one-shot forms: every stage enqueued back to back, one synchronisation ----
// Optimistic bet (single shard, R << N): instead of a full histogram pass, sample
// every stride-th row batch, guess the threshold a few sigma high, select a
// superset with it, then verify: the records' exact histogram must contain R rows
// and no slice may have overflowed.  The verified result is identical to the
// exact path's; a failed bet reruns the exact path.
static bool optimistic_eligible(hg_ctx* c, int64_t R, int* stride_out, u32* need_out) {
    if (!c->opt_enable || c->opt_consecutive_fail >= 2) return false;
    if (R * 8 > c->N || c->N < 65536) return false;
    make_geometry(c);
    const int stride = auto_stride(c, R);
    if (stride < 2) return false;
    const i64 sampled = sampled_rows(c, stride);
    const double fr = (double)R * (double)sampled / (double)c->N;   // expected sample count at the true cut
    const double need = fr + (double)c->opt_sigma * std::sqrt(fr) + 1.0;
    *stride_out = stride;
    *need_out = (u32)std::ceil(need);
    return true;
}

// R = N on one shard (the reference's CIFAR-10 setting): every row is a member of every ranked list, so nothing
// has to be selected or written down -- the ranking kernel walks the shard's rows directly, computing each row's
// distance and match bit from the codes and labels in both of its passes (counting, then stable placement).
static int enqueue_all_rows(hg_ctx* c, int64_t R) {
    make_geometry(c);
    HG_TRY(set_R(c, R, 1, 0));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(c->err.reserve(4)); HG_TRY(c->failq.reserve(qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->sl_cnt.reserve(qb));
    HG_TRY(c->cand.reserve(64));
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(c->want_lists ? slots * 4 : 16));
    HG_TRY(c->out_dist.reserve(c->want_lists ? slots : 16));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    c->optimistic = false;
    c->crow = R;
    c->cap = 0;
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    c->direct_rank = true;
    c->rec8 = false;
    const int rc = launch_rank(c, 0, nbits);
    c->direct_rank = false;
    HG_TRY(rc);
    c->lists_valid = c->want_lists;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
    return HG_OK;
}

static int enqueue_exact(hg_ctx* c, int64_t R) {
    if (c->N == c->n_total && R == c->N && c->opt_all_rows && c->LW <= 2 && c->NW <= 8)
        return enqueue_all_rows(c, R);                 // one-shot calls are single-shard
    // (k_rank_direct ranks ANY R from the rows themselves, but with one block per query it re-reads the whole database per
    // query and runs one wavefront per SIMD: measured against k_hist + k_select + k_rank_fused it loses for N/8 < R < N --
    // 15.7 vs 12.3 ms at N = 200k, R = 100k; 82 vs 70 ms at N = 1M, R = 500k -- so only "rank_direct" = 2 routes that regime to it)
    if (c->opt_rank_direct == 2 && c->N == c->n_total && R * 8 > c->N && c->opt_all_rows && !c->is_sub && rank_direct_tile(c, R) > 0)
        return enqueue_all_rows(c, R);
    HG_TRY(do_hist(c, 1));
    HG_TRY(do_plan(c, R, nullptr, 1, 0));
    return do_select(c);
}

// The exact sequence with its second pass on the matrix cores (one shard, R << N): full histogram -> plan (exact
// threshold t, and sstar = the last segment whose ties at t are still inside the quota) -> k_select_mx with T = t,
// fixed-capacity slices -> the bet's rank stage, which cuts the ties at the quota.  Nothing is guessed, so the only way
// this can fail is a slice overflowing its capacity (clustered rows): *err then, and the caller runs enqueue_exact.
static bool exact_mx_applies(const hg_ctx* c, int64_t R) {
    return c->opt_exact_mfma && c->opt_select_mfma && c->N == c->n_total && R * 8 <= c->N && c->N >= 65536 && !c->is_sub;
}
static int enqueue_exact_mx(hg_ctx* c, int64_t R) {
    HG_TRY(do_hist(c, 1, true, true));                 // per segment pair on the matrix cores where that applies
    HG_TRY(do_plan(c, R, nullptr, 1, 0));              // c->t, c->sstar; leaves optimistic = false, crow = R
    const Geo& g = c->geo;
    // in all the slices hold R records + the ties of one segment beyond the quota, but unevenly: segments up to sstar carry
    // ALL their rows at distance t (the cut bucket is typically the fullest), later ones none -- budget like the bet does
    const double mean = 0.1 * (double)c->cand_budget_x10 * (double)R / (double)g.S;
    u32 cap = (u32)std::ceil(mean + 6.0 * std::sqrt(mean) + 16.0);
    cap = (cap + 15u) & ~15u;
    c->optimistic = true;
    c->exact_mx = true;
    c->cap = cap;
    c->crow = (i64)g.S * cap;
    HG_HIP(hipMemsetAsync(c->failq.p, 0, (size_t)g.Qpad * 4, c->stream));
    const int rc = do_select(c);
    c->exact_mx = false;
    return rc;
}

static int enqueue_optimistic(hg_ctx* c, int64_t R, int stride, u32 need_cnt) {
    (void)need_cnt;
    // the guess stops at the cut, far below b/2 when R <= N/8 on any data whose queries resemble the database: the
    // sampled pass writes only those planes (130 -> 68 MB per launch at C2).  A cut beyond them reads as a thin
    // sample -- everything is taken, the slices overflow, the exact sequence answers.
    HG_TRY(do_hist(c, stride, false, false, c->NB / 2 + 2));
    HG_TRY(set_R(c, R, 1, 0));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->tguess.reserve(qb));
    HG_TRY(c->sl_start.reserve((size_t)g.S * qb)); HG_TRY(c->sl_tie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->failq.reserve(qb));
    HG_TRY(c->err.reserve(4));
    HG_TRY(c->sstar.reserve(qb));
    c->t_begin(KI_GUESS);
    const Geo gh = hist_geometry(c);                   // the sampled pass ran on coarser segments
    {   // lanes per query by the number of sampled segments each has to sum
        const int ratio = (int)(gh.L / g.L);
        const u32 srows = (u32)sampled_rows(c, stride);
#define HG_GUESS(P)                                                                                                     \
        hipLaunchKernelGGL(k_guess_direct<P>, dim3(grid_for(g.Qpad, WPB * (64 / P))), dim3(256), 0, c->stream,          \
                           c->hist.as<u32>(), gh.S, ratio, (double)c->opt_sigma, (i64)c->n_total, srows,                \
                           c->tguess.as<int>(), c->sstar.as<int>(), c->failq.as<u32>(), c->err.as<int>(), g)
        // (more lanes per query shorten a lane's share of a plane but scatter a wavefront's loads over more lines: with 64
        // lanes for every query that the chip has room for, Q = 1000 went 0.068 -> 0.090 ms, C3 0.032 -> 0.061)
        if (gh.S <= 64) HG_GUESS(4);
        else if (gh.S <= 512) HG_GUESS(16);
        else HG_GUESS(64);
#undef HG_GUESS
    }
    c->t_end();
    HG_TRY(c->check_launch("k_guess_direct"));
    c->err_zeroed = true;                              // launch_rank need not clear the lost-bet flag again
    // slice capacity: a guessed cut typically keeps 1.3-3 R rows (the guess overshoots by at most one
    // distance bucket, and cumulative counts grow ~2x per bucket in the tail where the cut lies; clustered
    // codes grow faster) -- budget 4 R per query over the S segments plus 6 sigma per slice.  HBM is
    // plentiful (2.5 GB at C2); an overflow only costs the exact rerun.
    const double mean = 0.1 * (double)c->cand_budget_x10 * (double)c->cap_boost * (double)R / (double)g.S;
    u32 cap = slice_capacity(c, mean);
    c->optimistic = true;
    c->cap = cap;
    c->crow = (i64)g.S * cap;
    c->stage = ST_DB | ST_Q | ST_PLAN;
    return do_select(c);
}

static int enqueue_exact(hg_ctx* c, int64_t R);

// Lost bets are per query (a short superset, an overflowed slice).  When only a few queries lost,
// rerun just those through the exact sequence in a child context that borrows the database
// tables, and patch their results into place.  *handled = false: too many, caller reruns all.
static int rerun_lost_queries(hg_ctx* c, int64_t R, bool lists, bool with_ap, bool* handled) {
    *handled = false;
    const Geo g = c->geo;
    std::vector<u32> bad((size_t)g.Q);
    HG_HIP(hipMemcpyAsync(bad.data(), c->qbad.p, (size_t)g.Q * 4, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    std::vector<u32> lost;
    for (int q = 0; q < g.Q; ++q) if (bad[(size_t)q]) lost.push_back((u32)q);
    const i64 nF = (i64)lost.size();
    if (nF == 0 || nF * 8 > g.Q) return HG_OK;
    if (!c->sub) {
        c->sub = new hg_ctx();
        c->sub->is_sub = true;
        c->sub->device = c->device;
        c->sub->stream = c->stream;                  // same stream: ordered with the parent's work
    }
    hg_ctx* s = c->sub;
    s->N = c->N; s->b = c->b; s->C = c->C; s->n_total = c->n_total; s->NW = c->NW; s->NB = c->NB; s->LW = c->LW;
    s->idx_base = c->idx_base;
    s->target_units = c->target_units; s->min_segment = c->min_segment; s->opt_enable = 0;
    // a handful of queries: the per-segment bookkeeping (k_hist_reduce, k_seg_layout walk S segments per query) costs more than
    // the pair passes themselves -- 2048 segments: 0.8 ms of a 1 ms rerun; 256 keep every CU busy and cost 0.1
    s->opt_max_segments = 256;
    s->timing = 0;
    s->db.borrow(c->db);
    s->dblab.borrow(c->dblab);
    s->Q = nF;
    HG_TRY(c->flist.reserve((size_t)nF * 4));
    HG_HIP(hipMemcpyAsync(c->flist.p, lost.data(), (size_t)nF * 4, hipMemcpyHostToDevice, c->stream));
    HG_TRY(s->qc.reserve((size_t)nF * c->NW * 4 + 64 * 4));
    HG_TRY(s->qlab.reserve((size_t)nF * c->LW * 8));
    auto move = [&](const void* src, void* dst, i64 rowbytes, int gather) {
        hipLaunchKernelGGL(k_move_rows, dim3((unsigned)nF), dim3(256), 0, c->stream, (const u8*)src, (u8*)dst,
                           c->flist.as<u32>(), rowbytes, gather);
    };
    move(c->qc.p, s->qc.p, (i64)c->NW * 4, 1);
    move(c->qlab.p, s->qlab.p, (i64)c->LW * 8, 1);
    HG_TRY(c->check_launch("k_move_rows"));
    s->stage = ST_DB | ST_Q;
    s->want_lists = lists;
    HG_TRY(enqueue_exact(s, R));
    if (with_ap) HG_TRY(do_ap(s));
    move(s->mbits.p, c->mbits.p, c->RW * 8, 0);
    if (with_ap) {
        move(s->ap.p, c->ap.p, 8, 0);
        move(s->rel.p, c->rel.p, 4, 0);
    }
    if (lists) {
        move(s->out_idx.p, c->out_idx.p, R * 4, 0);
        move(s->out_dist.p, c->out_dist.p, R, 0);
    }
    HG_TRY(c->check_launch("k_move_rows"));
    HG_TRY(c->sync());                               // `lost` (the H2D source) must outlive the copy
    c->opt_requeried += nF;
    *handled = true;
    return HG_OK;
}

// The bet's sequence for hg_map, enqueued on the stream: sampled histogram -> guess -> select -> verify + order ->
// AP -> flag, AP and hit counts into pinned memory.  Pure enqueue (no synchronisation, no allocation once the
// buffers are warm), so it can run under stream capture.
static int enqueue_bet_with_ap(hg_ctx* c, int64_t R, int stride, u32 need_cnt) {
    c->t_step_begin();
    HG_TRY(enqueue_optimistic(c, R, stride, need_cnt));
    HG_TRY(do_ap(c));
    const size_t Q = (size_t)c->geo.Q;
    char* pb = (char*)c->pin;                  // [flag 16 B][ap Q x 8][rel Q x 4]
    HG_HIP(hipMemcpyAsync(pb, c->err.p, 4, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(hipMemcpyAsync(pb + 16, c->ap.p, Q * 8, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(hipMemcpyAsync(pb + 16 + Q * 8, c->rel.p, Q * 4, hipMemcpyDeviceToHost, c->stream));
    c->t_step_end();
    return HG_OK;
}

// Second sighting of the same step (same tables, options, R, timing level; no buffer moved since): capture it.
static int capture_step(hg_ctx* c, int64_t R, int stride, u32 need_cnt) {
    c->drop_graph();
    const unsigned long long epoch0 = g_alloc_epoch.load();
    HG_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    c->capturing = true;
    const int rc = enqueue_bet_with_ap(c, R, stride, need_cnt);
    c->capturing = false;
    hipGraph_t gr = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &gr);
    if (rc != HG_OK || e != hipSuccess || !gr || g_alloc_epoch != epoch0) {
        if (gr) (void)hipGraphDestroy(gr);
        c->drop_graph();
        (void)hipGetLastError();
        if (rc != HG_OK) return rc;
        return fail(HG_ERR_HIP, "step capture failed: %s", e != hipSuccess ? hipGetErrorString(e) : "a buffer moved during capture");
    }
    hipGraphExec_t ex = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    if (e2 != hipSuccess) {
        (void)hipGraphDestroy(gr);
        c->drop_graph();
        return fail(HG_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
    }
    auto& sg = c->sg;
    sg.graph = gr; sg.exec = ex;
    sg.epoch = g_alloc_epoch; sg.cfg = c->cfg_epoch; sg.R = R; sg.timing = c->timing;
    sg.stage = c->stage; sg.optimistic = c->optimistic; sg.lists_valid = c->lists_valid; sg.cap = c->cap; sg.crow = c->crow;
    sg.RW = c->RW; sg.geo = c->geo;
    c->graph_captures++;
    return HG_OK;
}

static int run_oneshot(hg_ctx* c, int64_t R, bool lists, bool with_ap) {
    c->real_lists = false;
    int stride = 0;
    u32 need_cnt = 0;
    if (R < 1 || R > c->n_total)
        return fail(HG_ERR_ARG, "R=%lld outside 1..N (N=%lld rows in the database)", (long long)R, (long long)c->n_total);
    c->want_lists = lists;
    const bool bet = optimistic_eligible(c, R, &stride, &need_cnt);
    int flag = 0;
    if (bet) {
        c->opt_runs++;
        if (with_ap) {
            HG_TRY(ensure_pin(c, (size_t)c->Q * 12 + 16));
            auto& sg = c->sg;
            bool launched = false;
            // event-record nodes inside a graph turned out slow and unreliable on ROCm 7.2 (a replayed step took 1.9 ms
            // instead of 1.55, elapsed times came back for one replay in twenty): with kernel timing on, steps stay eager
            if (c->opt_graph && !lists && !c->is_sub && c->timing == 0) {
                const bool same = sg.exec && sg.epoch == g_alloc_epoch && sg.cfg == c->cfg_epoch && sg.R == R && sg.timing == c->timing;
                const bool seen = sg.seen_epoch == g_alloc_epoch && sg.seen_cfg == c->cfg_epoch && sg.seen_R == R && sg.seen_timing == c->timing;
                if (!same && seen) {
                    if (capture_step(c, R, stride, need_cnt) != HG_OK) c->opt_graph = 0;      // not fatal: stay eager from now on
                }
                if (c->sg.exec && c->sg.epoch == g_alloc_epoch && c->sg.cfg == c->cfg_epoch && c->sg.R == R && c->sg.timing == c->timing) {
                    HG_HIP(hipGraphLaunch(sg.exec, c->stream));
                    HG_TRY(c->sync());
                    c->t_collect_graph();
                    // what the captured enqueue functions leave behind on the host side
                    HG_TRY(set_R(c, R, 1, 0));
                    c->geo = sg.geo; c->RW = sg.RW; c->stage = sg.stage; c->optimistic = sg.optimistic; c->lists_valid = sg.lists_valid;
                    c->cap = sg.cap; c->crow = sg.crow; c->err_zeroed = false;
                    c->graph_replays++;
                    launched = true;
                }
            }
            if (!launched) {
                HG_TRY(enqueue_bet_with_ap(c, R, stride, need_cnt));
                HG_TRY(c->sync());
                sg.seen_epoch = g_alloc_epoch; sg.seen_cfg = c->cfg_epoch; sg.seen_R = R; sg.seen_timing = c->timing;
            }
            flag = *(const int*)c->pin;
            c->ap_staged = flag == 0;
        } else {
            HG_TRY(enqueue_optimistic(c, R, stride, need_cnt));
            HG_TRY(read_plan_flag(c, &flag));
        }
        if (!flag) { c->opt_consecutive_fail = 0; return HG_OK; }
        bool handled = false;                      // some queries lost their bet
        HG_TRY(rerun_lost_queries(c, R, lists, with_ap, &handled));
        if (handled) { c->opt_consecutive_fail = 0; return HG_OK; }
        // many queries lost.  Before paying for the exact two-pass sequence (3x the bet at C2), bet once more with
        // twice the safety margin and twice the record budget -- the verification is what makes either bet exact.
        // Still lost: the hits crowd into few segments (a database stored class by class: ten classes put ten times the
        // mean into a query's slices), which no margin on the CUT cures -- escalate the slices' capacity (x8, x64, until a
        // slice would hold its whole segment) and remember what worked for the next calls on this database.
        if (c->opt_second_bet) {
            const i64 sigma0 = c->opt_sigma, budget0 = c->cand_budget_x10, boost0 = c->cap_boost;
            for (int attempt = 0; attempt < 3; ++attempt) {
                if (attempt > 0) {
                    if (c->cap >= (u32)((c->geo.L + 15) & ~15ll)) break;               // a slice already holds a segment
                    if ((double)c->geo.Q * (double)c->crow * 8.0 * 8.0 > 64e9) break;  // the record rows would not fit comfortably
                    c->cap_boost *= 8;
                }
                c->opt_sigma = 2 * sigma0 + 2;
                c->cand_budget_x10 = 2 * budget0;
                c->opt_rebets++;
                c->want_lists = lists;
                int rc;
                if (with_ap) {
                    rc = enqueue_bet_with_ap(c, R, stride, need_cnt);
                    if (rc == HG_OK) rc = c->sync();
                    flag = *(const int*)c->pin;
                    c->ap_staged = rc == HG_OK && flag == 0;
                } else {
                    rc = enqueue_optimistic(c, R, stride, need_cnt);
                    if (rc == HG_OK) rc = read_plan_flag(c, &flag);
                }
                c->opt_sigma = sigma0;
                c->cand_budget_x10 = budget0;
                if (rc != HG_OK) { c->cap_boost = boost0; return rc; }
                // held with twice the budget of a first bet at this boost: the next call's first bet gets that budget
                // (a class-sorted database of tight clusters lost every first bet at x8 and won every second one)
                auto keep = [&] { if (c->cap_boost < 4096) c->cap_boost *= 2; c->opt_consecutive_fail = 0; c->cfg_epoch++; };
                if (!flag) { keep(); return HG_OK; }
                handled = false;
                HG_TRY(rerun_lost_queries(c, R, lists, with_ap, &handled));
                if (handled) { keep(); return HG_OK; }
            }
            c->cap_boost = boost0;                 // nothing helped: do not keep paying for big slices
        }
        c->opt_fallbacks++;                        // still too many: exact path for all
        c->opt_consecutive_fail++;
        c->want_lists = lists;
    }
    if (exact_mx_applies(c, R)) {
        c->want_lists = lists;
        c->t_step_begin();
        HG_TRY(enqueue_exact_mx(c, R));
        if (with_ap) HG_TRY(do_ap(c));
        c->t_step_end();
        HG_TRY(read_plan_flag(c, &flag));
        if (!flag) return HG_OK;
        c->want_lists = lists;                         // a slice overflowed: the vector-ALU select with exact-sized slices
    }
    c->t_step_begin();
    HG_TRY(enqueue_exact(c, R));
    if (with_ap) HG_TRY(do_ap(c));
    c->t_step_end();
    return check_plan_flag(c);
}

int hg_topr(hg_ctx* c, int64_t R) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_topr", "hg_set_database + hg_set_queries"));
    return run_oneshot(c, R, true, false);
}

int hg_map(hg_ctx* c, int64_t R, double* host_ap, int64_t* host_rel) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_map", "hg_set_database + hg_set_queries"));
    HG_TRY(run_oneshot(c, R, false, true));
    return hg_get_ap(c, host_ap, host_rel);
}

// ---- real-valued ranking: sample -> guess -> select -> 4-pass radix sort -> finish ----
// one attempt; *lost = some query came up short of R records or overflowed a slice (bet mode only)
static int real_attempt(hg_ctx* c, int64_t R, bool bet, double sigma, double budget, bool with_ap, int* lost) {
    ++c->real_attempts;
    c->real_lds_ranked = 0;
    c->real_no_cut = !bet;
    make_geometry(c);
    {   // Float rows are 4*bpad bytes (32x a 64-bit code): keep a segment's rows within ~512 KB so the few
        // segments an XCD works on at a time stay in its 4 MiB L2 while all query tiles pass over them.
        Geo& gg = c->geo;
        i64 L = (i64)c->opt_real_seg_bytes / ((i64)c->bpad * 4);
        L = L / 16 * 16;
        if (L < 64) L = 64;
        if (gg.L > L) {
            gg.L = L;
            gg.S = (int)((gg.N + L - 1) / L);
            gg.nUnits = (i64)gg.S * gg.nQT;
            gg.nBlk = (int)((gg.nUnits + WPB - 1) / WPB);
        }
    }
    HG_TRY(set_R(c, R, 1, 0));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->thr.reserve(qb)); HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->failq.reserve(qb));
    HG_TRY(c->tot.reserve(qb)); HG_TRY(c->err.reserve(4)); HG_TRY(c->qbad.reserve(qb));
    HG_HIP(hipMemsetAsync(c->failq.p, 0, qb, c->stream));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    if (bet) {
        // sample so that about 64 of a query's top R rows are in it; guess the cut `sigma` deviations deep
        i64 stride = (i64)((double)R / (double)c->opt_real_sample_hits);
        if (stride < 1) stride = 1;
        const i64 M = (c->N + stride - 1) / stride;
        const double fr = (double)R * (double)M / (double)c->N;
        const u32 rank_s = (u32)std::ceil(fr + sigma * std::sqrt(fr)) + 1u;
        // samp[q][mstride]: the matrix-core sample pass stores 16 samples at a time (rows 64-byte aligned), the vector kernels M densely
        const i64 mstride = c->bpad <= 128 && c->opt_real_mfma ? (M + 15) / 16 * 16 : M;
        HG_TRY(c->samp.reserve((size_t)g.Q * mstride * 4));
        HG_TRY(real_sample(c, M, stride, mstride));
        c->t_begin(KI_REAL_GUESS);
        if (M <= RG_MMAX) hipLaunchKernelGGL(k_real_guess_lds, dim3(g.Q), dim3(1024), 0, c->stream, c->samp.as<float>(), M, mstride, rank_s, c->thr.as<float>());
        else hipLaunchKernelGGL(k_real_guess, dim3(g.Q), dim3(256), 0, c->stream, c->samp.as<float>(), M, mstride, rank_s, c->thr.as<float>());
        c->t_end();
        HG_TRY(c->check_launch("k_real_guess"));
        const double mean = budget * (double)R / (double)g.S;
        u32 cap = (u32)std::ceil(mean + 6.0 * std::sqrt(mean) + 16.0);
        cap = (cap + 15u) & ~15u;                         // a multiple of the compact records' ring (16) and flush piece (8)
        const u32 whole = (u32)((g.L + 15) & ~15ll);      // (a slice never needs more than its segment's rows)
        c->cap = cap < whole ? cap : whole;
    } else {
        // no bet: every row becomes a record (thr = -inf), slices are whole segments
        std::vector<float> ninf((size_t)g.Q, -INFINITY);
        HG_HIP(hipMemcpyAsync(c->thr.p, ninf.data(), (size_t)g.Q * 4, hipMemcpyHostToDevice, c->stream));
        HG_HIP(hipStreamSynchronize(c->stream));
        c->cap = (u32)g.L;
    }
    c->crow = (i64)g.S * c->cap;
    const size_t rows = (size_t)g.Q * c->crow * 8;
    // the record rows; the global-memory ranking passes (a query whose records exceed the LDS, the exhaustive mode) need two
    // more buffers of that size -- a widened bet (run_real) only goes as far as the rows alone stay moderate
    if (bet && rows > (size_t)64 << 30) { *lost = 1; return HG_OK; }
    if (!bet && rows * 3 > (size_t)200 << 30)
        return fail(HG_ERR_NOMEM, "real-valued ranking: %zu GB of records needed (Q=%d, %lld per query)", rows * 3 >> 30, g.Q, (long long)c->crow);
    HG_TRY(c->cand.reserve(rows));
    HG_TRY(real_select(c));
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(c->out_idx.reserve(slots * 4));
    HG_TRY(c->scores.reserve(slots * 4));
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    if (c->real_filtered && bet && c->opt_real_sort_lds && g.S <= RK_SMAX && R <= RK_RMAX) {
        // a query's records fit the LDS of one workgroup: copy + select + counting passes + ranked list in one kernel
        constexpr int NA = 14336;
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_rank_lds<NA>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)real_rank_lds_bytes<NA>()));
        c->t_begin(KI_RADIX);
        hipLaunchKernelGGL(k_real_rank_lds<NA>, dim3(g.Q), dim3(1024), real_rank_lds_bytes<NA>(), c->stream, c->cand.as<u64>(), c->crow, c->cap,
                           c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->thr.as<float>(), c->out_idx.as<u32>(), c->scores.as<float>(),
                           c->dblab.as<u64>(), c->qlab.as<u64>(), c->mbits.as<u64>(), c->RW, c->err.as<int>(), c->qbad.as<u32>(), g);
        c->t_end();
        HG_TRY(c->check_launch("k_real_rank_lds"));
        int flag = 0;
        HG_TRY(read_plan_flag(c, &flag));
        if (!(flag & 2)) {
            c->real_lds_ranked = 1;
            *lost = flag & 1;
            c->stage = ST_DB | ST_Q | ST_SELECT;
            if (*lost) return HG_OK;
            c->stage |= ST_MATCH;                          // the rank kernel left the match bits too
            if (with_ap) HG_TRY(do_ap(c));
            return c->sync();
        }
        HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));    // some query's records exceed the LDS: the global-memory passes rank them all
    }
    if (rows * 3 > (size_t)200 << 30) {
        if (bet) { *lost = 1; return HG_OK; }
        return fail(HG_ERR_NOMEM, "real-valued ranking: %zu GB of records needed (Q=%d, %lld per query)", rows * 3 >> 30, g.Q, (long long)c->crow);
    }
    HG_TRY(c->sortA.reserve(rows)); HG_TRY(c->sortB.reserve(rows));
    const int nwav = c->crow >= 16384 ? 16 : 4;
    const size_t lds = (size_t)(nwav + 1) * 256 * 4;
    u64* bufs[2] = {c->sortA.as<u64>(), c->sortB.as<u64>()};
    const u64* in = c->cand.as<u64>();
    bool grouped = false;
    if (c->opt_real_groups && g.S <= 8192 && !bet && c->crow <= (i64)RG_MAXG * RG_CAP) {
        // every row a record (R = N on a CIFAR-sized database): split by score range into LDS-sized groups, order each group
        // in LDS (k_real_group_split / k_real_group_sort) -- two trips of the records through memory instead of the radix
        // passes' four, 3.1 -> 0.85 ms at C1; piled-up scores come back as bit 2 of the flag.  (A bet's list beyond the LDS --
        // 19 000 records in 489 short slices at R = 10 000 -- stays with the radix passes: 3.9 ms against 5.7 this way.)
        HG_TRY(c->gtab.reserve((size_t)g.Q * (RG_MAXG + 1) * 4));
        // (a group spans at least RG_CAP / 2 of cumulative count -- the largest bucket is at most RG_CAP / 2: at most 2 n / RG_CAP + 1 groups)
        const int maxg = (int)std::min<i64>(RG_MAXG, 2 * c->crow / RG_CAP + 2);
        static bool lds_set = false;
        if (!lds_set) {
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_group_sort), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)real_group_sort_lds()));
            lds_set = true;
        }
        c->t_begin(KI_RADIX);
        hipLaunchKernelGGL(k_real_group_split, dim3(g.Q), dim3(1024), real_group_split_lds(g.S), c->stream, c->cand.as<u64>(), c->crow, c->cap,
                           c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->tot.as<u32>(), c->sortA.as<u64>(), c->gtab.as<u32>(), c->crow, c->err.as<int>(), maxg, g);
        c->t_end();
        HG_TRY(c->check_launch("k_real_group_split"));
        // the sort writes the ranked lists and the match bits itself (k_real_finish and k_match are for the radix passes)
        HG_HIP(hipMemsetAsync(c->mbits.p, 0, (size_t)g.Q * c->RW * 8, c->stream));
        HG_HIP(hipMemsetAsync(c->qbad.p, 0, (size_t)g.Qpad * 4, c->stream));
        const GroupOut go{c->out_idx.as<u32>(), c->scores.as<float>(), c->mbits.as<u32>(), c->dblab.as<u64>(), c->qlab.as<u64>(), c->RW, g.R, g.LW, g.idx_base};
        c->t_begin(KI_RADIX);
        hipLaunchKernelGGL(k_real_group_sort, dim3(g.Q, maxg), dim3(1024), real_group_sort_lds(), c->stream, c->sortA.as<u64>(), c->gtab.as<u32>(),
                           go, c->crow, c->err.as<int>());
        c->t_end();
        HG_TRY(c->check_launch("k_real_group_sort"));
        int flag = 0;
        HG_TRY(read_plan_flag(c, &flag));
        if (flag & 4) HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
        else grouped = true;
    }
    c->real_grouped = grouped ? 1 : 0;
    if (grouped) {
        c->stage = ST_DB | ST_Q | ST_SELECT | ST_MATCH;
        if (with_ap) HG_TRY(do_ap(c));
        HG_TRY(read_plan_flag(c, lost));
        return HG_OK;
    }
    for (int pass = 0; pass < 4 && !grouped; ++pass) {
        RadixArgs ra{c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->tot.as<u32>(), c->cap, c->crow, c->crow, pass == 0, 32 + 8 * pass};
        u64* out = bufs[pass & 1];
        c->t_begin(KI_RADIX);
        if (nwav == 16) hipLaunchKernelGGL(k_radix_pass<16>, dim3(g.Q), dim3(1024), lds, c->stream, in, out, ra, g);
        else hipLaunchKernelGGL(k_radix_pass<4>, dim3(g.Q), dim3(256), lds, c->stream, in, out, ra, g);
        c->t_end();
        HG_TRY(c->check_launch("k_radix_pass"));
        in = out;
    }
    c->t_begin(KI_REAL_FINISH);
    const i64 nKB = grid_for(g.R);
    if (nKB * g.Q > 0x7FFFFFFFll) return fail(HG_ERR_ARG, "real-valued ranking: Q*R too large for one launch");
    hipLaunchKernelGGL(k_real_finish, dim3((unsigned)(nKB * g.Q)), dim3(256), 0, c->stream, in, c->crow, c->tot.as<u32>(),
                       c->out_idx.as<u32>(), c->scores.as<float>(), c->err.as<int>(), c->qbad.as<u32>(), (int)nKB,
                       c->real_filtered ? c->thr.as<float>() : nullptr, g);
    c->t_end();
    HG_TRY(c->check_launch("k_real_finish"));
    c->stage = ST_DB | ST_Q | ST_SELECT;
    HG_TRY(do_match(c));                               // label gather through the ranked idx list
    if (with_ap) HG_TRY(do_ap(c));
    HG_TRY(read_plan_flag(c, lost));
    return HG_OK;
}

static int run_real(hg_ctx* c, int64_t R, bool with_ap) {
    if (!c->bpad || !c->dbf.p || !c->qf.p || !c->dbf_resident || !c->qf_resident)
        return fail(HG_ERR_STATE, "real-valued ranking needs the float features on the GPU: load them with hg_set_database_f32 / "
                                  "hg_set_queries_f32 (option keep_floats = 1 if the database is a +-1 code)");
    if (c->n_total != c->N) return fail(HG_ERR_STATE, "real-valued ranking is single-shard");
    if (R < 1 || R > c->N) return fail(HG_ERR_ARG, "R=%lld outside 1..N (N=%lld rows in the database)", (long long)R, (long long)c->N);
    int lost = 0;
    c->real_attempts = 0;
    if (R * 8 <= c->N && c->N >= 65536) {              // bet on a sampled cut; retry once deeper, then give up betting
        const double boost0 = (double)c->real_cap_boost;
        HG_TRY(real_attempt(c, R, true, 6.0, 3.0 * boost0, with_ap, &lost));
        if (!lost) { c->real_lists = true; return HG_OK; }
        // a deeper cut with twice the budget; then -- features that follow the labels in a database stored class by class
        // put a query's top rows into a tenth of its slices -- eight and sixty-four times the slices' capacity, kept for
        // the next calls on this database (the exhaustive mode below writes EVERY pair down: 80 GB at 10k x 1M)
        for (int attempt = 0; attempt < 3; ++attempt) {
            if (attempt > 0) {
                if (c->cap >= (u32)((c->geo.L + 15) & ~15ll)) break;      // a slice already holds its segment
                c->real_cap_boost *= 8;
            }
            HG_TRY(real_attempt(c, R, true, 16.0, 6.0 * (double)c->real_cap_boost, with_ap, &lost));
            if (!lost) {
                if (c->real_cap_boost < 4096) c->real_cap_boost *= 2;     // (the budget that held: 6 = 2 x 3)
                c->real_lists = true;
                return HG_OK;
            }
        }
        c->real_cap_boost = (i64)boost0;
    }
    HG_TRY(real_attempt(c, R, false, 0.0, 0.0, with_ap, &lost));
    if (lost) return fail(HG_ERR_HIP, "real-valued ranking: internal error, exhaustive pass came up short");
    c->real_lists = true;
    return HG_OK;
}

int hg_topr_real(hg_ctx* c, int64_t R) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_topr_real", "hg_set_database_f32 + hg_set_queries_f32"));
    return run_real(c, R, false);
}

int hg_map_real(hg_ctx* c, int64_t R, double* host_ap, int64_t* host_rel) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_map_real", "hg_set_database_f32 + hg_set_queries_f32"));
    HG_TRY(run_real(c, R, true));
    return hg_get_ap(c, host_ap, host_rel);
}

int hg_get_topr_real(hg_ctx* c, uint32_t* host_idx, float* host_scores) {
    HG_TRY(need(c, ST_SELECT, "hg_get_topr_real", "hg_topr_real / hg_map_real"));
    if (!c->real_lists) return fail(HG_ERR_STATE, "hg_get_topr_real: the last ranking was not a real-valued one");
    const size_t slots = (size_t)c->geo.Q * c->geo.R;
    if (host_idx) HG_HIP(hipMemcpyAsync(host_idx, c->out_idx.p, slots * 4, hipMemcpyDeviceToHost, c->stream));
    if (host_scores) HG_HIP(hipMemcpyAsync(host_scores, c->scores.p, slots * 4, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

int hg_get_topr(hg_ctx* c, uint32_t* host_idx, uint8_t* host_dist) {
    HG_TRY(need(c, ST_SELECT, "hg_get_topr", "hg_select"));
    if (!c->lists_valid) return fail(HG_ERR_STATE, "ranked lists were not materialised by the last call (hg_map skips them; use hg_topr)");
    const size_t slots = (size_t)c->geo.Q * c->geo.R;
    if (host_idx) HG_HIP(hipMemcpyAsync(host_idx, c->out_idx.p, slots * 4, hipMemcpyDeviceToHost, c->stream));
    if (host_dist) HG_HIP(hipMemcpyAsync(host_dist, c->out_dist.p, slots, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

int hg_get_match(hg_ctx* c, uint8_t* host_imatch) {
    HG_TRY(need(c, ST_MATCH, "hg_get_match", "hg_match"));
    if (!host_imatch) return fail(HG_ERR_ARG, "hg_get_match: null pointer");
    const i64 Q = c->geo.Q, R = c->geo.R, RW = c->RW;
    std::vector<u64> bits((size_t)Q * RW);
    HG_HIP(hipMemcpyAsync(bits.data(), c->mbits.p, bits.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    for (i64 q = 0; q < Q; ++q)
        for (i64 k = 0; k < R; ++k) host_imatch[q * R + k] = (u8)((bits[q * RW + (k >> 6)] >> (k & 63)) & 1ull);
    return HG_OK;
}

int hg_get_ap(hg_ctx* c, double* host_ap, int64_t* host_rel) {
    HG_TRY(need(c, ST_AP, "hg_get_ap", "hg_ap"));
    const i64 Q = c->geo.Q;
    if (c->ap_staged && c->pin) {                      // the one-shot call already brought them over
        const char* pb = (const char*)c->pin;
        if (host_ap) memcpy(host_ap, pb + 16, (size_t)Q * 8);
        if (host_rel) {
            const u32* r = (const u32*)(pb + 16 + (size_t)Q * 8);
            for (i64 q = 0; q < Q; ++q) host_rel[q] = r[q];
        }
        return HG_OK;
    }
    // one batch of copies into pinned memory, one synchronisation; a deferred verdict rides along
    HG_TRY(ensure_pin(c, (size_t)Q * 12 + 16));
    char* pb = (char*)c->pin;
    if (c->verdict_pending) HG_HIP(hipMemcpyAsync(pb, c->err.p, 4, hipMemcpyDeviceToHost, c->stream));
    if (host_ap) HG_HIP(hipMemcpyAsync(pb + 16, c->ap.p, (size_t)Q * 8, hipMemcpyDeviceToHost, c->stream));
    if (host_rel) HG_HIP(hipMemcpyAsync(pb + 16 + (size_t)Q * 8, c->rel.p, (size_t)Q * 4, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    if (c->verdict_pending) { c->verdict_flag = *(const int*)pb; c->verdict_known = true; }
    if (host_ap) memcpy(host_ap, pb + 16, (size_t)Q * 8);
    if (host_rel) {
        const u32* r = (const u32*)(pb + 16 + (size_t)Q * 8);
        for (i64 q = 0; q < Q; ++q) host_rel[q] = r[q];
    }
    return HG_OK;
}

int hg_get_hist(hg_ctx* c, uint32_t* host_hist) {
    HG_TRY(need(c, ST_HIST, "hg_get_hist", "hg_hist"));
    if (!host_hist) return fail(HG_ERR_ARG, "hg_get_hist: null pointer");
    const Geo& g = c->geo;
    HG_HIP(hipMemcpy2DAsync(host_hist, (size_t)g.Q * 4, c->hown.p, (size_t)g.Qpad * 4, (size_t)g.Q * 4, (size_t)g.NB,
                            hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}


// ---- collectives: RCCL over xGMI, on the context's own stream (no PyTorch anywhere) ----------------------------
int hg_comm_unique_id(uint8_t* id) {
    if (!id) return fail(HG_ERR_ARG, "hg_comm_unique_id: null pointer");
    HG_TRY(rccl_load());
    ncclUniqueId u;
    HG_NCCL(g_rccl.GetUniqueId(&u));
    static_assert(sizeof u == HG_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return HG_OK;
}

int hg_comm_init(hg_ctx* c, const uint8_t* id, int rank, int world) {
    if (!c || !id) return fail(HG_ERR_ARG, "hg_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(HG_ERR_ARG, "hg_comm_init: rank %d of %d", rank, world);
    if (c->comm) return fail(HG_ERR_STATE, "hg_comm_init: the context already has a communicator (hg_comm_destroy first)");
    HG_TRY(c->use());
    HG_TRY(rccl_load());
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    // RCCL prints a version banner on stdout when a communicator comes up; stdout belongs to the caller (bench.py's
    // one JSON line): send whatever the library prints during the call to stderr instead
    fflush(stdout);
    const int saved = dup(1);
    if (saved >= 0) (void)dup2(2, 1);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    fflush(stdout);
    if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
    if (r != ncclSuccess) { c->comm = nullptr; return fail(HG_ERR_HIP, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    c->comm_rank = rank;
    c->comm_world = world;
    return HG_OK;
}

int hg_comm_destroy(hg_ctx* c) {
    if (!c) return fail(HG_ERR_ARG, "hg_comm_destroy: null context");
    if (!c->comm) return HG_OK;
    HG_TRY(c->use());
    HG_TRY(c->sync());
    ncclComm_t k = c->comm;
    c->comm = nullptr;
    c->comm_rank = 0; c->comm_world = 1;
    HG_NCCL(g_rccl.CommDestroy(k));
    return HG_OK;
}

int hg_comm_info(hg_ctx* c, int* rank, int* world) {
    if (!c) return fail(HG_ERR_ARG, "hg_comm_info: null context");
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    if (world) *world = c->comm ? c->comm_world : 0;      // 0: no communicator
    return HG_OK;
}

int hg_allgather(hg_ctx* c, int slot, const void* dev_src, int64_t nbytes, void** dev_gathered) {
    if (!c || !dev_src || !dev_gathered || nbytes < 1) return fail(HG_ERR_ARG, "hg_allgather: bad argument");
    if (slot < 0 || slot >= 4) return fail(HG_ERR_ARG, "hg_allgather: slot %d outside 0..3", slot);
    if (!c->comm) return fail(HG_ERR_STATE, "hg_allgather: no communicator (hg_comm_init)");
    HG_TRY(c->use());
    DevBuf& out = c->gathered[slot];
    HG_TRY(out.reserve((size_t)nbytes * c->comm_world));
    c->t_begin(KI_COMM);
    HG_NCCL(g_rccl.AllGather(dev_src, out.p, (size_t)nbytes, ncclUint8, c->comm, c->stream));
    c->t_end();
    *dev_gathered = out.p;
    return c->stage_end();
}

// The north star's exchange: every shard's ranked (dist, idx) lists all-gathered and merged (exactly one shard owns a
// slot, the others hold HG_IDX_NONE / 0xFF there).  hg_get_topr then returns the global lists on every rank.
int hg_allgather_topr(hg_ctx* c) {
    HG_TRY(need(c, ST_SELECT, "hg_allgather_topr", "hg_select / hg_rank"));
    if (!c->lists_valid) return fail(HG_ERR_STATE, "hg_allgather_topr: ranked lists were not materialised by the last call");
    if (!c->comm) return fail(HG_ERR_STATE, "hg_allgather_topr: no communicator (hg_comm_init)");
    const i64 n = (i64)c->geo.Q * c->geo.R;
    const int G = c->comm_world;
    HG_TRY(c->gath_idx.reserve((size_t)n * 4 * G));       // own landing zones: hg_allgather's slots may hold live data
    HG_TRY(c->gath_dist.reserve((size_t)n * G));
    c->t_begin(KI_COMM);
    HG_NCCL(g_rccl.AllGather(c->out_idx.p, c->gath_idx.p, (size_t)n * 4, ncclUint8, c->comm, c->stream));
    HG_NCCL(g_rccl.AllGather(c->out_dist.p, c->gath_dist.p, (size_t)n, ncclUint8, c->comm, c->stream));
    c->t_end();
    c->t_begin(KI_MERGE);
    hipLaunchKernelGGL(k_min_topr, dim3(grid_for(n)), dim3(256), 0, c->stream, c->gath_idx.as<u32>(), c->gath_dist.as<u8>(),
                       c->out_idx.as<u32>(), c->out_dist.as<u8>(), n, G);
    c->t_end();
    HG_TRY(c->check_launch("k_min_topr"));
    return c->stage_end();
}

// max over the ranks of one host double (step times of a benchmark), and a barrier: both one tiny all-reduce
int hg_allreduce_max_f64(hg_ctx* c, double* host_inout) {
    if (!c || !host_inout) return fail(HG_ERR_ARG, "hg_allreduce_max_f64: null argument");
    if (!c->comm) return fail(HG_ERR_STATE, "hg_allreduce_max_f64: no communicator (hg_comm_init)");
    HG_TRY(c->use());
    HG_TRY(c->comm_tmp.reserve(16));
    HG_HIP(hipMemcpyAsync(c->comm_tmp.p, host_inout, 8, hipMemcpyHostToDevice, c->stream));
    HG_NCCL(g_rccl.AllReduce(c->comm_tmp.p, c->comm_tmp.as<char>() + 8, 1, ncclFloat64, ncclMax, c->comm, c->stream));
    HG_HIP(hipMemcpyAsync(host_inout, c->comm_tmp.as<char>() + 8, 8, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

int hg_barrier(hg_ctx* c) {
    double x = 0.0;
    return hg_allreduce_max_f64(c, &x);
}

// Context-owned device scratch (grows only) and a stream-ordered device-to-device copy: what an in-process
// communicator (virtual shards of one GPU in the tests) needs to do hg_allgather's job without RCCL.
int hg_scratch(hg_ctx* c, int slot, int64_t nbytes, void** dev_ptr) {
    if (!c || !dev_ptr || nbytes < 1 || slot < 0 || slot >= 4) return fail(HG_ERR_ARG, "hg_scratch: bad argument");
    HG_TRY(c->use());
    HG_TRY(c->scratch[slot].reserve((size_t)nbytes));
    *dev_ptr = c->scratch[slot].p;
    return HG_OK;
}

int hg_memcpy_dtod(hg_ctx* c, void* dev_dst, const void* dev_src, int64_t nbytes) {
    if (!c || !dev_dst || !dev_src || nbytes < 0) return fail(HG_ERR_ARG, "hg_memcpy_dtod: bad argument");
    HG_TRY(c->use());
    if (nbytes) HG_HIP(hipMemcpyAsync(dev_dst, dev_src, (size_t)nbytes, hipMemcpyDeviceToDevice, c->stream));
    return c->stage_end();
}

// device <-> host copies of raw device addresses, complete on return (a stand-in communicator that goes through the
// host -- tests/file_comm.py, for multi-process dry runs on one GPU -- is their only user)
int hg_memcpy_dtoh(hg_ctx* c, void* host_dst, const void* dev_src, int64_t nbytes) {
    if (!c || !host_dst || !dev_src || nbytes < 0) return fail(HG_ERR_ARG, "hg_memcpy_dtoh: bad argument");
    HG_TRY(c->use());
    if (nbytes) HG_HIP(hipMemcpyAsync(host_dst, dev_src, (size_t)nbytes, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

int hg_memcpy_htod(hg_ctx* c, void* dev_dst, const void* host_src, int64_t nbytes) {
    if (!c || !dev_dst || !host_src || nbytes < 0) return fail(HG_ERR_ARG, "hg_memcpy_htod: bad argument");
    HG_TRY(c->use());
    if (nbytes) HG_HIP(hipMemcpyAsync(dev_dst, host_src, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    return c->sync();
}

int hg_synchronize(hg_ctx* c) {
    if (!c) return fail(HG_ERR_ARG, "hg_synchronize: null context");
    HG_TRY(c->use());
    return c->sync();
}

int hg_set_stream(hg_ctx* c, void* stream) {
    if (!c) return fail(HG_ERR_ARG, "hg_set_stream: null context");
    HG_TRY(c->use());
    HG_TRY(c->sync());                               // drain the old stream first
    c->drop_graph();
    c->cfg_epoch++;
    if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        HG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    if (c->sub) c->sub->stream = c->stream;
    return HG_OK;
}

int hg_set_option(hg_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return fail(HG_ERR_ARG, "hg_set_option: null argument");
    c->cfg_epoch++;                                    // whatever changes: a captured step is rebuilt
    if (!strcmp(key, "step_graph")) { c->opt_graph = value != 0; return HG_OK; }
    if (!strcmp(key, "stage_sync")) { c->stage_sync = value != 0; return HG_OK; }
    if (!strcmp(key, "target_units")) {
        if (value < 1) return fail(HG_ERR_ARG, "target_units must be >= 1");
        c->target_units = value;
    } else if (!strcmp(key, "min_segment")) {
        if (value < 16) return fail(HG_ERR_ARG, "min_segment must be >= 16");
        c->min_segment = value;
    } else if (!strcmp(key, "max_segments")) {
        if (value < 1) return fail(HG_ERR_ARG, "max_segments must be >= 1");
        c->opt_max_segments = value;
    } else if (!strcmp(key, "optimistic")) {
        c->opt_enable = value != 0;
        c->opt_consecutive_fail = c->shard_bet_fail = 0;
    } else if (!strcmp(key, "sample_stride")) {
        if (value < 0 || value > 1024) return fail(HG_ERR_ARG, "sample_stride must be 0 (auto) .. 1024");
        c->opt_stride = value;
    } else if (!strcmp(key, "guess_sigma")) {
        if (value < 0 || value > 64) return fail(HG_ERR_ARG, "guess_sigma must be 0..64");
        c->opt_sigma = value;
    } else if (!strcmp(key, "staged_lists")) {
        c->staged_lists = value != 0;
    } else if (!strcmp(key, "real_queries_per_lane")) {
        if (value != 1 && value != 2) return fail(HG_ERR_ARG, "real_queries_per_lane must be 1 or 2");
        c->opt_real_qpl = value;
    } else if (!strcmp(key, "rank_waves")) {
        if (value != 0 && value != 4 && value != 16) return fail(HG_ERR_ARG, "rank_waves must be 0, 4 or 16");
        c->opt_rank_waves = value;
    } else if (!strcmp(key, "all_rows_shortcut")) {
        c->opt_all_rows = value != 0;
    } else if (!strcmp(key, "sample_ratio")) {
        if (value < 1 || value > 64) return fail(HG_ERR_ARG, "sample_ratio must be 1..64");
        c->opt_sample_ratio = value;
    } else if (!strcmp(key, "defer_verdict")) {
        c->defer_verdict = value != 0;
    } else if (!strcmp(key, "rank_wave")) {
        if (value < 0 || value > 400) return fail(HG_ERR_ARG, "rank_wave must be 0 (off) or the LDS record capacity in tenths of R, <= 400");
        c->opt_rank_wave = value;
    } else if (!strcmp(key, "timing_every")) {
        if (value < 1 || value > 1024) return fail(HG_ERR_ARG, "timing_every must be 1..1024");
        c->opt_timing_every = value;
    } else if (!strcmp(key, "cap_boost")) {
        if (value < 1 || value > 4096) return fail(HG_ERR_ARG, "cap_boost must be 1..4096");
        c->cap_boost = value;
    } else if (!strcmp(key, "rank_direct_lds")) {
        if (value < 32 || value > 160) return fail(HG_ERR_ARG, "rank_direct_lds must be 32..160 (KB)");
        c->opt_rank_direct_lds = value;
    } else if (!strcmp(key, "rank_direct")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "rank_direct must be 0, 1 (R = N) or 2 (also N/8 < R < N)");
        c->opt_rank_direct = value;
    } else if (!strcmp(key, "rank_wave_max")) {
        if (value < 0 || value > 16128) return fail(HG_ERR_ARG, "rank_wave_max must be 0..16128 (a lane's chunk must fit its byte counters)");
        c->opt_rank_wave_max = value;
    } else if (!strcmp(key, "select_packed")) {
        c->opt_select_packed = value;
    } else if (!strcmp(key, "rank_lds")) {
        c->opt_rank_lds = value != 0;
    } else if (!strcmp(key, "rank_cnt")) {
        c->opt_rank_cnt = value != 0;
    } else if (!strcmp(key, "host_pack")) {
        c->opt_host_pack = value != 0;
    } else if (!strcmp(key, "keep_floats")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "keep_floats must be 0, 1 or 2");
        c->opt_keep_floats = value;
    } else if (!strcmp(key, "pack_threads")) {
        if (value < 0 || value > 1024) return fail(HG_ERR_ARG, "pack_threads must be 0..1024");
        c->opt_pack_threads = value;
    } else if (!strcmp(key, "compact_records")) {
        c->opt_compact = value != 0;
    } else if (!strcmp(key, "second_bet")) {
        c->opt_second_bet = value != 0;
    } else if (!strcmp(key, "lds_pad")) {
        if (value < 0 || value > 24 * 1024) return fail(HG_ERR_ARG, "lds_pad must be 0..24576");
        c->opt_lds_pad = value;
    } else if (!strcmp(key, "hist_mfma")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "hist_mfma must be 0, 1 or 2");
        c->opt_hist_mfma = value;
    } else if (!strcmp(key, "ap_recip")) {
        c->opt_ap_recip = value != 0;
    } else if (!strcmp(key, "exact_mfma")) {
        c->opt_exact_mfma = value != 0;
    } else if (!strcmp(key, "select_mfma")) {
        c->opt_select_mfma = value != 0;
    } else if (!strcmp(key, "probe_select")) {
        if (value && !kProbes)
            return fail(HG_ERR_ARG, "probe_select: this is the production build -- the probes live in libhashgan_amd_probe.so "
                                    "(python -m hashgan_amd.build --probes, HG_LIBRARY=<path>)");
        c->opt_probe = value;
    } else if (!strcmp(key, "select_qt")) {
        (void)value;                                   // retired (round 1 experiment): the tile count follows the code length
    } else if (!strcmp(key, "real_mfma")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "real_mfma must be 0, 1 or 2");
        c->opt_real_mfma = value;
    } else if (!strcmp(key, "real_groups")) {
        c->opt_real_groups = value != 0;
    } else if (!strcmp(key, "real_sort_lds")) {
        c->opt_real_sort_lds = value != 0;
    } else if (!strcmp(key, "real_sample_hits")) {
        if (value < 16 || value > 4096) return fail(HG_ERR_ARG, "real_sample_hits must be 16..4096");
        c->opt_real_sample_hits = value;
    } else if (!strcmp(key, "real_segment_bytes")) {
        if (value < 4096) return fail(HG_ERR_ARG, "real_segment_bytes must be >= 4096");
        c->opt_real_seg_bytes = value;
    } else if (!strcmp(key, "cand_budget_x10")) {
        if (value < 11 || value > 1000) return fail(HG_ERR_ARG, "cand_budget_x10 must be 11..1000");
        c->cand_budget_x10 = value;
    } else {
        return fail(HG_ERR_ARG, "hg_set_option: unknown key '%s'", key);
    }
    return HG_OK;
}

// Work buffers only grow (a big call leaves gigabytes behind for the next one to reuse); hg_trim
// gives everything but the resident tables back.
int hg_trim(hg_ctx* c) {
    if (!c) return fail(HG_ERR_ARG, "hg_trim: null context");
    HG_TRY(c->use());
    HG_TRY(c->sync());
    DevBuf* work[] = {&c->hist, &c->seglt, &c->segtie, &c->sl_start, &c->sl_tie, &c->sl_cnt, &c->cand, &c->out_idx,
                      &c->out_dist, &c->stage_in, &c->hwq, &c->samp, &c->sortA, &c->sortB, &c->gtab, &c->scores, &c->bigq, &c->mbits2,
                      &c->dbx, &c->qx, &c->dbx2, &c->qx2, &c->dbfx, &c->dbfb, &c->sampx};   // the images are rebuilt on demand
    for (auto* d : work) d->release();
    c->dbfx_valid = false;
    c->dbfb_valid = false;
    for (auto& d : c->gathered) d.release();
    for (auto& d : c->scratch) d.release();
    c->gath_idx.release(); c->gath_dist.release();
    c->dbx_valid = c->qx_valid = c->dbx2_valid = c->qx2_valid = false;
    c->dbx8.release(); c->dbx8_valid = false;
    c->dbx3.release(); c->dbx3_valid = false;
    if (c->sub) { hg_ctx* s = c->sub; c->sub = nullptr; (void)hg_destroy(s); }
    c->stage &= (ST_DB | ST_Q);
    c->lists_valid = false;
    c->real_lists = false;
    return HG_OK;
}

int hg_get_stat(hg_ctx* c, const char* key, int64_t* value) {
    if (!c || !key || !value) return fail(HG_ERR_ARG, "hg_get_stat: null argument");
    if (!strcmp(key, "optimistic_runs")) *value = c->opt_runs;
    else if (!strcmp(key, "optimistic_fallbacks")) *value = c->opt_fallbacks;
    else if (!strcmp(key, "optimistic_requeried")) *value = c->opt_requeried;
    else if (!strcmp(key, "optimistic_rebets")) *value = c->opt_rebets;
    else if (!strcmp(key, "cap_boost")) *value = c->cap_boost;
    else if (!strcmp(key, "real_cap_boost")) *value = c->real_cap_boost;
    else if (!strcmp(key, "real_grouped")) *value = c->real_grouped;
    else if (!strcmp(key, "last_optimistic")) *value = c->optimistic ? 1 : 0;
    else if (!strcmp(key, "real_attempts")) *value = c->real_attempts;
    else if (!strcmp(key, "real_filtered")) *value = c->real_filtered ? 1 : 0;
    else if (!strcmp(key, "real_lds_ranked")) *value = c->real_lds_ranked;
    else if (!strcmp(key, "device_bytes")) {
        DevBuf* all[] = {&c->db, &c->dblab, &c->qc, &c->qlab, &c->hist, &c->hown, &c->posbase, &c->seglt, &c->segtie,
                         &c->t, &c->tguess, &c->sstar, &c->cnt_lt, &c->quota, &c->tie_before, &c->n_lt, &c->err,
                         &c->sl_start, &c->sl_tie, &c->sl_cnt, &c->tot, &c->failq, &c->cand, &c->out_idx, &c->out_dist,
                         &c->mbits, &c->shapes, &c->ap, &c->rel, &c->stage_in, &c->badcnt, &c->qbad, &c->flist, &c->hwq,
                         &c->dbf, &c->qf, &c->samp, &c->thr, &c->sortA, &c->sortB, &c->gtab, &c->scores, &c->dbx, &c->qx, &c->bigq, &c->dbx2, &c->qx2, &c->mbits2,
                         &c->dbfx, &c->dbfb, &c->thr2, &c->xmax2, &c->dbx8, &c->dbx3, &c->sampx, &c->ap_recip, &c->part};
        i64 total = 0;
        for (auto* d : all) if (!d->borrowed) total += (i64)d->cap;
        *value = total;
    }
#ifdef HG_RANK_PROFILE
    else if (!strcmp(key, "dbg_hwq_ptr")) *value = (int64_t)(uintptr_t)c->hwq.p;
#endif
    else if (!strcmp(key, "db_nonbinary")) *value = c->census_db[0];
    else if (!strcmp(key, "db_zeros")) *value = c->census_db[1];
    else if (!strcmp(key, "db_minus_ones")) *value = c->census_db[2];
    else if (!strcmp(key, "q_nonbinary")) *value = c->census_q[0];
    else if (!strcmp(key, "q_zeros")) *value = c->census_q[1];
    else if (!strcmp(key, "q_minus_ones")) *value = c->census_q[2];
    else if (!strcmp(key, "probe_build")) *value = kProbes ? 1 : 0;
    else if (!strcmp(key, "db_floats")) *value = c->dbf_resident ? 1 : 0;
    else if (!strcmp(key, "q_floats")) *value = c->qf_resident ? 1 : 0;
    else if (!strcmp(key, "graph_replays")) *value = c->graph_replays;
    else if (!strcmp(key, "graph_captures")) *value = c->graph_captures;
    else if (!strcmp(key, "segments")) *value = c->geo.S;
    else if (!strcmp(key, "segment_rows")) *value = c->geo.L;
    else if (!strcmp(key, "slice_capacity")) *value = c->cap;
    else if (!strcmp(key, "record_row")) *value = c->crow;
    else return fail(HG_ERR_ARG, "hg_get_stat: unknown key '%s'", key);
    return HG_OK;
}

int hg_timing_enable(hg_ctx* c, int on) {
    if (!c) return fail(HG_ERR_ARG, "hg_timing_enable: null context");
    c->timing = on < 0 ? 0 : (on > 2 ? 2 : on);
    c->t_seq = 0;
    return HG_OK;
}

int hg_timing_reset(hg_ctx* c) {
    if (!c) return fail(HG_ERR_ARG, "hg_timing_reset: null context");
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->t_collect();
    for (int i = 0; i < KI_COUNT; ++i) { c->t_ms[i] = 0; c->t_n[i] = 0; }
    return HG_OK;
}

int hg_timing_read(hg_ctx* c, int cap, const char** names, double* total_ms, int64_t* launches, int* n) {
    if (!c || !n) return fail(HG_ERR_ARG, "hg_timing_read: null argument");
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->t_collect();                                    // events recorded since the last read
    int k = 0;
    for (int i = 0; i < KI_COUNT && k < cap; ++i) {
        if (!c->t_n[i]) continue;
        if (names) names[k] = kKernelNames[i];
        if (total_ms) total_ms[k] = c->t_ms[i];
        if (launches) launches[k] = c->t_n[i];
        ++k;
    }
    *n = k;
    return HG_OK;
}

}  // extern "C"
