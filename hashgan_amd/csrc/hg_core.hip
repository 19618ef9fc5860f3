// libhashgan_amd.so -- context, tables, options / statistics / timing, label match, AP and the downloads (see hg_ctx.hpp
// for the map of the translation units).
#include "hg_ctx.hpp"
#include "hg_host_pack.hpp"

#include <exception>
#include <map>
#include <mutex>
#include <thread>
#include <unistd.h>

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

const char* const kKernelNames[KI_COUNT] = {"k_hist", "k_hist_reduce", "k_plan", "k_seg_counts", "k_seg_layout", "k_guess",
                                            "k_select", "k_rank_hist", "k_order", "k_rank_fused", "k_match", "k_ap", "k_merge", "k_pack",
                                            "k_real_sample", "k_real_guess", "k_real_select", "k_radix_pass", "k_real_finish", "k_select_mx", "k_rank_cnt", "rccl_allgather", "step_gpu_span", "k_real_rescore"};
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



std::atomic<unsigned long long> g_alloc_epoch{1};
const char* const kHostPhaseNames[HP_COUNT] = {"init", "devmalloc", "devfree", "hostmalloc", "hostfree", "destroy", "stream", "event", "sync", "pack", "thread"};
std::atomic<long long> g_host_ns[HP_COUNT], g_host_calls[HP_COUNT], g_host_max_ns[HP_COUNT];

// ---- the process-wide cache of device blocks, pinned blocks and streams (hg_ctx.hpp) ----------------------------------
namespace {
struct BlockCache {
    std::mutex m;
    std::multimap<std::pair<int, size_t>, void*> blocks;   // (device, bytes) -> block; pinned blocks: device -1
    size_t dev_bytes = 0, pin_bytes = 0;
    std::multimap<int, hipStream_t> streams;
    long long hits = 0, misses = 0;
    static size_t limit(const char* env, size_t dflt_mb) {
        const char* v = getenv(env);
        return (size_t)(v ? atoll(v) : (long long)dflt_mb) << 20;
    }
    void* take(int device, size_t want, size_t* got) {
        std::lock_guard<std::mutex> lk(m);
        static const bool trace = getenv("HG_CACHE_TRACE") != nullptr;
        auto it = blocks.lower_bound({device, want});
        if (it == blocks.end() || it->first.first != device || it->first.second > 2 * want + ((size_t)1 << 20)) {
            ++misses;
            if (trace) fprintf(stderr, "[hg cache] miss: %s block of %zu bytes (cached: %zu device, %zu pinned bytes in %zu blocks)\n",
                               device < 0 ? "pinned" : "device", want, dev_bytes, pin_bytes, blocks.size());
            return nullptr;
        }
        void* p = it->second;
        *got = it->first.second;
        (device < 0 ? pin_bytes : dev_bytes) -= *got;
        blocks.erase(it);
        ++hits;
        return p;
    }
    static void free_block(int device, void* p) {
        if (device < 0) { HostTimer t_(HP_HOSTFREE); (void)hipHostFree(p); }
        else { HostTimer t_(HP_DEVFREE); (void)hipFree(p); }
    }
    void give(int device, void* p, size_t bytes) {
        static const size_t dev_limit = limit("HG_CACHE_MB", 49152), pin_limit = limit("HG_PIN_CACHE_MB", 512);
        const size_t lim = device < 0 ? pin_limit : dev_limit;
        // (hipFree waits for the device; a cached block must be just as free of readers and writers before anybody else takes it)
        (void)hipDeviceSynchronize();
        if (bytes > lim) { free_block(device, p); return; }
        std::vector<std::pair<int, void*>> evict;
        {
            std::lock_guard<std::mutex> lk(m);
            blocks.insert({{device, bytes}, p});
            size_t& total = device < 0 ? pin_bytes : dev_bytes;
            total += bytes;
            while (total > lim) {                              // the largest block of this kind goes first
                auto hi = blocks.upper_bound({device, ~(size_t)0});
                if (hi == blocks.begin()) break;
                --hi;
                if ((hi->first.first < 0) != (device < 0)) break;
                total -= hi->first.second;
                evict.push_back({hi->first.first, hi->second});
                blocks.erase(hi);
            }
        }
        for (auto& e : evict) free_block(e.first, e.second);
    }
    void release_all() {
        std::vector<std::pair<int, void*>> all;
        std::vector<std::pair<int, hipStream_t>> st;
        {
            std::lock_guard<std::mutex> lk(m);
            for (auto& b : blocks) all.push_back({b.first.first, b.second});
            blocks.clear();
            dev_bytes = pin_bytes = 0;
            for (auto& x : streams) st.push_back({x.first, x.second});
            streams.clear();
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto& b : all) { if (b.first >= 0) (void)hipSetDevice(b.first); free_block(b.first, b.second); }
        for (auto& x : st) { (void)hipSetDevice(x.first); (void)hipStreamDestroy(x.second); }
        (void)hipSetDevice(cur);
    }
};
BlockCache& cache() { static BlockCache* c = new BlockCache(); return *c; }      // (never destroyed: contexts may outlive static destructors)
}  // namespace

void* cache_take_dev(int device, size_t want, size_t* got) { return cache().take(device, want, got); }
void cache_give_dev(int device, void* p, size_t bytes) { cache().give(device, p, bytes); }
void* cache_take_pin(size_t want, size_t* got) { return cache().take(-1, want, got); }
void cache_give_pin(void* p, size_t bytes) { cache().give(-1, p, bytes); }
hipStream_t cache_take_stream(int device) {
    BlockCache& c = cache();
    std::lock_guard<std::mutex> lk(c.m);
    auto it = c.streams.find(device);
    if (it == c.streams.end()) return nullptr;
    hipStream_t s = it->second;
    c.streams.erase(it);
    return s;
}
void cache_give_stream(int device, hipStream_t s) {
    BlockCache& c = cache();
    {
        std::lock_guard<std::mutex> lk(c.m);
        if (c.streams.count(device) < 8) { c.streams.insert({device, s}); return; }
    }
    (void)hipStreamDestroy(s);
}
void cache_release_all() { cache().release_all(); }
static long long cache_stat(int which) {
    BlockCache& c = cache();
    std::lock_guard<std::mutex> lk(c.m);
    return which == 0 ? (long long)c.dev_bytes : which == 1 ? (long long)c.pin_bytes : which == 2 ? c.hits : which == 3 ? c.misses : (long long)c.streams.size();
}
static int stream_create(int device, hipStream_t* out) {
    *out = cache_take_stream(device);
    if (*out) return HG_OK;
    HG_HIP(host_timed(HP_STREAM, [&] { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); }));
    return HG_OK;
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


int launch_min_topr(hg_ctx* c, const u32* idx_all, const u8* dist_all, i64 n, int G) {
    c->t_begin(KI_MERGE);
    hipLaunchKernelGGL(k_min_topr, dim3(grid_for(n)), dim3(256), 0, c->stream, idx_all, dist_all, c->out_idx.as<u32>(), c->out_dist.as<u8>(), n, G);
    c->t_end();
    return c->check_launch("k_min_topr");
}

// The plan kernel flags R > (rows in the histograms it saw); read it back with the results.
int read_plan_flag(hg_ctx* c, int* flag) {
    HG_HIP(hipMemcpyAsync(flag, c->err.p, 4, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

// The verdict words, the APs and the hit counts of a one-shot call live side by side in one device block -- in the layout of the
// pinned staging block -- so that they come home in ONE copy (three blit kernels and their gaps cost a short step 10-15 us: C3 0.214
// -> 0.20 ms).  err, ap and rel become views; whoever reserves more than a view holds later simply gets a buffer of its own again.
int ensure_out_block(hg_ctx* c) {
    const size_t Q = (size_t)c->Q;
    char* base = (char*)c->outblk.p;
    if (base && c->outblk_q == (i64)Q && c->err.p == base && c->ap.p == base + 16 && c->rel.p == base + 16 + Q * 8) return HG_OK;
    HG_TRY(c->sync());                                 // (nothing may still be writing the old buffers)
    HG_TRY(c->outblk.reserve(16 + Q * 12 + 64));
    HG_HIP(hipMemsetAsync(c->outblk.p, 0, 16, c->stream));
    c->err.view(c->outblk, 0, 16);
    c->ap.view(c->outblk, 16, Q * 8);
    c->rel.view(c->outblk, 16 + Q * 8, Q * 4);
    c->outblk_q = (i64)Q;
    return HG_OK;
}

int ensure_pin(hg_ctx* c, size_t need_b) {
    if (c->pin_cap >= need_b) return HG_OK;
    pin_free(c->pin, c->pin_cap);
    c->pin = nullptr; c->pin_cap = 0;
    HG_HIP(pin_alloc(&c->pin, need_b, &c->pin_cap));
    ++g_alloc_epoch;                                   // captured downloads point into the old block
    return HG_OK;
}


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
    HostTimer t_init(HP_INIT);
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
    const int rc_s = stream_create(device, &c->stream);
    if (rc_s != HG_OK) { delete c; return rc_s; }
    *out = c;
    return HG_OK;
}

int hg_destroy(hg_ctx* c) {
    if (!c) return HG_OK;
    HostTimer t_destroy(HP_DESTROY);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->t_collect();
    c->drop_graph();
    for (auto e : c->pool) (void)hipEventDestroy(e);
    DevBuf* all[] = {&c->db, &c->dblab, &c->qc, &c->qlab, &c->hist, &c->hown, &c->posbase, &c->seglt, &c->segtie,
                     &c->t, &c->tguess, &c->sstar, &c->cnt_lt, &c->quota, &c->tie_before, &c->n_lt, &c->err, &c->sl_start,
                     &c->sl_tie, &c->sl_cnt, &c->tot, &c->failq, &c->cand, &c->out_idx, &c->out_dist, &c->mbits,
                     &c->shapes, &c->ap, &c->rel, &c->stage_in, &c->badcnt, &c->qbad, &c->flist, &c->hwq, &c->dbf, &c->qf, &c->samp, &c->thr,
                     &c->sortA, &c->sortB, &c->gtab, &c->scores, &c->dbx, &c->qx, &c->bigq, &c->mbits2, &c->dbfx, &c->dbfb, &c->thr2, &c->xmax2, &c->dbx8, &c->dbx3, &c->dbx4, &c->sampx, &c->ap_recip, &c->part, &c->dbytes, &c->outblk, &c->beyond, &c->cntq, &c->hist2, &c->krows};
    for (auto* d : all) d->release();
    for (auto& d : c->gathered) d.release();
    for (auto& d : c->scratch) d.release();
    c->comm_tmp.release(); c->gath_idx.release(); c->gath_dist.release();
    c->obuf[0].release(); c->obuf[1].release();
    comm_release(c);
    if (c->sub) { hg_ctx* s = c->sub; c->sub = nullptr; (void)hg_destroy(s); }
    pin_free(c->pin, c->pin_cap);
    for (auto& m : c->mslot) { pin_free(m.pin, m.cap); if (m.ev) (void)hipEventDestroy(m.ev); m.pin = nullptr; m.ev = nullptr; }
    pin_free(c->hpk, c->hpk_cap);
    pin_free(c->fstage, c->fstage_cap);
    for (auto& q : c->qstage) { pin_free(q.pin, q.cap); if (q.ev) (void)hipEventDestroy(q.ev); q.pin = nullptr; q.ev = nullptr; }
    if (c->stream2_ev) (void)hipEventDestroy(c->stream2_ev);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); cache_give_stream(c->device, c->stream2); }
    for (auto& e : c->fstage_ev) if (e) (void)hipEventDestroy(e);
    if (c->stream && c->own_stream && !c->is_sub) cache_give_stream(c->device, c->stream);   // (synchronised above)
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
    c->dbx3_valid = false;
    c->dbx4_valid = false;
    c->dbx8_valid = false;
    c->opt_consecutive_fail = c->shard_bet_fail = 0;    // a new database: earlier lost bets say nothing about it
    c->cap_boost = c->real_cap_boost = 1;
    c->crowd_probed = false;
    c->cfg_epoch++;
    c->db_gen++;
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
static int ensure_fstage(hg_ctx* c) {
    constexpr int NSL = 4;
    const size_t CH = (size_t)16 << 20;
    if (!c->fstage) {
        HG_HIP(pin_alloc(&c->fstage, CH * NSL, &c->fstage_cap));
        for (int k = 0; k < NSL; ++k) HG_HIP(hipEventCreateWithFlags(&c->fstage_ev[k], hipEventDisableTiming));
    }
    return HG_OK;
}
static int ensure_stream2(hg_ctx* c) {
    if (!c->stream2) HG_TRY(stream_create(c->device, &c->stream2));
    if (!c->stream2_ev) HG_HIP(hipEventCreateWithFlags(&c->stream2_ev, hipEventDisableTiming));
    return HG_OK;
}
static int stage_floats(hg_ctx* c, const float* x, i64 n, int b, int bpad, DevBuf& feats, hipStream_t stream, int which_pool) {
    constexpr int NSL = 4;
    const size_t CH = (size_t)16 << 20;
    HG_TRY(ensure_fstage(c));
    const i64 rows_per = (i64)(CH / ((size_t)bpad * 4));
    int slot = 0;
    bool used[NSL] = {false, false, false, false};
    try {
        for (i64 r0 = 0; r0 < n; r0 += rows_per, slot = (slot + 1) % NSL) {
            const i64 r1 = r0 + rows_per < n ? r0 + rows_per : n;
            if (used[slot]) HG_HIP(hipEventSynchronize(c->fstage_ev[slot]));
            float* st = (float*)((char*)c->fstage + (size_t)slot * CH);
            host_copy_rows(x, r0, r1, b, bpad, st, 0, which_pool);
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
        pin_free(c->hpk, c->hpk_cap);
        c->hpk = nullptr; c->hpk_cap = 0;
        HG_HIP(pin_alloc(&c->hpk, need_b, &c->hpk_cap));
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
        HG_TRY(ensure_stream2(c));
        HG_TRY(feats.reserve(fb_f + 256));
        try {
            HostTimer t_thr(HP_THREAD);                  // ("thread": starting the staging thread)
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
        HostTimer t_pack(HP_PACK);                       // ("pack": the packing pool's pass over the arrays, prefixes shipped meanwhile)
        host_pack_ship(x, lab, n, b, C, hc, hl, &cs, 0,
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
    c->dbx3_valid = false;
    c->dbx4_valid = false;
    c->dbx8_valid = false;
    c->dbfx_valid = false;
    c->dbfb_valid = false;
    c->opt_consecutive_fail = c->shard_bet_fail = 0;
    c->cap_boost = c->real_cap_boost = 1;
    c->crowd_probed = false;
    c->cfg_epoch++;
    c->db_gen++;
    return HG_OK;
}

// hg_set_queries: dst_a[i] = src_a[(i / dw) * sw + i % dw] (packed code rows: the unused upper half of an odd row's last 64-bit word is dropped),
// dst_b[j] = src_b[j] (label words) -- sources in pinned host memory
static __global__ __launch_bounds__(256) void k_copy_in(const u32* __restrict__ src_a, u32* __restrict__ dst_a, const i64 na, const int sw, const int dw,
                                                        const u32* __restrict__ src_b, u32* __restrict__ dst_b, const i64 nb) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < na) { const i64 r = i / dw; dst_a[i] = src_a[r * sw + (i - r * dw)]; }
    else if (i - na < nb) dst_b[i - na] = src_b[i - na];
}

// A new query table of the SAME size on an unchanged database and configuration keeps hg_map_begin's licence to enqueue blind:
// every buffer of the step is sized by Q, nothing about the bet depends on what the queries are (its verdict is checked on the
// GPU either way), so a caller that hands over batch after batch keeps two steps in flight.
static void queries_replaced(hg_ctx* c, i64 old_q, bool had_q) {
    const bool carry = had_q && old_q == c->Q && c->map_warm_R >= 0 && c->map_warm_cfg == c->cfg_epoch;
    c->cfg_epoch++;
    c->q_gen++;
    if (carry) c->map_warm_cfg = c->cfg_epoch;
}

int hg_set_queries_f32(hg_ctx* c, const float* host_x, const int64_t* host_labels, int64_t Q, int64_t* bad_codes,
                       int64_t* bad_labels) {
    HG_TRY(need(c, ST_DB, "hg_set_queries_f32", "hg_set_database"));
    if (Q < 1 || !host_x || !host_labels) return fail(HG_ERR_ARG, "hg_set_queries_f32: need Q >= 1 and data");
    if (Q > 0x7FFFFFC0ll) return fail(HG_ERR_ARG, "hg_set_queries_f32: Q too large");
    const i64 old_q = c->Q;
    const bool had_q = (c->stage & ST_Q) != 0;
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
    queries_replaced(c, old_q, had_q);
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
    const i64 old_q = c->Q;
    const bool had_q = (c->stage & ST_Q) != 0;
    c->Q = Q;
    // through a pinned block of the context's own (two, alternating), copied from there by the stream: the call returns without
    // waiting for whatever the stream still holds -- a step of hg_map_begin on the previous batch -- and the caller's arrays
    // are free the moment it does
    const int W = (c->b + 63) / 64;
    const size_t cb = (size_t)Q * W * 8, lb = (size_t)Q * c->LW * 8;
    hg_ctx::QStage& qs = c->qstage[c->qstage_next];
    c->qstage_next ^= 1;
    if (qs.used) HG_HIP(hipEventSynchronize(qs.ev));   // (the copy out of this block two loads ago)
    if (qs.cap < cb + lb) {
        pin_free(qs.pin, qs.cap);
        qs.pin = nullptr; qs.cap = 0;
        HG_HIP(pin_alloc(&qs.pin, cb + lb, &qs.cap));
    }
    if (!qs.ev) HG_HIP(hipEventCreateWithFlags(&qs.ev, hipEventDisableTiming));
    memcpy(qs.pin, codes, cb);
    memcpy((char*)qs.pin + cb, labels, lb);
    HG_TRY(c->qc.reserve((size_t)Q * c->NW * 4 + 64 * 4));
    HG_TRY(c->qlab.reserve(lb));
    // fetched by a KERNEL out of the pinned block (it is device-addressable: loads over the link), like k_copy_out the other way: a
    // copy-engine transfer of these 160 KB sits ~20 us on the stream behind a cross-queue barrier -- per step, for a caller that hands
    // over a new batch per step (a small table only: a big one is the copy engines' work)
    if (cb + lb <= ((size_t)4 << 20)) {
        const i64 nc = (i64)Q * c->NW, nl = (i64)Q * c->LW * 2;
        hipLaunchKernelGGL(k_copy_in, dim3(grid_for(nc + nl)), dim3(256), 0, c->stream, (const u32*)qs.pin, c->qc.as<u32>(), nc, 2 * W, c->NW,
                           (const u32*)((const char*)qs.pin + cb), c->qlab.as<u32>(), nl);
        HG_TRY(c->check_launch("k_copy_in"));
    } else {
        HG_TRY(upload_codes(c, c->qc, (const uint64_t*)qs.pin, Q, W, c->NW));
        HG_HIP(hipMemcpyAsync(c->qlab.p, (const char*)qs.pin + cb, lb, hipMemcpyHostToDevice, c->stream));
    }
    HG_HIP(hipEventRecord(qs.ev, c->stream));
    qs.used = true;
    c->stage = ST_DB | ST_Q;
    c->qx_valid = false;
    c->qf_resident = false;                            // packed input: no float table
    queries_replaced(c, old_q, had_q);
    return HG_OK;
}

extern "C++" int do_match(hg_ctx* c) {
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

// k_ap's tables for the current R: the flattened summation trees and the reciprocals of the ranks (also read by
// k_rank_cnt's AP epilogue); *use_recip: lists beyond 2^20 divide
extern "C++" int ensure_ap_tables(hg_ctx* c, bool* use_recip_out) {
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
        // (AP_RECIP_SLACK more: ap_eval2's masked slots past the last rank still load a -- finite -- entry)
        HG_TRY(c->ap_recip.reserve((size_t)(g.R + 1 + AP_RECIP_SLACK) * 8));
        hipLaunchKernelGGL(k_recip_table, dim3(grid_for(g.R + 1 + AP_RECIP_SLACK)), dim3(256), 0, c->stream, c->ap_recip.as<double>(), (i64)g.R + AP_RECIP_SLACK);
        HG_TRY(c->check_launch("k_recip_table"));
        c->recip_for_R = g.R;
    }
    HG_TRY(c->ap.reserve((size_t)g.Q * 8));
    HG_TRY(c->rel.reserve((size_t)g.Q * 4));
    *use_recip_out = use_recip;
    return HG_OK;
}

// only: optional device flags [Q] -- evaluate just the flagged queries
extern "C++" int do_ap_range(hg_ctx* c, i64 q0, i64 nq, const u32* only) {      // k_ap's block index is the query: a range is a pointer offset
    c->ap_staged = false;
    const Geo& g = c->geo;
    bool use_recip = false;
    HG_TRY(ensure_ap_tables(c, &use_recip));
    c->t_begin(KI_AP);
    // few queries with long lists: four times the threads per query (see k_ap)
    const bool wide = nq * 2 < (i64)c->n_cu * 8 && g.R > 2 * AP_CHUNK && c->opt_ap_wide;
    if (nq > 0 && wide)
        hipLaunchKernelGGL(k_ap<512>, dim3((unsigned)nq), dim3(512), 0, c->stream, c->mbits.as<u64>() + (size_t)q0 * c->RW, c->RW, g.R,
                           c->shapes.as<ApShape>(), use_recip ? c->ap_recip.as<double>() : (const double*)nullptr,
                           c->ap.as<double>() + q0, c->rel.as<u32>() + q0, only ? only + q0 : (const u32*)nullptr);
    else if (nq > 0)
        hipLaunchKernelGGL(k_ap<AP_THREADS>, dim3((unsigned)nq), dim3(AP_THREADS), 0, c->stream, c->mbits.as<u64>() + (size_t)q0 * c->RW, c->RW, g.R,
                           c->shapes.as<ApShape>(), use_recip ? c->ap_recip.as<double>() : (const double*)nullptr,
                           c->ap.as<double>() + q0, c->rel.as<u32>() + q0, only ? only + q0 : (const u32*)nullptr);
    c->t_end();
    HG_TRY(c->check_launch("k_ap"));
    c->stage |= ST_AP;
    return HG_OK;
}

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

int hg_get_topr_real(hg_ctx* c, uint32_t* host_idx, float* host_scores) {
    HG_TRY(need(c, ST_SELECT, "hg_get_topr_real", "hg_topr_real / hg_map_real"));
    if (!c->real_lists) return fail(HG_ERR_STATE, "hg_get_topr_real: no real-valued ranked lists (the last ranking was not real-valued, or it was an hg_map_real, "
                                                  "which skips them: use hg_topr_real, or option real_map_lists = 1)");
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
    if (c->stream && c->own_stream) cache_give_stream(c->device, c->stream);
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        HG_TRY(stream_create(c->device, &c->stream));
        c->own_stream = true;
    }
    if (c->sub) c->sub->stream = c->stream;
    return HG_OK;
}

int hg_set_option(hg_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return fail(HG_ERR_ARG, "hg_set_option: null argument");
    if (!strcmp(key, "handicap_next_bet")) {           // test hook: not a configuration change (a blind hg_map_begin stays blind -- and loses)
        if (value < 0 || value > 64) return fail(HG_ERR_ARG, "handicap_next_bet must be 0..64");
        c->handicap_next = value;
        return HG_OK;
    }
    c->cfg_epoch++;                                    // whatever changes: a captured step is rebuilt
    if (!strcmp(key, "step_graph")) { c->opt_graph = value != 0; return HG_OK; }
    if (!strcmp(key, "stage_sync")) { c->stage_sync = value != 0; return HG_OK; }
    if (!strcmp(key, "defer_verdict")) { c->defer_verdict = value != 0; return HG_OK; }
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
    } else if (!strcmp(key, "all_rows_shortcut")) {
        c->opt_all_rows = value != 0;
    } else if (!strcmp(key, "timing_every")) {
        if (value < 1 || value > 1024) return fail(HG_ERR_ARG, "timing_every must be 1..1024");
        c->opt_timing_every = value;
    } else if (!strcmp(key, "cap_boost")) {
        if (value < 1 || value > 4096) return fail(HG_ERR_ARG, "cap_boost must be 1..4096");
        // a caller that WIDENS the slices is retrying a lost sharded bet within the same call (sharded.evaluate_shard): that attempt
        // does not count towards hg_bet_eligible's "two calls in a row lost their bets"
        if (value > c->cap_boost && c->shard_bet_fail > 0) c->shard_bet_fail--;
        c->cap_boost = value;
    } else if (!strcmp(key, "crowd_probe")) {
        c->opt_crowd_probe = value != 0;
    } else if (!strcmp(key, "fuse_ap")) {
        c->opt_fuse_ap = value != 0;
    } else if (!strcmp(key, "rank_dense")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "rank_dense must be 0 or 1 (2 is accepted and means 1)");
        c->opt_rank_dense = value;
    } else if (!strcmp(key, "inline_leftovers")) {
        c->opt_inline_leftovers = value != 0;
    } else if (!strcmp(key, "rank_slices")) {
        if (value < 0) return fail(HG_ERR_ARG, "rank_slices must be >= 0 (the smallest R it takes; 0: off)");
        c->opt_rank_slices = value;
    } else if (!strcmp(key, "rank_dense_gbm")) {
        if (value < -1 || value > 1) return fail(HG_ERR_ARG, "rank_dense_gbm must be -1, 0 or 1");
        c->opt_rank_dense_gbm = value;
    } else if (!strcmp(key, "dense_budget_mb")) {
        if (value < 1) return fail(HG_ERR_ARG, "dense_budget_mb must be >= 1");
        c->opt_dense_budget_mb = value;
    } else if (!strcmp(key, "select_packed")) {
        c->opt_select_packed = value;
    } else if (!strcmp(key, "rank_lds")) {
        // the bet's rank stage with a query's records resident in LDS: 2 = k_rank_lean where it applies, else k_rank_cnt (default);
        // 1 = k_rank_cnt only; 0 = neither: k_rank_fused walks the records in global memory
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "rank_lds must be 0, 1 or 2");
        c->opt_rank_cnt = value >= 1;
        c->opt_rank_lean = value >= 2;
    } else if (!strcmp(key, "host_pack")) {
        c->opt_host_pack = value != 0;
    } else if (!strcmp(key, "keep_floats")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "keep_floats must be 0, 1 or 2");
        c->opt_keep_floats = value;
    } else if (!strcmp(key, "compact_records")) {
        c->opt_compact = value != 0;
    } else if (!strcmp(key, "second_bet")) {
        c->opt_second_bet = value != 0;
    } else if (!strcmp(key, "hist_mfma")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "hist_mfma must be 0, 1 or 2");
        c->opt_hist_mfma = value;
    } else if (!strcmp(key, "ap_recip")) {
        c->opt_ap_recip = value != 0;
    } else if (!strcmp(key, "select_mfma")) {
        c->opt_select_mfma = value != 0;
#if HG_PROBES
    } else if (!strcmp(key, "probe_select")) {         // (libhashgan_amd_probe.so only: python -m hashgan_amd.build --probes)
        c->opt_probe = value;
#endif
    } else if (!strcmp(key, "real_mfma")) {
        if (value < 0 || value > 2) return fail(HG_ERR_ARG, "real_mfma must be 0, 1 or 2");
        c->opt_real_mfma = value;
    } else if (!strcmp(key, "real_sample_half")) {
        c->opt_real_sample_h = value != 0;
    } else if (!strcmp(key, "real_second_sample")) {
        c->opt_real_second = value != 0;
    } else if (!strcmp(key, "ap_wide")) {
        c->opt_ap_wide = value != 0;
    } else if (!strcmp(key, "real_map_lists")) {
        c->opt_real_map_lists = value != 0;
    } else if (!strcmp(key, "real_whole_rounds")) {
        if (value < 0 || value > 8) return fail(HG_ERR_ARG, "real_whole_rounds must be 0 .. 8");
        c->opt_real_rounds = value;
    } else if (!strcmp(key, "real_groups")) {
        c->opt_real_groups = value != 0;
    } else if (!strcmp(key, "real_sort_lds")) {
        c->opt_real_sort_lds = value != 0;
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
                      &c->dbx, &c->qx, &c->dbfx, &c->dbfb, &c->sampx, &c->dbytes};   // the images are rebuilt on demand
    for (auto* d : work) d->release();
    c->dbfx_valid = false;
    c->dbfb_valid = false;
    for (auto& d : c->gathered) d.release();
    for (auto& d : c->scratch) d.release();
    c->gath_idx.release(); c->gath_dist.release();
    c->obuf[0].release(); c->obuf[1].release();
    c->dbx_valid = c->qx_valid = false;
    c->dbx8.release(); c->dbx8_valid = false;
    c->dbx3.release(); c->dbx3_valid = false;
    c->dbx4.release(); c->dbx4_valid = false;
    if (c->sub) { hg_ctx* s = c->sub; c->sub = nullptr; (void)hg_destroy(s); }
    c->stage &= (ST_DB | ST_Q);
    c->lists_valid = false;
    c->real_lists = false;
    cache_release_all();                               // (a caller that trims wants the memory back at the RUNTIME -- another framework in the process -- not in this library's cache)
    return HG_OK;
}

int hg_preload(hg_ctx* c) {
    if (!c) return fail(HG_ERR_ARG, "hg_preload: null context");
    HG_TRY(c->use());
    HG_TRY(ensure_stream2(c));
    HG_TRY(ensure_fstage(c));
    HG_TRY(ensure_pin(c, (size_t)1 << 20));
    hipFuncAttributes a;
    HG_HIP(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ap<AP_THREADS>)));
    HG_TRY(preload_seq()); HG_TRY(preload_valu()); HG_TRY(preload_mx()); HG_TRY(preload_mx1()); HG_TRY(preload_real());
    host_pack_warm();
    return HG_OK;
}

int hg_release_cache(void) {
    cache_release_all();
    return HG_OK;
}

int hg_get_stat(hg_ctx* c, const char* key, int64_t* value) {
    if (!c || !key || !value) return fail(HG_ERR_ARG, "hg_get_stat: null argument");
    if (!strcmp(key, "optimistic_runs")) *value = c->opt_runs;
    else if (!strcmp(key, "optimistic_fallbacks")) *value = c->opt_fallbacks;
    else if (!strcmp(key, "optimistic_requeried")) *value = c->opt_requeried;
    else if (!strcmp(key, "optimistic_rebets")) *value = c->opt_rebets;
    else if (!strcmp(key, "rank_leftovers")) *value = c->opt_leftover;
    else if (!strcmp(key, "select_variant")) *value = c->last_select;
    else if (!strcmp(key, "rank_variant")) *value = c->last_rank;
    else if (!strcmp(key, "ap_fused")) *value = c->ap_fused ? 1 : 0;
    else if (!strcmp(key, "cap_boost")) *value = c->cap_boost;
    else if (!strcmp(key, "crowding_x100")) *value = c->crowd_x100;
    else if (!strcmp(key, "real_cap_boost")) *value = c->real_cap_boost;
    else if (!strcmp(key, "last_optimistic")) *value = c->optimistic ? 1 : 0;
    else if (!strcmp(key, "real_attempts")) *value = c->real_attempts;
    else if (!strcmp(key, "real_requeried")) *value = c->real_requeried;
    // how the last real-valued ranking ran: bit 0 = bf16 filter + exact rescoring, bit 1 = ranked by the LDS-resident kernel,
    // bit 2 = record lists beyond the LDS ordered group by group (k_real_group_*), bit 3 = the filter ran in IEEE half (else bfloat16)
    else if (!strcmp(key, "real_path")) *value = (c->real_filtered ? 1 : 0) | (c->real_lds_ranked ? 2 : 0) | (c->real_grouped ? 4 : 0) | (c->real_filtered && c->dbfb_half ? 8 : 0);
    else if (!strcmp(key, "device_bytes")) {
        DevBuf* all[] = {&c->db, &c->dblab, &c->qc, &c->qlab, &c->hist, &c->hown, &c->posbase, &c->seglt, &c->segtie,
                         &c->t, &c->tguess, &c->sstar, &c->cnt_lt, &c->quota, &c->tie_before, &c->n_lt, &c->err,
                         &c->sl_start, &c->sl_tie, &c->sl_cnt, &c->tot, &c->failq, &c->cand, &c->out_idx, &c->out_dist,
                         &c->mbits, &c->shapes, &c->ap, &c->rel, &c->stage_in, &c->badcnt, &c->qbad, &c->flist, &c->hwq,
                         &c->dbf, &c->qf, &c->samp, &c->thr, &c->sortA, &c->sortB, &c->gtab, &c->scores, &c->dbx, &c->qx, &c->bigq, &c->mbits2,
                         &c->dbfx, &c->dbfb, &c->thr2, &c->xmax2, &c->dbx8, &c->dbx3, &c->dbx4, &c->sampx, &c->ap_recip, &c->part, &c->dbytes, &c->outblk, &c->beyond, &c->cntq, &c->hist2, &c->krows,
                         &c->comm_tmp, &c->gath_idx, &c->gath_dist, &c->obuf[0], &c->obuf[1]};
        i64 total = 0;
        for (auto* d : all) if (!d->borrowed) total += (i64)d->cap;
        for (auto& d : c->gathered) if (!d.borrowed) total += (i64)d.cap;
        for (auto& d : c->scratch) if (!d.borrowed) total += (i64)d.cap;
        *value = total;
    }
#ifdef HG_RANK_PROFILE
#ifdef HG_RANK_PROFILE                        // (tools/rank_phase_profile.py: where k_rank_cnt left its phase timestamps)
    else if (!strcmp(key, "dbg_hwq_ptr")) *value = (int64_t)(uintptr_t)c->hwq.p;
#endif
#endif
    else if (!strcmp(key, "cache_device_bytes")) *value = cache_stat(0);      // the process-wide block cache (hg_ctx.hpp)
    else if (!strcmp(key, "cache_pinned_bytes")) *value = cache_stat(1);
    else if (!strcmp(key, "cache_hits")) *value = cache_stat(2);
    else if (!strcmp(key, "cache_misses")) *value = cache_stat(3);
    else if (!strcmp(key, "cache_streams")) *value = cache_stat(4);
    else if (!strncmp(key, "host_", 5)) {              // process-wide host-side phase timers (hg_ctx.hpp, HostPhase)
        const char* rest = key + 5;
        int kind = -1;                                 // 0 total us, 1 calls, 2 longest call us
        if (!strncmp(rest, "us_", 3)) { kind = 0; rest += 3; }
        else if (!strncmp(rest, "n_", 2)) { kind = 1; rest += 2; }
        else if (!strncmp(rest, "max_us_", 7)) { kind = 2; rest += 7; }
        int ph = -1;
        for (int i = 0; i < HP_COUNT; ++i) if (!strcmp(rest, kHostPhaseNames[i])) ph = i;
        if (kind < 0 || ph < 0) return fail(HG_ERR_ARG, "hg_get_stat: unknown key '%s'", key);
        *value = kind == 0 ? g_host_ns[ph].load() / 1000 : kind == 1 ? g_host_calls[ph].load() : g_host_max_ns[ph].load() / 1000;
    }
    else if (!strcmp(key, "graph_replays")) *value = c->graph_replays;
    else if (!strcmp(key, "cut_beyond_planes")) {      // a download: asked for after a lost owner-routed bet only
        u32 v = 0;
        if (c->beyond.p) {
            HG_HIP(hipMemcpyAsync(&v, c->beyond.p, 4, hipMemcpyDeviceToHost, c->stream));
            HG_TRY(c->sync());
        }
        *value = v;
    }
    else if (!strcmp(key, "map_async_steps")) *value = c->map_async_steps;
    else if (!strcmp(key, "map_async_redone")) *value = c->map_async_redone;
    else if (!strcmp(key, "segments")) *value = c->geo.S;
    else if (!strcmp(key, "records_kept")) {
        // records the last bet's select pass left in the slices, over all live queries (a download of the slice counts: a
        // measurement read after the step, never part of one) -- kept / (Q R) is what the guess's margin costs
        const Geo& g = c->geo;
        *value = -1;
        if (c->optimistic && c->sl_cnt.p && c->sl_cnt.cap >= (size_t)g.S * g.Qpad * 4) {
            HG_TRY(c->use());
            HG_TRY(c->sync());
            std::vector<u32> h((size_t)g.S * g.Qpad);
            HG_HIP(hipMemcpy(h.data(), c->sl_cnt.p, h.size() * 4, hipMemcpyDeviceToHost));
            i64 total = 0;
            for (int s = 0; s < g.S; ++s) for (int q = 0; q < g.Q; ++q) total += h[(size_t)s * g.Qpad + q];
            *value = total;
        }
    }
    else return fail(HG_ERR_ARG, "hg_get_stat: unknown key '%s'", key);
    return HG_OK;
}

// What hg_set_database_f32 / hg_set_queries_f32 saw in the float table they were handed: entries outside {-1, 0, +1}, zeros, minus
// ones, and whether the floats are resident on the device -- from which the caller tells +-1 codes (Hamming ranking), {0,1} bits
// and real-valued features (inner-product ranking) apart.
int hg_get_census(hg_ctx* c, int queries, int64_t out[4]) {
    if (!c || !out) return fail(HG_ERR_ARG, "hg_get_census: null argument");
    const i64* cs = queries ? c->census_q : c->census_db;
    out[0] = cs[0]; out[1] = cs[1]; out[2] = cs[2];
    out[3] = (queries ? c->qf_resident : c->dbf_resident) ? 1 : 0;
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
