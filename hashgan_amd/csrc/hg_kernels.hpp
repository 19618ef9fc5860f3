// Device code of the retrieval-evaluation path for gfx950 (MI355X, CDNA4).
//
// Work decomposition of the two pair passes (hist, select):
//   lane   <-> one query          (64 queries per wavefront, codes in VGPRs)
//   wave   <-> one unit = (segment s of the database shard, query tile qt)
//   loop   <-> database rows of the segment, WAVE-UNIFORM: the row's code words
//              come in through scalar loads (s_load_dwordx16) and feed
//              v_xor_b32 / v_bcnt_u32_b32 as SGPR operands.
// So a (query, row) pair costs 2*NW VALU ops for the distance (NW = 32-bit words
// per code) and no cross-lane traffic at all; rows are visited in index order
// by every lane, which is what makes the canonical order (distance asc, index
// asc) fall out of plain per-lane counters.
//
// Replaces lib/metric.py:13-23 of the reference (np.dot -> np.argsort -> label
// match -> AP) for binary codes; see DESIGN.md for the full mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hg {

typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned char u8;
typedef long long i64;

constexpr int WPB = 4;                 // wavefronts per 256-thread block (each works on its own unit / query)
constexpr u32 IDX_NONE = 0xFFFFFFFFu;  // slot of the ranked list owned by another shard
constexpr int AP_CHUNK = 8192;         // NumPy's reduction buffer (elements) -- np.sum order
constexpr int AP_LEAF = 128;           // NumPy's pairwise-sum block
constexpr int AP_THREADS = 128;

struct Geo {
    int Q, Qpad, nQT;   // queries, padded to 64, query tiles
    int NW, NB;         // 32-bit words per code, distance buckets (b + 1)
    int LW;             // 64-bit words per label row
    int S;              // segments of the shard
    i64 N, L;           // shard rows, rows per segment (multiple of 16)
    i64 R;              // ranked-list length
    u32 idx_base;       // global index of shard row 0
    i64 nUnits;         // S * nQT
    int wpb;            // wavefronts (= units) per block of the launch this Geo goes to
    int nBlk;           // logical blocks = ceil(nUnits / wpb); the grid is padded to a multiple of 8
};

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8).  Give
// every XCD one contiguous range of logical blocks, i.e. a contiguous range of
// database segments, so that its private 4 MiB L2 holds just that slice of the
// shard while all query tiles stream over it.  Speed only; any mapping is correct.
__device__ __forceinline__ int logical_block(int nBlk) {
    const int per = (int)gridDim.x >> 3;
    const int lb = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    return lb < nBlk ? lb : -1;
}

// pb[] in k_order is written by one lane and read by others of the SAME wavefront:
// LDS operations of a wave complete in issue order, so no s_barrier is needed --
// only the compiler has to keep the program order of the accesses.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NW>
__device__ __forceinline__ u32 hamming(const u32 (&qw)[NW], const u32* __restrict__ row) {
    u32 d = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) d += __builtin_popcount(qw[w] ^ row[w]);
    return d;
}

// Rows per scalar-load batch: the batch is fetched with s_load_dwordx16 bursts
// BEFORE any of it is used, so one s_waitcnt covers it; the other wavefronts
// of the SIMD fill the wait.
template <int NW> struct Batch { static constexpr int rows = NW <= 2 ? 16 : (NW <= 4 ? 8 : 4); };

// ----------------------------------------------------------------------------
// K1  distance histogram.   metric.py:13 (the Q x N similarity matrix), never
// materialised: every pair's distance goes straight into the lane's (= query's)
// private histogram column in LDS, h[d][lane] -- bank = lane % 32, so the
// ds_add_u32 stream is conflict free.  Output hist[s][d][q], q fastest.
// ----------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void k_hist(const u32* __restrict__ qc, const u32* __restrict__ db,
                                              u32* __restrict__ hist, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 unit = (i64)lb * g.wpb + wave;
    if (unit >= g.nUnits) return;
    const int s = (int)(unit / g.nQT);
    const int qt = (int)(unit - (i64)s * g.nQT);
    const int q = qt * 64 + lane;

    u32 qw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) qw[w] = q < g.Q ? qc[(i64)q * NW + w] : 0u;

    u32* h = lds + wave * g.NB * 64;
    for (int d = 0; d < g.NB; ++d) h[d * 64 + lane] = 0u;

    const i64 lo = (i64)s * g.L;
    const i64 hi = lo + g.L < g.N ? lo + g.L : g.N;
    const u32* __restrict__ p = db + lo * NW;
    i64 n = lo;
    constexpr int B = Batch<NW>::rows;
    for (; n + B <= hi; n += B, p += B * NW) {
        u32 c[B * NW];
#pragma unroll
        for (int i = 0; i < B * NW; ++i) c[i] = p[i];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            u32 d = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) d += __builtin_popcount(qw[w] ^ c[j * NW + w]);
            atomicAdd(&h[d * 64 + lane], 1u);
        }
    }
    for (; n < hi; ++n, p += NW) {
        const u32 d = hamming<NW>(qw, p);
        atomicAdd(&h[d * 64 + lane], 1u);
    }
    u32* __restrict__ out = hist + (i64)s * g.NB * g.Qpad + q;
    for (int d = 0; d < g.NB; ++d) out[(i64)d * g.Qpad] = h[d * 64 + lane];
}

// K2a  Hown[d][q] = sum over segments of hist[s][d][q].
__global__ __launch_bounds__(256) void k_hist_reduce(const u32* __restrict__ hist, u32* __restrict__ hown, const Geo g) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 plane = (i64)g.NB * g.Qpad;
    if (i >= plane) return;
    u32 acc = 0;
    for (int s = 0; s < g.S; ++s) acc += hist[(i64)s * plane + i];
    hown[i] = acc;
}

// ----------------------------------------------------------------------------
// K2b  per-query plan.   metric.py:14 + the [0:R] cut at :19, as a counting
// argument over the b+1 possible distances: t = smallest d with
// #(dist <= d over ALL shards) >= R; everything closer than t is in the top R,
// of the rows at exactly t the first `quota` in (shard, index) order are.
// hall: G gathered shard histograms [G][NB][Qpad] (or this shard's own, G = 1).
// ----------------------------------------------------------------------------
struct Plan {
    int* t;          // threshold distance
    u32* cnt_lt;     // rows closer than t, all shards   (= global position of the first tie)
    u32* quota;      // ties at t kept, all shards        (= R - cnt_lt)
    u32* tie_before; // ties at t owned by lower-ranked shards
    u32* n_lt;       // rows closer than t in THIS shard
    u32* posbase;    // [NB][Qpad] global position of this shard's first row in bucket d (d <= t)
    int* err;        // set when R exceeds the total row count
};

__global__ __launch_bounds__(256) void k_plan(const u32* __restrict__ hown, const u32* __restrict__ hall, int G, int rank,
                                              Plan pl, const Geo g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.Q) return;
    const i64 plane = (i64)g.NB * g.Qpad;
    u64 cum = 0;
    u32 nlt = 0;
    int t = -1;
    for (int d = 0; d < g.NB && t < 0; ++d) {
        const i64 o = (i64)d * g.Qpad + q;
        u64 all = 0, before = 0;
        if (G > 1) {
            for (int r = 0; r < G; ++r) {
                const u32 v = hall[(i64)r * plane + o];
                all += v;
                if (r < rank) before += v;
            }
        } else {
            all = hown[o];
        }
        pl.posbase[o] = (u32)(cum + before);
        if (cum + all >= (u64)g.R) {
            t = d;
            pl.cnt_lt[q] = (u32)cum;
            pl.quota[q] = (u32)((u64)g.R - cum);
            pl.tie_before[q] = (u32)before;
        } else {
            nlt += hown[o];
            cum += all;
        }
    }
    if (t < 0) {  // R > total rows: caller error, keep the device state harmless
        atomicExch(pl.err, 1);
        pl.cnt_lt[q] = 0; pl.quota[q] = 0; pl.tie_before[q] = 0; nlt = 0;
    }
    pl.t[q] = t;
    pl.n_lt[q] = nlt;
}

// K2c  per (segment, query): rows closer than t and rows at t in that segment.
__global__ __launch_bounds__(256) void k_seg_counts(const u32* __restrict__ hist, const int* __restrict__ tq,
                                                    u32* __restrict__ seglt, u32* __restrict__ segtie, const Geo g) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;  // over S * Qpad
    if (i >= (i64)g.S * g.Qpad) return;
    const int s = (int)(i / g.Qpad);
    const int q = (int)(i - (i64)s * g.Qpad);
    u32 lt = 0, tie = 0;
    if (q < g.Q) {
        const int t = tq[q];
        const u32* __restrict__ hp = hist + (i64)s * g.NB * g.Qpad + q;
        for (int d = 0; d < t; ++d) lt += hp[(i64)d * g.Qpad];
        if (t >= 0) tie = hp[(i64)t * g.Qpad];
    }
    seglt[i] = lt;
    segtie[i] = tie;
}

// K2d  exclusive prefix over segments (in place), one thread per query.
__global__ __launch_bounds__(256) void k_seg_prefix(u32* __restrict__ seglt, u32* __restrict__ segtie, const Geo g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.Qpad) return;
    u32 a = 0, b = 0;
    for (int s = 0; s < g.S; ++s) {
        const i64 o = (i64)s * g.Qpad + q;
        const u32 x = seglt[o], y = segtie[o];
        seglt[o] = a; segtie[o] = b;
        a += x; b += y;
    }
}

// ----------------------------------------------------------------------------
// K3  select.   Second pass over the pairs.  A lane walks its query through the
// segment in index order and appends
//   rows closer than t  -> scratch list scr[q][..]   (index order; K4 buckets them)
//   rows at t           -> straight to their final slots out_idx[q][cnt_lt + tie rank]
//                          while the (shard-global) tie rank is below the quota.
// Both streams start at offsets known from the histograms, so the result does
// not depend on scheduling: no atomics, no cross-lane traffic.
// ----------------------------------------------------------------------------
struct SelectArgs {
    const int* t;
    const u32* cnt_lt;
    const u32* quota;
    const u32* tie_before;
    const u32* seglt;    // [S][Qpad] exclusive prefix
    const u32* segtie;   // [S][Qpad] exclusive prefix
};

template <int NW>
__global__ __launch_bounds__(256) void k_select(const u32* __restrict__ qc, const u32* __restrict__ db,
                                                const SelectArgs a, u32* __restrict__ scr_all,
                                                u32* __restrict__ out_idx, u8* __restrict__ out_dist, const Geo g) {
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 unit = (i64)lb * g.wpb + wave;
    if (unit >= g.nUnits) return;
    const int s = (int)(unit / g.nQT);
    const int qt = (int)(unit - (i64)s * g.nQT);
    const int q = qt * 64 + lane;
    const bool live = q < g.Q;

    u32 qw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) qw[w] = live ? qc[(i64)q * NW + w] : 0u;
    const int t = live ? a.t[q] : -1;                 // -1: nothing is ever selected
    const u32 quota = live ? a.quota[q] : 0u;
    const i64 so = (i64)s * g.Qpad + q;
    u32 ltpos = a.seglt[so];
    u32 tierank = (live ? a.tie_before[q] : 0u) + a.segtie[so];
    const u32 cntlt = live ? a.cnt_lt[q] : 0u;
    const i64 rowoff = (i64)(live ? q : 0) * g.R;
    u32* __restrict__ scr = scr_all + rowoff;
    u32* __restrict__ oi = out_idx + rowoff + cntlt;    // first tie slot of this query
    u8* __restrict__ od = out_dist + rowoff + cntlt;

    const i64 lo = (i64)s * g.L;
    const i64 hi = lo + g.L < g.N ? lo + g.L : g.N;
    const u32* __restrict__ p = db + lo * NW;
    i64 n = lo;

#define HG_SELECT_ONE(D, NIDX)                                          \
    {                                                                   \
        const int d = (int)(D);                                         \
        if (d <= t) {                                                   \
            const u32 gi = g.idx_base + (u32)(NIDX);                    \
            if (d < t) {                                                \
                scr[ltpos] = gi;                                        \
                ++ltpos;                                                \
            } else {                                                    \
                if (tierank < quota) {                                  \
                    oi[tierank] = gi;                                   \
                    od[tierank] = (u8)d;                                \
                }                                                       \
                ++tierank;                                              \
            }                                                           \
        }                                                               \
    }

    constexpr int B = Batch<NW>::rows;
    for (; n + B <= hi; n += B, p += B * NW) {
        u32 c[B * NW];
#pragma unroll
        for (int i = 0; i < B * NW; ++i) c[i] = p[i];
        u32 dd[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            u32 d = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) d += __builtin_popcount(qw[w] ^ c[j * NW + w]);
            dd[j] = d;
        }
#pragma unroll
        for (int j = 0; j < B; ++j) HG_SELECT_ONE(dd[j], n + j)
    }
    for (; n < hi; ++n, p += NW) HG_SELECT_ONE(hamming<NW>(qw, p), n)
#undef HG_SELECT_ONE
}

// ----------------------------------------------------------------------------
// K4  order.   Stable counting sort of a query's scratch list (rows closer than
// t, index order) by distance into the final slots: one wavefront per query,
// 64 entries per step.  The distance is recomputed from the row's code (one
// 4*NW-byte gather per entry) instead of being carried through scratch.
// Rank among equal-distance lanes of a step: bit-sliced match over the NBITS
// bits of d (ballots), then popcount below the lane; per-bucket running
// positions live in the wave's LDS row pb[d].
// ----------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void k_order(const u32* __restrict__ qc, const u32* __restrict__ db,
                                               const u32* __restrict__ scr, const u32* __restrict__ n_lt,
                                               const int* __restrict__ tq, const u32* __restrict__ posbase,
                                               u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                               int nbits, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
    if (q >= g.Q) return;
    u32* pb = lds + wave * g.NB;
    const int t = tq[q];
    for (int d = lane; d < t; d += 64) pb[d] = posbase[(i64)d * g.Qpad + q];
    wave_lds_sync();
    u32 qw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) qw[w] = qc[(i64)q * NW + w];
    const u32 cnt = n_lt[q];
    const u32* __restrict__ src = scr + (i64)q * g.R;
    u32* __restrict__ oi = out_idx + (i64)q * g.R;
    u8* __restrict__ od = out_dist + (i64)q * g.R;
    const u64 below = (1ull << lane) - 1ull;

    for (u32 base = 0; base < cnt; base += 64) {
        const u32 i = base + lane;
        const bool valid = i < cnt;
        u32 gi = 0, d = 0;
        if (valid) {
            gi = src[i];
            d = hamming<NW>(qw, db + (i64)(gi - g.idx_base) * NW);
        }
        u64 peers = __ballot(valid);
        for (int k = 0; k < nbits; ++k) {
            const bool bit = (d >> k) & 1u;
            const u64 m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        const u32 rank = (u32)__popcll(peers & below);
        const u32 npeer = (u32)__popcll(peers);
        if (valid) {
            const u32 start = pb[d];
            oi[start + rank] = gi;
            od[start + rank] = (u8)d;
            if (rank == npeer - 1) pb[d] = start + npeer;   // last peer advances the bucket
        }
        wave_lds_sync();
    }
}

// ----------------------------------------------------------------------------
// K5  label match.   metric.py:17-19: slot k of query q matches iff the ranked
// row shares a positive label with the query.  One bit per slot, 64 slots per
// wavefront via ballot.  Slots owned by another shard (IDX_NONE) give 0.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match(const u32* __restrict__ out_idx, const u64* __restrict__ dblab,
                                               const u64* __restrict__ qlab, u64* __restrict__ mbits,
                                               i64 RW, int nKB, const Geo g) {
    const int q = (int)(blockIdx.x / (u32)nKB);          // nKB = ceil(R / 256) blocks per query
    const i64 k = (i64)(blockIdx.x - (u32)q * (u32)nKB) * 256 + threadIdx.x;
    bool m = false;
    if (k < g.R) {
        const u32 gi = out_idx[(i64)q * g.R + k];
        if (gi != IDX_NONE) {
            const u64* __restrict__ dl = dblab + (i64)(gi - g.idx_base) * g.LW;
            const u64* __restrict__ ql = qlab + (i64)q * g.LW;
            u64 any = 0;
            for (int w = 0; w < g.LW; ++w) any |= dl[w] & ql[w];
            m = any != 0;
        }
    }
    const u64 word = __ballot(m);
    if ((threadIdx.x & 63) == 0 && (k >> 6) < RW) mbits[(i64)q * RW + (k >> 6)] = word;
}

// OR of G shards' bit rows (disjoint by construction).
__global__ __launch_bounds__(256) void k_or_bits(const u64* __restrict__ all, u64* __restrict__ out, i64 n, int G) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 v = 0;
    for (int r = 0; r < G; ++r) v |= all[(i64)r * n + i];
    out[i] = v;
}

// min over G shards' ranked lists: exactly one shard owns a slot, the others hold IDX_NONE / 0xFF.
__global__ __launch_bounds__(256) void k_min_topr(const u32* __restrict__ idx_all, const u8* __restrict__ dist_all,
                                                  u32* __restrict__ idx, u8* __restrict__ dist, i64 n, int G) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 v = IDX_NONE;
    u8 dv = 0xFF;
    for (int r = 0; r < G; ++r) {
        const u32 x = idx_all[(i64)r * n + i];
        if (x < v) { v = x; dv = dist_all[(i64)r * n + i]; }
    }
    idx[i] = v;
    dist[i] = dv;
}

// ----------------------------------------------------------------------------
// K6  average precision.   metric.py:20-23 in float64, bit-exact to NumPy:
//   px[k]  = cumsum(imatch)[k] / (k + 1)        one correctly rounded division
//   AP     = np.sum(px * imatch) / rel
// np.sum adds 8192-element chunks left to right, each chunk by pairwise
// summation: blocks of <= 128 elements through 8 strided accumulators combined
// as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail, halves split at
// n/2 rounded down to a multiple of 8.  The split tree depends only on the
// chunk length, so the host flattens it once per R into a leaf table and a
// postfix program (ApShape); a thread evaluates one leaf, thread 0 the program.
// ----------------------------------------------------------------------------
struct ApShape {               // tree of one chunk length (<= AP_CHUNK elements)
    int n;                     // chunk length
    int n_leaves;              // <= 128
    int n_prog;                // <= 255
    unsigned short leaf_start[AP_LEAF];
    unsigned short leaf_len[AP_LEAF];
    short prog[2 * AP_LEAF];   // >= 0: push leaf, -1: add the two on top
};

__device__ __forceinline__ u32 count_bits_below(const u64* cw, int e) {  // bits [0, e) of the chunk
    u32 c = 0;
    const int full = e >> 6;
    for (int w = 0; w < full; ++w) c += (u32)__popcll(cw[w]);
    const int rem = e & 63;
    if (rem) c += (u32)__popcll(cw[full] & ((1ull << rem) - 1ull));
    return c;
}

__global__ __launch_bounds__(AP_THREADS) void k_ap(const u64* __restrict__ mbits, i64 RW, i64 R,
                                                   const ApShape* __restrict__ shapes,  // [0] full chunk, [1] last chunk
                                                   double* __restrict__ ap, u32* __restrict__ rel) {
    __shared__ u64 cw[AP_CHUNK / 64];
    __shared__ double leafsum[AP_LEAF];
    __shared__ double stk[16];
    __shared__ u32 s_before;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const u64* __restrict__ row = mbits + (i64)q * RW;
    if (tid == 0) s_before = 0;
    double total = 0.0;        // thread 0 only
    const i64 n_chunks = (R + AP_CHUNK - 1) / AP_CHUNK;
    for (i64 c = 0; c < n_chunks; ++c) {
        const i64 cb = c * AP_CHUNK;
        const bool last = (c == n_chunks - 1);
        const ApShape* __restrict__ sh = shapes + ((last && (R - cb) != AP_CHUNK) ? 1 : 0);
        const int n = (int)(R - cb < AP_CHUNK ? R - cb : AP_CHUNK);
        const i64 w = (cb >> 6) + tid;
        cw[tid] = (w < RW) ? row[w] : 0ull;       // AP_THREADS == AP_CHUNK / 64
        __syncthreads();
        const u32 before = s_before;
        if (tid < sh->n_leaves) {
            const int ls = sh->leaf_start[tid], ll = sh->leaf_len[tid];
            u32 cnt = before + count_bits_below(cw, ls);
            int e = ls;                              // element index inside the chunk
            // value of element e (elements are consumed in increasing order)
            auto next = [&]() -> double {
                const bool bit = (cw[e >> 6] >> (e & 63)) & 1ull;
                double v = 0.0;
                if (bit) { ++cnt; v = (double)cnt / (double)(cb + e + 1); }
                ++e;
                return v;
            };
            double res;
            if (ll < 8) {
                res = 0.0;
                for (int i = 0; i < ll; ++i) res += next();
            } else {
                double r0 = next(), r1 = next(), r2 = next(), r3 = next();
                double r4 = next(), r5 = next(), r6 = next(), r7 = next();
                int i = 8;
                for (; i < ll - (ll % 8); i += 8) {
                    r0 += next(); r1 += next(); r2 += next(); r3 += next();
                    r4 += next(); r5 += next(); r6 += next(); r7 += next();
                }
                res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                for (; i < ll; ++i) res += next();
            }
            leafsum[tid] = res;
        }
        __syncthreads();
        if (tid == 0) {
            int sp = 0;
            for (int i = 0; i < sh->n_prog; ++i) {
                const int op = sh->prog[i];
                if (op >= 0) stk[sp++] = leafsum[op];
                else { --sp; stk[sp - 1] = stk[sp - 1] + stk[sp]; }
            }
            total = (c == 0) ? stk[0] : total + stk[0];
            s_before = before + count_bits_below(cw, n);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const u32 r = s_before;
        rel[q] = r;
        ap[q] = r ? total / (double)r : __longlong_as_double(0x7FF8000000000000ll);
    }
}

// fill helpers
__global__ __launch_bounds__(256) void k_fill_u32(u32* __restrict__ p, u32 v, i64 n) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace hg
