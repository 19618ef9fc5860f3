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
constexpr int TAIL_WORDS = 64;         // words appended to an exported histogram: [0] overflow flag, [1] sampled rows
// Measurement probes of the matrix-core select kernels (SelArgs::probe) exist only in the probe build
// (`python -m hashgan_amd.build --probes` -> libhashgan_amd_probe.so, -DHG_PROBES=1); the production
// kernels carry none of their branches.
#ifndef HG_PROBES
#define HG_PROBES 0
#endif
constexpr bool kProbes = HG_PROBES != 0;

struct Geo {
    int Q, Qpad, nQT;   // queries, padded to 64, query tiles
    int NW, NB;         // 32-bit words per code, distance buckets (b + 1)
    int LW;             // 64-bit words per label row
    int S;              // segments of the shard
    i64 N, L;           // shard rows, rows per segment (multiple of 32)
    i64 R;              // ranked-list length
    u32 idx_base;       // global index of shard row 0
    i64 nUnits;         // S * nQT
    int hist_stride;    // k_hist visits every hist_stride-th row batch (1 = all rows; >1 = sampling pass)
    int wpb;            // wavefronts (= units) per block of the launch this Geo goes to
    int nBlk;           // logical blocks = ceil(nUnits / wpb); the grid is padded to a multiple of 8
    int hcap;           // sampled pass of the one-shot bet: only the distance planes [0, hcap) are written and read (0: all)
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
    const i64 step = (i64)B * g.hist_stride;
    // Software prefetch: the next batch's scalar loads are issued right after the FIRST row of the
    // current batch has been consumed (so the s_waitcnt that guards the current batch has just
    // retired and covers nothing else) and have the other B-1 rows of work to land.
    if (n + B <= hi) {
        u32 c[B * NW];
#pragma unroll
        for (int i = 0; i < B * NW; ++i) c[i] = p[i];
        for (; n + B <= hi; n += step, p += step * NW) {
            const bool more = n + step + B <= hi;
            {
                u32 d = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) d += __builtin_popcount(qw[w] ^ c[w]);
                atomicAdd(&h[d * 64 + lane], 1u);
            }
            __builtin_amdgcn_sched_barrier(0);
            u32 cn[B * NW];
            if (more) {
#pragma unroll
                for (int i = 0; i < B * NW; ++i) cn[i] = p[step * NW + i];
            } else {
#pragma unroll
                for (int i = 0; i < B * NW; ++i) cn[i] = 0u;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 1; j < B; ++j) {
                u32 d = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) d += __builtin_popcount(qw[w] ^ c[j * NW + w]);
                atomicAdd(&h[d * 64 + lane], 1u);
            }
#pragma unroll
            for (int i = 0; i < B * NW; ++i) c[i] = cn[i];
        }
    }
    if (g.hist_stride == 1) {   // ragged tail of the segment (the sampling pass skips it)
        for (; n < hi; ++n, p += NW) {
            const u32 d = hamming<NW>(qw, p);
            atomicAdd(&h[d * 64 + lane], 1u);
        }
    }
    u32* __restrict__ out = hist + (i64)s * g.NB * g.Qpad + q;
    for (int d = 0; d < g.NB; ++d) out[(i64)d * g.Qpad] = h[d * 64 + lane];
}

// K2a  Hown[d][q] = sum over segments of hist[s][d][q].
static __global__ __launch_bounds__(256) void k_hist_reduce(const u32* __restrict__ hist, u32* __restrict__ hown, const Geo g) {
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

static __global__ __launch_bounds__(256) void k_plan(const u32* __restrict__ hown, const u32* __restrict__ hall, int G, int rank,
                                              Plan pl, const Geo g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const i64 plane = (i64)g.NB * g.Qpad + TAIL_WORDS;     // stride between the gathered shard histograms
    if (q == 0) {                                           // a shard reported overflowed slices: the whole bet is off
        u32 flag = hown[plane - TAIL_WORDS];
        if (G > 1) for (int r = 0; r < G; ++r) flag |= hall[(i64)r * plane + plane - TAIL_WORDS];
        if (flag) atomicExch(pl.err, 1);
    }
    if (q >= g.Q) return;
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
static __global__ __launch_bounds__(256) void k_seg_counts(const u32* __restrict__ hist, const int* __restrict__ tq,
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

// K2d  one thread per query walks the segments in order and turns the counts
// into the exact layout of the query's record row: segment s writes its
// (closer-than-t rows + kept ties) at sl_start[s][q]; sl_tie[s][q] = how many of
// the segment's ties are still inside the quota (ties are ranked shard-globally:
// lower-ranked shards first, then index order).  tot[q] = records of the row.
static __global__ __launch_bounds__(256) void k_seg_layout(const u32* __restrict__ seglt, const u32* __restrict__ segtie,
                                                    const u32* __restrict__ quota, const u32* __restrict__ tie_before,
                                                    u32* __restrict__ sl_start, u32* __restrict__ sl_tie,
                                                    u32* __restrict__ tot, int* __restrict__ sstar, int ratio, const Geo g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.Qpad) return;
    u32 pos = 0;
    u64 tierank = q < g.Q ? tie_before[q] : 0;
    const u64 qt = q < g.Q ? quota[q] : 0;
    int last = -1;                                      // last segment that still contributes ties at the cut
    for (int s = 0; s < g.S; ++s) {
        const i64 o = (i64)s * g.Qpad + q;
        const u32 lt = seglt[o], tie = segtie[o];
        const u64 room = tierank < qt ? qt - tierank : 0;
        const u32 keep = tie < room ? tie : (u32)room;
        sl_start[o] = pos;
        sl_tie[o] = keep;
        if (keep) last = s;
        pos += lt + keep;
        tierank += tie;
    }
    tot[q] = pos;
    // the matrix-core select collects distance t up to here (its `sstar`); `ratio` select segments per segment of this pass
    if (sstar && q < g.Q) sstar[q] = last < 0 ? -1 : (last + 1) * ratio - 1;
}

// K2e  optimistic plan: threshold guess from SAMPLED histograms (this shard's, or the G
// gathered ones).  With f = sampled rows / all rows, the guess is the smallest T whose sample
// count reaches f*R + sigma*sqrt(f*R) + 1, i.e. #(dist <= T over all rows) >= R all but
// certainly.  The guess is only a performance bet: the records' exact histogram goes through
// k_plan afterwards (k_rank_fused) and a lost bet reruns the exact path.
// The cut bucket T is usually far larger than what is still missing below it (its first `quota` rows
// in index order are all that can enter the list), so the guess is two-dimensional: T, and the last
// segment `sstar` up to which rows AT distance T are still collected -- the smallest prefix of the
// database (lower-ranked shards first, then this shard's segments in order) whose sampled count of
// {dist < T} + {dist == T inside the prefix} reaches the same `need`.  Segments past sstar select
// dist < T only.  Exactness is untouched: the records at distance T form a prefix in index order,
// so either that prefix holds the true quota (the plan finds t = T with the right first rows) or the
// records come up short of R and the bet is lost.
// hseg: this shard's per-segment sample histograms [Sh][NB][Qpad] (k_hist's raw output), Sh segments
// of `ratio` select-segments each.
static __global__ __launch_bounds__(256) void k_guess(const u32* __restrict__ hs, const u32* __restrict__ hall, int G, int rank,
                                               const u32* __restrict__ hseg, int Sh, int ratio,
                                               double sigma, i64 n_total, int* __restrict__ T, int* __restrict__ sstar,
                                               const Geo g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.Q) return;
    const i64 plane = (i64)g.NB * g.Qpad + TAIL_WORDS;
    // sampled rows over all shards -> sample count that makes #(dist <= T over all rows) >= R all but certain
    u64 sampled = 0;
    if (G > 1) for (int r = 0; r < G; ++r) sampled += hall[(i64)r * plane + plane - TAIL_WORDS + 1];
    else sampled = hs[plane - TAIL_WORDS + 1];
    const double fr = (double)g.R * (double)sampled / (double)n_total;
    const double needd = fr + sigma * sqrt(fr) + 1.0;
    const u64 need = (u64)ceil(needd);
    u64 cum = 0, below = 0;
    int t = g.NB - 1;                                  // sample too thin: take everything
    bool found = false;
    for (int d = 0; d < g.NB; ++d) {
        const i64 o = (i64)d * g.Qpad + q;
        below = cum;
        if (G > 1) for (int r = 0; r < G; ++r) cum += hall[(i64)r * plane + o];
        else cum += hs[o];
        if (cum >= need) { t = d; found = true; break; }
    }
    T[q] = t;
    int ss = g.S - 1;                                  // default: collect distance T everywhere
    if (found) {
        u64 have = below;                              // {dist < T} everywhere ...
        const i64 ot = (i64)t * g.Qpad + q;
        if (G > 1) for (int r = 0; r < rank; ++r) have += hall[(i64)r * plane + ot];   // ... + {dist == T} on lower shards
        if (have >= need) {
            ss = -1;                                   // the lower shards already hold the prefix
        } else {
            for (int sh = 0; sh < Sh; ++sh) {
                have += hseg[((i64)sh * g.NB + t) * g.Qpad + q];
                if (have >= need) { ss = (sh + 1) * ratio - 1; break; }
            }
            if (ss > g.S - 1) ss = g.S - 1;
        }
    }
    sstar[q] = ss;
}

// One-shot, single-shard form of k_hist_reduce + k_guess: sums the sampled pass's per-segment histograms
// itself, bucket by bucket, and stops at the guessed cut -- a third of the planes at C2, no reduced copy,
// one launch less.  Also clears the bet's per-query overflow flags and the lost-bet flag (two fills less).
// PARTS lanes per query (4, 16 or 64 -- more when the sampled pass has many segments): 64 / PARTS queries per wavefront
template <int PARTS>
__global__ __launch_bounds__(256) void k_guess_direct(const u32* __restrict__ hseg, int Sh, int ratio, double sigma,
                                                      i64 n_total, u32 sampled, int* __restrict__ T, int* __restrict__ sstar,
                                                      u32* __restrict__ failq, int* __restrict__ err, u32* __restrict__ crowd, const Geo g) {
    constexpr int QPW = 64 / PARTS;                    // lane = part * QPW + query-in-wave
    const int lane = threadIdx.x & 63, part = lane / QPW;
    const int q = (blockIdx.x * WPB + (threadIdx.x >> 6)) * QPW + (lane % QPW);
    if (part == 0 && q < g.Qpad) failq[q] = 0u;
    if (q == 0 && part == 0) { err[0] = 0; err[1] = 0; }    // the lost-bet flag and the fused step's leftover count
    const bool live = q < g.Q;
    const int qq = live ? q : 0;                       // dead lanes follow query 0 (shuffles need every lane)
    const double fr = (double)g.R * (double)sampled / (double)n_total;
    const u64 need = (u64)ceil(fmax(1.0, fr + sigma * sqrt(fr) + 1.0));     // (sigma < 0: the "handicap_next_bet" test hook)
    const i64 plane = (i64)g.NB * g.Qpad;
    const int per = (Sh + PARTS - 1) / PARTS;          // part p sums segments [p * per, (p + 1) * per)
    const int s0 = part * per < Sh ? part * per : Sh, s1 = s0 + per < Sh ? s0 + per : Sh;
    u64 cum = 0, below = 0;
    int t = g.NB - 1;                                  // sample too thin (or the cut beyond the planes the pass wrote): take everything
    bool found = false;
    const int dn = g.hcap > 0 && g.hcap < g.NB ? g.hcap : g.NB;
    // The walk over the distances is a chain of L2 round trips (a plane's counts must be in before the next is worth
    // reading): two planes per step, eight segments of each in flight -- 16 loads per trip instead of 4 (0.038 -> 0.025 ms
    // at C2, where a lane sums 13 segments per plane and stops at the 19th; C5 0.083 -> 0.049).
    constexpr int PL = 4;                              // planes per trip (round 3: 2 -> 4, 32 loads in flight: 0.0255 -> 0.0205 ms at C2, C5 0.047 -> 0.034)
    u32 gmax = 0, gtot = 0;                            // crowding probe: the fullest sampled segment of the last plane group, the group's total
    for (int d = 0; d < dn && !found; d += PL) {
        u32 cs[PL];
        const u32* __restrict__ col[PL];
        gmax = 0;
#pragma unroll
        for (int p = 0; p < PL; ++p) { cs[p] = 0; col[p] = hseg + (i64)(d + p < dn ? d + p : d) * g.Qpad + qq; }
        for (int sh = s0; sh < s1; sh += 8) {
            u32 v[PL][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const i64 o = (i64)(sh + k < s1 ? sh + k : 0) * plane;     // past the part: any valid segment, not counted
#pragma unroll
                for (int p = 0; p < PL; ++p) v[p][k] = col[p][o];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                u32 sk = 0;
#pragma unroll
                for (int p = 0; p < PL; ++p) { const u32 x = sh + k < s1 ? v[p][k] : 0u; cs[p] += x; sk += x; }
                gmax = sk > gmax ? sk : gmax;
            }
        }
#pragma unroll
        for (int off = QPW; off < 64; off <<= 1)                          // sums over the query's parts
#pragma unroll
            for (int p = 0; p < PL; ++p) cs[p] += (u32)__shfl_xor((int)cs[p], off);
        gtot = 0;
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            gtot += cs[p];
            if (!found && d + p < dn) {
                below = cum;
                cum += cs[p];
                if (cum >= need) { t = d + p; found = true; }
            }
        }
    }
    if (crowd) {
        // How unevenly do the rows near this query spread over the database?  The plane group that holds the cut carries
        // most of the sampled mass below it: its fullest segment against the group's total, summed over the queries --
        // the host turns sum(max) * segments / sum(total) into the slices' width for this database (a database stored class
        // by class puts a query's near rows into its class's tenth of the segments: run_oneshot used to find that out by
        // losing two bets).  First call on a database only.
#pragma unroll
        for (int off = QPW; off < 64; off <<= 1) {
            const u32 m = (u32)__shfl_xor((int)gmax, off);
            gmax = gmax > m ? gmax : m;
        }
        if (live && found && part == 0) { atomicAdd(crowd, gmax); atomicAdd(crowd + 1, gtot); }
    }
    // all parts of a query agree on t; a wave's queries may stop at different d: the shuffles below only pair
    // lanes of the same query, which left the loop together
    int ss = g.S - 1;                                  // default: collect distance T everywhere
    if (found) {
        // {dist < T} everywhere + {dist == T} up to a segment: first segment (in order) where the count reaches need
        const u32* __restrict__ col = hseg + (i64)t * g.Qpad + qq;
        u32 mine = 0;
        for (int sh = s0; sh < s1; ++sh) mine += col[(i64)sh * plane];
        u32 incl = mine;                               // inclusive prefix over the query's parts (lanes QPW apart)
#pragma unroll
        for (int off = QPW; off < 64; off <<= 1) {
            const u32 v = (u32)__shfl_up((int)incl, off);
            if (lane >= off) incl += v;
        }
        u64 have = below + (u64)(incl - mine);
        int cand = 0x7FFFFFFF;
        for (int sh = s0; sh < s1; ++sh) {
            have += col[(i64)sh * plane];
            if (have >= need) { cand = sh; break; }
        }
#pragma unroll
        for (int off = QPW; off < 64; off <<= 1) {
            const int m = __shfl_xor(cand, off);
            cand = cand < m ? cand : m;
        }
        if (cand != 0x7FFFFFFF) ss = (cand + 1) * ratio - 1;
        if (ss > g.S - 1) ss = g.S - 1;
    }
    if (live && part == 0) { T[q] = t; sstar[q] = ss; }
}

// ----------------------------------------------------------------------------
// K3  select.   The pass over the pairs that produces ranked-list members.  A
// lane walks its query through the segment in index order; every row with
//   dist <  T,  or  dist == T and still inside the lane's tie allowance
// becomes one 8-byte record {idx:32 | dist:8 | match:1} appended to the lane's
// slice of the query's record row.  The match bit (metric.py:17-19: the row
// shares a positive label with the query) is computed here because the row's
// label words are wave-uniform scalars at this point -- no gather later.
//   exact mode      T = t from the full histogram, slices are exact-sized, ties
//                   limited per slice (k_seg_layout): the row ends up holding
//                   precisely this shard's members of the top R.
//   optimistic mode T = guess, fixed-capacity slices, all ties: a superset of
//                   the members unless a slice overflows (-> fail flag).
// Rows are visited in index order and slices are per (query, segment), so the
// records of a row, read slice by slice, are in index order: no atomics, no
// cross-lane traffic, result independent of scheduling.
// ----------------------------------------------------------------------------
struct SelArgs {
    const int* T;          // [Qpad] threshold (t or guess)
    const u32* sl_start;   // [S][Qpad] exact mode
    const u32* sl_tie;     // [S][Qpad] exact mode
    u32* sl_cnt;           // [S][Qpad] out: records in the slice
    u32* fail;             // [Qpad] out, optimistic mode: a slice overflowed
    u32 cap;               // optimistic mode: slice capacity (records)
    i64 crow;              // record-row stride
    int optimistic;
    const int* sstar;      // optimistic mode: [Qpad] last segment that still collects dist == T (k_guess)
    int probe;             // measurement probes of the matrix-core kernels (option "probe_select"): 2 no drain,
                           // 4 no record stores, 8 no emit -- each breaks the bet on purpose (exact rerun follows)
};

constexpr int sel_batch_rows(int nw) {          // rows per scalar-load batch: <= 64 SGPRs of code words, <= 32 rows
    int r = 64 / nw, p = 1;
    while (p * 2 <= r) p *= 2;
    return p > 32 ? 32 : (p < 2 ? 2 : p);
}

__device__ __forceinline__ u64 make_rec(u32 idx, u32 d, bool m) {
    return (u64)idx | ((u64)(d | (m ? 0x100u : 0u)) << 32);
}

// Hot loop of k_select, per (query, row) pair -- 2*NW + 1 VALU ops and NO branch:
//   dp = popcount(q ^ row) - T - 1      the -T-1 rides in as the first v_bcnt accumulator,
//                                        so the sign bit of dp IS the test (dist <= T);
//   hm = alignbit(hm, dp, 31)            = (hm << 1) | sign(dp): one op shifts the lane's hit
//                                        mask and appends this row's bit.
// After a batch of <= 32 rows every lane holds the bit mask of ITS hits in the batch.  Hits are
// rare (R/N per pair), so instead of branching on every row whose 64 lanes contain a hit, the
// wave drains the masks afterwards: per round every lane with hits left takes its earliest one
// (different lanes, different rows), re-reads that row's words with a per-lane load, and writes
// the record.  Rounds per batch = max hits of any one lane (~2 at R/N = 0.5%), instead of one
// divergent block per hit row (~10 per batch).
// LW = 0: labels wider than 128 classes, match bit left 0 (k_match runs later).
// OPT: optimistic mode (compile-time so the drain carries no tie logic).
template <int NW, int LW, bool OPT>
__global__ __launch_bounds__(256) void k_select(const u32* __restrict__ qc, const u64* __restrict__ qlab,
                                                const u32* __restrict__ db, const u64* __restrict__ dblab,
                                                const SelArgs a, u64* __restrict__ cand, const Geo g) {
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
    constexpr int LWA = LW > 0 ? LW : 1;

    u32 qw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) qw[w] = live ? qc[(i64)q * NW + w] : 0u;
    u64 ql[LWA];
#pragma unroll
    for (int w = 0; w < LWA; ++w) ql[w] = (LW > 0 && live) ? qlab[(i64)q * LW + w] : 0ull;
    // -1: nothing is ever selected.  Optimistic mode: past the query's last tie-collecting segment
    // only rows strictly closer than the guessed cut are taken.
    const int T = live ? (OPT ? a.T[q] - (s > a.sstar[q] ? 1 : 0) : a.T[q]) : -1;
    const u32 bias = (u32)(-T - 1);                   // dist + bias < 0  <=>  dist <= T
    const i64 so = (i64)s * g.Qpad + q;
    u32 start, tielim = 0xFFFFFFFFu;
    if (OPT) start = (u32)s * a.cap;
    else { start = a.sl_start[so]; tielim = a.sl_tie[so]; }
    u32 ties = 0;
    u64* __restrict__ row = cand + (i64)(live ? q : 0) * a.crow;
    u64* __restrict__ wp = row + start;                    // next record of this lane's slice
    u32 room = OPT ? a.cap : 0xFFFFFFFFu;                  // optimistic: slice capacity (exact-mode slices are exact-sized)
    u32 dropped = 0;

    const i64 lo = (i64)s * g.L;
    const i64 hi = lo + g.L < g.N ? lo + g.L : g.N;
    const u32* __restrict__ p = db + lo * NW;
    i64 n = lo;

    // Drain the hit masks of a window of up to 128 rows that starts at row n0 (= uniform pointers
    // wp0 / wl0 into the code and label tables).  The window is two halves of <= 64 rows: bit
    // cntA-1-j of hmA <-> row n0+j, bit cntB-1-j of hmB <-> row n0+cntA+j, so within a half the
    // highest set bit is the lane's earliest hit and half A precedes half B.  Per round every lane
    // with hits left handles its earliest one; rounds = the largest hit count of any lane.
    auto drain = [&](u64 hmA, u64 hmB, i64 n0, const u32* __restrict__ wp0, const u64* __restrict__ wl0,
                     int cntA, int cntB) {
        while (__any((hmA | hmB) != 0ull)) {
            if ((hmA | hmB) != 0ull) {
                const bool inA = hmA != 0ull;
                u64 cur = inA ? hmA : hmB;
                const int k = 63 - __clzll((long long)cur);
                cur ^= 1ull << k;
                if (inA) hmA = cur; else hmB = cur;
                const u32 j = inA ? (u32)(cntA - 1 - k) : (u32)(cntA + cntB - 1 - k);  // row inside the window
                const u32* __restrict__ rp = wp0 + j * NW;        // per-lane re-read; the window was just streamed (L2)
                u32 d = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) d += __builtin_popcount(qw[w] ^ rp[w]);
                bool keep = true;
                if (!OPT) {
                    keep = (int)d < T;
                    if (!keep) { keep = ties < tielim; ++ties; }
                }
                if (keep) {
                    u64 any = 0;
                    if (LW > 0) {
                        const u64* __restrict__ lp = wl0 + j * LWA;
#pragma unroll
                        for (int w = 0; w < LWA; ++w) any |= lp[w] & ql[w];
                    }
                    if (!OPT || room) {
                        *wp = make_rec(g.idx_base + (u32)n0 + j, d, any != 0);
                        ++wp;
                        --room;
                    } else {
                        ++dropped;
                    }
                }
            }
        }
    };

    constexpr int B = sel_batch_rows(NW);                  // rows per scalar-load batch
    constexpr int HB = B >= 32 ? 2 : 1;                    // batches per 64-bit half mask
    constexpr int WROWS = 2 * HB * B;                      // rows per drain window (128 for short codes)
    const u64* __restrict__ pl = dblab + lo * LWA;
    for (; n + WROWS <= hi; n += WROWS, p += WROWS * NW, pl += WROWS * LWA) {
        u64 half[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u64 hm64 = 0;
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) {
                u32 c[B * NW];
#pragma unroll
                for (int i = 0; i < B * NW; ++i) c[i] = p[(h * HB + kb) * B * NW + i];
                u32 hm = 0;
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    u32 dp = bias;
#pragma unroll
                    for (int w = 0; w < NW; ++w) dp += __builtin_popcount(qw[w] ^ c[j * NW + w]);
                    hm = __builtin_amdgcn_alignbit(hm, dp, 31);
                }
                hm64 = (hm64 << B) | hm;                   // earlier batch in the higher bits
            }
            half[h] = hm64;
        }
        if (__builtin_expect(__any((half[0] | half[1]) != 0ull), 0))
            drain(half[0], half[1], n, p, pl, HB * B, HB * B);
    }
    if (n < hi) {                                          // ragged end of the segment: < WROWS rows
        u64 hmA = 0, hmB = 0;
        const int cnt = (int)(hi - n);
        const int cntA = cnt < 64 ? cnt : 64, cntB = cnt - cntA;
        for (int j = 0; j < cnt; ++j) {
            const u32 dp = bias + hamming<NW>(qw, p + j * NW);
            if (j < 64) hmA = (hmA << 1) | (u64)(dp >> 31);
            else hmB = (hmB << 1) | (u64)(dp >> 31);
        }
        drain(hmA, hmB, n, p, pl, cntA, cntB);
    }

    a.sl_cnt[so] = (u32)(wp - (row + start));
    if (OPT && dropped && live) a.fail[q] = 1u;        // several lanes may store the same 1
}

// ----------------------------------------------------------------------------
// K3d  select, dense regime (exact mode, R/N >= 1/4 -- e.g. the reference's own CIFAR-10
// configuration R = N).  When most pairs are selected the output is the bottleneck, so the
// mapping is transposed: lane <-> database row (64 consecutive rows), the query is wave-
// uniform (code, threshold and label words in SGPRs), the wave loops over the 64 queries of
// its tile.  Kept rows are compacted with a ballot + prefix popcount, so a query's records
// leave as ONE coalesced store per 64 rows, in index order; tie ranks are a running uniform
// count.  Same record rows, slices and counts as k_select: downstream is unchanged.
// ----------------------------------------------------------------------------
template <int NW, int LW>
__global__ __launch_bounds__(256) void k_select_dense(const u32* __restrict__ qc, const u64* __restrict__ qlab,
                                                      const u32* __restrict__ db, const u64* __restrict__ dblab,
                                                      const SelArgs a, u64* __restrict__ cand, const Geo g) {
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 unit = (i64)lb * g.wpb + wave;
    if (unit >= g.nUnits) return;
    const int s = (int)(unit / g.nQT);
    const int qt = (int)(unit - (i64)s * g.nQT);
    constexpr int LWA = LW > 0 ? LW : 1;
    const i64 lo = (i64)s * g.L;
    const i64 hi = lo + g.L < g.N ? lo + g.L : g.N;
    const u64 below = (1ull << lane) - 1ull;
    const int qend = (qt + 1) * 64 < g.Q ? (qt + 1) * 64 : g.Q;
    for (int q = qt * 64; q < qend; ++q) {                  // wave-uniform
        const int T = a.T[q];
        const i64 so = (i64)s * g.Qpad + q;
        const u32 start = a.sl_start[so], tielim = a.sl_tie[so];
        u32 qw[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) qw[w] = qc[(i64)q * NW + w];
        u64 ql[LWA];
#pragma unroll
        for (int w = 0; w < LWA; ++w) ql[w] = LW > 0 ? qlab[(i64)q * LW + w] : 0ull;
        u64* __restrict__ row = cand + (i64)q * a.crow;
        u32 pos = start, ties = 0;
        for (i64 n0 = lo; n0 < hi; n0 += 64) {
            const i64 n = n0 + lane;
            const bool valid = n < hi;
            u32 d = 0;
            if (valid) {
#pragma unroll
                for (int w = 0; w < NW; ++w) d += __builtin_popcount(qw[w] ^ db[n * NW + w]);
            }
            const bool is_tie = valid && (int)d == T;
            const u64 tmask = __ballot(is_tie);
            const bool keep = valid && ((int)d < T || (is_tie && ties + (u32)__popcll(tmask & below) < tielim));
            ties += (u32)__popcll(tmask);
            const u64 kmask = __ballot(keep);
            if (keep) {
                u64 any = 0;
                if (LW > 0) {
#pragma unroll
                    for (int w = 0; w < LWA; ++w) any |= dblab[n * LWA + w] & ql[w];
                }
                row[pos + (u32)__popcll(kmask & below)] = make_rec(g.idx_base + (u32)n, d, any != 0);
            }
            pos += (u32)__popcll(kmask);
        }
        if (lane == 0) a.sl_cnt[so] = pos - start;
    }
}

// ----------------------------------------------------------------------------
// K4  order (exact mode with several shards: the plan comes from gathered histograms).
// metric.py:14 finished: a query's records (index order) go to their final rank
// positions, canonical order, by a stable counting sort over the distance -- one
// wavefront per query, 64 records per step:
//   dist <  t : position = running start of the bucket (LDS row pb[d]) + rank
//               among the step's lanes of the same distance (bit-sliced match
//               over the bits of d via ballots, popcount below the lane);
//   dist == t : position = cnt_lt + shard-global tie rank, kept while the rank
//               is below the quota;
//   dist >  t : dropped (only optimistic supersets contain such rows).
// The record's match bit lands in the query's bit row at that position (LDS
// bitmap, written out coalesced), so label matching costs no extra pass; the
// idx/dist lists are written only when the caller wants them.
// ----------------------------------------------------------------------------
struct OrdArgs {
    const int* t;
    const u32* cnt_lt;
    const u32* quota;
    const u32* tie_before;
    const u32* posbase;    // [NB][Qpad]
    const u32* tot;        // [Qpad] records of the query (one dense run: exact-mode rows)
    i64 crow;
    int want_lists;
    int bits_lds;          // bit row fits the wave's LDS share; else global atomics on a zeroed row
    i64 RW;                // 64-bit words per bit row
};

static __global__ __launch_bounds__(256) void k_order(const u64* __restrict__ cand, const OrdArgs a,
                                               u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                               u32* __restrict__ mbits32, int nbits, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
    if (q >= g.Q) return;
    const int bmw = a.bits_lds ? (int)(2 * a.RW) : 0;       // 32-bit words of the LDS bitmap
    u32* pb = lds + wave * (g.NB + bmw);
    u32* bm = pb + g.NB;
    const int t = a.t[q];
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;
    if (t < 0) return;                                      // flagged query: the host reruns the exact path
    for (int d = lane; d < t; d += 64) pb[d] = a.posbase[(i64)d * g.Qpad + q];
    for (int w = lane; w < bmw; w += 64) bm[w] = 0u;
    wave_lds_sync();
    const u32 cntlt = a.cnt_lt[q], quota = a.quota[q], tiebef = a.tie_before[q];
    const u64* __restrict__ row = cand + (i64)q * a.crow;
    u32* __restrict__ oi = out_idx + (i64)q * g.R;
    u8* __restrict__ od = out_dist + (i64)q * g.R;
    const u64 below = (1ull << lane) - 1ull;
    u32 tie_run = 0;
    {
        const u32 cnt = a.tot[q];
        const u64* __restrict__ sl = row;
        for (u32 base = 0; base < cnt; base += 64) {
            const u32 i = base + lane;
            const bool valid = i < cnt;
            const u64 rec = valid ? sl[i] : 0ull;
            const u32 gi = (u32)rec;
            const u32 meta = (u32)(rec >> 32);
            const u32 d = meta & 0xFFu;
            const bool is_lt = valid && (int)d < t;
            const bool is_tie = valid && (int)d == t;
            u64 peers = __ballot(is_lt);
            for (int k = 0; k < nbits; ++k) {
                const bool bit = (d >> k) & 1u;
                const u64 m = __ballot(is_lt && bit);
                peers &= bit ? m : ~m;
            }
            const u64 tmask = __ballot(is_tie);
            u32 pos = IDX_NONE;
            if (is_lt) {
                const u32 rank = (u32)__popcll(peers & below);
                const u32 npeer = (u32)__popcll(peers);
                const u32 start = pb[d];
                pos = start + rank;
                if (rank == npeer - 1) pb[d] = start + npeer;   // last peer advances the bucket
            } else if (is_tie) {
                const u32 gr = tiebef + tie_run + (u32)__popcll(tmask & below);
                if (gr < quota) pos = cntlt + gr;
            }
            tie_run += (u32)__popcll(tmask);
            if (pos != IDX_NONE) {
                if (a.want_lists) { oi[pos] = gi; od[pos] = (u8)d; }
                if (meta & 0x100u) {
                    if (a.bits_lds) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
                    else atomicOr(&grow[pos >> 5], 1u << (pos & 31));
                }
            }
            wave_lds_sync();
        }
    }
    if (a.bits_lds) {
        wave_lds_sync();
        for (int w = lane; w < bmw; w += 64) grow[w] = bm[w];
    }
}

// ----------------------------------------------------------------------------
// K4f  verify + plan + order in one launch (optimistic mode, single shard).
// One 256-thread block per query; the query's slices are split into four
// contiguous ranges, one per wavefront:
//   phase 1  every wave histograms the distances of its records
//   phase 2  bucket totals -> threshold t, quota; per-wave bucket starts  (= k_plan, G = 1)
//            fewer than R records, or an overflowed slice: *err = 1, the host reruns the
//            exact path
//   phase 3  every wave places its records exactly like k_order; match bits go to a
//            block-wide LDS bitmap, written out coalesced.
// Ranks are stable across the four ranges because wave w's bucket d starts after the
// bucket-d records of waves < w (and ties are ranked the same way).
// ----------------------------------------------------------------------------
struct RankArgs {
    const u32* sl_cnt;     // [S][Qpad]   (slice mode)
    const u32* tot;        // [Qpad]      (dense mode: one run of tot[q] records, walked in chunks of `cap`)
    const u32* fail;       // [Qpad]
    int* err;
    u32* qbad;             // [Q] out: 1 = this query's bet was lost (rerun it exactly), 0 = ranked
    // several shards: the plan needs the gathered histograms, so the kernel runs twice --
    // mode 1 = histogram phase only (per-wave histograms -> hwq, shard totals -> hown, then the
    // exchange and k_plan), mode 2 = placement phase only, plan taken from the arrays below.
    int mode;              // 0 fused (one shard), 1 histogram phase, 2 placement phase
    u32* hwq;              // [Q][NWAV][NB] per-wave histograms between the two phases
    u32* hown;             // [NB][Qpad] (+ tail) shard histogram out (mode 1)
    const int* xt;         // external plan (mode 2): threshold
    const u32* xcnt_lt;
    const u32* xquota;
    const u32* xtie_before;
    const u32* xposbase;   // [NB][Qpad]
    u32 cap;
    i64 crow;
    int dense;
    int want_lists;
    int bits_lds;
    i64 RW;
    const u32* only;       // optional [Q]: handle only the flagged queries (the rest were ranked by k_rank_cnt / k_rank_lean)
    // direct mode (R = N on one shard): there are no records -- "record" i of a query is row i of the shard,
    // its distance and match bit are computed from the codes and labels on the fly, in both passes
    int direct;
    int rec8;              // records are one byte {match:1 | dist:7} (compact select) instead of 8 bytes
    const u32* db;         // [N][NW]
    const u64* dblab;      // [N][LW]
    const u32* qc;         // [Q][NW]
    const u64* qlab;       // [Q][LW]
};

template <int NWAV>   // wavefronts per query: 4, or 16 for long lists
__global__ __launch_bounds__(NWAV * 64) void k_rank_fused(const u64* __restrict__ cand, const RankArgs a,
                                                    u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                                    u32* __restrict__ mbits32, int nbits, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = g.NB;
    constexpr int nthr = NWAV * 64;
    constexpr int nwav = NWAV;
    const int bmw = a.bits_lds ? (int)(2 * a.RW) : 0;
    u32* hw = lds;                    // [nwav][NB]  per-wave histograms, then per-wave bucket positions
    u32* tot = hw + nwav * NB;        // [NB]     bucket totals, then global bucket starts
    u32* misc = tot + NB;             // [8]      t, cnt_lt, quota
    u32* bm = misc + 8;               // [bmw]
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;
    if (a.only && !a.only[q]) return;
    if (a.fail[q]) {                                  // a slice of this query overflowed
        if (tid == 0) {
            if (a.mode == 1 || a.mode == 3) atomicOr(&a.hown[(i64)NB * g.Qpad], 1u);   // tail word 0: this shard lost the bet
            else { atomicExch(a.err, 1); a.qbad[q] = 1u; }
        }
        if (a.mode == 1 || a.mode == 3) for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = 0u;
        return;
    }
    for (int i = tid; i < (nwav + 1) * NB + 8 + bmw; i += nthr) lds[i] = 0u;
    __syncthreads();
    const u64* __restrict__ row = cand + (i64)q * a.crow;
    // slice mode: S slices of capacity cap; dense mode: the run of tot[q] records cut into chunks of cap
    const u32 dense_tot = a.direct ? (u32)g.N : (a.dense ? a.tot[q] : 0u);
    const int nsl = a.dense ? (int)((dense_tot + a.cap - 1) / a.cap) : g.S;
    const int s0 = (int)((i64)nsl * wave / nwav), s1 = (int)((i64)nsl * (wave + 1) / nwav);
    // A wave walks its slices 64 records per step.  Global-load latency, not work, bounds this
    // kernel, so the walker runs one step ahead: the next step's records (and the next slice's
    // count) are requested before the current step is processed.  Records are read up to the slice
    // CAPACITY and masked by the count afterwards, so the two loads do not depend on each other.
    struct Walk { int s; u32 base, cnt; };
    // slice counts of this wave's range, 64 at a time in one vector load (lane i <-> slice c0 + i);
    // a slice's count is then a v_readlane away instead of a dependent scalar load per slice
    int c0 = s0;
    u32 cnts = (!a.dense && s0 + lane < s1) ? a.sl_cnt[(i64)(s0 + lane) * g.Qpad + q] : 0u;
    auto slice_cnt = [&](int s) -> u32 {
        if (s >= s1) return 0u;
        if (a.dense) {
            const u32 left = dense_tot - (u32)s * a.cap;
            return left < a.cap ? left : a.cap;
        }
        if (s >= c0 + 64 || s < c0) {                  // uniform: refill the window (ranges longer than 64 slices)
            c0 = s;
            cnts = (s + lane < s1) ? a.sl_cnt[(i64)(s + lane) * g.Qpad + q] : 0u;
        }
        return (u32)__builtin_amdgcn_readlane((int)cnts, s - c0);
    };
    auto first = [&]() { Walk w{s0, 0u, slice_cnt(s0)}; return w; };
    auto next = [&](Walk w) {
        w.base += 64;
        if (w.base >= a.cap || w.base >= w.cnt) { ++w.s; w.base = 0; w.cnt = slice_cnt(w.s); }
        return w;
    };
    u32 dq[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};    // direct mode: the query's code and label words
    u64 dl[2] = {0ull, 0ull};
    if (a.direct) {
        for (int w = 0; w < g.NW && w < 8; ++w) dq[w] = a.qc[(i64)q * g.NW + w];
        for (int w = 0; w < g.LW && w < 2; ++w) dl[w] = a.qlab[(i64)q * g.LW + w];
    }
    auto fetch = [&](const Walk& w) -> u64 {
        const u32 i = w.base + lane;
        if (!(w.s < s1 && i < a.cap)) return 0ull;
        // dense mode: the run ends at tot[q], and the buffer may end with the last query's run -- never read past it
        // (slice mode reads up to the slice's capacity, which is allocated; the count masks the rest afterwards)
        if (a.dense && !a.direct && (u32)w.s * a.cap + i >= dense_tot) return 0ull;
        if (a.direct) {
            const i64 n = (i64)w.s * a.cap + i;               // row of the shard
            if (n >= g.N) return 0ull;
            u32 d = 0;
            for (int k = 0; k < g.NW && k < 8; ++k) d += (u32)__builtin_popcount(dq[k] ^ a.db[n * g.NW + k]);
            u64 any = 0;
            for (int k = 0; k < g.LW && k < 2; ++k) any |= dl[k] & a.dblab[n * g.LW + k];
            return make_rec(g.idx_base + (u32)n, d, any != 0);
        }
        if (a.rec8) {
            const u32 m = ((const u8*)cand)[(i64)q * a.crow + (i64)w.s * a.cap + i];
            return (u64)((m & 0x7Fu) | ((m >> 7) << 8)) << 32;
        }
        return row[(i64)w.s * a.cap + i];
    };
    // phase 1
    u32* myh = hw + wave * NB;
    if (a.mode == 2) {                                // histograms were computed by the mode-1 launch
        for (int i = tid; i < nwav * NB; i += nthr) hw[i] = a.hwq[(i64)q * nwav * NB + i];
    } else {
        Walk w = first();
        u64 rec = fetch(w);
        Walk w1 = next(w);
        u64 rec1 = fetch(w1);
        Walk w2 = next(w1);
        u64 rec2 = fetch(w2);
        while (w.s < s1) {
            const Walk w3 = next(w2);                 // three steps of records in flight per wavefront
            const u64 rec3 = fetch(w3);
            if (w.base + lane < w.cnt) {
                const u32 d = (u32)(rec >> 32) & 0xFFu;
                if (d < (u32)NB) atomicAdd(&myh[d], 1u);
            }
            w = w1; rec = rec1;
            w1 = w2; rec1 = rec2;
            w2 = w3; rec2 = rec3;
        }
    }
    __syncthreads();
    // phase 2
    for (int d = tid; d < NB; d += nthr) {
        u32 acc = 0;
        for (int w = 0; w < nwav; ++w) acc += hw[w * NB + d];
        tot[d] = acc;
    }
    __syncthreads();
    if (a.mode == 1) {                                // hand the histograms over and stop
        for (int i = tid; i < nwav * NB; i += nthr) a.hwq[(i64)q * nwav * NB + i] = hw[i];
        for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = tot[d];
        return;
    }
    if (a.mode == 3) {                                // counts for the merge, before the plan turns tot[] into starts
        for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = tot[d];
        __syncthreads();
    }
    if (a.mode == 2) {
        if (tid == 0) {
            const int t = a.xt[q];
            int dmin = 0;
            while (dmin < NB - 1 && tot[dmin] == 0u) ++dmin;
            misc[0] = (u32)t;
            misc[1] = a.xcnt_lt[q];
            misc[2] = a.xquota[q];
            misc[3] = (u32)dmin;
            a.qbad[q] = t < 0 ? 1u : 0u;
        }
        __syncthreads();
        for (int d = tid; d < NB; d += nthr)          // my rows of bucket d start here in the global list
            tot[d] = a.xposbase[(i64)d * g.Qpad + q];
        __syncthreads();
    }
    if ((a.mode == 0 || a.mode == 3) && tid == 0) {
        // mode 3 (local ranking for k_merge_ranked): rank whatever this shard has, up to R -- never "lost" here
        u64 want = (u64)g.R;
        if (a.mode == 3) {
            u64 have = 0;
            for (int d = 0; d < NB; ++d) have += tot[d];
            if (have < want) want = have;
        }
        u64 cum = 0;
        int t = -1, dmin = -1;
        if (want > 0)
            for (int d = 0; d < NB; ++d) {
                const u32 c = tot[d];
                if (c && dmin < 0) dmin = d;
                tot[d] = (u32)cum;                    // global start of bucket d
                if (cum + c >= want) { t = d; break; }
                cum += c;
            }
        misc[0] = (u32)t;
        misc[1] = (u32)cum;                           // cnt_lt
        misc[2] = (u32)(want - cum);                  // quota
        misc[3] = (u32)(dmin < 0 ? 0 : dmin);         // smallest distance present
        if (a.mode == 0) {
            if (t < 0) atomicExch(a.err, 1);          // the superset is too small: bet lost
            a.qbad[q] = t < 0 ? 1u : 0u;
        }
    }
    __syncthreads();
    const int t = (int)misc[0];
    if (t < 0) {
        if (a.mode == 3) for (int w = tid; w < (int)(2 * a.RW); w += nthr) grow[w] = 0u;   // nothing to rank: an empty bitmap
        return;
    }
    const u32 tie0 = a.mode == 2 ? a.xtie_before[q] : 0u;    // ties owned by lower-ranked shards
    for (int d = tid; d <= t && d < NB; d += nthr) {  // per-wave starts: bucket start + records of earlier waves
        u32 run = d < t ? tot[d] : tie0;              // for d == t the "start" is the tie rank offset
        for (int w = 0; w < nwav; ++w) {
            const u32 h = hw[w * NB + d];
            hw[w * NB + d] = run;
            run += h;
        }
    }
    __syncthreads();
    // phase 3
    u32* pb = hw + wave * NB;
    const u32 cntlt = misc[1], quota = misc[2];
    // ranked distances of this query span [dmin, t): match on d - dmin, usually 4 bits instead of nbits
    const u32 dmin = misc[3];
    int kb = 0;
    while (kb < nbits && (int)(dmin + (1u << kb)) < t) ++kb;
    u32 tie_run = pb[t];                              // ties owned by earlier waves
    u32* __restrict__ oi = out_idx + (i64)q * g.R;
    u8* __restrict__ od = out_dist + (i64)q * g.R;
    const u64 below = (1ull << lane) - 1ull;
    {
        Walk w = first();
        u64 rec = fetch(w);
        Walk w1 = next(w);
        u64 rec1 = fetch(w1);
        Walk w2 = next(w1);
        u64 rec2 = fetch(w2);
        while (w.s < s1) {
            const Walk w3 = next(w2);
            const u64 rec3 = fetch(w3);
            const bool valid = w.base + lane < w.cnt;
            const u32 gi = (u32)rec;
            const u32 meta = (u32)(rec >> 32);
            const u32 d = meta & 0xFFu;
            const bool is_lt = valid && (int)d < t;
            const bool is_tie = valid && (int)d == t;
            u64 peers = __ballot(is_lt);
            const u32 key = d - dmin;
            for (int k = 0; k < kb; ++k) {
                const bool bit = (key >> k) & 1u;
                const u64 m = __ballot(is_lt && bit);
                peers &= bit ? m : ~m;
            }
            const u64 tmask = __ballot(is_tie);
            u32 pos = IDX_NONE;
            if (is_lt) {
                const u32 rank = (u32)__popcll(peers & below);
                const u32 npeer = (u32)__popcll(peers);
                const u32 start = pb[d];
                pos = start + rank;
                if (rank == npeer - 1) pb[d] = start + npeer;
            } else if (is_tie) {
                const u32 gr = tie_run + (u32)__popcll(tmask & below);
                if (gr < quota) pos = cntlt + gr;
            }
            tie_run += (u32)__popcll(tmask);
            if (pos != IDX_NONE) {
                if (a.want_lists) { oi[pos] = gi; od[pos] = (u8)d; }
                if (meta & 0x100u) {
                    if (a.bits_lds) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
                    else atomicOr(&grow[pos >> 5], 1u << (pos & 31));
                }
            }
            wave_lds_sync();
            w = w1; rec = rec1;
            w1 = w2; rec1 = rec2;
            w2 = w3; rec2 = rec3;
        }
    }
    if (a.bits_lds) {
        __syncthreads();
        for (int w = tid; w < bmw; w += nthr) grow[w] = bm[w];
    }
}

// ----------------------------------------------------------------------------
// K4m  merge of per-shard rankings (sharded bet, AP only).  Every shard has ranked its own records
// (k_rank_cnt / k_rank_lean / k_rank_fused mode 3): a match bitmap in LOCAL rank order -- distance ascending, index
// ascending -- and its per-distance record counts.  Shards own contiguous index ranges, so the global order
// is: for each distance d, shard 0's bucket d, then shard 1's, ...  One wavefront per query (lane r <->
// shard r) derives the global cut from the gathered counts exactly like k_plan and stitches the global
// bitmap together from bit ranges of the local ones.  A shard's records of global rank < R all have local
// rank < R, so the local top-R lists suffice.  No second pass over the records, one exchange less.
//   hall: [G][NB * Qpad + TAIL_WORDS] gathered counts (+ overflow flag in tail word 0)
//   ball: [G][Q * RW] gathered local bitmaps (64-bit words)
// ----------------------------------------------------------------------------
// use_lds: the G local bitmap rows of the query are first copied into LDS with all loads in flight (the
// stitching reads them bit range by bit range, one dependent load per 64 bits otherwise).
// q0, q1: the queries this launch merges (a rank of the sharded bet takes its own share of them: hg_merge_ap_part).
// Where the gathered tables live: all-gathered whole tables (every rank holds every query's rows of every shard), or the
// owner-routed blocks of an all-to-all (this rank holds only ITS queries' rows of every shard -- hg_pack_ranked_by_owner).
struct MergeSrc {
    i64 hstride;     // u32 words between two shards' count tables
    int hq;          // queries per row of a count table (a table is [NB][hq], q fastest)
    i64 tail_off;    // word offset of a shard's tail (overflow flag in word 0) inside its block
    i64 bstride;     // u64 words between two shards' bitmap tables
    int qoff;        // first query the tables hold (row index = q - qoff)
};
static __global__ __launch_bounds__(256) void k_merge_ranked(const u32* __restrict__ hall, const u64* __restrict__ ball, int G,
                                                      i64 RW, u64* __restrict__ out, int* __restrict__ err,
                                                      u32* __restrict__ qbad, int use_lds, const Geo g, const int q0, const int q1, const MergeSrc ms) {
    extern __shared__ __attribute__((aligned(16))) u64 mrows[];       // [WPB][G][RW] when use_lds
    const int lane = threadIdx.x & 63;
    const int q = q0 + blockIdx.x * WPB + (threadIdx.x >> 6);
    if (q >= q1) return;
    const int wv = threadIdx.x >> 6;
    u64* lrows = mrows + (i64)wv * G * RW;
    // the query's record counts of all shards, [G][NB], fetched with every load in flight
    u32* lcnt = (u32*)(mrows + (use_lds ? (i64)WPB * G * RW : 0)) + (i64)wv * G * g.NB;
    for (int i = lane; i < G * g.NB; i += 64) {
        const int r = i / g.NB, d = i - r * g.NB;
        lcnt[i] = hall[(i64)r * ms.hstride + (i64)d * ms.hq + (q - ms.qoff)];
    }
    if (use_lds) {
        for (i64 i = lane; i < (i64)G * RW; i += 64) {
            const i64 r = i / RW, w = i - r * RW;
            lrows[i] = ball[r * ms.bstride + (i64)(q - ms.qoff) * RW + w];
        }
    }
    wave_lds_sync();
    const bool mine = lane < G;                                       // lane r speaks for shard r (G <= 64)
    if (q == q0 && mine && hall[(i64)lane * ms.hstride + ms.tail_off]) atomicExch(err, 1);   // a slice overflowed somewhere
    u64* __restrict__ orow = out + (i64)q * RW;
    u64 acc = 0;                // output bits not yet written (wave-uniform), `fill` of them
    int fill = 0;
    i64 wi = 0;                 // next output word
    u64 cum = 0;                // records of all shards closer than d
    u32 loff = 0;               // this shard's records closer than d = where its bucket d starts in its bitmap
    bool done = false;
    for (int d = 0; d < g.NB && !done; ++d) {
        const u32 c = mine ? lcnt[lane * g.NB + d] : 0u;
        u32 tot = c, pre = c;                                          // wave sum and inclusive prefix over the shards
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = (u32)__shfl_up((int)pre, off);
            if (lane >= off) pre += v;
        }
        tot = (u32)__shfl((int)pre, 63);
        u32 take = c;
        if (cum + tot >= (u64)g.R) {                                   // the cut falls into this distance: ties by (shard, index)
            const u64 quota = (u64)g.R - cum;
            const u64 before = (u64)(pre - c);
            take = before >= quota ? 0u : (u32)((quota - before) < c ? (quota - before) : c);
            done = true;
        }
        for (int r = 0; r < G; ++r) {                                  // append shard r's `take` bits of bucket d
            const u32 n = (u32)__builtin_amdgcn_readlane((int)take, r);
            const u32 so = (u32)__builtin_amdgcn_readlane((int)loff, r);
            const u64* src = use_lds ? lrows + (i64)r * RW : ball + (i64)r * ms.bstride + (i64)(q - ms.qoff) * RW;
            for (u32 k0 = 0; k0 < n; k0 += 64) {
                const u32 k = k0 + lane;
                const bool bit = k < n && ((src[(so + k) >> 6] >> ((so + k) & 63)) & 1ull);
                const u64 chunk = __ballot(bit);
                const int nv = (int)(n - k0 < 64 ? n - k0 : 64);
                acc |= chunk << fill;
                if (fill + nv >= 64) {
                    if (lane == 0) orow[wi] = acc;
                    ++wi;
                    acc = fill ? chunk >> (64 - fill) : 0ull;
                    fill = fill + nv - 64;
                } else {
                    fill += nv;
                }
            }
        }
        loff += c;
        cum += tot;
    }
    if (fill && wi < RW) { if (lane == 0) orow[wi] = acc; ++wi; }
    for (i64 w = wi + lane; w < RW; w += 64) orow[w] = 0ull;
    if (lane == 0) {
        const bool lost = cum < (u64)g.R;                              // fewer than R records over all shards: bet lost
        qbad[q] = lost ? 1u : 0u;
        if (lost) atomicExch(err, 1);
    }
}

// ----------------------------------------------------------------------------
// K4o  the sharded bet's exchanges ROUTED BY QUERY OWNER (round 4).  The queries are split over the ranks exactly like
// the per-query stages (hashgan_amd.sharded.shard_bounds: rank o owns [q0(o), q0(o) + nq(o))); what a stage needs of a
// query it needs from every shard, but only on the query's owner -- so the tables travel by all-to-all, one block per
// destination, instead of every rank receiving every query's rows of every shard (80 MB of ingress per GPU and step at
// C4 / 8 GPUs with all-gathers, 10.6 MB this way).
// ----------------------------------------------------------------------------
struct Owners { int G, per, extra, width; };          // shard_bounds(Q, G): per = Q / G, the first `extra` ranks own one more
__host__ __device__ inline int owner_q0(const Owners& w, int o) { return o * w.per + (o < w.extra ? o : w.extra); }
__host__ __device__ inline int owner_nq(const Owners& w, int o) { return w.per + (o < w.extra ? 1 : 0); }
__host__ __device__ inline int owner_of(const Owners& w, int q) {
    const int big = w.extra * (w.per + 1);            // queries owned by the ranks that own per + 1
    return q < big ? q / (w.per + 1) : w.extra + (w.per ? (q - big) / w.per : 0);
}
inline Owners make_owners(i64 Q, int G) { Owners w; w.G = G; w.per = (int)(Q / G); w.extra = (int)(Q % G); w.width = w.per + (w.extra ? 1 : 0); return w; }

// Exchange 1 out: the sampled shard histogram, cut by owner.  Block o = u32 [4 + HC * width]: [1] = rows this shard's sampled
// pass visited, [4 + d * width + i] = sampled rows at distance d < HC of query q0(o) + i.
static __global__ __launch_bounds__(256) void k_pack_sample_owner(const u32* __restrict__ hown, const Owners w, const int HC,
                                                                   u32* __restrict__ out, const Geo g) {
    const i64 blk = 4 + (i64)HC * w.width;
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (i64)w.G * HC * w.width) return;
    const int o = (int)(idx / ((i64)HC * w.width));
    const int rem = (int)(idx - (i64)o * HC * w.width);
    const int d = rem / w.width, i = rem - d * w.width;
    out[o * blk + 4 + rem] = i < owner_nq(w, o) ? hown[(i64)d * g.Qpad + owner_q0(w, o) + i] : 0u;
    if (rem < 4) out[o * blk + rem] = rem == 1 ? hown[(i64)g.NB * g.Qpad + 1] : 0u;
}

// On the owner: the guess of ITS queries from the G received blocks (k_guess's arithmetic), answered per shard -- block r
// = u32 [width][4] {T, sampled rows below the cut's ties that precede shard r's own (= k_guess's `have` before its
// segments), the sample count the cut must reach, found}.
static __global__ __launch_bounds__(256) void k_guess_owner(const u32* __restrict__ recv, const Owners w, const int HC, const int nq,
                                                             const double sigma, const i64 n_total, u32* __restrict__ out, const Geo g) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.width) return;
    const i64 blk = 4 + (i64)HC * w.width;
    u64 sampled = 0;
    for (int r = 0; r < w.G; ++r) sampled += recv[r * blk + 1];
    const double fr = (double)g.R * (double)sampled / (double)n_total;
    const u64 need = (u64)ceil(fr + sigma * sqrt(fr) + 1.0);
    u64 cum = 0, below = 0;
    int t = g.NB - 1;                                  // sample too thin (or the cut beyond the planes sent): take everything
    bool found = false;
    if (i < nq) {
        for (int d = 0; d < HC; ++d) {
            below = cum;
            for (int r = 0; r < w.G; ++r) cum += recv[r * blk + 4 + (i64)d * w.width + i];
            if (cum >= need) { t = d; found = true; break; }
        }
    }
    u64 have = below;
    for (int r = 0; r < w.G; ++r) {
        u32* o4 = out + ((i64)r * w.width + i) * 4;
        o4[0] = (u32)t;
        o4[1] = (u32)(have > 0xFFFFFFFFull ? 0xFFFFFFFFull : have);
        o4[2] = (u32)(need > 0xFFFFFFFFull ? 0xFFFFFFFFull : need);
        o4[3] = found ? 1u : 0u;
        if (found) have += recv[r * blk + 4 + (i64)t * w.width + i];
    }
}

// On every rank: the owners' answers (block o = the queries o owns) + this shard's per-segment sampled histograms ->
// the shared cut T and this shard's sstar, exactly as k_guess derives them from all-gathered tables.
static __global__ __launch_bounds__(256) void k_guess_finish(const u32* __restrict__ ans, const Owners w, const u32* __restrict__ hseg,
                                                              const int Sh, const int ratio, int* __restrict__ T, int* __restrict__ sstar,
                                                              u32* __restrict__ beyond, const Geo g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.Q) return;
    const int o = owner_of(w, q), i = q - owner_q0(w, o);
    const u32* a4 = ans + ((i64)o * w.width + i) * 4;
    const int t = (int)a4[0];
    const u64 need = a4[2];
    int ss = g.S - 1;                                  // default: collect distance T everywhere
    // the owner found no cut within the b/2 + 2 planes it was sent: this query takes every row up to the last plane and overflows
    // whatever the slices' capacity -- the caller reads *beyond (stat "cut_beyond_planes") and goes to the exact sequence at once
    if (!a4[3]) *beyond = 1u;
    if (a4[3]) {
        u64 have = a4[1];
        if (have >= need) {
            ss = -1;                                   // the lower shards already hold the prefix
        } else {
            for (int sh = 0; sh < Sh; ++sh) {
                have += hseg[((i64)sh * g.NB + t) * g.Qpad + q];
                if (have >= need) { ss = (sh + 1) * ratio - 1; break; }
            }
            if (ss > g.S - 1) ss = g.S - 1;
        }
    }
    T[q] = t;
    sstar[q] = ss;
}

// Exchange 2 out: this shard's per-distance record counts and its match bitmap in LOCAL rank order (hg_select_ranked), cut
// by owner.  Block o = u32 counts [cw = NB * width rounded up to even], u32 tail [TAIL_WORDS] ([0] = this shard's overflow
// flag), u64 bits [width][RW].
static __global__ __launch_bounds__(256) void k_pack_ranked_owner(const u32* __restrict__ hown, const u64* __restrict__ mbits, const Owners w,
                                                                   const i64 RW, const i64 cw, u32* __restrict__ out, const Geo g) {
    const i64 blk32 = cw + TAIL_WORDS + 2 * (i64)w.width * RW;       // u32 words per block
    const i64 ncnt = (i64)w.G * (cw + TAIL_WORDS), nbit = (i64)w.G * w.width * RW;
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (idx < ncnt) {
        const int o = (int)(idx / (cw + TAIL_WORDS));
        const i64 rem = idx - (i64)o * (cw + TAIL_WORDS);
        u32 v = 0;
        if (rem < (i64)g.NB * w.width) {
            const int d = (int)(rem / w.width), i = (int)(rem - (i64)d * w.width);
            if (i < owner_nq(w, o)) v = hown[(i64)d * g.Qpad + owner_q0(w, o) + i];
        } else if (rem >= cw) {
            v = hown[(i64)g.NB * g.Qpad + (rem - cw)];               // the tail as it is
        }
        out[o * blk32 + rem] = v;
    } else if (idx < ncnt + nbit) {
        const i64 k = idx - ncnt;
        const int o = (int)(k / ((i64)w.width * RW));
        const i64 rem = k - (i64)o * w.width * RW;
        const int i = (int)(rem / RW);
        u64* bits = (u64*)(out + o * blk32 + cw + TAIL_WORDS);
        bits[rem] = i < owner_nq(w, o) ? mbits[(i64)(owner_q0(w, o) + i) * RW + (rem - (i64)i * RW)] : 0ull;
    }
}

// ----------------------------------------------------------------------------
// K5  label match.   metric.py:17-19: slot k of query q matches iff the ranked
// row shares a positive label with the query.  One bit per slot, 64 slots per
// wavefront via ballot.  Slots owned by another shard (IDX_NONE) give 0.
// ----------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_match(const u32* __restrict__ out_idx, const u64* __restrict__ dblab,
                                               const u64* __restrict__ qlab, u64* __restrict__ mbits,
                                               i64 RW, int nKB, const Geo g) {
    const int q = (int)(blockIdx.x / (u32)nKB);          // nKB = ceil(R / 256) blocks per query
    const i64 k = (i64)(blockIdx.x - (u32)q * (u32)nKB) * 256 + threadIdx.x;
    bool m = false;
    if (k < g.R) {
        const u32 gi = out_idx[(i64)q * g.R + k];
        // (a query that lost its bet leaves its list row as it found it -- stale words of an earlier allocation; its
        // bits are recomputed by the rerun, but the gather must not follow them out of the table)
        if (gi != IDX_NONE && (i64)(gi - g.idx_base) < g.N) {
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

// tail of an exported histogram: [0] overflow flag (cleared), [1] rows the pass visited, the rest 0
static __global__ void k_set_tail(u32* __restrict__ tail, u32 visited) {
    if (threadIdx.x < TAIL_WORDS) tail[threadIdx.x] = threadIdx.x == 1 ? visited : 0u;
}

// OR of G shards' bit rows (disjoint by construction).
static __global__ __launch_bounds__(256) void k_or_bits(const u64* __restrict__ all, u64* __restrict__ out, i64 n, int G) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 v = 0;
    for (int r = 0; r < G; ++r) v |= all[(i64)r * n + i];
    out[i] = v;
}

// min over G shards' ranked lists: exactly one shard owns a slot, the others hold IDX_NONE / 0xFF.
static __global__ __launch_bounds__(256) void k_min_topr(const u32* __restrict__ idx_all, const u8* __restrict__ dist_all,
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
// node table (ApShape); a thread evaluates one leaf, then the internal nodes are summed level by level.
// ----------------------------------------------------------------------------
struct ApShape {               // tree of one chunk length (<= AP_CHUNK elements)
    int n;                     // chunk length
    int n_leaves;              // <= 128
    int n_prog;                // <= 255
    unsigned short leaf_start[AP_LEAF];
    unsigned short leaf_len[AP_LEAF];
    short prog[2 * AP_LEAF];   // >= 0: push leaf, -1: add the two on top (the tree in postfix; kept for reference)
    // the same tree by levels: node ids 0 .. n_leaves-1 are the leaves, internal node k has id n_leaves + k,
    // children nl/nr (always left + right: the order of NumPy's additions), height nh = 1 + max(children)
    int n_nodes;               // internal nodes = n_leaves - 1
    int max_h;
    short nl[AP_LEAF], nr[AP_LEAF];
    unsigned char nh[AP_LEAF];
};

__device__ __forceinline__ u32 count_bits_below(const u64* cw, int e) {  // bits [0, e) of the chunk
    u32 c = 0;
    const int full = e >> 6;
    for (int w = 0; w < full; ++w) c += (u32)__popcll(cw[w]);
    const int rem = e & 63;
    if (rem) c += (u32)__popcll(cw[full] & ((1ull << rem) - 1ull));
    return c;
}

// recip[k] = RN(1 / k), k = 1 .. n: what turns k_ap's division into three multiply-adds (below)
static __global__ __launch_bounds__(256) void k_recip_table(double* __restrict__ recip, i64 n) {
    const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
    if (k <= n) recip[k] = k ? 1.0 / (double)k : 0.0;
}

// LDS of one AP evaluation (k_ap: static arrays; k_rank_cnt's epilogue: carved out of its counters, free by then)
struct ApLds {
    u64* cw;       // [AP_CHUNK / 64] the chunk's match bits
    double* tree;  // [2 * AP_LEAF]   leaf sums, then the sums of the internal nodes
    u32* wpre;     // [AP_CHUNK / 64 + 1] matches in the chunk's words before word w
    u32* sb;       // [2] matches before the chunk; first wavefront's total
};
constexpr int AP_LDS_BYTES = AP_CHUNK / 8 + 2 * AP_LEAF * 8 + (AP_CHUNK / 64 + 1) * 4 + 12;    // cw, tree, wpre (+ pad to 8), sb
__device__ __forceinline__ ApLds ap_lds_at(u8* base) {     // base 8-byte aligned
    ApLds l;
    l.cw = (u64*)base;
    l.tree = (double*)(base + AP_CHUNK / 8);
    l.wpre = (u32*)(base + AP_CHUNK / 8 + 2 * AP_LEAF * 8);
    l.sb = l.wpre + AP_CHUNK / 64 + 2;
    return l;
}

// The AP of one query by a block of NT threads (NT >= 128, every thread of the block calls this -- it has barriers);
// word(w) = 64-bit word w of the query's match-bit row (0 beyond RW).  Thread 0 returns through *ap_out / *rel_out.
template <int NT, class WordFn>
__device__ __forceinline__ void ap_eval(const WordFn& word_at, const i64 RW, const i64 R, const ApShape* __restrict__ shapes,
                                        const double* __restrict__ recip, const ApLds& L, const int tid,
                                        double* __restrict__ ap_out, u32* __restrict__ rel_out) {
    static_assert(NT >= AP_CHUNK / 64 && NT % 64 == 0, "one thread per word of a chunk");
    u64* cw = L.cw;
    double* tree = L.tree;
    u32* wpre = L.wpre;
    u32& s_before = L.sb[0];
    u32& s_w0 = L.sb[1];
    if (tid == 0) s_before = 0;
    double total = 0.0;        // thread 0 only
    const i64 n_chunks = (R + AP_CHUNK - 1) / AP_CHUNK;
    for (i64 c = 0; c < n_chunks; ++c) {
        const i64 cb = c * AP_CHUNK;
        const bool last = (c == n_chunks - 1);
        const ApShape* __restrict__ sh = shapes + ((last && (R - cb) != AP_CHUNK) ? 1 : 0);
        const int n = (int)(R - cb < AP_CHUNK ? R - cb : AP_CHUNK);
        if (tid < AP_CHUNK / 64) {   // load the chunk's words and prefix their popcounts (two wavefronts, 64 words each)
            const i64 w = (cb >> 6) + tid;
            const u64 word = (w < RW) ? word_at(w) : 0ull;
            cw[tid] = word;
            u32 incl = (u32)__popcll(word);
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = (u32)__shfl_up((int)incl, off);
                if ((tid & 63) >= off) incl += v;
            }
            if (tid == 63) s_w0 = incl;
            wpre[tid + 1] = incl;                              // the second wave's values still lack the first's total
            if (tid == 0) wpre[0] = 0u;
        }
        __syncthreads();
        if (tid >= 64 && tid < AP_CHUNK / 64) wpre[tid + 1] += s_w0;
        __syncthreads();
        const u32 before = s_before;
        const bool sparse = wpre[AP_CHUNK / 64] * 4u < (u32)n;   // block-uniform: fewer than one slot in four matches
        // value of element e of the chunk: (matches up to and including e) / (its 1-based rank), 0 without a match
        auto val = [&](const int e) -> double {
            const u64 word = cw[e >> 6];
            const int bpos = e & 63;
            if (!((word >> bpos) & 1ull)) return 0.0;
            const u32 cnt = before + wpre[e >> 6] + (u32)__popcll(word & ((2ull << bpos) - 1ull));
            const double a = (double)cnt, b = (double)(cb + e + 1);
            if (!recip) return a / b;
            // A division is a dozen double-rate instructions around v_rcp_f64 per element (0.069 -> 0.060 ms at C2 without).
            // With y = RN(1 / b) from a table shared by all queries, q = RN(a y), r = a - b q (exact in one fma) and
            // q' = RN(q + r y) is the correctly rounded a / b (Markstein's final step; checked on the GPU against the
            // division for ALL 1 <= a <= b <= 131072 and 1.7e10 random pairs below 2^31: tools/ap_div_check.hip).
            const double y = recip[cb + e + 1];
            const double q = a * y;
            return __builtin_fma(__builtin_fma(-q, b, a), y, q);
        };
        // Eight lanes per leaf: NumPy's leaf sum runs eight strided accumulators (element i -> r[i % 8], in order) and
        // combines them as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).  Lane j of a leaf's group owns r_j; the xor butterfly
        // 1, 2, 4 is exactly that tree (IEEE addition is commutative); lane 0 adds the < 8 tail elements in order.
        const int nl = sh->n_leaves;
        for (int l0 = 0; l0 < nl; l0 += NT / 8) {
            const int leaf = l0 + (tid >> 3), j = tid & 7;
            const bool act = leaf < nl;
            const int ls = act ? sh->leaf_start[leaf] : 0, ll = act ? sh->leaf_len[leaf] : 0;
            const int body = ll - (ll % 8);
            double r = 0.0;
            if (ll >= 8 && !sparse) {
                for (int e = ls + j; e < ls + body; e += 8) r += val(e);      // 0.0 + v == v: the first add is exact
            } else if (ll >= 8) {
                // r_j adds elements ls + j, ls + j + 8, ... in order; only MATCHING slots contribute (r + 0.0 == r exactly,
                // r >= 0), and they are few -- one in ten with ten classes -- so the lane walks the set bits of its stride-8
                // sub-mask instead of all 16 slots when the chunk is sparse (same additions in the same order; dense chunks --
                // trained codes put matches first -- keep the plain loop, which does not diverge)
#pragma unroll
                for (int sp = 0; sp < AP_LEAF / 64; ++sp) {
                    const int e0 = ls + 64 * sp;                              // first element of this 64-element span of the leaf
                    if (64 * sp >= body) break;
                    const int wi = e0 >> 6, bs = e0 & 63;
                    u64 win = cw[wi] >> bs;                                   // the span's 64 match bits (it may straddle two words)
                    if (bs) win |= cw[wi + 1 < AP_CHUNK / 64 ? wi + 1 : wi] << (64 - bs);
                    const int nv = body - 64 * sp;                            // valid elements of the span (a multiple of 8)
                    if (nv < 64) win &= (1ull << nv) - 1ull;
                    u64 m = (win >> j) & 0x0101010101010101ull;               // elements j, j + 8, ... of the span
                    while (m) {
                        const int p = __builtin_ctzll(m);
                        m &= m - 1ull;
                        r += val(e0 + j + p);
                    }
                }
            }
            r += __shfl_xor(r, 1);
            r += __shfl_xor(r, 2);
            r += __shfl_xor(r, 4);
            if (act && j == 0) {
                double res = ll >= 8 ? r : 0.0;
                for (int e = ls + (ll >= 8 ? body : 0); e < ls + ll; ++e) res += val(e);
                tree[leaf] = res;
            }
        }
        __syncthreads();
        // the tree, level by level: all nodes of one height are independent (thread k owns internal node k);
        // every addition is left + right exactly as in NumPy's recursion, only the schedule differs
        for (int hgt = 1; hgt <= sh->max_h; ++hgt) {
            if (tid < sh->n_nodes && sh->nh[tid] == hgt) tree[sh->n_leaves + tid] = tree[sh->nl[tid]] + tree[sh->nr[tid]];
            __syncthreads();
        }
        if (tid == 0) {
            const double chunk_sum = tree[sh->n_leaves + sh->n_nodes - 1];    // the root is the last node (or the only leaf)
            total = (c == 0) ? chunk_sum : total + chunk_sum;
            s_before = before + wpre[n >> 6] + ((n & 63) ? (u32)__popcll(cw[n >> 6] & ((1ull << (n & 63)) - 1ull)) : 0u);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const u32 r = s_before;
        *rel_out = r;
        *ap_out = r ? total / (double)r : __longlong_as_double(0x7FF8000000000000ll);
    }
}

// ap_eval with a third of the instructions (round 4; k_rank_lean's epilogue).  Same arithmetic, same order of additions:
//   * a lane (leaf, j) walks its stride-8 elements incrementally: 32 window bits per step come from two LDS reads and a
//     funnel shift, the running match count from v_bcnt of the window's bytes (no per-element prefix lookup, no 64-bit
//     masks), the rank as a double from b += 8.0, the reciprocal through an immediate offset from one per-lane pointer;
//     a slot without a match contributes a = 0 -> q = 0 -> +0.0, and r + 0.0 == r exactly, so no branch per element;
//   * the < 8 tail elements of a leaf are evaluated by lanes 0..6 of its group at once and added by lane 0 in order (DPP);
//   * the eight accumulators combine through DPP (xor 1, xor 2, half mirror) instead of ds_bpermute;
//   * the tree above the leaves is walked by wavefront 0 alone (its tables prefetched at the start): no block barrier per level.
// Needs the reciprocal table (with AP_RECIP_SLACK finite entries past R: masked slots still load); without one, ap_eval.
constexpr int AP_RECIP_SLACK = 160;
template <int CTRL> __device__ __forceinline__ double ap_dpp_mov(const double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int NT, class WordFn>
__device__ __forceinline__ void ap_eval2(const WordFn& word_at, const i64 RW, const i64 R, const ApShape* __restrict__ shapes,
                                         const double* __restrict__ recip, const ApLds& L, const int tid,
                                         double* __restrict__ ap_out, u32* __restrict__ rel_out) {
    static_assert(NT >= AP_CHUNK / 64 && NT % 64 == 0, "one thread per word of a chunk");
    u64* cw = L.cw;
    const u32* cw32 = (const u32*)L.cw;
    double* tree = L.tree;
    u32* wpre = L.wpre;
    u32& s_before = L.sb[0];
    u32& s_w0 = L.sb[1];
    if (tid == 0) s_before = 0;
    double total = 0.0;        // thread 0 only
    const i64 n_chunks = (R + AP_CHUNK - 1) / AP_CHUNK;
    for (i64 c = 0; c < n_chunks; ++c) {
        const i64 cb = c * AP_CHUNK;
        const bool last = (c == n_chunks - 1);
        const ApShape* __restrict__ sh = shapes + ((last && (R - cb) != AP_CHUNK) ? 1 : 0);
        const int n = (int)(R - cb < AP_CHUNK ? R - cb : AP_CHUNK);
        // the tree's tables (wavefront 0 walks it at the end) and the first leaves, requested before anything waits
        const int nn = sh->n_nodes, nlv = sh->n_leaves, max_h = sh->max_h;
        int tl[2] = {0, 0}, tr[2] = {0, 0}, th[2] = {0, 0};
        if (tid < 64) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (tid + 64 * k < nn) { tl[k] = sh->nl[tid + 64 * k]; tr[k] = sh->nr[tid + 64 * k]; th[k] = sh->nh[tid + 64 * k]; }
        }
        if (tid < AP_CHUNK / 64) {   // load the chunk's words and prefix their popcounts (two wavefronts, 64 words each)
            const i64 w = (cb >> 6) + tid;
            const u64 word = (w < RW) ? word_at(w) : 0ull;
            cw[tid] = word;
            u32 incl = (u32)__popcll(word);
            incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);
            incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);
            incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);
            incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);
            incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);
            incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);
            if (tid == 63) s_w0 = incl;
            wpre[tid + 1] = incl;                              // the second wave's values still lack the first's total
            if (tid == 0) wpre[0] = 0u;
        }
        __syncthreads();
        const u32 before = s_before;
        const u32 w0tot = s_w0;
        for (int l0 = 0; l0 < nlv; l0 += NT / 8) {
            const int leaf = l0 + (tid >> 3), j = tid & 7;
            const bool act = leaf < nlv;
            const int ls = act ? sh->leaf_start[leaf] : 0, ll = act ? sh->leaf_len[leaf] : 0;
            const int body = ll - (ll % 8);
            // matches before the leaf: (earlier chunks) + (words before its first) + (bits below it in that word); ls is a multiple of 8
            const int wi = ls >> 6;
            const u64 fw = cw[wi];
            u32 P = before + wpre[wi] + (wi > 64 ? w0tot : 0u) + (u32)__popcll(fw & ((1ull << (ls & 63)) - 1ull));
            const u32 lmask = (2u << j) - 1u;                  // bits 0..j of a byte: the elements of a group of eight up to the lane's
            const int wd = ls >> 5, bs = ls & 31;              // the leaf's first dword of the bitmap, its first bit inside it
            const double* __restrict__ rp = recip + (cb + ls + j + 1);
            double b = (double)(cb + ls + j + 1);
            double r = 0.0;
            for (int i4 = 0; __any(32 * i4 < body); ++i4) {    // 32 elements of the leaf = 4 of the lane's per step
                const u32 lo = cw32[wd + i4], hi = cw32[wd + i4 + 1];
                u32 W = __builtin_amdgcn_alignbit(hi, lo, (u32)bs);      // elements 32 i4 .. 32 i4 + 31 of the leaf
                const int nb = body - 32 * i4;
                W = nb >= 32 ? W : nb <= 0 ? 0u : W & ((1u << nb) - 1u);
                const double y0 = rp[0], y1 = rp[8], y2 = rp[16], y3 = rp[24];
                const double ys[4] = {y0, y1, y2, y3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 byte = (W >> (8 * k)) & 0xFFu;
                    const u32 cnt = P + (u32)__builtin_popcount(byte & lmask);
                    P += (u32)__builtin_popcount(byte);
                    const u32 bit = (byte >> j) & 1u;
                    const double a = (double)(cnt * bit);
                    const double q = a * ys[k];
                    r += __builtin_fma(__builtin_fma(-q, b, a), ys[k], q);
                    b += 8.0;
                }
                rp += 32;
            }
            r += ap_dpp_mov<0xB1>(r);                          // quad_perm [1, 0, 3, 2]: lane ^ 1
            r += ap_dpp_mov<0x4E>(r);                          // quad_perm [2, 3, 0, 1]: lane ^ 2
            r += ap_dpp_mov<0x141>(r);                         // row_half_mirror: the other quad of the eight
            // tail: element body + j of the leaf, for j < ll - body
            {
                const int tw = (ls + body) >> 5, tb = (ls + body) & 31;
                const u32 byte = __builtin_amdgcn_alignbit(cw32[tw + 1], cw32[tw], (u32)tb) & ((1u << (ll - body)) - 1u);
                const u32 cnt = P + (u32)__builtin_popcount(byte & lmask);
                const u32 bit = (byte >> j) & 1u;
                const double a = (double)(cnt * bit);
                const double bt = (double)(cb + ls + body + j + 1);
                const double y = recip[cb + ls + body + j + 1];
                const double q = a * y;
                const double v = __builtin_fma(__builtin_fma(-q, bt, a), y, q);      // +0.0 where there is no element
                double res = r;                                // (ll < 8: r is 0.0)
                res += v;                                      // lane 0 of the group: element body + 0
                res += ap_dpp_mov<0x101>(v);                   // row_shl:1 .. 7: the values of lanes j + 1 .. j + 7
                res += ap_dpp_mov<0x102>(v);
                res += ap_dpp_mov<0x103>(v);
                res += ap_dpp_mov<0x104>(v);
                res += ap_dpp_mov<0x105>(v);
                res += ap_dpp_mov<0x106>(v);
                res += ap_dpp_mov<0x107>(v);
                if (act && j == 0) tree[leaf] = res;
            }
        }
        __syncthreads();
        if (tid < 64) {
            // the tree, level by level: all nodes of one height are independent; every addition is left + right exactly as in
            // NumPy's recursion, only the schedule differs
            for (int hgt = 1; hgt <= max_h; ++hgt) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (tid + 64 * k < nn && th[k] == hgt) tree[nlv + tid + 64 * k] = tree[tl[k]] + tree[tr[k]];
                wave_lds_sync();
            }
            if (tid == 0) {
                const double chunk_sum = tree[nlv + nn - 1];    // the root is the last node (or the only leaf)
                total = (c == 0) ? chunk_sum : total + chunk_sum;
                s_before = before + wpre[n >> 6] + ((n >> 6) > 64 ? w0tot : 0u) + ((n & 63) ? (u32)__popcll(cw[n >> 6] & ((1ull << (n & 63)) - 1ull)) : 0u);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const u32 r = s_before;
        *rel_out = r;
        *ap_out = r ? total / (double)r : __longlong_as_double(0x7FF8000000000000ll);
    }
}

// only: optional [Q] -- evaluate just the flagged queries (the rest got their AP from k_rank_cnt's epilogue)
// NT threads per block: 128 when the queries alone fill the GPU; 512 for few queries with long lists (the reference's CIFAR-10 evaluation:
// 1000 queries x 54 000 ranks -- at 128 threads that is two wavefronts per SIMD walking seven chunks one after the other)
template <int NT>
static __global__ __launch_bounds__(NT) void k_ap(const u64* __restrict__ mbits, i64 RW, i64 R,
                                                   const ApShape* __restrict__ shapes,  // [0] full chunk, [1] last chunk
                                                   const double* __restrict__ recip,    // [R + 1 + AP_RECIP_SLACK] or null
                                                   double* __restrict__ ap, u32* __restrict__ rel, const u32* __restrict__ only) {
    __shared__ __attribute__((aligned(8))) u8 aplds[AP_LDS_BYTES];
    const int q = blockIdx.x;
    if (only && !only[q]) return;
    const u64* __restrict__ row = mbits + (i64)q * RW;
    // with the table of reciprocals: the rank kernels' epilogue (ap_eval2: the same additions in the same order, a third of the instructions)
    if (recip) ap_eval2<NT>([&](const i64 w) { return row[w]; }, RW, R, shapes, recip, ap_lds_at(aplds), (int)threadIdx.x, ap + q, rel + q);
    else ap_eval<NT>([&](const i64 w) { return row[w]; }, RW, R, shapes, recip, ap_lds_at(aplds), (int)threadIdx.x, ap + q, rel + q);
}

// ----------------------------------------------------------------------------
// K0  binarise + pack on device.  The step upstream of the metric: forward_all()
// (main.py:151-158) hands float32 features [n][b] and integer labels [n][C]; the
// reference never binarises (tanh outputs go straight into np.dot), the hashing
// evaluation does: bit j = (x[j] > 0).  One wavefront per row, 64 columns per ballot.
// Also counts entries outside {-1, 0, +1}, zeros and minus ones (codes) / entries outside {0, 1} (labels), so
// the host can tell +-1 codes from {0,1} bits from anything else (mixtures rank differently under np.dot than
// under a Hamming distance -- metric.py:13) instead of silently ranking something else.
// ----------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_pack_sign_f32(const float* __restrict__ x, i64 ld, u32* __restrict__ out, i64 n, int b,
                                                       int NW, unsigned long long* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n) return;
    u32 nbad = 0, nzero = 0, nneg = 0;
    for (int c0 = 0; c0 < b; c0 += 64) {
        const int col = c0 + lane;
        const float v = col < b ? x[r * ld + col] : 1.0f;     // ld: row pitch of the (zero-padded) feature table
        nbad += !(v == 1.0f || v == -1.0f || v == 0.0f);
        nzero += v == 0.0f;
        nneg += v == -1.0f;
        const u64 word = __ballot(col < b && v > 0.0f);
        const int w = c0 >> 5;
        if (lane == 0) {
            out[r * NW + w] = (u32)word;
            if (w + 1 < NW) out[r * NW + w + 1] = (u32)(word >> 32);
        }
    }
    // bad[0]: entries outside {-1, 0, +1}; bad[2]: zeros; bad[3]: minus ones (bad[1] belongs to the labels).
    // all +-1  <=>  bad[0] == 0 and bad[2] == 0;   all {0,1}  <=>  bad[0] == 0 and bad[3] == 0
    if (nbad) atomicAdd(bad, (unsigned long long)nbad);
    if (nzero) atomicAdd(bad + 2, (unsigned long long)nzero);
    if (nneg) atomicAdd(bad + 3, (unsigned long long)nneg);
}

static __global__ __launch_bounds__(256) void k_pack_labels_i64(const long long* __restrict__ lab, u64* __restrict__ out, i64 n, int C,
                                                         int LW, unsigned long long* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n) return;
    u32 nbad = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int col = c0 + lane;
        const long long v = col < C ? lab[r * C + col] : 0;
        nbad += !(v == 0 || v == 1);
        const u64 word = __ballot(v != 0);
        if (lane == 0) out[r * LW + (c0 >> 6)] = word;
    }
    if (nbad) atomicAdd(bad + 1, (unsigned long long)nbad);
}

// Row gather / scatter between a full query set and the compacted set of queries whose bet
// was lost: block i moves row (gather ? list[i] -> i : i -> list[i]) of `rowbytes` bytes.
static __global__ __launch_bounds__(256) void k_move_rows(const u8* __restrict__ src, u8* __restrict__ dst,
                                                   const u32* __restrict__ list, i64 rowbytes, int gather) {
    const i64 i = blockIdx.x;
    const i64 r = list[i];
    const u8* __restrict__ s = src + (gather ? r : i) * rowbytes;
    u8* __restrict__ d = dst + (gather ? i : r) * rowbytes;
    if (((rowbytes | (i64)(size_t)s | (i64)(size_t)d) & 3) == 0) {
        const u32* __restrict__ s4 = (const u32*)s;
        u32* __restrict__ d4 = (u32*)d;
        for (i64 k = threadIdx.x; k < rowbytes / 4; k += 256) d4[k] = s4[k];
    } else {
        for (i64 k = threadIdx.x; k < rowbytes; k += 256) d[k] = s[k];
    }
}

// fill helpers
static __global__ __launch_bounds__(256) void k_fill_u32(u32* __restrict__ p, u32 v, i64 n) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// What a rank of the sharded bet hands to the all-gather after merging + evaluating ITS queries [q0, q0 + nq): pairs of
// doubles, pair i < nq = {AP, hit count} of query q0 + i, pairs nq .. width - 1 zero, pair `width` = {lost-bet flag, nq}.
static __global__ __launch_bounds__(256) void k_pack_part(const double* __restrict__ ap, const u32* __restrict__ rel, const int* __restrict__ err,
                                                   const i64 q0, const i64 nq, const i64 width, double* __restrict__ out) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i > width) return;
    double a = 0.0, b = 0.0;
    if (i < nq) { a = ap[q0 + i]; b = (double)rel[q0 + i]; }
    else if (i == width) { a = *err ? 1.0 : 0.0; b = (double)nq; }
    out[2 * i] = a;
    out[2 * i + 1] = b;
}

}  // namespace hg
