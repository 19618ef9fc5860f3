// hashgan_amd -- verify + plan + order of the bet with the query's records resident in LDS.
//
// k_rank_fused walks a query's records twice from global memory, 64 records of ONE slice per step:
// with slices of ~64 +- 8 records its steps are two-thirds full and the kernel waits on one
// 512-byte load per wavefront most of the time (74 % of its wave-cycles, profiles/r01_v5_pmc.txt).
// Here the block first copies all the query's records into LDS, compacted in slice (= index)
// order and reduced to what ranking needs -- 2 bytes {dist, match} per record, plus the 4-byte
// index only when the ranked lists are wanted -- with several slices' loads in flight per
// wavefront; histogram, plan and the stable placement (k_rank_fused's, metric.py:14,19) then run
// on dense 64-record steps out of LDS.  Queries with more records than the LDS holds are flagged
// in big[] and left to k_rank_fused (launched afterwards with only = big).
#pragma once
#include "hg_kernels.hpp"

namespace hg {

struct RankLdsArgs {
    const u32* sl_cnt;     // [S][Qpad]
    const u32* fail;       // [Qpad]
    int* err;
    u32* qbad;             // [Q]
    u32* big;              // [Q] out: 1 = not handled here (too many records for the LDS)
    u32 cap;
    i64 crow;
    int want_lists;
    int rec8;              // records are one byte {match:1 | dist:7} (compact select, no lists) instead of 8-byte {idx, dist, match}
    i64 RW;
    int lds_recs;          // record capacity of the LDS arrays
    // several shards (k_rank_fused's modes): 0 fused; 1 histogram phase only (per-wave histograms -> hwq,
    // shard totals -> hown); 2 placement with the plan computed from the gathered histograms
    int mode;
    u32* hwq;              // [Q][NWAV][NB]
    u32* hown;             // [NB][Qpad] (+ tail)
    const int* xt;
    const u32* xcnt_lt;
    const u32* xquota;
    const u32* xtie_before;
    const u32* xposbase;   // [NB][Qpad]
    int nbc;               // k_rank_cnt: distances that have counters (0: all up to 127); a record beyond them sends the query to k_rank_fused
    // k_rank_cnt, mode 0: the AP of every query it ranks comes out of its epilogue (the bitmap is still in LDS) -- metric.py:20-23
    const ApShape* ap_shapes;   // null: no AP here (k_ap runs later)
    const double* ap_recip;     // [R + 1] or null
    double* ap;                 // [Q]
    u32* rel;                   // [Q]
    u32* nleft;                 // count of queries left to the general kernel (big[q] = 1): the host launches it only if > 0
    // k_rank_cnt: the nbc counters cover the distances [max(0, cut[q] - nbc + 1), cut[q]] -- every record of a bet is within
    // its query's cut (the guess, or the exact threshold), and a list that reaches further down than the 16 (32) distances
    // this kernel places leaves it anyway (null: the counters cover [0, nbc))
    const int* cut;
    int spec_pieces;       // k_rank_lean: 16-byte pieces of every slice fetched before the slice counts are known
    int il;                // k_rank_lean: the record rows are interleaved (rec8_at)
};

template <int NWAV>
__global__ __launch_bounds__(NWAV * 64) void k_rank_lds(const u64* __restrict__ cand, const RankLdsArgs a,
                                                        u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                                        u32* __restrict__ mbits32, int nbits, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = g.NB, S = g.S;
    constexpr int nthr = NWAV * 64;
    const int bmw = (int)(2 * a.RW);
    // LDS carve (32-bit words)
    u32* hw = lds;                         // [NWAV][NB]
    u32* tot = hw + NWAV * NB;             // [NB]
    u32* misc = tot + NB;                  // [8]
    u32* wsum = misc + 8;                  // [NWAV + 1] scan scratch
    u32* bm = wsum + NWAV + 8;             // [bmw]
    u32* pref = bm + bmw;                  // [S + 1] exclusive prefix of the slice counts
    unsigned short* rec16 = (unsigned short*)(pref + ((S + 2) & ~1));     // [lds_recs]
    u32* idx32 = (u32*)(rec16 + a.lds_recs);                              // [lds_recs] (lists only)
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;

    if (a.mode == 2) { if (a.big[q]) return; }        // flagged by the histogram launch: k_rank_fused's
    else if (tid == 0) a.big[q] = 0u;
    if (a.fail[q]) {                                  // a slice of this query overflowed
        if (tid == 0) {
            if (a.mode == 1 || a.mode == 3) atomicOr(&a.hown[(i64)NB * g.Qpad], 1u);   // tail word 0: this shard lost the bet
            else { atomicExch(a.err, 1); a.qbad[q] = 1u; }
        }
        if (a.mode == 1 || a.mode == 3) for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = 0u;
        return;
    }
    for (int i = tid; i < (NWAV + 1) * NB + 8 + NWAV + 8 + bmw; i += nthr) lds[i] = 0u;

    // ---- slice counts -> exclusive prefix (thread t owns a run of consecutive slices) ----
    const int per = (S + nthr - 1) / nthr;
    const int sb = tid * per, se = sb + per < S ? sb + per : S;
    u32 mine = 0;
    for (int s = sb; s < se; ++s) mine += a.sl_cnt[(i64)s * g.Qpad + q];
    u32 incl = mine;                                  // inclusive scan over the wavefront's lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = (u32)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    __syncthreads();                                  // the zero fill above is done
    if (lane == 63) wsum[wave + 1] = incl;
    __syncthreads();
    u32 wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w + 1];
    u32 run = wbase + incl - mine;
    for (int s = sb; s < se; ++s) {
        pref[s] = run;
        run += a.sl_cnt[(i64)s * g.Qpad + q];
    }
    if (tid == nthr - 1) pref[S] = run;               // the last thread's run ends at the total (empty runs included)
    __syncthreads();
    const u32 n = pref[S];
    if (n > (u32)a.lds_recs) {                        // too many records for the LDS: k_rank_fused takes this query
        if (tid == 0) a.big[q] = 1u;
        return;
    }

    // ---- copy the records into LDS: wave w takes slices w, w + NWAV, ...; many slices' loads in flight ----
    const u64* __restrict__ row = cand + (i64)q * a.crow;
    if (a.rec8) {
        // compact records, one byte {match:1 | dist:7} each: a lane fetches TWO (one 16-bit load covers 128 records of a
        // slice; capacities are multiples of 16, so the odd byte past the count is still inside the slice)
        const u8* __restrict__ row8 = (const u8*)cand + (i64)q * a.crow;
        constexpr int NSL = 16;
        for (int s = wave; s < S; s += NSL * NWAV) {
            u32 p[NSL], c[NSL], v[NSL];
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const int sk = s + k * NWAV;
                const bool ok = sk < S;
                p[k] = ok ? pref[sk] : 0u;
                c[k] = ok ? pref[sk + 1] - p[k] : 0u;
                const u8* r = row8 + (i64)(ok ? sk : s) * a.cap;
                v[k] = 2u * (u32)lane < c[k] ? (u32)*(const unsigned short*)(r + 2 * lane) : 0u;
            }
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const u32 i0 = 2u * lane, m0 = v[k] & 0xFFu, m1 = v[k] >> 8;
                if (i0 < c[k]) rec16[p[k] + i0] = (unsigned short)((m0 & 0x7Fu) | ((m0 >> 7) << 8));
                if (i0 + 1 < c[k]) rec16[p[k] + i0 + 1] = (unsigned short)((m1 & 0x7Fu) | ((m1 >> 7) << 8));
                if (c[k] > 128) {                     // long slices (rare)
                    const u8* r = row8 + (i64)(s + k * NWAV) * a.cap;
                    for (u32 i = lane + 128; i < c[k]; i += 64) {
                        const u32 m = r[i];
                        rec16[p[k] + i] = (unsigned short)((m & 0x7Fu) | ((m >> 7) << 8));
                    }
                }
            }
        }
    } else {
        constexpr int NSL = 8;                        // slices per iteration: 2 NSL loads in flight per wavefront
        for (int s = wave; s < S; s += NSL * NWAV) {
            u32 p[NSL], c[NSL];
            u64 v0[NSL], v1[NSL];
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const int sk = s + k * NWAV;
                const bool ok = sk < S;
                p[k] = ok ? pref[sk] : 0u;
                c[k] = ok ? pref[sk + 1] - p[k] : 0u;
                const u64* r = row + (i64)(ok ? sk : s) * a.cap;
                v0[k] = (u32)lane < c[k] ? r[lane] : 0ull;
                v1[k] = (u32)lane + 64 < c[k] ? r[lane + 64] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const u32 i0 = lane, i1 = lane + 64;
                if (i0 < c[k]) { rec16[p[k] + i0] = (unsigned short)(v0[k] >> 32); if (a.want_lists) idx32[p[k] + i0] = (u32)v0[k]; }
                if (i1 < c[k]) { rec16[p[k] + i1] = (unsigned short)(v1[k] >> 32); if (a.want_lists) idx32[p[k] + i1] = (u32)v1[k]; }
                if (c[k] > 128) {                     // long slices (rare)
                    const u64* r = row + (i64)(s + k * NWAV) * a.cap;
                    for (u32 i = lane + 128; i < c[k]; i += 64) {
                        const u64 v = r[i];
                        rec16[p[k] + i] = (unsigned short)(v >> 32);
                        if (a.want_lists) idx32[p[k] + i] = (u32)v;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 1: per-wave histograms over contiguous quarters of the record list ----
    const u32 r0w = (u32)((u64)n * wave / NWAV), r1w = (u32)((u64)n * (wave + 1) / NWAV);
    u32* myh = hw + wave * NB;
    if (a.mode == 2) {                                // computed by the histogram launch
        for (int i = tid; i < NWAV * NB; i += nthr) hw[i] = a.hwq[(i64)q * NWAV * NB + i];
    } else {
        for (u32 i = r0w + lane; i < r1w; i += 64) {
            const u32 d = rec16[i] & 0xFFu;
            if (d < (u32)NB) atomicAdd(&myh[d], 1u);
        }
    }
    __syncthreads();
    // ---- phase 2: totals, threshold, quota, per-wave bucket starts (k_rank_fused's) ----
    for (int d = tid; d < NB; d += nthr) {
        u32 acc = 0;
        for (int w = 0; w < NWAV; ++w) acc += hw[w * NB + d];
        tot[d] = acc;
    }
    __syncthreads();
    if (a.mode == 1) {                                // hand the histograms over and stop
        for (int i = tid; i < NWAV * NB; i += nthr) a.hwq[(i64)q * NWAV * NB + i] = hw[i];
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
        for (int d = tid; d < NB; d += nthr) tot[d] = a.xposbase[(i64)d * g.Qpad + q];   // my rows of bucket d start here
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
        u32 acc = d < t ? tot[d] : tie0;              // for d == t the "start" is the tie rank offset
        for (int w = 0; w < NWAV; ++w) {
            const u32 h = hw[w * NB + d];
            hw[w * NB + d] = acc;
            acc += h;
        }
    }
    __syncthreads();
    // ---- phase 3: stable placement, 64 records per step ----
    u32* pb = hw + wave * NB;
    const u32 cntlt = misc[1], quota = misc[2], dmin = misc[3];
    int kb = 0;
    while (kb < nbits && (int)(dmin + (1u << kb)) < t) ++kb;
    u32 tie_run = pb[t];                              // ties owned by earlier waves
    u32* __restrict__ oi = out_idx + (i64)q * g.R;
    u8* __restrict__ od = out_dist + (i64)q * g.R;
    const u64 below = (1ull << lane) - 1ull;
    for (u32 base = r0w; base < r1w; base += 64) {
        const u32 i = base + lane;
        const bool valid = i < r1w;
        const u32 meta = valid ? (u32)rec16[i] : 0u;
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
            if (a.want_lists) { oi[pos] = idx32[i]; od[pos] = (u8)d; }
            if (meta & 0x100u) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
        }
        wave_lds_sync();
    }
    __syncthreads();
    for (int w = tid; w < bmw; w += nthr) grow[w] = bm[w];
}

}  // namespace hg
