// hashgan_amd -- verify + plan + order of the bet as a per-thread counting sort (metric.py:14 and the [0:R] cut at
// :19, for one query per block, records resident in LDS).
//
// k_rank_lds places 64 records per wavefront step with a bit-sliced ballot match (which lanes hold the same
// distance?) -- ~70 vector + ~50 scalar instructions per step, 8.5 k scalar instructions per query
// (profiles/r02_compact_pmc_summary.txt).  A ranked list only spans about ten distinct distances, so the classic
// counting sort is far cheaper: every thread owns a CONTIGUOUS chunk of the query's records (index order is the
// tie order, so contiguous chunks keep the sort stable),
//   count    one fire-and-forget LDS add per record into the thread's own byte counter of that distance,
//   totals   per distance: 256 byte counters = 64 dwords, one v_sad_u8 + a wave reduction -> the plan (threshold
//            t, quota, bucket starts) exactly as k_plan / k_rank_fused compute it,
//   scan     per distance in [dmin, t]: exclusive prefix of the thread counters -> 16-bit offsets inside the bucket,
//   place    every thread walks its chunk again: rank = bucket start + its running offset (one returning LDS add);
//            the match bit goes into the LDS bitmap at that rank.
// ~14 instructions per 64 records instead of ~120.  Needs every record's distance <= 127 (records are kept as one byte,
// byte counters exist for 128 distances) and a ranked list that spans <= 16 distances; other queries are flagged in
// big[] and left to k_rank_fused.
// Modes 0 (single shard, fused) and 3 (local ranking for hg_merge_ranked) of k_rank_fused.
#pragma once
#include "hg_kernels.hpp"
#include "hg_rank_lds.hpp"

namespace hg {

inline __host__ __device__ int rank_cnt_maxb(int NB) { return NB <= 65 ? 16 : 32; }    // distances a ranked list may span on this path

struct RankCntLds { int cnt, off, tot, misc, wsum, bm, pref, rec, idx, total; };     // byte offsets
__host__ __device__ inline RankCntLds rank_cnt_layout(int NB, i64 RW, int S, int recs, int want_lists) {
    RankCntLds l;
    const int RC_MAXB = rank_cnt_maxb(NB);
    l.cnt = 0;                                   // [NB][64] u32: byte counter of thread 4 i + j = byte j of dword i
    l.off = l.cnt + (NB < 128 ? NB : 128) * 256; // (distances beyond 127 never reach this path)   [RC_MAXB + 1][128] u32: 16-bit offset of thread 2 i + j = half j of dword i
    l.tot = l.off + (RC_MAXB + 1) * 512;         // [NB] u32   (offsets row RC_MAXB is a dummy: records beyond the cut add there)
    l.misc = l.tot + NB * 4;                     // [8] u32
    l.wsum = l.misc + 32;                        // [8] u32
    l.bm = l.wsum + 32;                          // [2 RW] u32
    l.pref = l.bm + (int)(2 * RW) * 4;           // [S + 2] u32
    l.rec = l.pref + ((S + 2) & ~1) * 4;         // [recs] u8 {match:1 | dist:7}
    l.idx = l.rec + ((recs + 7) & ~7);           // [recs] u32 (lists only)
    l.total = l.idx + (want_lists ? recs * 4 : 0);
    return l;
}

__global__ __launch_bounds__(256) void k_rank_cnt(const u64* __restrict__ cand, const RankLdsArgs a,
                                                  u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                                  u32* __restrict__ mbits32, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 rlds[];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = g.NB, S = g.S;
    const int NBc = NB < 128 ? NB : 128;             // distances that have counters (records beyond 127 leave this path)
    constexpr int nthr = 256, NWAV = 4;
    const int RC_MAXB = rank_cnt_maxb(NB);
    const int bmw = (int)(2 * a.RW);
    const RankCntLds L = rank_cnt_layout(NB, a.RW, S, a.lds_recs, a.want_lists);
    u32* cnt32 = (u32*)(rlds + L.cnt);
    u32* off32 = (u32*)(rlds + L.off);
    u32* tot = (u32*)(rlds + L.tot);
    u32* misc = (u32*)(rlds + L.misc);
    u32* wsum = (u32*)(rlds + L.wsum);
    u32* bm = (u32*)(rlds + L.bm);
    u32* pref = (u32*)(rlds + L.pref);
    u8* rec8 = rlds + L.rec;
    u32* idx32 = (u32*)(rlds + L.idx);
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;

    if (tid == 0) a.big[q] = 0u;
    if (a.fail[q]) {                                  // a slice of this query overflowed
        if (tid == 0) {
            if (a.mode == 3) atomicOr(&a.hown[(i64)NB * g.Qpad], 1u);   // tail word 0: this shard lost the bet
            else { atomicExch(a.err, 1); a.qbad[q] = 1u; }
        }
        if (a.mode == 3) for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = 0u;
        return;
    }
    for (int i = tid; i < (L.pref - L.cnt) / 4; i += nthr) ((u32*)rlds)[i] = 0u;       // counters, offsets, totals, misc, bitmap

    // ---- slice counts -> exclusive prefix (thread t owns a run of consecutive slices) ----
    const int per = (S + nthr - 1) / nthr;
    const int sb = tid * per, se = sb + per < S ? sb + per : S;
    u32 mine = 0;
    for (int s = sb; s < se; ++s) mine += a.sl_cnt[(i64)s * g.Qpad + q];
    u32 incl = mine;
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
    if (tid == nthr - 1) pref[S] = run;
    __syncthreads();
    const u32 n = pref[S];
    if (n > (u32)a.lds_recs) {                        // too many records for the LDS: k_rank_fused takes this query
        if (tid == 0) a.big[q] = 1u;
        return;
    }

    // ---- copy the records into LDS, compacted in slice (= index) order ----
    const u64* __restrict__ row = cand + (i64)q * a.crow;
    if (a.rec8) {
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
                const u32 i0 = 2u * lane;
                if (i0 < c[k]) rec8[p[k] + i0] = (u8)v[k];
                if (i0 + 1 < c[k]) rec8[p[k] + i0 + 1] = (u8)(v[k] >> 8);
                if (c[k] > 128) {
                    const u8* r = row8 + (i64)(s + k * NWAV) * a.cap;
                    for (u32 i = lane + 128; i < c[k]; i += 64) rec8[p[k] + i] = r[i];
                }
            }
        }
    } else {
        constexpr int NSL = 8;
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
                // 8-byte records {idx:32 | dist:8 | match:1} -> {match:1 | dist:7} (this path: b < 128, so a distance fits 7 bits)
                if (i0 < c[k]) { rec8[p[k] + i0] = (u8)(((u32)(v0[k] >> 32) & 0x7Fu) | ((u32)(v0[k] >> 33) & 0x80u)); if (a.want_lists) idx32[p[k] + i0] = (u32)v0[k]; }
                if (i1 < c[k]) { rec8[p[k] + i1] = (u8)(((u32)(v1[k] >> 32) & 0x7Fu) | ((u32)(v1[k] >> 33) & 0x80u)); if (a.want_lists) idx32[p[k] + i1] = (u32)v1[k]; }
                if ((i0 < c[k] && ((u32)(v0[k] >> 32) & 0x80u)) || (i1 < c[k] && ((u32)(v1[k] >> 32) & 0x80u))) misc[7] = 1u;   // a distance beyond 127
                if (c[k] > 128) {
                    const u64* r = row + (i64)(s + k * NWAV) * a.cap;
                    for (u32 i = lane + 128; i < c[k]; i += 64) {
                        const u64 v = r[i];
                        rec8[p[k] + i] = (u8)(((u32)(v >> 32) & 0x7Fu) | ((u32)(v >> 33) & 0x80u));
                        if ((u32)(v >> 32) & 0x80u) misc[7] = 1u;
                        if (a.want_lists) idx32[p[k] + i] = (u32)v;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (misc[7]) {                                    // only codes of >= 128 bits ranked against far-away rows: the general kernel
        if (tid == 0) a.big[q] = 1u;
        return;
    }

    // ---- count: thread `tid` owns records [i0, i1); chunk length = 4 (mod 8) bytes so that the 64 lanes of a wavefront
    // read 64 different LDS banks (stride = chunk / 4 dwords, odd) and every chunk starts on a dword ----
    u32 chunk = (n + nthr - 1) / nthr;
    chunk += (4u - (chunk & 7u)) & 7u;
    if (chunk > 252u) {                               // a byte counter could wrap: not this path's case
        if (tid == 0) a.big[q] = 1u;
        return;
    }
    const u32 i0 = (u32)tid * chunk < n ? (u32)tid * chunk : n;
    const u32 i1 = i0 + chunk < n ? i0 + chunk : n;
    const u32* rec32 = (const u32*)rec8;              // four records per read (i0 is a multiple of 4)
    {
        const u32 one = 1u << (8 * (tid & 3));
#pragma unroll 2
        for (u32 i = i0; i < i1; i += 4) {
            const u32 v = rec32[i >> 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 d = (v >> (8 * j)) & 0x7Fu;
                if (i + j < i1 && d < (u32)NBc) atomicAdd(&cnt32[d * 64 + (tid >> 2)], one);
            }
        }
    }
    __syncthreads();
    // ---- totals per distance: thread = (distance d, quarter j) sums 16 dwords of byte counters; 64 distances per round ----
    for (int d0 = 0; d0 < NBc; d0 += 64) {
        const int d = d0 + (tid >> 2), j = tid & 3;
        u32 s = 0;
        if (d < NBc) {
#pragma unroll
            for (int k = 0; k < 16; ++k) s += __builtin_amdgcn_sad_u8(cnt32[d * 64 + j * 16 + k], 0u, 0u);
        }
        s += (u32)__shfl_xor((int)s, 1);
        s += (u32)__shfl_xor((int)s, 2);
        if (d < NBc && j == 0) tot[d] = s;
    }
    __syncthreads();
    if (a.mode == 3) {                                // counts for the merge, before the plan turns tot[] into starts
        for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = tot[d];
        __syncthreads();
    }
    // ---- plan (k_plan for one shard), by wavefront 0: lane l speaks for distances l, l + 64, ... ----
    if (wave == 0) {
        // mode 3 (local ranking for k_merge_ranked): rank whatever this shard has, up to R -- never "lost" here
        u64 want = (u64)g.R;
        if (a.mode == 3) {
            u32 have = 0;
            for (int d = lane; d < NB; d += 64) have += tot[d];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) have += (u32)__shfl_xor((int)have, off);
            if ((u64)have < want) want = have;
        }
        u32 base = 0;                                 // records closer than the current group of 64 distances
        int t = -1, dmin = -1;
        u32 cntlt = 0;
        for (int d0 = 0; d0 < NBc && t < 0 && want > 0; d0 += 64) {
            const int d = d0 + lane;
            const u32 c = d < NBc ? tot[d] : 0u;
            u32 inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = (u32)__shfl_up((int)inc, off);
                if (lane >= off) inc += v;
            }
            const u32 start = base + inc - c;         // global start of bucket d
            const u64 present = __ballot(c != 0u);
            if (dmin < 0 && present) dmin = d0 + (int)__builtin_ctzll(present);
            const u64 reached = __ballot((u64)base + inc >= want && d < NBc);
            if (reached) {
                const int lt = (int)__builtin_ctzll(reached);
                t = d0 + lt;
                cntlt = (u32)__shfl((int)start, lt);
                if (lane <= lt) tot[d] = start;       // starts of the buckets up to the cut
            } else {
                if (d < NBc) tot[d] = start;
                base += (u32)__shfl((int)inc, 63);
            }
        }
        if (lane == 0) {
            misc[0] = (u32)t;
            misc[1] = cntlt;
            misc[2] = (u32)(want - (u64)cntlt);       // quota
            misc[3] = (u32)(dmin < 0 ? 0 : dmin);     // smallest distance present
            if (a.mode == 0) {
                if (t < 0) atomicExch(a.err, 1);      // the superset is too small: bet lost
                a.qbad[q] = t < 0 ? 1u : 0u;
            }
        }
    }
    __syncthreads();
    const int t = (int)misc[0];
    if (t < 0) {
        if (a.mode == 3) for (int w = tid; w < bmw; w += nthr) grow[w] = 0u;   // nothing to rank: an empty bitmap
        return;
    }
    const u32 cntlt = misc[1], quota = misc[2];
    const int dmin = (int)misc[3];
    const int nbk = t - dmin + 1;
    if (nbk > RC_MAXB) {                              // a list spanning many distances: the general kernel
        if (tid == 0) { a.big[q] = 1u; if (a.mode == 0) a.qbad[q] = 0u; }
        return;
    }
    // ---- scan: offsets of every thread inside each bucket [dmin, t] ----
    for (int k = wave; k < nbk; k += NWAV) {
        const u32 x = cnt32[(dmin + k) * 64 + lane];
        const u32 s = __builtin_amdgcn_sad_u8(x, 0u, 0u);
        u32 inc = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = (u32)__shfl_up((int)inc, off);
            if (lane >= off) inc += v;
        }
        const u32 o0 = inc - s, o1 = o0 + (x & 0xFFu), o2 = o1 + ((x >> 8) & 0xFFu), o3 = o2 + ((x >> 16) & 0xFFu);
        off32[k * 128 + 2 * lane] = o0 | (o1 << 16);                 // threads 4 lane, 4 lane + 1
        off32[k * 128 + 2 * lane + 1] = o2 | (o3 << 16);             // threads 4 lane + 2, 4 lane + 3
    }
    __syncthreads();
    // ---- place: four records per round -- their returning LDS adds are issued back to back (same-thread adds to one
    // offset stay in order), so a round pays one LDS round trip, not four; records beyond the cut add to a dummy row ----
    {
        u32* __restrict__ oi = out_idx + (i64)q * g.R;
        u8* __restrict__ od = out_dist + (i64)q * g.R;
        const int sh = 16 * (tid & 1);
        const u32 one = 1u << sh;
        for (u32 i = i0; i < i1; i += 4) {
            u32 meta[4], r[4], st[4];
            const u32 v = rec32[i >> 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                             // -> {dist:8 | match at bit 8}; 0xFFFF: past the chunk
                const u32 m = (v >> (8 * j)) & 0xFFu;
                meta[j] = i + j < i1 ? (m & 0x7Fu) | ((m >> 7) << 8) : 0xFFFFu;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = (int)(meta[j] & 0xFFu);
                const bool in = d <= t && meta[j] != 0xFFFFu;
                const int k = in ? d - dmin : RC_MAXB;
                r[j] = atomicAdd(&off32[k * 128 + (tid >> 1)], one);
                st[j] = tot[in ? d : 0];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = (int)(meta[j] & 0xFFu);
                const bool in = d <= t && meta[j] != 0xFFFFu;
                const u32 rk = (r[j] >> sh) & 0xFFFFu;
                u32 pos = IDX_NONE;
                if (in) {
                    if (d < t) pos = st[j] + rk;
                    else if (rk < quota) pos = cntlt + rk;
                }
                if (pos != IDX_NONE) {
                    if (a.want_lists) { oi[pos] = idx32[i + j]; od[pos] = (u8)d; }
                    if (meta[j] & 0x100u) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
                }
            }
        }
    }
    __syncthreads();
    for (int w = tid; w < bmw; w += nthr) grow[w] = bm[w];
}

}  // namespace hg
