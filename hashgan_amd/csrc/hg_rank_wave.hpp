// hashgan_amd -- the bet's rank stage with ONE WAVEFRONT PER QUERY (metric.py:14 and the [0:R] cut at :19; one-byte
// records, no ranked lists, a list that fits the wavefront's share of the LDS).
//
// k_rank_cnt gives a query a block of four wavefronts and walks ~10 phases separated by block barriers: two dependent
// global round trips (slice counts, then the slices), LDS counting, the plan by wavefront 0 alone, offsets, placement.
// A block lives ~27 us whatever it ranks (measured: 1280 queries = one round of blocks 0.033 ms, 10 240 queries 0.194 ms;
// R = 100, 170 records per query: still 0.11 ms), five blocks fit a CU, and that concurrency / latency ratio IS the
// kernel's throughput.  Here a query is one wavefront: the same counting sort (per-lane contiguous chunks, byte
// counters, 16-bit offsets, one returning LDS add per record, four in flight), but nothing waits for another wavefront --
// no barrier, the plan's scans run where the data is -- and a query costs 5 KB + its records of LDS instead of a 32 KB
// block, so 10 (C2: 6500 records) to 25 (a sharded rank's 800) independent queries are in flight per CU instead of 5.
// Modes 0 (single shard, fused) and 3 (local ranking for hg_merge_ranked) like k_rank_cnt; what it declines (a list longer
// than its LDS share, a distance without a counter, a list spanning more than 16 / 32 distances) is flagged in big[] and
// left to k_rank_fused, exactly like k_rank_cnt's leftovers.
#pragma once
#include "hg_rank_cnt.hpp"

namespace hg {

constexpr int RW_SMAX = 1024;                 // slices per query this kernel takes (the prefix array lives in LDS)

struct RankWaveLds { int cnt, off, tot, bm, pref, rec, per_wave; };      // byte offsets inside a wavefront's region
__host__ __device__ inline RankWaveLds rank_wave_layout(int NB, i64 RW, int S, int recs, int nbc) {
    RankWaveLds l;
    const int RC_MAXB = rank_cnt_maxb(NB);
    const int NBall = NB < 128 ? NB : 128;
    const int NBc = nbc > 0 && nbc < NBall ? nbc : NBall;
    l.cnt = 0;                                    // [NBc][16] u32: byte counter of lane 4 i + j = byte j of dword i
    l.off = l.cnt + NBc * 64;                     // [RC_MAXB + 1][32] u32: 16-bit offset of lane 2 i + j = half j of dword i (row RC_MAXB: dummy)
    l.tot = l.off + (RC_MAXB + 1) * 128;          // [NBall] u32: totals, then bucket starts
    l.bm = l.tot + NBall * 4;                     // [2 RW] u32
    l.pref = l.bm + (int)(2 * RW) * 4;            // [S + 1] u32
    l.rec = (l.pref + (S + 1) * 4 + 15) & ~15;    // [recs] u8 {match:1 | dist:7}
    l.per_wave = (l.rec + recs + 15) & ~15;
    return l;
}

// Blocks are `blockDim.x / 64` independent wavefronts (the launcher picks 1 or 2: whatever packs the CU's LDS best).
static __global__ __launch_bounds__(256) void k_rank_wave(const u8* __restrict__ cand8, const RankLdsArgs a, u32* __restrict__ mbits32, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 wlds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (q >= g.Q) return;                                         // (no block barrier anywhere below)
    const int NB = g.NB, S = g.S;
    const int NBall = NB < 128 ? NB : 128;
    const int NBc = a.nbc > 0 && a.nbc < NBall ? a.nbc : NBall;
    const int RC_MAXB = rank_cnt_maxb(NB);
    const int bmw = (int)(2 * a.RW);
    const RankWaveLds L = rank_wave_layout(NB, a.RW, S, a.lds_recs, a.nbc);
    u8* base = wlds + (size_t)wave * L.per_wave;
    u32* cnt32 = (u32*)(base + L.cnt);
    u32* off32 = (u32*)(base + L.off);
    u32* tot = (u32*)(base + L.tot);
    u32* bm = (u32*)(base + L.bm);
    u32* pref = (u32*)(base + L.pref);
    u8* rec8 = base + L.rec;
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;

    if (lane == 0) a.big[q] = 0u;
    if (a.fail[q]) {                                              // a slice of this query overflowed
        if (lane == 0) {
            if (a.mode == 3) atomicOr(&a.hown[(i64)NB * g.Qpad], 1u);       // tail word 0: this shard lost the bet
            else { atomicExch(a.err, 1); a.qbad[q] = 1u; }
        }
        if (a.mode == 3) for (int d = lane; d < NB; d += 64) a.hown[(i64)d * g.Qpad + q] = 0u;
        return;
    }
    for (int i = lane; i < L.pref / 4; i += 64) ((u32*)base)[i] = 0u;       // counters, offsets, totals, bitmap

    // ---- slice counts -> exclusive prefix (lane l owns a run of consecutive slices) ----
    const int per = (S + 63) / 64;
    const int sb = lane * per, se = sb + per < S ? sb + per : S;
    u32 mine = 0;
    for (int s = sb; s < se; ++s) mine += a.sl_cnt[(i64)s * g.Qpad + q];
    u32 incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = (u32)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    const u32 n = (u32)__shfl((int)incl, 63);
    if (n > (u32)a.lds_recs) {                                    // does not fit this wavefront's LDS share
        if (lane == 0) a.big[q] = 1u;
        return;
    }
    {
        u32 run = incl - mine;
        for (int s = sb; s < se; ++s) {
            pref[s] = run;
            run += a.sl_cnt[(i64)s * g.Qpad + q];
        }
        if (lane == 63) pref[S] = n;
    }
    wave_lds_sync();

    // ---- copy the query's records into LDS, compacted in slice (= index) order: 16 slices' loads in flight ----
    const u8* __restrict__ row8 = cand8 + (i64)q * a.crow;
    {
        constexpr int NSL = 16;
        for (int s = 0; s < S; s += NSL) {
            u32 p[NSL], c[NSL], v[NSL];
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const int sk = s + k;
                const bool ok = sk < S;
                p[k] = ok ? pref[sk] : 0u;
                c[k] = ok ? pref[sk + 1] - p[k] : 0u;
                const u8* r = row8 + (i64)(ok ? sk : s) * a.cap;
                v[k] = 2u * (u32)lane < c[k] ? (u32)*(const unsigned short*)(r + 2 * lane) : 0u;
            }
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const u32 j0 = 2u * lane;
                if (j0 < c[k]) rec8[p[k] + j0] = (u8)v[k];
                if (j0 + 1 < c[k]) rec8[p[k] + j0 + 1] = (u8)(v[k] >> 8);
                if (c[k] > 128) {
                    const u8* r = row8 + (i64)(s + k) * a.cap;
                    for (u32 i = lane + 128; i < c[k]; i += 64) rec8[p[k] + i] = r[i];
                }
            }
        }
    }
    wave_lds_sync();

    // ---- count: lane l owns records [i0, i1); chunk length = 4 (mod 8) bytes: 64 lanes read 64 different banks ----
    const u32* rec32 = (const u32*)rec8;
    u32 chunk = (n + 63) / 64;
    chunk += (4u - (chunk & 7u)) & 7u;                            // <= 252 + 4: lds_recs keeps n <= 252 * 64
    const u32 i0 = (u32)lane * chunk < n ? (u32)lane * chunk : n;
    const u32 i1 = i0 + chunk < n ? i0 + chunk : n;
    bool beyond = false;
    {
        const u32 one = 1u << (8 * (lane & 3));
#pragma unroll 2
        for (u32 i = i0; i < i1; i += 4) {
            const u32 v = rec32[i >> 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 d = (v >> (8 * j)) & 0x7Fu;
                if (i + j < i1) {
                    if (d < (u32)NBc) atomicAdd(&cnt32[d * 16 + (lane >> 2)], one);
                    else beyond = true;                           // a distance without a counter: the general kernel
                }
            }
        }
    }
    if (__any(beyond)) { if (lane == 0) a.big[q] = 1u; return; }
    wave_lds_sync();
    // totals per distance: lane d sums the 16 dwords of byte counters of distance d
    for (int d0 = 0; d0 < NBc; d0 += 64) {
        const int d = d0 + lane;
        if (d < NBc) {
            u32 sm = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) sm += __builtin_amdgcn_sad_u8(cnt32[d * 16 + k], 0u, 0u);
            tot[d] = sm;
        }
    }
    wave_lds_sync();
    if (a.mode == 3) for (int d = lane; d < NB; d += 64) a.hown[(i64)d * g.Qpad + q] = d < NBall ? tot[d] : 0u;

    // ---- plan (k_plan for one shard): lane l speaks for distances l, l + 64, ... ----
    u64 want = (u64)g.R;
    if (a.mode == 3 && (u64)n < want) want = n;                   // local ranking: whatever this shard has, up to R
    u32 pbase = 0;
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
        const u32 start = pbase + inc - c;                        // global start of bucket d
        const u64 present = __ballot(c != 0u);
        if (dmin < 0 && present) dmin = d0 + (int)__builtin_ctzll(present);
        const u64 reached = __ballot((u64)pbase + inc >= want && d < NBc);
        if (reached) {
            const int lt = (int)__builtin_ctzll(reached);
            t = d0 + lt;
            cntlt = (u32)__shfl((int)start, lt);
            if (lane <= lt) tot[d] = start;                       // starts of the buckets up to the cut
        } else {
            if (d < NBc) tot[d] = start;
            pbase += (u32)__shfl((int)inc, 63);
        }
    }
    if (a.mode == 0 && lane == 0) {
        if (t < 0) atomicExch(a.err, 1);                          // the superset is too small: bet lost
        a.qbad[q] = t < 0 ? 1u : 0u;
    }
    if (t < 0) {
        if (a.mode == 3) for (int w = lane; w < bmw; w += 64) grow[w] = 0u;      // nothing to rank: an empty bitmap
        return;
    }
    const u32 quota = (u32)(want - (u64)cntlt);
    if (dmin < 0) dmin = 0;
    const int nbk = t - dmin + 1;
    if (nbk > RC_MAXB) {                                          // a list spanning many distances: the general kernel
        if (lane == 0) { a.big[q] = 1u; if (a.mode == 0) a.qbad[q] = 0u; }
        return;
    }
    wave_lds_sync();

    // ---- offsets of every lane inside each bucket [dmin, t] ----
    for (int k = 0; k < nbk; ++k) {
        const u32 x = (cnt32[(dmin + k) * 16 + (lane >> 2)] >> (8 * (lane & 3))) & 0xFFu;
        u32 inc = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = (u32)__shfl_up((int)inc, off);
            if (lane >= off) inc += v;
        }
        ((unsigned short*)off32)[k * 64 + lane] = (unsigned short)(inc - x);
    }
    wave_lds_sync();
    // ---- place: four records per round, their returning LDS adds issued back to back (same-lane adds stay in order) ----
    {
        const int sh = 16 * (lane & 1);
        const u32 one = 1u << sh;
        for (u32 i = i0; i < i1; i += 4) {
            u32 meta[4], r[4], st[4];
            const u32 v = rec32[i >> 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                         // -> {dist:8 | match at bit 8}; 0xFFFF: past the chunk
                const u32 m = (v >> (8 * j)) & 0xFFu;
                meta[j] = i + j < i1 ? (m & 0x7Fu) | ((m >> 7) << 8) : 0xFFFFu;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = (int)(meta[j] & 0xFFu);
                const bool in = d <= t && meta[j] != 0xFFFFu;
                const int k = in ? d - dmin : RC_MAXB;
                r[j] = atomicAdd(&off32[k * 32 + (lane >> 1)], one);
                st[j] = in ? tot[d] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = (int)(meta[j] & 0xFFu);
                const bool in = d <= t && meta[j] != 0xFFFFu;
                const u32 rk = (r[j] >> sh) & 0xFFFFu;
                if (in && (meta[j] & 0x100u)) {
                    const u32 pos = st[j] + rk;
                    if (d < t || pos - cntlt < quota) atomicOr(&bm[pos >> 5], 1u << (pos & 31));      // ties: the first `quota` in index order
                }
            }
        }
    }
    wave_lds_sync();
    for (int w = lane; w < bmw; w += 64) grow[w] = bm[w];
}

}  // namespace hg
