// hashgan_amd -- verify + plan + order of the bet as a per-thread counting sort (metric.py:14 and the [0:R] cut at
// :19, for one query per block, records resident in LDS).
//
// Round 1's k_rank_lds placed 64 records per wavefront step with a bit-sliced ballot match (which lanes hold the same
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

namespace hg {

// Arguments of the LDS-resident rank kernels of the bet (k_rank_cnt, k_rank_lean)
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
};


inline __host__ __device__ int rank_cnt_maxb(int NB) { return NB <= 65 ? 16 : 32; }    // distances a ranked list may span on this path

struct RankCntLds { int cnt, off, tot, misc, wsum, bm, pref, rec, idx, total; };     // byte offsets
__host__ __device__ inline RankCntLds rank_cnt_layout(int NB, i64 RW, int S, int recs, int want_lists, int nbc = 0) {
    RankCntLds l;
    const int RC_MAXB = rank_cnt_maxb(NB);
    l.cnt = 0;                                   // [NB][64] u32: byte counter of thread 4 i + j = byte j of dword i
    const int NBc = nbc > 0 && nbc < (NB < 128 ? NB : 128) ? nbc : (NB < 128 ? NB : 128);
    l.off = l.cnt + (NBc * 256 > AP_LDS_BYTES + 8 ? NBc * 256 : (AP_LDS_BYTES + 15) & ~15);   // (distances beyond the counters never reach this path; the AP epilogue reuses the counters)   [RC_MAXB + 1][128] u32: 16-bit offset of thread 2 i + j = half j of dword i
    l.tot = l.off + (RC_MAXB + 1) * 512;         // [NB] u32   (offsets row RC_MAXB is a dummy: records beyond the cut add there)
    l.misc = l.tot + NB * 4;                     // [8] u32
    l.wsum = l.misc + 32;                        // [8] u32, then done[32] and tilecnt[32] (several tiles: per-bucket progress)
    l.bm = (l.wsum + 32 + 256 + 7) & ~7;         // [2 RW] u32 (read as u64 words by the AP epilogue)
    l.pref = l.bm + (int)(2 * RW) * 4;           // [S + 2] u32
    l.rec = l.pref + ((S + 2) & ~1) * 4;         // [recs] u8 {match:1 | dist:7}
    l.idx = l.rec + ((recs + 7) & ~7);           // [recs] u32 (lists only)
    l.total = l.idx + (want_lists ? recs * 4 : 0);
    return l;
}

#ifndef HG_RANK_WAVES
#define HG_RANK_WAVES 5        // wavefronts per SIMD the kernel is compiled for (blocks per CU, with its LDS)
#endif
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HG_RANK_WAVES, HG_RANK_WAVES))) void k_rank_cnt(const u64* __restrict__ cand, const RankLdsArgs a,
                                                  u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                                  u32* __restrict__ mbits32, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 rlds[];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = g.NB, S = g.S;
    const int NBall = NB < 128 ? NB : 128;
    const int NBc = a.nbc > 0 && a.nbc < NBall ? a.nbc : NBall;   // distances that have counters (records beyond them leave this path) ...
    int wlo = 0;                                                  // ... starting at this one
    if (a.cut) { const int T = a.cut[q]; wlo = T - (NBc - 1) > 0 ? T - (NBc - 1) : 0; }
    constexpr int nthr = 256, NWAV = 4;
    const int RC_MAXB = rank_cnt_maxb(NB);
    const int bmw = (int)(2 * a.RW);
    const RankCntLds L = rank_cnt_layout(NB, a.RW, S, a.lds_recs, a.want_lists, a.nbc);
    u32* cnt32 = (u32*)(rlds + L.cnt);
    u32* off32 = (u32*)(rlds + L.off);
    u32* tot = (u32*)(rlds + L.tot);
    u32* misc = (u32*)(rlds + L.misc);
    u32* wsum = (u32*)(rlds + L.wsum);
    u32* done = wsum + 8;
    u32* tilecnt = done + 32;
    u32* bm = (u32*)(rlds + L.bm);
    u32* pref = (u32*)(rlds + L.pref);
    u8* rec8 = rlds + L.rec;
    u32* idx32 = (u32*)(rlds + L.idx);
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;

#ifdef HG_RANK_PROFILE
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    int tkn = 0;
#define HG_TK() do { if (tid == 0 && q < 4096) a.hwq[(i64)q * 16 + tkn] = (u32)(__builtin_amdgcn_s_memtime() - tk0); ++tkn; } while (0)
#else
#define HG_TK() do {} while (0)
#endif
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

    // One-byte records: the first 96 bytes of the first 128 slices are fetched NOW, before anybody knows how many records
    // a slice holds (thread = half a slice, three 16-byte loads): they fly while the slice counts make their own round trip
    // through memory and the prefix is scanned -- one global latency instead of three in a row (profile of round 3: the
    // dependent copy was 37 % of a block's life).  Bytes past a slice's count are simply not used.
    // (slices of up to 48 bytes -- many short segments -- take one thread each: twice as many slices covered)
    constexpr int SPEC_B = 48;                                     // bytes per thread
    const int sp_tps = (a.cap & ~15u) > (u32)SPEC_B ? 2 : 1;       // threads per slice
    const int SPEC_SL = nthr / sp_tps;                             // slices covered
    const int sp_s = sp_tps == 2 ? tid >> 1 : tid, sp_h = sp_tps == 2 ? tid & 1 : 0;
    uint4 spec[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if (a.rec8 && sp_s < S) {
        const u8* r = (const u8*)cand + (i64)q * a.crow + (i64)sp_s * a.cap + sp_h * SPEC_B;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if ((u32)(sp_h * SPEC_B + 16 * j + 16) <= a.cap) spec[j] = *(const uint4*)(r + 16 * j);
    }

    // ---- slice counts -> exclusive prefix (thread t owns a run of consecutive slices) ----
    const int per = (S + nthr - 1) / nthr;
    const int sb = tid * per, se = sb + per < S ? sb + per : S;
    u32 mine = 0;
    u32 first = 0;                                    // (one slice per thread is the usual case: its count is not read twice)
    for (int s = sb; s < se; ++s) {
        const u32 v = a.sl_cnt[(i64)s * g.Qpad + q];
        if (s == sb) first = v;
        mine += v;
    }
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
        run += s == sb ? first : a.sl_cnt[(i64)s * g.Qpad + q];
    }
    if (tid == nthr - 1) pref[S] = run;
    __syncthreads();
    HG_TK();                                          // 0: slice counts + prefix
    const u32 n = pref[S];
    // The records are ranked in tiles of up to TC (they need not all fit the LDS): pass A counts every tile into the
    // totals, the plan follows, pass B re-reads the tiles in order and places them, carrying per-bucket progress.
    // One tile (the usual case, ~1.3 R records) is copied and counted once.
    u32 TC = (u32)a.lds_recs & ~7u;
    if (TC > 252u * nthr) TC = 252u * nthr;           // a thread's chunk must fit its byte counters
    const u32 ntile = n ? (n + TC - 1) / TC : 1u;
    const u64* __restrict__ row = cand + (i64)q * a.crow;
    const u8* __restrict__ row8 = (const u8*)cand + (i64)q * a.crow;

    // ---- copy records [T0, T0 + TC) of the query's list into LDS, compacted in slice (= index) order ----
    auto copy_all = [&]() {                           // single tile: many slices' loads in flight per wavefront
        if (a.rec8) {
            // the prefetched bytes: thread (slice sp_s, half sp_h) owns records [48 sp_h, 48 sp_h + 48) of its slice
            if (sp_s < S) {
                const u32 p0 = pref[sp_s], c0 = pref[sp_s + 1] - p0;
                const u32 b0 = (u32)(sp_h * SPEC_B);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32 w[4] = {spec[j].x, spec[j].y, spec[j].z, spec[j].w};
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const u32 i = b0 + 16 * j + e;
                        if (i < c0) rec8[p0 + i] = (u8)(w[e >> 2] >> (8 * (e & 3)));
                    }
                }
            }
            // what they do not cover: records beyond the 96th of a slice (or beyond its capacity's last whole 16 bytes)
            const u32 cov = (a.cap & ~15u) < (u32)(sp_tps * SPEC_B) ? (a.cap & ~15u) : (u32)(sp_tps * SPEC_B);     // piece k (16 bytes) was loaded iff 16 k + 16 <= cap
            // (lane l looks at slice wave + 4 l, all at once; the few slices that need more are then copied by the whole wavefront)
            const int Sp = S < SPEC_SL ? S : SPEC_SL;                     // the prefetched slices
            for (int sbase = wave; sbase < Sp; sbase += NWAV * 64) {
                const int s2 = sbase + NWAV * lane;
                const u32 p2 = s2 < Sp ? pref[s2] : 0u, c2 = s2 < Sp ? pref[s2 + 1] - p2 : 0u;
                const u32 lo2 = cov;
                u64 m = __ballot(c2 > lo2);
                while (m) {
                    const int l = (int)__builtin_ctzll(m);
                    m &= m - 1;
                    const u32 pp = (u32)__builtin_amdgcn_readlane((int)p2, l), cc = (u32)__builtin_amdgcn_readlane((int)c2, l);
                    const u32 ll = (u32)__builtin_amdgcn_readlane((int)lo2, l);
                    const u8* r = row8 + (i64)(sbase + NWAV * l) * a.cap;
                    for (u32 i = ll + lane; i < cc; i += 64) rec8[pp + i] = r[i];
                }
            }
        }
        if (a.rec8) {                                 // slices beyond the prefetched 128: sixteen slices' loads in flight per wavefront
            constexpr int NSL = 16;
            for (int s = SPEC_SL + wave; s < S; s += NSL * NWAV) {
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
                    const u32 j0 = 2u * lane;
                    if (j0 < c[k]) rec8[p[k] + j0] = (u8)v[k];
                    if (j0 + 1 < c[k]) rec8[p[k] + j0 + 1] = (u8)(v[k] >> 8);
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
                    const u32 j0 = lane, j1 = lane + 64;
                    // 8-byte records {idx:32 | dist:8 | match:1} -> {match:1 | dist:7}; a distance beyond 127 leaves this path
                    if (j0 < c[k]) { rec8[p[k] + j0] = (u8)(((u32)(v0[k] >> 32) & 0x7Fu) | ((u32)(v0[k] >> 33) & 0x80u)); if (a.want_lists) idx32[p[k] + j0] = (u32)v0[k]; }
                    if (j1 < c[k]) { rec8[p[k] + j1] = (u8)(((u32)(v1[k] >> 32) & 0x7Fu) | ((u32)(v1[k] >> 33) & 0x80u)); if (a.want_lists) idx32[p[k] + j1] = (u32)v1[k]; }
                    if ((j0 < c[k] && ((u32)(v0[k] >> 32) & 0x80u)) || (j1 < c[k] && ((u32)(v1[k] >> 32) & 0x80u))) misc[7] = 1u;
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
    };
    auto copy_tile = [&](const u32 T0, const u32 T1) {   // several tiles: slices are long; 4 slices x 4 loads in flight per wavefront
        constexpr int NS = 4, NU = 4;
        for (int s0 = wave; s0 < S; s0 += NS * NWAV) {
            u32 p[NS], lo[NS], hi[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int s = s0 + k * NWAV;
                const u32 pp = s < S ? pref[s] : 0u, e = s < S ? pref[s + 1] : 0u;
                const bool in = e > T0 && pp < T1 && e > pp;
                p[k] = pp;
                lo[k] = in ? (pp > T0 ? pp : T0) - pp : 0u;          // the slice's records inside the tile: [lo, hi)
                hi[k] = in ? (e < T1 ? e : T1) - pp : 0u;
            }
            if (a.rec8) {
                u32 base[NS], more = 0;
#pragma unroll
                for (int k = 0; k < NS; ++k) { base[k] = lo[k] & ~3u; more |= hi[k] > base[k] ? 1u : 0u; }
                while (more) {                                        // wave-uniform: rounds of 1024 records per slice
                    u32 v[NS][NU];
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        const u8* r = row8 + (i64)(s0 + k * NWAV < S ? s0 + k * NWAV : 0) * a.cap;
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const u32 i4 = base[k] + 256u * u + 4u * lane;       // four records per load (slices start 16-byte aligned)
                            v[k][u] = i4 < hi[k] ? *(const u32*)(r + i4) : 0u;
                        }
                    }
                    more = 0;
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const u32 i4 = base[k] + 256u * u + 4u * lane;
#pragma unroll
                            for (u32 j = 0; j < 4; ++j) {
                                const u32 i = i4 + j;
                                if (i >= lo[k] && i < hi[k]) rec8[p[k] + i - T0] = (u8)(v[k][u] >> (8 * j));
                            }
                        }
                        base[k] += 256u * NU;
                        more |= hi[k] > base[k] ? 1u : 0u;
                    }
                    more = (u32)__any((int)more);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const u64* r = row + (i64)(s0 + k * NWAV < S ? s0 + k * NWAV : 0) * a.cap;
                    for (u32 i = lo[k] + lane; i < hi[k]; i += 64) {
                        const u64 v = r[i];
                        rec8[p[k] + i - T0] = (u8)(((u32)(v >> 32) & 0x7Fu) | ((u32)(v >> 33) & 0x80u));
                        if ((u32)(v >> 32) & 0x80u) misc[7] = 1u;
                        if (a.want_lists) idx32[p[k] + i - T0] = (u32)v;
                    }
                }
            }
        }
    };

    // per tile of m records: thread `tid` owns records [i0, i1) of the tile; chunk length = 4 (mod 8) bytes so that the 64 lanes
    // of a wavefront read 64 different LDS banks (stride = chunk / 4 dwords, odd) and every chunk starts on a dword
    const u32* rec32 = (const u32*)rec8;              // four records per read
    u32 i0 = 0, i1 = 0;
    auto count_tile = [&](const u32 m) {
        u32 chunk = (m + nthr - 1) / nthr;
        chunk += (4u - (chunk & 7u)) & 7u;
        i0 = (u32)tid * chunk < m ? (u32)tid * chunk : m;
        i1 = i0 + chunk < m ? i0 + chunk : m;
        const u32 one = 1u << (8 * (tid & 3));
        if (i1 - i0 <= 32u) {                         // the usual chunk (~26 records): all its words in flight, then the adds
            u32 wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wv[k] = i0 + 4u * k < i1 ? rec32[(i0 >> 2) + k] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32 d = ((wv[k] >> (8 * j)) & 0x7Fu) - (u32)wlo;      // relative to the first counter (wraps below it)
                    if (i0 + 4u * k + j < i1) {
                        if (d < (u32)NBc) atomicAdd(&cnt32[d * 64 + (tid >> 2)], one);
                        else misc[7] = 1u;
                    }
                }
            }
            return;
        }
#pragma unroll 2
        for (u32 i = i0; i < i1; i += 4) {
            const u32 v = rec32[i >> 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 d = ((v >> (8 * j)) & 0x7Fu) - (u32)wlo;
                if (i + j < i1) {
                    if (d < (u32)NBc) atomicAdd(&cnt32[d * 64 + (tid >> 2)], one);
                    else misc[7] = 1u;                                   // a distance without a counter: the general kernel
                }
            }
        }
    };
    // totals per distance, added to tot[]: thread = (distance d, quarter j) sums 16 dwords of byte counters; 64 distances per round
    auto add_totals = [&]() {
        for (int d0 = 0; d0 < NBc; d0 += 64) {
            const int d = d0 + (tid >> 2), j = tid & 3;
            u32 sm = 0;
            if (d < NBc) {
#pragma unroll
                for (int k = 0; k < 16; ++k) sm += __builtin_amdgcn_sad_u8(cnt32[d * 64 + j * 16 + k], 0u, 0u);
            }
            sm += (u32)__shfl_xor((int)sm, 1);
            sm += (u32)__shfl_xor((int)sm, 2);
            if (d < NBc && j == 0 && wlo + d < NB) tot[wlo + d] += sm;
        }
    };
    auto zero_counters = [&]() {
        for (int i = tid; i < NBc * 64; i += nthr) cnt32[i] = 0u;
    };

    // ---- pass A: totals ----
    if (ntile == 1) {
        copy_all();
        __syncthreads();
        HG_TK();                                      // 1: copy
        if (misc[7]) { if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); } return; }        // a distance beyond 127: the general kernel
        count_tile(n);
        __syncthreads();
        HG_TK();                                      // 2: count
        if (misc[7]) { if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); } return; }
        add_totals();
        __syncthreads();
        HG_TK();                                      // 3: totals
    } else {
        for (u32 tl = 0; tl < ntile; ++tl) {
            const u32 T0 = tl * TC, T1 = T0 + TC < n ? T0 + TC : n;
            copy_tile(T0, T1);
            __syncthreads();
            if (misc[7]) { if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); } return; }
            count_tile(T1 - T0);
            __syncthreads();
            if (misc[7]) { if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); } return; }
            add_totals();
            __syncthreads();
            zero_counters();
            __syncthreads();
        }
    }
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
            const int d = wlo + d0 + lane;                  // (tot[] is indexed by the distance itself)
            const bool has = d0 + lane < NBc && d < NB;
            const u32 c = has ? tot[d] : 0u;
            u32 inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = (u32)__shfl_up((int)inc, off);
                if (lane >= off) inc += v;
            }
            const u32 start = base + inc - c;         // global start of bucket d
            const u64 present = __ballot(c != 0u);
            if (dmin < 0 && present) dmin = wlo + d0 + (int)__builtin_ctzll(present);
            const u64 reached = __ballot((u64)base + inc >= want && has);
            if (reached) {
                const int lt = (int)__builtin_ctzll(reached);
                t = wlo + d0 + lt;
                cntlt = (u32)__shfl((int)start, lt);
                if (lane <= lt) tot[d] = start;       // starts of the buckets up to the cut
            } else {
                if (has) tot[d] = start;
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
    HG_TK();                                          // 4: plan
    const int t = (int)misc[0];
    if (t < 0) {
        if (a.mode == 3) for (int w = tid; w < bmw; w += nthr) grow[w] = 0u;   // nothing to rank: an empty bitmap
        return;
    }
    const u32 cntlt = misc[1], quota = misc[2];
    const int dmin = (int)misc[3];
    const int nbk = t - dmin + 1;
    if (nbk > RC_MAXB) {                              // a list spanning many distances: the general kernel
        if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); if (a.mode == 0) a.qbad[q] = 0u; }
        return;
    }

    // ---- pass B: per tile, offsets of every thread inside each bucket [dmin, t], then placement ----
    u32* __restrict__ oi = out_idx + (i64)q * g.R;
    u8* __restrict__ od = out_dist + (i64)q * g.R;
    for (u32 tl = 0; tl < ntile; ++tl) {
        const u32 T0 = tl * TC, T1 = T0 + TC < n ? T0 + TC : n;
        if (ntile > 1) {                              // (one tile: records and counters are still in place)
            copy_tile(T0, T1);
            __syncthreads();
            count_tile(T1 - T0);
            __syncthreads();
        }
        for (int k = wave; k < nbk; k += NWAV) {
            const u32 x = cnt32[(dmin + k - wlo) * 64 + lane];
            const u32 sm = __builtin_amdgcn_sad_u8(x, 0u, 0u);
            u32 inc = sm;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = (u32)__shfl_up((int)inc, off);
                if (lane >= off) inc += v;
            }
            const u32 o0 = inc - sm, o1 = o0 + (x & 0xFFu), o2 = o1 + ((x >> 8) & 0xFFu), o3 = o2 + ((x >> 16) & 0xFFu);
            off32[k * 128 + 2 * lane] = o0 | (o1 << 16);                 // threads 4 lane, 4 lane + 1
            off32[k * 128 + 2 * lane + 1] = o2 | (o3 << 16);             // threads 4 lane + 2, 4 lane + 3
            if (lane == 63) tilecnt[k] = inc;                            // the tile's records of this distance
        }
        __syncthreads();
        HG_TK();                                      // 5: offsets
        // place: four records per round -- their returning LDS adds are issued back to back (same-thread adds to one
        // offset stay in order), so a round pays one LDS round trip, not four; records beyond the cut add to a dummy row
        {
            const int sh = 16 * (tid & 1);
            const u32 one = 1u << sh;
            constexpr int PG = 4;                                         // records per round (8 = two words per round was slower: 14.5 k vs 13.1 k cycles for the phase)
            for (u32 i = i0; i < i1; i += PG) {
                u32 meta[PG], r[PG], st[PG];
                const u32 v0 = rec32[i >> 2], v1 = PG > 4 && i + 4 < i1 ? rec32[(i >> 2) + 1] : 0u;
#pragma unroll
                for (int j = 0; j < PG; ++j) {                            // -> {dist:8 | match at bit 8}; 0xFFFF: past the chunk
                    const u32 m = ((j < 4 ? v0 : v1) >> (8 * (j & 3))) & 0xFFu;
                    meta[j] = i + j < i1 ? (m & 0x7Fu) | ((m >> 7) << 8) : 0xFFFFu;
                }
#pragma unroll
                for (int j = 0; j < PG; ++j) {
                    const int d = (int)(meta[j] & 0xFFu);
                    const bool in = d <= t && meta[j] != 0xFFFFu;
                    const int k = in ? d - dmin : RC_MAXB;
                    r[j] = atomicAdd(&off32[k * 128 + (tid >> 1)], one);
                }
#pragma unroll
                for (int j = 0; j < PG; ++j) {
                    const int d = (int)(meta[j] & 0xFFu);
                    const bool in = d <= t && meta[j] != 0xFFFFu;
                    st[j] = (in ? tot[d] : 0u) + (ntile > 1 ? done[in ? d - dmin : 0] : 0u);      // bucket start + what earlier tiles placed there
                }
#pragma unroll
                for (int j = 0; j < PG; ++j) {
                    const int d = (int)(meta[j] & 0xFFu);
                    const bool in = d <= t && meta[j] != 0xFFFFu;
                    const u32 rk = (r[j] >> sh) & 0xFFFFu;
                    u32 pos = IDX_NONE;
                    if (in) {
                        if (d < t) pos = st[j] + rk;
                        else if (st[j] - cntlt + rk < quota) pos = st[j] + rk;      // ties: st = cntlt + ties of earlier tiles
                    }
                    if (pos != IDX_NONE) {
                        if (a.want_lists) { oi[pos] = idx32[i + j]; od[pos] = (u8)d; }
                        if (meta[j] & 0x100u) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
                    }
                }
            }
        }
        __syncthreads();
        HG_TK();                                      // 6: place
        if (ntile > 1) {
            if (tid < nbk) done[tid] += tilecnt[tid];
            zero_counters();
            __syncthreads();
        }
    }
    for (int w = tid; w < bmw; w += nthr) grow[w] = bm[w];
    HG_TK();                                          // 7: bitmap out
    if (a.ap_shapes) {
        // metric.py:20-23 while the bitmap is in LDS: k_ap's very arithmetic (ap_eval), its scratch carved out of the counters
        __syncthreads();                              // the last tile's counters and offsets are no longer read
        const u64* bm64 = (const u64*)bm;
        ap_eval<nthr>([&](const i64 w) { return bm64[w]; }, a.RW, g.R, a.ap_shapes, a.ap_recip, ap_lds_at(rlds + L.cnt), tid, a.ap + q, a.rel + q);
    }
}

}  // namespace hg
