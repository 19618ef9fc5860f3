// hashgan_amd -- the bet's rank stage, lean form: verify + plan + stable counting sort + AP of one query per block
// (metric.py:14, the [0:R] cut at :19 and :20-23), for one-byte records, no ranked lists, at most 256 slices per query.
//
// k_rank_cnt is bound by its vector instructions, not by latency (profiles/r04_pmc_c2.txt: 2575 vector instructions per
// wavefront, 67 % of the SIMD issue cycles at five blocks per CU).  This kernel does the same counting sort with a third
// fewer of them:
//   * the first `spec` 16-byte pieces of every slice (what a slice holds but for a 4-sigma exception; bytes past a slice's
//     count are fetched and not used) are requested before anything else, one piece per thread and load -- no dependent
//     round trip; a slice with more is finished by its own thread once the counts are known;
//   * slices are compacted into LDS piece by piece (one ds_write_b128 per 16 records; the unused tail of a slice's last
//     piece is overwritten with a distance just beyond the query's cut, which sends it to a dummy bucket) instead of byte by
//     byte (48 predicated one-byte stores per thread);
//   * a thread's chunk of the compacted array (whole pieces) stays in registers from the count to the placement;
//   * prefix sums run on DPP row shifts (six v_add_dpp) instead of six ds_bpermute round trips;
//   * the bucket starts are folded into the threads' 16-bit offsets, so a record's returning LDS add IS its rank: no
//     second lookup, and "inside the cut" is `rank < R` for ties and closer rows alike;
//   * the plan is computed by every wavefront for itself (a 64-lane scan of <= 34 totals): no broadcast, one barrier less.
// What it declines (a record outside the counters' window, a list spanning more than 16 / 32 distances, a cut beyond 126) is
// flagged in big[] for k_rank_fused exactly like k_rank_cnt's leftovers; shapes it does not take at all (lists wanted,
// 8-byte records, more than 256 slices or 1024 pieces per query) stay with k_rank_cnt.
#pragma once
#include "hg_rank_cnt.hpp"

namespace hg {

#ifndef HG_RL_INFLIGHT
#define HG_RL_INFLIGHT 4
#endif
constexpr int RL_NPT = 4;                       // 16-byte pieces of the record row a thread fetches
constexpr int RL_MAX_PIECES = 256 * RL_NPT;

struct RankLeanLds { int cnt, off, tot, misc, bm, pref, rec, total; };     // byte offsets
__host__ __device__ inline RankLeanLds rank_lean_layout(int NB, i64 RW, int S, int rec_bytes, int nbc) {
    RankLeanLds l;
    const int RC_MAXB = rank_cnt_maxb(NB);
    const int NBall = NB < 128 ? NB : 128;
    const int NBc = nbc > 0 && nbc < NBall ? nbc : NBall;
    l.cnt = 0;                                   // [NBc + 1][64] u32: byte counter of thread 4 i + j = byte j of dword i; row NBc: padding
    int cb = (NBc + 1) * 256;
    if (cb < AP_LDS_BYTES + 8) cb = AP_LDS_BYTES + 8;          // (the AP epilogue reuses the counters)
    l.off = (cb + 15) & ~15;                     // [RC_MAXB + 1][128] u32: 16-bit rank of thread 2 i + j = half j of dword i; row RC_MAXB: dummy
    l.tot = l.off + (RC_MAXB + 1) * 512;         // [NB] u32
    l.misc = l.tot + ((NB + 3) & ~3) * 4;        // [16] u32: 7 = a record outside the window
    l.bm = l.misc + 64;                          // [2 RW] u32 + 16 bytes of slack
    l.pref = (l.bm + (int)(2 * RW) * 4 + 16 + 15) & ~15;       // (the zero fill ends here)  [S] u32 {first piece:16 | records:16}, then [4] pieces per wavefront
    l.rec = (l.pref + (S + 4) * 4 + 15) & ~15;   // the compacted records, whole 16-byte pieces
    l.total = l.rec + rec_bytes;
    return l;
}

template <int CTRL, int ROWS> __device__ __forceinline__ u32 rl_dpp_add(const u32 v) {
    return v + (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false);
}
// inclusive prefix sum over the 64 lanes: four row shifts, then lane 15 -> next row, lane 31 -> the upper half
__device__ __forceinline__ u32 rl_scan(u32 v) {
    v = rl_dpp_add<0x111, 0xF>(v);
    v = rl_dpp_add<0x112, 0xF>(v);
    v = rl_dpp_add<0x114, 0xF>(v);
    v = rl_dpp_add<0x118, 0xF>(v);
    v = rl_dpp_add<0x142, 0xA>(v);
    v = rl_dpp_add<0x143, 0xC>(v);
    return v;
}

static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HG_RANK_WAVES, HG_RANK_WAVES)))
void k_rank_lean(const u8* __restrict__ cand8, const RankLdsArgs a, u32* __restrict__ mbits32, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 rlds[];
    // the grid is padded to a multiple of 8: consecutive queries share an XCD, and with it the L2 lines that hold eight queries' pieces side by side
    const int q = logical_block(g.Q);
    if (q < 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = g.NB, S = g.S;
    const int NBall = NB < 128 ? NB : 128;
    const int NBc = a.nbc > 0 && a.nbc < NBall ? a.nbc : NBall;   // distances that have counters (<= 63: the launcher checks)
    const int RC_MAXB = rank_cnt_maxb(NB);
    const int bmw = (int)(2 * a.RW);
    constexpr int nthr = 256;
    const RankLeanLds L = rank_lean_layout(NB, a.RW, S, a.lds_recs, a.nbc);
    u32* cnt32 = (u32*)(rlds + L.cnt);
    u32* off32 = (u32*)(rlds + L.off);
    u32* tot = (u32*)(rlds + L.tot);
    u32* misc = (u32*)(rlds + L.misc);
    u32* bm = (u32*)(rlds + L.bm);
    u32* pref = (u32*)(rlds + L.pref);
    u8* rec8 = rlds + L.rec;
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;

#ifdef HG_RANK_PROFILE
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    int tkn = 0;
#define HG_TKL() do { if (tid == 0 && q < 4096) a.hwq[(i64)q * 16 + tkn] = (u32)(__builtin_amdgcn_s_memtime() - tk0); ++tkn; } while (0)
#else
#define HG_TKL() do {} while (0)
#endif

    // ---- everything this block reads from memory, requested at once ----
    const u32 failed = a.fail[q];
    const int T = a.cut ? a.cut[q] : NB - 1;                      // every record's distance is <= T
    const u32 c_s = tid < S ? a.sl_cnt[(i64)tid * g.Qpad + q] : 0u;
    const int PPS = (int)(a.cap >> 4);                            // pieces per slice (cap is a multiple of 16)
    const int PSP = a.spec_pieces;                                // pieces per slice fetched up front (<= PPS, S * PSP <= RL_MAX_PIECES)
    const int TP = S * PSP;
    const u32 inv = (65536u + (u32)PSP - 1u) / (u32)PSP;          // p / PSP = (p * inv) >> 16 for p < 1024, PSP <= 64
    // piece ps of slice s is row16[s * PPS + ps]
    const uint4* __restrict__ row16 = (const uint4*)(cand8 + (i64)q * a.crow);
    uint4 spec[RL_NPT];
#pragma unroll
    for (int k = 0; k < RL_NPT; ++k) {
        const u32 p = (u32)(tid + nthr * k);
        const u32 s = (p * inv) >> 16, ps = p - s * (u32)PSP;
        spec[k] = p < (u32)TP ? row16[s * (u32)PPS + ps] : uint4{0u, 0u, 0u, 0u};
    }
    for (int i = tid * 16; i < L.pref; i += nthr * 16) *(uint4*)(rlds + i) = uint4{0u, 0u, 0u, 0u};      // counters, ranks, totals, misc, bitmap

    if (tid == 0) a.big[q] = 0u;
    if (failed) {                                     // a slice of this query overflowed
        if (tid == 0) {
            if (a.mode == 3) atomicOr(&a.hown[(i64)NB * g.Qpad], 1u);   // tail word 0: this shard lost the bet
            else { atomicExch(a.err, 1); a.qbad[q] = 1u; }
        }
        if (a.mode == 3) for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = 0u;
        return;
    }
    const int wlo = a.cut ? (T - (NBc - 1) > 0 ? T - (NBc - 1) : 0) : 0;        // the counters cover distances [wlo, wlo + NBc)
    const u32 pad = (u32)(wlo + NBc);                 // a distance no record has (> T): what fills the tail of a slice's last piece

    // ---- pieces per slice -> exclusive prefix (thread = slice) ----
    const u32 pc = (c_s + 15u) >> 4;
    const u32 incl = rl_scan(pc);
    if (lane == 63) pref[S + wave] = incl;          // (outside the region being zeroed)
    __syncthreads();                                  // (the zero fill is done, too)
    u32 wbase = 0, n16 = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const u32 v = pref[S + w];
        wbase += w < wave ? v : 0u;
        n16 += v;
    }
    if (tid < S) pref[tid] = (wbase + incl - pc) | (c_s << 16);
    if (T > 126 || n16 * 16u > (u32)a.lds_recs || n16 > (u32)RL_MAX_PIECES) {     // (block-uniform) a cut the padding cannot top, more records than the LDS or the threads' registers hold
        if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); }
        return;
    }
    __syncthreads();
    HG_TKL();                                         // 0: loads issued, slice prefix

    // ---- compaction: piece ps of slice s lands at piece (first piece of s) + ps ----
    const u32 padw = pad * 0x01010101u;
    // `valid` (> 0) records of the slice from this piece on; the unused tail of its last piece becomes padding
    auto put_piece = [&](const uint4 v, const int valid, const u32 dst) {
        u32 w[4] = {v.x, v.y, v.z, v.w};
        if (valid < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nb = valid - 4 * i;
                const u32 keep = nb >= 4 ? 0xFFFFFFFFu : nb <= 0 ? 0u : (1u << (8 * nb)) - 1u;
                w[i] = (w[i] & keep) | (padw & ~keep);
            }
        }
        *(uint4*)(rec8 + 16u * dst) = uint4{w[0], w[1], w[2], w[3]};
    };
#pragma unroll
    for (int k = 0; k < RL_NPT; ++k) {
        const u32 p = (u32)(tid + nthr * k);
        if (p < (u32)TP) {
            const u32 s = (p * inv) >> 16, ps = p - s * (u32)PSP;
            const u32 e = pref[s];
            const int valid = (int)(e >> 16) - (int)(16u * ps);
            if (valid > 0) put_piece(spec[k], valid, (e & 0xFFFFu) + ps);
        }
    }
    if (pc > (u32)PSP) {                              // rare: a slice longer than what was fetched up front -- its own thread finishes it
        const u32 first = wbase + incl - pc;
        for (u32 ps = (u32)PSP; ps < pc; ++ps) put_piece(row16[(u32)tid * (u32)PPS + ps], (int)c_s - (int)(16u * ps), first + ps);
    }
    __syncthreads();
    HG_TKL();                                         // 1: compaction

    // ---- the thread's chunk: PPT whole pieces, kept in registers ----
    const int PPT = (int)((n16 + nthr - 1) / nthr);   // 1..4 (n16 <= 1024), block-uniform
    const u32 p0 = (u32)tid * (u32)PPT;
    uint4 ch[RL_NPT];
#pragma unroll
    for (int k = 0; k < RL_NPT; ++k)
        if (k < PPT && p0 + k < n16) ch[k] = *(const uint4*)(rec8 + 16u * (p0 + k));
        else ch[k] = uint4{0u, 0u, 0u, 0u};
    // count: one fire-and-forget LDS add per record into the thread's byte counter of that distance (row NBc: padding, and
    // whatever lies outside the window -- such a record makes the query leave this path)
    {
        const u32 one = 1u << (8 * (tid & 3));
        u8* cbase = (u8*)cnt32 + (tid >> 2) * 4;
        u32 mx = 0;
#pragma unroll
        for (int k = 0; k < RL_NPT; ++k) {
            if (k < PPT && p0 + k < n16) {
                const u32 w[4] = {ch[k].x, ch[k].y, ch[k].z, ch[k].w};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const u32 d = (w[i >> 2] >> (8 * (i & 3))) & 0x7Fu;
                    const u32 rel = d - (u32)wlo;                    // wraps below the window
                    mx = rel > mx ? rel : mx;
                    const u32 row = rel < (u32)NBc ? rel : (u32)NBc;
                    atomicAdd((u32*)(cbase + row * 256u), one);
                }
            }
        }
        if (mx > (u32)NBc) misc[7] = 1u;
    }
    __syncthreads();
    HG_TKL();                                         // 2: count
    if (misc[7]) { if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); } return; }
    // totals per distance: thread = (distance d, quarter j) sums 16 dwords of byte counters
    {
        const int d = tid >> 2, j = tid & 3;
        u32 sm = 0;
        if (d < NBc) {
            const uint4* c4 = (const uint4*)(cnt32 + d * 64 + j * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 v = c4[k];
                sm += __builtin_amdgcn_sad_u8(v.x, 0u, 0u) + __builtin_amdgcn_sad_u8(v.y, 0u, 0u) + __builtin_amdgcn_sad_u8(v.z, 0u, 0u) +
                      __builtin_amdgcn_sad_u8(v.w, 0u, 0u);
            }
        }
        sm += (u32)__builtin_amdgcn_update_dpp(0, (int)sm, 0xB1, 0xF, 0xF, false);       // quad_perm [1, 0, 3, 2]
        sm += (u32)__builtin_amdgcn_update_dpp(0, (int)sm, 0x4E, 0xF, 0xF, false);       // quad_perm [2, 3, 0, 1]
        if (d < NBc && j == 0 && wlo + d < NB) tot[wlo + d] = sm;
    }
    __syncthreads();
    HG_TKL();                                         // 3: totals
    if (a.mode == 3) for (int d = tid; d < NB; d += nthr) a.hown[(i64)d * g.Qpad + q] = tot[d];

    // ---- plan (k_plan for one shard), by every wavefront for itself: lane l speaks for distance wlo + l ----
    const bool has = lane < NBc && wlo + lane < NB;
    const u32 ctot = has ? tot[wlo + lane] : 0u;
    const u32 cinc = rl_scan(ctot);
    const u32 cstart = cinc - ctot;                   // global start of the lane's bucket
    u64 want = (u64)g.R;
    if (a.mode == 3) {                                // local ranking for k_merge_ranked: whatever this shard has, up to R
        const u32 have = (u32)__builtin_amdgcn_readlane((int)cinc, 63);
        if ((u64)have < want) want = have;
    }
    const u64 present = __ballot(ctot != 0u);
    const u64 reached = want > 0 ? __ballot(has && (u64)cinc >= want) : 0ull;
    const int t = reached ? wlo + (int)__builtin_ctzll(reached) : -1;
    if (tid == 0 && a.mode == 0) {
        if (t < 0) atomicExch(a.err, 1);              // the superset is too small: bet lost
        a.qbad[q] = t < 0 ? 1u : 0u;
    }
    if (t < 0) {
        if (a.mode == 3) for (int w = tid; w < bmw; w += nthr) grow[w] = 0u;   // nothing to rank: an empty bitmap
        return;
    }
    const int dmin = wlo + (int)__builtin_ctzll(present);          // (present != 0: something reached want > 0)
    const int nbk = t - dmin + 1;
    if (nbk > RC_MAXB) {                              // a list spanning many distances: the general kernel
        if (tid == 0) { a.big[q] = 1u; if (a.nleft) atomicAdd(a.nleft, 1u); }
        return;
    }
    // ranks: row k = bucket dmin + k; a thread's entry = the bucket's start + the records of that distance in earlier chunks
    for (int k = wave; k < nbk; k += 4) {
        const int row = dmin + k - wlo;
        const u32 x = cnt32[row * 64 + lane];         // threads 4 lane .. 4 lane + 3
        const u32 sm = __builtin_amdgcn_sad_u8(x, 0u, 0u);
        const u32 inc = rl_scan(sm);
        const u32 o0 = (u32)__builtin_amdgcn_readlane((int)cstart, row) + inc - sm;
        const u32 o1 = o0 + (x & 0xFFu), o2 = o1 + ((x >> 8) & 0xFFu), o3 = o2 + ((x >> 16) & 0xFFu);
        *(uint2*)(off32 + k * 128 + 2 * lane) = uint2{o0 | (o1 << 16), o2 | (o3 << 16)};
    }
    if (tid < 128) off32[RC_MAXB * 128 + tid] = (u32)want * 0x00010001u;        // dummy row: ranks from `want` on -- never inside the cut
    __syncthreads();
    HG_TKL();                                         // 4: plan + ranks

    // ---- place: a record's returning add is its rank; four adds in flight (same-thread adds to one entry stay in order) ----
    {
        const u32 sh = 16u * (u32)(tid & 1);
        const u32 one = 1u << sh;
        u8* obase = (u8*)off32 + (tid >> 1) * 4;
        const u32 wantu = (u32)want;
#pragma unroll
        for (int k = 0; k < RL_NPT; ++k) {
            if (k < PPT && p0 + k < n16) {
                const u32 w[4] = {ch[k].x, ch[k].y, ch[k].z, ch[k].w};
                constexpr int GR = HG_RL_INFLIGHT;                // returning adds in flight per thread (4, 8 or 16: one to four dwords of the piece)
#pragma unroll
                for (int i0 = 0; i0 < 16; i0 += GR) {
                    u32 r[GR];
#pragma unroll
                    for (int e = 0; e < GR; ++e) {
                        const int d = (int)((w[(i0 + e) >> 2] >> (8 * ((i0 + e) & 3))) & 0x7Fu);
                        const int kk = d > t ? RC_MAXB : d - dmin;
                        r[e] = atomicAdd((u32*)(obase + kk * 512), one);
                    }
#pragma unroll
                    for (int e = 0; e < GR; ++e) {
                        const u32 pos = (r[e] >> sh) & 0xFFFFu;
                        if (((w[(i0 + e) >> 2] >> (8 * ((i0 + e) & 3) + 7)) & 1u) && pos < wantu) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
                    }
                }
            }
        }
    }
    __syncthreads();
    HG_TKL();                                         // 5: place
    for (int w = tid; w < bmw; w += nthr) grow[w] = bm[w];
    HG_TKL();                                         // 6: bitmap out
    if (a.ap_shapes) {
        // metric.py:20-23 while the bitmap is in LDS: k_ap's very arithmetic (ap_eval), its scratch carved out of the counters
        const u64* bm64 = (const u64*)bm;
        if (a.ap_recip) ap_eval2<nthr>([&](const i64 w) { return bm64[w]; }, a.RW, g.R, a.ap_shapes, a.ap_recip, ap_lds_at(rlds + L.cnt), tid, a.ap + q, a.rel + q);
        else ap_eval<nthr>([&](const i64 w) { return bm64[w]; }, a.RW, g.R, a.ap_shapes, a.ap_recip, ap_lds_at(rlds + L.cnt), tid, a.ap + q, a.rel + q);
        HG_TKL();                                     // 7: AP
    }
}

}  // namespace hg
