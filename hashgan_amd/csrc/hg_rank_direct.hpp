// hashgan_amd -- the dense regime in ONE kernel: when R is a large share of N (the reference's own CIFAR-10 setting is
// R = N, lib/metric.py:14,19 with cfg.DATA.MAP_R = DB_SIZE) nothing needs selecting -- most or all rows are members of
// the ranked list -- so a block ranks ITS QUERY's rows straight from the packed tables: no histogram pass, no plan, no
// records in HBM.
//
// It is k_rank_cnt's counting sort with the rows themselves as the records: a tile of rows is read coalesced, every
// row's Hamming distance (xor + popcount, metric.py:13) and label-match bit (metric.py:17-19) become one byte in LDS,
// threads then own contiguous chunks of the tile --
//   pass A   per tile: fill, count into per-thread byte counters, add to the totals per distance;
//   plan     threshold t, tie quota, bucket starts (k_plan's arithmetic) from the totals of ALL rows: exact, nothing guessed;
//   pass B   per tile: fill again, count, per-bucket prefix over the threads, place (one returning LDS add per row) --
//            the match bit goes to its rank in the LDS bitmap, and (hg_topr) index and distance to the ranked lists.
// Canonical order (distance, index) by construction: tiles and chunks are in index order, the sort is stable.
// Used for R = N when the counters, the R-bit bitmap and a tile of >= 8192 rows fit 80 KB of LDS (two blocks per CU):
// C1 (Q = 1000, N = R = 54 000, b = 32) 0.48 -> 0.25 ms against k_rank_fused's 64-wide ballot ranking over the rows; Q = 10k,
// N = R = 54 000, b = 64: 5.5 -> 2.85 ms.  The kernel ranks ANY R, but one block per query re-reads the whole database for
// every query and a CU holds few blocks: for N/8 < R < N it loses to k_hist + k_select + k_rank_fused (N = 200k, R = 100k:
// 15.7 vs 12.3 ms; N = 1M, R = 500k: 82 vs 70 ms) and is only taken on request ("rank_direct" = 2).
#pragma once
#include "hg_kernels.hpp"

namespace hg {

struct RankDirectArgs {
    const u32* qc;         // [Q][NW]
    const u64* qlab;       // [Q][LW]
    const u32* db;         // [N][NW]
    const u64* dblab;      // [N][LW]
    int* err;
    u32* qbad;             // [Q]
    i64 RW;                // 64-bit words per bitmap row
    int tile_rows;         // rows per tile (a multiple of 8, <= 252 * 256)
    int want_lists;
    // hg_map: the AP of the query leaves with its ranking, out of the bitmap in LDS (metric.py:20-23; ap_eval2) -- no k_ap launch
    const ApShape* ap_shapes;   // null: no AP here
    const double* ap_recip;     // [R + 1 + AP_RECIP_SLACK]
    double* ap;                 // [Q]
    u32* rel;                   // [Q]
};

struct RankDirectLds { int cnt, off, tot, misc, done, tilecnt, qsh, bm, rec, total; };      // byte offsets
__host__ __device__ inline RankDirectLds rank_direct_layout(int NB, i64 RW, int tile_rows) {
    RankDirectLds l;
    const int NBc = NB < 128 ? NB : 128;
    l.cnt = 0;                                    // [NBc][64] u32: byte counter of thread 4 i + j = byte j of dword i   (the AP epilogue's scratch afterwards)
    l.off = l.cnt + (NBc * 256 > AP_LDS_BYTES + 8 ? NBc * 256 : (AP_LDS_BYTES + 8 + 15) & ~15);   // [NBc + 1][128] u32: 16-bit offset of thread 2 i + j (row NBc: dummy for rows beyond the cut)
    l.tot = l.off + (NBc + 1) * 512;              // [NBc] u32: totals, then bucket starts
    l.misc = l.tot + NBc * 4;                     // [16] u32
    l.done = l.misc + 64;                         // [NBc + 1] u32: rows of each bucket placed by earlier tiles
    l.tilecnt = l.done + (NBc + 1) * 4;           // [NBc + 1] u32
    l.qsh = l.tilecnt + (NBc + 1) * 4;            // the query: 8 code words, 2 label words
    l.bm = (l.qsh + 48 + 15) & ~15;               // [2 RW] u32
    l.rec = (l.bm + (int)(2 * RW) * 4 + 15) & ~15;   // [tile_rows] u8 {match:1 | dist:7}
    l.total = l.rec + ((tile_rows + 15) & ~15);
    return l;
}

static __global__ __launch_bounds__(256) void k_rank_direct(const RankDirectArgs a, u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                                     u32* __restrict__ mbits32, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 dlds[];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nthr = 256, NWAV = 4;
    const int NB = g.NB, NW = g.NW, LW = g.LW;
    const int NBc = NB < 128 ? NB : 128;
    const int bmw = (int)(2 * a.RW);
    const RankDirectLds L = rank_direct_layout(NB, a.RW, a.tile_rows);
    u32* cnt32 = (u32*)(dlds + L.cnt);
    u32* off32 = (u32*)(dlds + L.off);
    u32* tot = (u32*)(dlds + L.tot);
    u32* misc = (u32*)(dlds + L.misc);
    u32* done = (u32*)(dlds + L.done);
    u32* tilecnt = (u32*)(dlds + L.tilecnt);
    u32* qsh = (u32*)(dlds + L.qsh);
    u64* qlsh = (u64*)(dlds + L.qsh + 32);
    u32* bm = (u32*)(dlds + L.bm);
    u8* rec8 = dlds + L.rec;
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;

    for (int i = tid; i < L.rec / 4; i += nthr) ((u32*)dlds)[i] = 0u;          // counters, offsets, totals, progress, bitmap
    __syncthreads();
    if (tid < NW) qsh[tid] = a.qc[(i64)q * NW + tid];
    if (tid >= 32 && tid < 32 + LW) qlsh[tid - 32] = a.qlab[(i64)q * LW + tid - 32];
    __syncthreads();

    const u32 N = (u32)g.N;
    const u32 TC = (u32)a.tile_rows;
    const u32 ntile = (N + TC - 1) / TC;

    // rows [T0, T1) -> one byte each in LDS; consecutive threads take consecutive rows (coalesced).  Eight rows' loads per
    // thread are in flight at once: a block is alone on its CU (its LDS), one wavefront per SIMD -- without that every row
    // would pay a full memory latency (first version: 239 ms for 10k queries x 1M rows).
    auto fill = [&](const u32 T0, const u32 T1) {
        constexpr int UF = 8;
        for (u32 i0f = T0 + tid; i0f < T1; i0f += UF * nthr) {
            if (NW <= 2 && LW == 1) {                  // the usual widths: straight-line loads
                u32 c0[UF], c1[UF];
                u64 lb[UF];
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const u32 i = i0f + u * nthr;
                    const bool ok = i < T1;
                    const u32* __restrict__ row = a.db + (i64)(ok ? i : T0) * NW;
                    c0[u] = row[0];
                    c1[u] = NW == 2 ? row[1] : 0u;
                    lb[u] = a.dblab[ok ? i : T0];
                }
                const u32 q0 = qsh[0], q1 = NW == 2 ? qsh[1] : 0u;
                const u64 ql0 = qlsh[0];
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const u32 i = i0f + u * nthr;
                    const u32 d = (u32)__builtin_popcount(q0 ^ c0[u]) + (u32)__builtin_popcount(q1 ^ c1[u]);
                    if (i < T1) rec8[i - T0] = (u8)(d | ((lb[u] & ql0) ? 0x80u : 0u));
                }
            } else {
                for (int u = 0; u < UF; ++u) {
                    const u32 i = i0f + u * nthr;
                    if (i >= T1) break;
                    u32 d = 0;
                    const u32* __restrict__ row = a.db + (i64)i * NW;
                    for (int w = 0; w < NW; ++w) d += (u32)__builtin_popcount(qsh[w] ^ row[w]);
                    u64 any = 0;
                    const u64* __restrict__ lrow = a.dblab + (i64)i * LW;
                    for (int w = 0; w < LW; ++w) any |= qlsh[w] & lrow[w];
                    rec8[i - T0] = (u8)(d | (any ? 0x80u : 0u));
                }
            }
        }
    };
    const u32* rec32 = (const u32*)rec8;
    u32 i0 = 0, i1 = 0;
    auto count_tile = [&](const u32 m) {                 // thread `tid` owns rows [i0, i1) of the tile: chunk = 4 (mod 8) bytes (bank spread)
        u32 chunk = (m + nthr - 1) / nthr;
        chunk += (4u - (chunk & 7u)) & 7u;
        i0 = (u32)tid * chunk < m ? (u32)tid * chunk : m;
        i1 = i0 + chunk < m ? i0 + chunk : m;
        const u32 one = 1u << (8 * (tid & 3));
#pragma unroll 2
        for (u32 i = i0; i < i1; i += 4) {
            const u32 v = rec32[i >> 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 d = (v >> (8 * j)) & 0x7Fu;
                if (i + j < i1) atomicAdd(&cnt32[d * 64 + (tid >> 2)], one);
            }
        }
    };
    auto add_totals = [&]() {                            // thread = (distance d, quarter j) sums 16 dwords of byte counters
        for (int d0 = 0; d0 < NBc; d0 += 64) {
            const int d = d0 + (tid >> 2), j = tid & 3;
            u32 sm = 0;
            if (d < NBc) {
#pragma unroll
                for (int k = 0; k < 16; ++k) sm += __builtin_amdgcn_sad_u8(cnt32[d * 64 + j * 16 + k], 0u, 0u);
            }
            sm += (u32)__shfl_xor((int)sm, 1);
            sm += (u32)__shfl_xor((int)sm, 2);
            if (d < NBc && j == 0) tot[d] += sm;
        }
    };
    auto zero_counters = [&]() {
        for (int i = tid; i < NBc * 64; i += nthr) cnt32[i] = 0u;
    };

    // ---- pass A: totals per distance over all rows ----
    for (u32 tl = 0; tl < ntile; ++tl) {
        const u32 T0 = tl * TC, T1 = T0 + TC < N ? T0 + TC : N;
        fill(T0, T1);
        __syncthreads();
        count_tile(T1 - T0);
        __syncthreads();
        add_totals();
        __syncthreads();
        if (ntile > 1) {
            zero_counters();
            __syncthreads();
        }
    }
    // ---- plan (k_plan for one shard), by wavefront 0: lane l speaks for distances l, l + 64 ----
    if (wave == 0) {
        const u64 want = (u64)g.R;                       // R <= N: the cut always exists
        u32 base = 0;
        int t = -1, dmin = -1;
        u32 cntlt = 0;
        for (int d0 = 0; d0 < NBc && t < 0; d0 += 64) {
            const int d = d0 + lane;
            const u32 c = d < NBc ? tot[d] : 0u;
            u32 inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = (u32)__shfl_up((int)inc, off);
                if (lane >= off) inc += v;
            }
            const u32 start = base + inc - c;            // global start of bucket d
            const u64 present = __ballot(c != 0u);
            if (dmin < 0 && present) dmin = d0 + (int)__builtin_ctzll(present);
            const u64 reached = __ballot((u64)base + inc >= want && d < NBc);
            if (reached) {
                const int lt = (int)__builtin_ctzll(reached);
                t = d0 + lt;
                cntlt = (u32)__shfl((int)start, lt);
                if (lane <= lt) tot[d] = start;
            } else {
                if (d < NBc) tot[d] = start;
                base += (u32)__shfl((int)inc, 63);
            }
        }
        if (lane == 0) {
            misc[0] = (u32)t;
            misc[1] = cntlt;
            misc[2] = (u32)(want - (u64)cntlt);          // quota of the tie bucket
            misc[3] = (u32)(dmin < 0 ? 0 : dmin);
            a.qbad[q] = t < 0 ? 1u : 0u;
            if (t < 0) atomicExch(a.err, 1);             // (cannot happen for R <= N; the caller would rerun)
        }
    }
    __syncthreads();
    const int t = (int)misc[0];
    if (t < 0) return;
    const u32 cntlt = misc[1], quota = misc[2];
    const int dmin = (int)misc[3];
    const int nbk = t - dmin + 1;

    // ---- pass B: per tile, offsets of every thread inside each bucket [dmin, t], then placement ----
    u32* __restrict__ oi = out_idx + (i64)q * g.R;
    u8* __restrict__ od = out_dist + (i64)q * g.R;
    for (u32 tl = 0; tl < ntile; ++tl) {
        const u32 T0 = tl * TC, T1 = T0 + TC < N ? T0 + TC : N;
        if (ntile > 1) {                                 // (one tile: rows and counters are still in place)
            fill(T0, T1);
            __syncthreads();
            count_tile(T1 - T0);
            __syncthreads();
        }
        for (int k = wave; k < nbk; k += NWAV) {
            const u32 x = cnt32[(dmin + k) * 64 + lane];
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
            if (lane == 63) tilecnt[k] = inc;                            // the tile's rows at this distance
        }
        __syncthreads();
        {
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
                    const int k = in ? d - dmin : NBc;
                    r[j] = atomicAdd(&off32[k * 128 + (tid >> 1)], one);
                    st[j] = in ? tot[d] + done[k] : 0u;                   // bucket start + what earlier tiles placed there
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = (int)(meta[j] & 0xFFu);
                    const bool in = d <= t && meta[j] != 0xFFFFu;
                    const u32 rk = (r[j] >> sh) & 0xFFFFu;
                    if (in && (d < t || st[j] - cntlt + rk < quota)) {    // ties: the first `quota` in index order
                        const u32 pos = st[j] + rk;
                        if (a.want_lists) { oi[pos] = g.idx_base + T0 + i + j; od[pos] = (u8)d; }
                        if (meta[j] & 0x100u) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
                    }
                }
            }
        }
        __syncthreads();
        if (ntile > 1) {
            for (int k = tid; k < nbk; k += nthr) done[k] += tilecnt[k];
            zero_counters();                             // (the offset rows are rewritten by the next tile's scan; the dummy row's content is never used)
            __syncthreads();
        }
    }
    for (int w = tid; w < bmw; w += nthr) grow[w] = bm[w];
    if (a.ap_shapes) {
        __syncthreads();                                 // (the counters -- the AP's scratch from here on -- are no longer read)
        const u64* bm64 = (const u64*)bm;
        ap_eval2<nthr>([&](const i64 w) { return bm64[w]; }, a.RW, g.R, a.ap_shapes, a.ap_recip, ap_lds_at(dlds + L.cnt), tid, a.ap + q, a.rel + q);
    }
}

}  // namespace hg
