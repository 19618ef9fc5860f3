// hashgan_amd -- the dense regime N/8 < R <= N through a byte matrix: when R is a large share of N
// (/root/reference/lib/metric.py:14,19 with cfg.DATA.MAP_R of the order of DB_SIZE) most rows of the database are members
// of every ranked list, so selecting them costs more than it saves -- 8-byte records of R = N/2 rows per query were 40 GB
// written and read again at Q = 10k, N = 1M (68.8 ms in round 3), and k_rank_direct re-reads 16 bytes of codes + labels per
// row and pass for every query through the L2 (82 ms).  Here
//   k_dense_bytes   writes D[q][i] = {match:1 | dist:7} for EVERY pair, one byte: a thread keeps four rows in registers,
//                   walks the queries with scalar loads and stores one dword per query -- a wavefront's store is 256
//                   contiguous bytes (metric.py:13 distance by xor + popcount, metric.py:17-19 label match);
//   k_rank_dense    one block per query, THREAD = a contiguous range of rows: it streams its range of D's row in 16-byte
//                   pieces, eight in flight; pass A counts into the thread's private column of LDS counters [dist][thread] (conflict
//                   free: the bank is the lane), the block adds the totals, plans the cut (k_plan's arithmetic: exact,
//                   nothing guessed) and turns the counters into global ranks (bucket start + the earlier threads'
//                   rows at that distance); pass B streams the range again and every row at or below the cut takes
//                   its rank with ONE returning LDS add on its own column -- a member iff rank < R, ties cut in index
//                   order by construction -- and drops its match bit into the R-bit LDS bitmap; the AP leaves from the
//                   epilogue (ap_eval2) like everywhere else.
// Canonical order (distance, index): thread ranges are in index order and a thread walks its rows in order; no atomic's
// result depends on an order of arrival (a column has one writer).  D is Q x N bytes in HBM (10 GB at Q = 10k, N = 1M;
// queries are chunked to a budget), written once and read twice: 30 bytes of HBM traffic per 16 pairs.
// Layout of a query's row of D: thread tau of k_rank_dense owns rows [tau Lr, (tau + 1) Lr), Lr = 16 P, and the row is
// stored piece-major -- D[q][p][tau][16]: the 16-byte piece p of every thread's range side by side -- so that a
// wavefront's load of "my next piece" is 1 KB of contiguous memory and k_dense_bytes' stores are too (a first version
// with each thread's range contiguous in memory had every lane on its own cache line: 32.8 ms for the ranking at
// Q = 10k, N = 1M, R = 500k).
#pragma once
#include "hg_kernels.hpp"

namespace hg {

constexpr int RD_THREADS = 256;
#ifndef HG_RD_PF
#define HG_RD_PF 8
#endif
constexpr int RD_PF = HG_RD_PF;           // 16-byte pieces a thread has in flight while it works on as many (16 measured the same: the loads' latency is covered)
// (a row past N carries the distance b with the match bit clear: see padrow in k_rank_dense; no test per row; codes of <= 126 bits)

// rows per thread range (a multiple of 16) for a database of N rows
__host__ __device__ inline i64 rank_dense_pieces(i64 N) { return ((N + RD_THREADS - 1) / RD_THREADS + 15) / 16; }

// grid.x = 4 P blocks: block bx covers quarter (bx & 3) of piece p = bx >> 2 -- threads t = 0..1023 of a piece are
// (tau = t >> 2, j = t & 3): rows tau Lr + 16 p + 4 j .. + 3, stored as the dword at p * 4096 + 4 t of the query's row.
template <int NW, int LW>
static __global__ __launch_bounds__(256) void k_dense_bytes(const u32* __restrict__ qc, const u64* __restrict__ qlab,
                                                            const u32* __restrict__ db, const u64* __restrict__ dblab,
                                                            u8* __restrict__ D, const i64 N, const i64 Npad, const int q0, const int nq, const int qper, const u32 padbyte) {
    const i64 P = Npad / (RD_THREADS * 16);
    const i64 p = blockIdx.x >> 2;
    const int t = (blockIdx.x & 3) * 256 + threadIdx.x;
    const i64 r0 = (i64)(t >> 2) * (P * 16) + p * 16 + (t & 3) * 4;
    u32 c[4][NW];
    u64 l[4][LW];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const i64 r = r0 + u < N ? r0 + u : N - 1;
#pragma unroll
        for (int k = 0; k < NW; ++k) c[u][k] = db[r * NW + k];
#pragma unroll
        for (int k = 0; k < LW; ++k) l[u][k] = dblab[r * LW + k];
    }
    u32 padmask = 0, padval = 0;                                      // rows past N: the pad byte whatever was computed
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (r0 + u >= N) { padmask |= 0xFFu << (8 * u); padval |= padbyte << (8 * u); }
    const int qa = blockIdx.y * qper, qe = qa + qper < nq ? qa + qper : nq;
    u8* __restrict__ out = D + (i64)qa * Npad + p * 4096 + (i64)t * 4;
    for (int qi = qa; qi < qe; ++qi, out += Npad) {
        const i64 q = q0 + qi;                                         // wave-uniform: the query's words come through scalar loads
        u32 qw[NW];
        u64 ql[LW];
#pragma unroll
        for (int k = 0; k < NW; ++k) qw[k] = qc[q * NW + k];
#pragma unroll
        for (int k = 0; k < LW; ++k) ql[k] = qlab[q * LW + k];
        u32 word = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32 d = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) d += (u32)__builtin_popcount(c[u][k] ^ qw[k]);
            u64 any = 0;
#pragma unroll
            for (int k = 0; k < LW; ++k) any |= l[u][k] & ql[k];
            word |= (d | (any ? 0x80u : 0u)) << (8 * u);
        }
        *(u32*)out = (word & ~padmask) | padval;
    }
}

struct RankDenseArgs {
    const u8* D;           // [nq][Npad] {match:1 | dist:7}, this launch's queries, piece-major (see above)
    i64 Npad;              // bytes per query: 4096 P
    int q0;                // first query of the launch
    int* err;
    u32* qbad;             // [Q]
    i64 RW;                // 64-bit words per bitmap row
    const ApShape* ap_shapes;   // null: no AP here
    const double* ap_recip;     // [R + 1 + AP_RECIP_SLACK]
    double* ap;                 // [Q]
    u32* rel;                   // [Q]
    // SLICES: the rows are the one-byte records of a bet (k_select_mx3 / mx4) instead of the byte matrix -- slice s of query q
    // is sl_cnt[s][q] bytes at cand8 + q * crow + s * cap, in index order; thread = (slice, part of it)
    const u8* cand8;
    const u32* sl_cnt;          // [S][Qpad]
    const u32* fail;            // [Qpad]: a slice of the query overflowed (the bet is lost for it)
    u32 cap;
    i64 crow;
    i64 rw_part;                // DENSE, 0 or the 64-bit bitmap words of ONE of gridDim.y blocks per query: block y ranks everything but keeps
                                // (in LDS) and writes only the ranks [y * 64 rw_part, (y + 1) * 64 rw_part) -- an R-bit bitmap beyond one block's LDS
    const u32* only;            // SLICES, optional [Q]: rank only the flagged queries (what k_rank_lean declined)
    int nrows;                  // counter rows (distances 0 .. nrows - 1; row `nrows`: pad bytes).  DENSE: b + 1.  SLICES: the bet's cut never exceeds
                                // b/2 + 1 (the sampled pass stops there; a thinner sample takes everything, overflows and is flagged): b/2 + 2 rows
};

// GBM: the R-bit bitmap stays in global memory (zeroed by the caller; members' match bits arrive by fire-and-forget
// atomic ORs -- ~5 % of the rows -- and k_ap evaluates it afterwards) instead of LDS: the block is then its counter
// columns alone, (b + 2) KB, and two to four blocks share a CU where a 62 KB bitmap (R = 500k) leaves room for one.
struct RankDenseLds { int cnt, tot, misc, bm, total; };              // byte offsets
__host__ __device__ inline RankDenseLds rank_dense_layout(int NR, i64 RW, bool gbm) {     // NR: counter rows in all (DENSE: b + 1; SLICES: nrows + 1, the pad row)
    RankDenseLds l;
    l.cnt = 0;                                   // [NR][256] u32: thread tid's counter of distance d at d * 256 + tid (the AP epilogue's scratch afterwards)
    int cb = NR * RD_THREADS * 4;
    if (cb < AP_LDS_BYTES + 8) cb = (AP_LDS_BYTES + 8 + 15) & ~15;
    l.tot = l.cnt + cb;                          // [NR] u32: totals, then bucket starts
    l.misc = l.tot + ((NR * 4 + 15) & ~15);      // [16] u32
    l.bm = l.misc + 64;                          // [2 RW + 1] u32 (the last word takes the ORs of rows beyond the cut)
    l.total = l.bm + (gbm ? 0 : (((int)(2 * RW) + 1) * 4 + 15) & ~15);
    return l;
}

// SLICES (the bet with a long list, R beyond k_rank_lean's LDS): the same two passes over the query's RECORDS.  The record rows
// are a superset of the top R by construction of the bet; what this kernel verifies is what every rank kernel of the bet
// verifies -- no slice overflowed, and the records number at least R (else the plan finds no cut: the bet is lost and the
// caller escalates).  k_rank_cnt ranks such lists tile by tile (compaction, per-tile scans: 2.8 ms at Q = 10k, R = 50 000).
template <bool LISTS, bool GBM, bool SLICES>
static __global__ __launch_bounds__(RD_THREADS) void k_rank_dense(const RankDenseArgs a, u32* __restrict__ out_idx, u8* __restrict__ out_dist,
                                                                  u32* __restrict__ mbits32, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 dlds[];
    constexpr int nthr = RD_THREADS, NWAV = RD_THREADS / 64;
    const int q = a.q0 + blockIdx.x;
    if (SLICES && a.only && !a.only[q]) return;          // (block-uniform) not one of the flagged queries
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = a.nrows;                          // distances with a counter row; row NB takes the pad bytes
    const int bmw = (int)(2 * a.RW);
    // The pad bytes' row.  SLICES: one past the distances.  DENSE: the LAST distance's own row, b -- pad rows have the highest
    // indices, so inside that bucket they rank behind every real row and beyond N >= R: never members, and the cut is found as
    // without them (one KB less of LDS: C1 gets four blocks per CU instead of three -- 1000 queries in one round).
    const int padrow = SLICES ? NB : NB - 1;
    const int NR = padrow + 1 > NB ? padrow + 1 : NB;
    const i64 RWl = a.rw_part ? a.rw_part : a.RW;      // bitmap words this block holds
    const u32 rk_lo = a.rw_part ? (u32)(blockIdx.y * a.rw_part * 64) : 0u;
    const RankDenseLds L = rank_dense_layout(NR, RWl, GBM);
    u32* cnt = (u32*)(dlds + L.cnt);
    u32* tot = (u32*)(dlds + L.tot);
    u32* misc = (u32*)(dlds + L.misc);
    u32* bm = (u32*)(dlds + L.bm);
    u32* __restrict__ grow = mbits32 + (i64)q * 2 * a.RW;

    for (int i = tid; i < NR * nthr; i += nthr) cnt[i] = 0u;
    const int bml = (int)(2 * RWl);                    // its words
    if (!GBM) for (int i = tid; i <= bml; i += nthr) bm[i] = 0u;
    if (tid < NB) tot[tid] = 0u;

    // DENSE: thread tid owns rows [tid Lr, (tid + 1) Lr), Lr = 16 P: piece p of its range is the uint4 at p * 256 + tid of the query's row of D.
    // SLICES: thread tid = (slice tid / parts, part tid % parts) owns a run of whole 16-byte pieces of that slice, contiguous in memory.
    i64 P, pstride;
    const uint4* __restrict__ drow;
    u32 tailkeep[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // SLICES: the valid bytes of the thread's LAST piece
    if (SLICES) {
        if (a.fail[q]) {                                 // (block-uniform) a slice overflowed: lost
            if (tid == 0) { atomicExch(a.err, 1); a.qbad[q] = 1u; }
            return;
        }
        const int parts = nthr / g.S > 0 ? nthr / g.S : 1;             // the launcher keeps S <= 256
        const int sl = tid / parts, part = tid - sl * parts;
        const u32 c = sl < g.S ? a.sl_cnt[(i64)sl * g.Qpad + q] : 0u;
        const u32 np = (c + 15u) >> 4;                                 // pieces of the slice (c <= cap: the select clamps the count)
        const u32 pa = np * (u32)part / (u32)parts, pb = np * (u32)(part + 1) / (u32)parts;
        P = (i64)(pb - pa);
        pstride = 1;
        drow = (const uint4*)(a.cand8 + (i64)q * a.crow + (i64)(sl < g.S ? sl : 0) * a.cap) + pa;
        if (pb == np && (c & 15u)) {                                   // the slice's last piece is partial: bytes past the count become pad rows
            const int valid = (int)(c & 15u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nb = valid - 4 * i;
                tailkeep[i] = nb >= 4 ? 0xFFFFFFFFu : nb <= 0 ? 0u : (1u << (8 * nb)) - 1u;
            }
        }
    } else {
        P = a.Npad / (nthr * 16);
        pstride = nthr;
        drow = (const uint4*)(a.D + (i64)blockIdx.x * a.Npad) + tid;
    }
    const u32 padw = (u32)padrow * 0x01010101u;
    // (SLICES: a record beyond the counter rows cannot come from a bet that holds -- it would index past the columns: clamp it onto the pad row)
    auto rowof = [&](const u32 d) -> u32 { return SLICES ? (d < (u32)NB ? d : (u32)padrow) : d; };
    // piece p of the thread's range, pad bytes in place
    auto piece = [&](const i64 p) -> uint4 {
        uint4 v = drow[(p < P ? p : 0) * pstride];
        if (SLICES && p == P - 1) {
            v.x = (v.x & tailkeep[0]) | (padw & ~tailkeep[0]); v.y = (v.y & tailkeep[1]) | (padw & ~tailkeep[1]);
            v.z = (v.z & tailkeep[2]) | (padw & ~tailkeep[2]); v.w = (v.w & tailkeep[3]) | (padw & ~tailkeep[3]);
        }
        return v;
    };
    u32* mycnt = cnt + tid;
    __syncthreads();

    // ---- pass A: the thread's rows per distance: extract, address, add -- nothing else ----
    {
        uint4 v[RD_PF], nx[RD_PF];
#pragma unroll
        for (int k = 0; k < RD_PF; ++k) v[k] = piece(k);
        for (i64 p0 = 0; p0 < P; p0 += RD_PF) {
#pragma unroll
            for (int k = 0; k < RD_PF; ++k) nx[k] = piece(p0 + RD_PF + k);
#pragma unroll
            for (int k = 0; k < RD_PF; ++k) {
                if (p0 + k < P) {
                    const u32 w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                    u32* ad[16];                                        // (addresses first, adds after: sixteen independent chains for the one wavefront a SIMD has)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ad[e] = mycnt + rowof((w4[e >> 2] >> (8 * (e & 3))) & 0x7Fu) * nthr;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int e = 0; e < 16; ++e) atomicAdd(ad[e], 1u);
                }
            }
#pragma unroll
            for (int k = 0; k < RD_PF; ++k) v[k] = nx[k];
        }
    }
    __syncthreads();
    // totals per distance: wavefront w sums the columns of distances w, w + 4, ...
    for (int d = wave; d < NB; d += NWAV) {
        u32 s = 0;
#pragma unroll
        for (int r = 0; r < nthr / 64; ++r) s += cnt[d * nthr + r * 64 + lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += (u32)__shfl_xor((int)s, off);
        if (lane == 0) tot[d] = s;
    }
    __syncthreads();
    // ---- plan (k_plan for one shard), by wavefront 0: lane l speaks for distances l, l + 64 ----
    if (wave == 0) {
        const u64 want = (u64)g.R;                       // R <= N: the cut always exists
        u32 base = 0;
        int t = -1, dmin = -1;
        u32 cntlt = 0;
        for (int d0 = 0; d0 < NB && t < 0; d0 += 64) {
            const int d = d0 + lane;
            const u32 c = d < NB ? tot[d] : 0u;
            u32 inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = (u32)__shfl_up((int)inc, off);
                if (lane >= off) inc += v;
            }
            const u32 start = base + inc - c;            // global start of bucket d
            const u64 present = __ballot(c != 0u);
            if (dmin < 0 && present) dmin = d0 + (int)__builtin_ctzll(present);
            const u64 reached = __ballot((u64)base + inc >= want && d < NB);
            if (reached) {
                const int lt = (int)__builtin_ctzll(reached);
                t = d0 + lt;
                cntlt = (u32)__shfl((int)start, lt);
                if (lane <= lt) tot[d] = start;
            } else {
                if (d < NB) tot[d] = start;
                base += (u32)__shfl((int)inc, 63);
            }
        }
        if (lane == 0) {
            misc[0] = (u32)t;
            misc[1] = cntlt;
            misc[3] = (u32)(dmin < 0 ? 0 : dmin);
            a.qbad[q] = t < 0 ? 1u : 0u;
            if (t < 0) atomicExch(a.err, 1);             // (cannot happen for R <= N; the caller would rerun)
        }
    }
    __syncthreads();
    const int t = (int)misc[0];
    if (t < 0) return;
    // counters -> global ranks: start of the bucket + the rows of earlier threads at that distance; the columns of
    // distances beyond the cut (and of the pad rows) start at 2^31: whatever such a row draws is no rank below R
    for (int d = wave; d < NR; d += NWAV) {
        if (d <= t) {
            u32 carry = tot[d];
#pragma unroll
            for (int r = 0; r < nthr / 64; ++r) {
                const u32 x = cnt[d * nthr + r * 64 + lane];
                u32 inc = x;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const u32 v = (u32)__shfl_up((int)inc, off);
                    if (lane >= off) inc += v;
                }
                cnt[d * nthr + r * 64 + lane] = carry + inc - x;
                carry += (u32)__shfl((int)inc, 63);
            }
        } else {
#pragma unroll
            for (int r = 0; r < nthr / 64; ++r) cnt[d * nthr + r * 64 + lane] = 0x80000000u;
        }
    }
    __syncthreads();

    // ---- pass B: every row draws a rank from its thread's column (one returning LDS add, no test); members (rank < R)
    //      that match leave their bit ----
    {
        const u32 R = (u32)g.R;
        const u32 rk_hi = a.rw_part ? (rk_lo + (u32)(a.rw_part * 64) < R ? rk_lo + (u32)(a.rw_part * 64) : (rk_lo < R ? R : rk_lo)) : R;
        const u32 span = rk_hi - rk_lo;                  // this block's ranks: [rk_lo, rk_lo + span)
        u32* __restrict__ oi = out_idx + (i64)q * g.R;
        u8* __restrict__ od = out_dist + (i64)q * g.R;
        const u32 row0 = g.idx_base + (u32)((i64)tid * P * 16);
        uint4 v[RD_PF], nx[RD_PF];
#pragma unroll
        for (int k = 0; k < RD_PF; ++k) v[k] = piece(k);
        for (i64 p0 = 0; p0 < P; p0 += RD_PF) {
#pragma unroll
            for (int k = 0; k < RD_PF; ++k) nx[k] = piece(p0 + RD_PF + k);
#pragma unroll
            for (int k = 0; k < RD_PF; ++k) {
                if (p0 + k < P) {
                    const u32 w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                    u32 pos[16];
                    u32* ad[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) ad[e] = mycnt + rowof((w4[e >> 2] >> (8 * (e & 3))) & 0x7Fu) * nthr;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int e = 0; e < 16; ++e) pos[e] = atomicAdd(ad[e], 1u);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const u32 w = w4[e >> 2];
                        const u32 ps = pos[e] - rk_lo;                 // (wraps for ranks below the block's window: not < span)
                        const bool hit = (w & (0x80u << (8 * (e & 3)))) != 0u && ps < span;
                        if (hit) {
                            if (GBM) atomicOr(&grow[ps >> 5], 1u << (ps & 31));
                            else atomicOr(&bm[ps >> 5], 1u << (ps & 31));
                        }
                        if (LISTS && ps < span) {
                            oi[ps + rk_lo] = row0 + (u32)((p0 + k) * 16 + e);
                            od[ps + rk_lo] = (u8)((w >> (8 * (e & 3))) & 0x7Fu);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < RD_PF; ++k) v[k] = nx[k];
        }
    }
    if (GBM) return;
    __syncthreads();
    for (int w = tid; w < bml; w += nthr)
        if ((int)(rk_lo >> 5) + w < bmw) grow[(rk_lo >> 5) + w] = bm[w];
    if (a.ap_shapes && !a.rw_part) {
        __syncthreads();                                 // (the counters -- the AP's scratch from here on -- are no longer read)
        const u64* bm64 = (const u64*)bm;
        ap_eval2<nthr>([&](const i64 w) { return bm64[w]; }, a.RW, g.R, a.ap_shapes, a.ap_recip, ap_lds_at(dlds + L.cnt), tid, a.ap + q, a.rel + q);
    }
}

}  // namespace hg
