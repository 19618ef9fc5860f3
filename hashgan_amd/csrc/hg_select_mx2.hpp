// hashgan_amd -- matrix-core select for codes of up to 64 bits with TWO rows per accumulator.
//
// k_select_mx leaves the vector ALU one op per (query, row) pair: the sign of an accumulator is one hit
// bit.  For short codes an f32 accumulator has room for two distances, and then one VALU op harvests
// two bits.  The MX block scale makes it free: a lane's 32 fp4 A elements are one scale block, so the
// k-half of lane-half 0 carries code word w of row a at scale 2^0 and the k-half of lane-half 1 the
// same word of ANOTHER row b at scale 2^11, against the query word on both halves:
//     acc = C + sum_k a_k s_k + 2^11 sum_k b_k s_k,        s_k = 2 q_k - 1 in {-1, +1}
// With  C = 2^23 + (T - pop(q) + 2^j)(1 + 2^11)  the accumulator is the integer
//     2^23 + [T - dist(a) + 2^j] + 2^11 [T - dist(b) + 2^j],
// exact in f32 (every partial sum is an integer below 2^24), its significand bits ARE that integer,
// and because |T - dist| <= 65 < 2^7 <= 2^j bit j says dist(a) <= T and bit 11 + j says dist(b) <= T.
// Registers r, r+1, r+2, r+3 of a group use j = 7, 8, 9, 10, so  m = (acc & K_j) | m  -- ONE
// v_and_or_b32 per accumulator -- gathers eight hit bits of four registers in one mask; one more op
// joins two groups.  Per 32 rows a lane spends 18 VALU ops instead of 32; the MFMA count per row is
// unchanged (one K=64 instruction per 32-bit word and 32 rows), and halves for codes of <= 32 bits.
//
// Everything else is k_select_mx's: the same lane <-> (query, segment) mapping, windows of 128 rows per
// half staged through LDS by direct-to-LDS loads (image, packed codes, labels), the word-granular
// push / dense emit drain, the same records, slices and counts.
//
// Bit layout of a harvested 16-row mask (rows rho = 0..15 of the group, earliest row = highest bit):
//     rho 0..7  <-> bits 25..18   (field 1: register 8u + 7 - rho,  second row of the accumulator)
//     rho 8..15 <-> bits 14..7    (field 0: register 8u + 15 - rho, first row)
// A 32-row tile gives two of them (u = 0, 1: registers 0..7 and 8..15); seven more ops squeeze them into one
// 32-bit word with bit k <-> row 31 - k -- k_select_mx's convention, so push and emit are the same code.
#pragma once
#include "hg_select_mx.hpp"

namespace hg {

constexpr int M2_QT = 2;                 // query tiles (of 32) per wavefront
constexpr int M2_WT = 4;                 // 32-row tiles per window
constexpr int M2_WROWS = 32 * M2_WT;     // rows per lane-half per window (= MX_WROWS)

// row rho (0..31) of a 32-row group -> (register r, field f)
__host__ __device__ inline void m2_place(int rho, int& r, int& f) {
    const int u = rho >> 4, p = rho & 15;
    f = p < 8 ? 1 : 0;
    r = 8 * u + (p < 8 ? 7 - p : 15 - p);
}

// Database image: groups of 32 rows; chunk (group G, word w, field f, register r) = 16 bytes at
// (((G * NW + w) * 2 + f) * 16 + r) * 16 holding code word w of the row that m2_place() puts at (r, f).
static __global__ __launch_bounds__(256) void k_expand_db2(const u32* __restrict__ db, uint4* __restrict__ dbx, i64 N, i64 n32, int NW) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n32 * NW) return;
    const i64 row = i / NW;
    const int w = (int)(i - row * NW);
    const u32 x = row < N ? db[row * NW + w] : 0u;
    int r, f;
    m2_place((int)(row & 31), r, f);
    dbx[(((row >> 5) * NW + w) * 2 + f) * 16 + r] = expand_word(x, false);
}

// Query image: chunk (query tile qt, word w, lane = 32 kb + j) = word w of query 32 qt + j for BOTH k-halves,
// as s = 2 q - 1: bit 1 -> +1.0 (0x2), bit 0 -> -1.0 (0xA).
static __global__ __launch_bounds__(256) void k_expand_queries2(const u32* __restrict__ qc, uint4* __restrict__ qx, i64 Q, i64 qpad, int NW) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= qpad * NW) return;
    const i64 q = i / NW;
    const int w = (int)(i - q * NW);
    uint4 o = {0u, 0u, 0u, 0u};
    if (q < Q) {
        const uint4 e = expand_word(qc[q * NW + w], false);          // bit -> 0x2 / 0x0
        o.x = 0xAAAAAAAAu ^ (e.x << 2); o.y = 0xAAAAAAAAu ^ (e.y << 2);   // 0xA ^ 0x8 = 0x2 where the bit is set
        o.z = 0xAAAAAAAAu ^ (e.z << 2); o.w = 0xAAAAAAAAu ^ (e.w << 2);
    }
    const i64 base = ((q >> 5) * NW + w) * 64 + (q & 31);
    qx[base] = o;
    qx[base + 32] = o;
}

struct Mx2Lds { int a, codes, labels, stage, qcodes, qlabels, queue, rings, total; };
__host__ __device__ inline Mx2Lds mx2_lds_layout(int NW, int LW, bool compact) {
    Mx2Lds l;
    l.a = 0;
    l.codes = M2_WT * NW * 1024;
    l.labels = l.codes + 2 * M2_WROWS * NW * 4;
    l.stage = l.labels + 2 * M2_WROWS * LW * 8;
    l.stage = (l.stage + 1023) & ~1023;
    l.qcodes = 2 * l.stage;
    l.qlabels = l.qcodes + WPB * 32 * M2_QT * NW * 4;
    l.queue = l.qlabels + WPB * 32 * M2_QT * LW * 8;
    l.rings = l.queue + WPB * mx_qcap(M2_QT, compact) * 8;
    l.total = l.rings + WPB * mx_ring_bytes(M2_QT, compact);
    return l;
}

// Geo as set by the launcher: g.nQT = query blocks (of 256 queries) per segment pair, g.nBlk = blocks; g.L % 32 == 0.
template <int NW, int LW, bool COMPACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NW == 2 ? 3 : 4, NW == 2 ? 3 : 4)))
void k_select_mx2(const u32* __restrict__ qc, const u64* __restrict__ qlab, const u8* __restrict__ qx,
                  const u32* __restrict__ db, const u8* __restrict__ dbx, const u64* __restrict__ dblab,
                  const SelArgs a, u64* __restrict__ cand, const Geo g) {
    static_assert(NW <= 2, "two distances per accumulator need |T - dist| < 2^7");
    extern __shared__ __attribute__((aligned(1024))) u8 mxlds[];
    constexpr int QT = M2_QT, WQ = 32 * QT;
    constexpr int CB = NW * 4, LB = LW * 8;
    constexpr int LWA = LW > 0 ? LW : 1;
    const Mx2Lds L = mx2_lds_layout(NW, LW, COMPACT);

    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;                                   // whole block: no barrier is skipped by a part of it
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQB = g.nQT;
    const int sp = lb / nQB;                             // segment pair
    const int qb = lb - sp * nQB;                        // block of 256 queries
    const int h = lane >> 5, j = lane & 31;

    const int s = 2 * sp + h;                            // this lane's segment
    const bool seg_ok = s < g.S;
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 nwin = ((len0 > len1 ? len0 : len1) + M2_WROWS - 1) / M2_WROWS;
    const i64 NG = (g.N + 31) >> 5;                      // 32-row groups in the image

    // ---- query side ----
    const int q0w = (qb * WPB + wave) * WQ;               // first query of this wavefront
    {
        u32* qcl = (u32*)(mxlds + L.qcodes + wave * WQ * CB);
        for (int e = lane; e < WQ * NW; e += 64) {
            const i64 q = q0w + e / NW;
            qcl[e] = q < g.Q ? qc[q * NW + (e % NW)] : 0u;
        }
        if (LW > 0) {
            u64* qll = (u64*)(mxlds + L.qlabels + wave * WQ * LB);
            for (int e = lane; e < WQ * LW; e += 64) {
                const i64 q = q0w + e / LWA;
                qll[e] = q < g.Q ? qlab[q * LW + (e % LWA)] : 0ull;
            }
        }
    }
    i32x4 bq[QT][NW];
    f32x16 biasv[QT];
    MxDrain<NW, LW, QT, M2_WROWS, COMPACT> dr;                        // slice cursors, hit queue, record rings (hg_mx_drain.hpp)
    dr.init(mxlds, MxDrainLds{L.qcodes, L.qlabels, L.queue, L.rings, L.codes, L.labels}, wave, lane, qb, sp, a.cap, a.crow, a.probe,
            g.idx_base, g.L, cand);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        const bool live = q < g.Q && seg_ok;
        int pop = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) pop += __builtin_popcount(q < g.Q ? qc[(i64)q * NW + w] : 0u);
#pragma unroll
        for (int w = 0; w < NW; ++w) bq[t][w] = *(const i32x4*)(qx + (((i64)(q0w / 32 + t) * NW + w) * 64 + lane) * 16);
        const int T = live ? a.T[q] - (s > a.sstar[q] ? 1 : 0) : -1;
        const float base = (float)(T - pop);             // T - dist = base + sum_k x_k s_k;  dead lane: T = -1, never >= 0
#pragma unroll
        for (int r = 0; r < 16; ++r) biasv[t][r] = 8388608.0f + (base + (float)(128 << (r & 3))) * 2049.0f;
        dr.set_live(t, live);
    }

    // ---- window staging (k_select_mx's, with 32-row tiles) ----
    const int ah = (j >> 2) & 1;                                     // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                           // the accumulator register of that row
    const i64 ag0 = (ah ? lo1 : lo0) >> 5;                           // first 32-row group of that segment
    auto stage_window = [&](const i64 win, const int buf) {
        u8* st = mxlds + buf * L.stage;
        for (int c = wave; c < M2_WT * NW; c += WPB) {
            const int T = c / NW, w = c - T * NW;
            i64 G = ag0 + win * M2_WT + T;
            G = G < NG ? G : NG - 1;                                 // past the end: any valid group (masked later)
            const u8* src = dbx + ((((G * NW + w) * 2 + h) * 16 + ar) * 16);     // field = k-half = lane-half of the A operand
            HG_GLDS16(src, st + L.a + c * 1024);
        }
        constexpr int CPH = (M2_WROWS * CB + 1023) / 1024, LPH = LW > 0 ? (M2_WROWS * LB + 1023) / 1024 : 0;
        for (int c = wave; c < 2 * (CPH + LPH); c += WPB) {
            const int hh = c & 1, k = c >> 1;
            const bool is_lab = k >= CPH;
            const int piece = is_lab ? k - CPH : k;
            const int rowb = is_lab ? LB : CB;
            const i64 seg_lo = hh ? lo1 : lo0;
            const i64 off = (seg_lo + win * M2_WROWS) * rowb + piece * 1024 + lane * 16;
            const u8* tab = is_lab ? (const u8*)dblab : (const u8*)db;
            const u8* src = tab + (off < g.N * rowb ? off : 0);      // see k_select_mx
            u8* dst = st + (is_lab ? L.labels : L.codes) + hh * M2_WROWS * rowb + piece * 1024;
            if (piece * 1024 + lane * 16 < M2_WROWS * rowb) HG_GLDS16(src, dst);
        }
    };

    const int scale_b = 0x7F7F7F7F;                                  // E8M0 2^0
    const int scale_a = h ? 0x8A8A8A8A : 0x7F7F7F7F;                 // the second row of the accumulator rides at 2^11
    auto issue = [&](const i32x4 (&af)[NW], const int t) -> f32x16 {
        f32x16 acc = biasv[t];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const i32x8 A = {af[w].x, af[w].y, af[w].z, af[w].w, 0, 0, 0, 0};
            const i32x8 B = {bq[t][w].x, bq[t][w].y, bq[t][w].z, bq[t][w].w, 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, scale_a, 0, scale_b);
        }
        return acc;
    };
    auto load_a = [&](i32x4 (&af)[NW], const u8* st, const int T) {
#pragma unroll
        for (int w = 0; w < NW; ++w) af[w] = *(const i32x4*)(st + L.a + ((T * NW + w) * 64 + lane) * 16);
    };
    // 16 accumulators -> the two mask words of the tile: 16 v_and_or_b32 + 2 v_lshl_or_b32
    auto harvest = [&](const f32x16& acc, u32& word) {
        u32 m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const u32 K = (1u << (7 + (r & 3))) | (1u << (18 + (r & 3)));
            m[r >> 2] = (__float_as_uint(acc[r]) & K) | m[r >> 2];
        }
        const u32 w0 = (m[1] << 4) | m[0], w1 = (m[3] << 4) | m[2];   // bits 25..18 = rows 0..7, bits 14..7 = rows 8..15
        // compact each to 16 bits (bit k <-> row 15 - k) and join: bit k of the result <-> row 31 - k of the tile
        const u32 c0 = ((w0 >> 7) & 0xFFu) | ((w0 >> 10) & 0xFF00u), c1 = ((w1 >> 7) & 0xFFu) | ((w1 >> 10) & 0xFF00u);
        word = (c0 << 16) | c1;
        asm volatile("" : "+v"(word));                               // pin here (pure ops would sink to the drain)
    };
    if (nwin > 0) stage_window(0, 0);
    for (i64 win = 0; win < nwin; ++win) {
        const int buf = (int)(win & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (win + 1 < nwin) stage_window(win + 1, buf ^ 1);
        const u8* st = mxlds + buf * L.stage;

        u32 m[QT][M2_WT];
        i32x4 acur[NW], anext[NW];
        load_a(acur, st, 0);
#pragma unroll
        for (int T = 0; T < M2_WT; ++T) {
            load_a(anext, st, T + 1 < M2_WT ? T + 1 : T);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x16 acc = issue(acur, t);                   // one accumulator set, as in k_select_mx (b = 32: 0.877 -> 0.838 ms)
                harvest(acc, m[t][T]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int w = 0; w < NW; ++w) acur[w] = anext[w];
        }
        const i64 left = mylen - win * M2_WROWS;                     // valid rows of this lane in the window
        if (left < M2_WROWS) {
#pragma unroll
            for (int T = 0; T < M2_WT; ++T) {
                const i64 v = left - 32 * T;                         // valid rows among the 32 of tile T
                const u32 keep = v >= 32 ? 0xFFFFFFFFu : (v <= 0 ? 0u : ~(0xFFFFFFFFu >> (int)v));
#pragma unroll
                for (int t = 0; t < QT; ++t) m[t][T] &= keep;
            }
        }
        if (kProbes && (a.probe & 2)) {                                      // measurement probe: no drain
#pragma unroll
            for (int t = 0; t < QT; ++t) if (m[t][0] == 0x12345678u && m[t][3] == 0x1234567u) dr.flags |= 0x100u << t;
        } else {
            dr.drain_window(m, win, st);
        }
    }
    dr.finish();

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        if (seg_ok && q < g.Qpad) {
            const bool live = q < g.Q;
            a.sl_cnt[(i64)s * g.Qpad + q] = live ? dr.cnt[t] : 0u;
            if (dr.lost(t) && live) a.fail[q] = 1u;
        }
    }
}

}  // namespace hg
