// hashgan_amd -- optimistic select on the matrix cores (gfx950).
//
// The reference computes the query x database similarity as a dot product of +-1 codes
// (/root/reference/lib/metric.py:13, np.dot(query, database.T)); k_select restates it as
// xor + popcount on the vector ALU, 2*NW + 1 integer ops per (query, row) pair, which is what
// bounds the whole evaluation.  Here the dot product goes back to where dot products are cheap:
//     dist(q, x) = popcount(q) + sum_k x_k * (1 - 2 q_k),      x_k in {0, 1},  (1 - 2 q_k) in {+1, -1}
// is an inner product over the code bits whose factors fit the 4-bit float format (E2M1 holds
// 0, +1, -1 exactly) and whose f32 accumulation is exact (|sum| <= 256).  One
// v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4, block scales = 1) per 64 code bits produces a
// 32-row x 32-query tile of distances; the per-query constant popcount(q) - T - 1 rides in as
// the C operand, so the SIGN of every accumulator IS the test dist <= T and the vector ALU is
// left with ONE op per pair (v_alignbit: shift the lane's hit mask, append the sign) instead of
// 2*NW + 1.  What leaves the kernel -- the 8-byte records in index order, the (segment, query)
// slices, their counts and overflow flags -- is exactly k_select<OPT>'s, so the ranking stage
// cannot tell which kernel ran.
//
// Mapping.  D[i][j] = sum_k A[i][k] B[k][j]: A rows = database rows, B columns = queries.  A
// lane holds column j = lane & 31 of D, i.e. ONE query, and 16 of the tile's 32 rows:
// register r of lane-half h = lane >> 5 is row (r & 3) + 8 (r >> 2) + 4 h.  The kernel feeds
// the A rows of half h from segment 2 sp + h, 16 consecutive rows per tile, in the order that
// makes register r <-> the segment's row 16 tile + r: each lane walks ITS segment in index
// order, exactly like a k_select lane, and its records land in the (segment, query) slice.
// The K order of an inner product is free as long as A and B agree: both images are produced
// by the same expand_word() below.
//
// Data movement.  A block = 4 wavefronts = one segment pair x 128 QT queries (QT tiles of 32
// per wave; QT = 2).  The database streams through LDS in windows of 8 row tiles (128 rows per half; 4 tiles for
// codes longer than 64 bits -- mx_wt() below): the
// fp4 image of the rows (A fragments, lane-linear) plus their packed codes and labels (for the
// drain's exact distance and match bit) are copied global -> LDS by direct-to-LDS loads, one
// window ahead, shared by the four waves; one barrier per window.  Hits leave through a
// two-phase drain (push / emit, below) twice per window.
#pragma once
#include "hg_kernels.hpp"
#include "hg_mx_drain.hpp"

namespace hg {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MX_WT = 8;                // row tiles per window
constexpr int MX_WROWS = 16 * MX_WT;    // rows per lane-half per window
// Window length and query tiles by code length.  What decides is how many blocks a CU holds (160 KB of LDS): with 8-tile
// windows a block of 65..128-bit codes (two MFMAs per tile, twice the image) needs 62 KB -- two per CU, 2 wavefronts per
// SIMD -- and four query tiles of 129..256-bit codes 124 KB, one per CU.  Windows of 4 tiles bring the former back to
// 40 KB (four per CU: C5 1.59 -> 1.08 ms), and two query tiles + 4-tile windows the latter to 64 KB (two per CU, the
// 256 registers of 2 wavefronts per SIMD: b = 255 4.72 -> 1.87 ms).  A 4-tile window is one half-window drain.
#ifndef HG_MX_WT_LONG
#define HG_MX_WT_LONG 4
#endif
#ifndef HG_MX_WT_XL
#define HG_MX_WT_XL 4
#endif
#ifndef HG_MX_QT_XL
#define HG_MX_QT_XL 2
#endif
__host__ __device__ constexpr int mx_wt(int NW) { return NW <= 2 ? MX_WT : NW <= 4 ? HG_MX_WT_LONG : HG_MX_WT_XL; }
__host__ __device__ constexpr int mx_qt(int NW) { return NW <= 4 ? 2 : HG_MX_QT_XL; }

// 8 code bits -> 8 nibbles, bit j at bit 4 j
__device__ __forceinline__ u32 spread8(u32 y) {
    y = (y | (y << 12)) & 0x000F000Fu;
    y = (y | (y << 6)) & 0x03030303u;
    y = (y | (y << 3)) & 0x11111111u;
    return y;
}
// one 32-bit code word -> 32 fp4 values (16 bytes).  Database side: bit -> 0.0 / 1.0 (0x0 / 0x2);
// query side: bit -> +1.0 / -1.0 (0x2 / 0xA).
__device__ __forceinline__ uint4 expand_word(u32 x, bool query_side) {
    uint4 o;
    const u32 s0 = spread8(x & 255u), s1 = spread8((x >> 8) & 255u), s2 = spread8((x >> 16) & 255u), s3 = spread8(x >> 24);
    if (query_side) {
        o.x = 0x22222222u | (s0 << 3); o.y = 0x22222222u | (s1 << 3); o.z = 0x22222222u | (s2 << 3); o.w = 0x22222222u | (s3 << 3);
    } else {
        o.x = s0 << 1; o.y = s1 << 1; o.z = s2 << 1; o.w = s3 << 1;
    }
    return o;
}

// Database image, in A-fragment order: rows in groups of 16; chunk (group G, mfma m, k-half kb, row ar)
// = 16 bytes at (((G * NM + m) * 2 + kb) * 16 + ar) * 16 holding code word 2 m + kb of row 16 G + ar.
static __global__ __launch_bounds__(256) void k_expand_db(const u32* __restrict__ db, uint4* __restrict__ dbx, i64 N, i64 n16, int NW, int NM) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    const int wpr = 2 * NM;
    if (i >= n16 * wpr) return;
    const i64 r = i / wpr;
    const int wd = (int)(i - r * wpr);
    const u32 x = (r < N && wd < NW) ? db[r * NW + wd] : 0u;
    const i64 G = r >> 4;
    const int ar = (int)(r & 15), m = wd >> 1, kb = wd & 1;
    dbx[((G * NM + m) * 2 + kb) * 16 + ar] = expand_word(x, false);
}

// Query image, in B-fragment order: chunk (query tile qt, mfma m, lane = 32 kb + j) at ((qt * NM + m) * 64 + lane) * 16
static __global__ __launch_bounds__(256) void k_expand_queries(const u32* __restrict__ qc, uint4* __restrict__ qx, i64 Q, i64 qpad, int NW, int NM) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    const int wpr = 2 * NM;
    if (i >= qpad * wpr) return;
    const i64 q = i / wpr;
    const int wd = (int)(i - q * wpr);
    const bool ok = q < Q && wd < NW;
    const uint4 z = {0u, 0u, 0u, 0u};
    const i64 qt = q >> 5;
    const int j = (int)(q & 31), m = wd >> 1, kb = wd & 1;
    qx[(qt * NM + m) * 64 + kb * 32 + j] = ok ? expand_word(qc[q * NW + wd], true) : z;
}

struct MxLds {                 // byte offsets inside the block's dynamic LDS
    int a, codes, labels;      // inside one stage
    int stage;                 // stage size
    int qcodes, qlabels;       // query tables (after the two stages)
    int queue;                 // per-wave hit queues (mx_qcap entries of 8 bytes each)
    int rings;                 // per-wave slice rings (compact records only)
    int total;
};
__host__ __device__ inline MxLds mx_lds_layout(int NW, int LW, int QT, bool compact) {
    const int QBLK = WPB * 32 * QT;            // queries per block
    const int NM = (NW + 1) / 2;
    const int MX_WT = mx_wt(NW), MX_WROWS = 16 * MX_WT;
    MxLds l;
    l.a = 0;
    l.codes = MX_WT * NM * 1024;
    l.labels = l.codes + 2 * MX_WROWS * NW * 4;
    l.stage = l.labels + 2 * MX_WROWS * LW * 8;
    l.stage = (l.stage + 1023) & ~1023;
    l.qcodes = 2 * l.stage;
    l.qlabels = l.qcodes + QBLK * NW * 4;
    l.queue = l.qlabels + QBLK * LW * 8;
    l.rings = l.queue + WPB * mx_qcap(QT, compact) * 8;
    l.total = l.rings + WPB * mx_ring_bytes(QT, compact);
    return l;
}

#define HG_GLDS16(src, dst)                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),          \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// Geo as set by the launcher: g.nQT = query blocks (of 128 QT queries) per segment pair, g.nBlk = blocks.
// QT: query tiles (of 32) per wavefront -- 2: 256 queries per block (what the launcher picks, mx_qt(); codes of up to 128
//     bits run 4 wavefronts per SIMD in 128 registers, longer ones 2 per SIMD); 4: 512 queries per block (HG_MX_QT_XL = 4).
// COMPACT: one-byte records through per-slice LDS rings (AP only -- hg_mx_drain.hpp); else 8-byte records with the index.
template <int NW, int LW, int QT, bool COMPACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NW > 4 ? 2 : 4, NW > 4 ? 2 : 4)))
void k_select_mx(const u32* __restrict__ qc, const u64* __restrict__ qlab, const u8* __restrict__ qx,
                 const u32* __restrict__ db, const u8* __restrict__ dbx, const u64* __restrict__ dblab,
                 const SelArgs a, u64* __restrict__ cand, const Geo g) {
    extern __shared__ __attribute__((aligned(1024))) u8 mxlds[];
    constexpr int WQ = 32 * QT;                          // queries per wavefront
    constexpr int NM = (NW + 1) / 2;
    constexpr int CB = NW * 4, LB = LW * 8;
    constexpr int LWA = LW > 0 ? LW : 1;
    constexpr int MX_WT = mx_wt(NW), MX_WROWS = 16 * MX_WT, NWORD = MX_WT / 2;     // (shadow the defaults: this code length's window)
    const MxLds L = mx_lds_layout(NW, LW, QT, COMPACT);

    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;                                   // whole block: no barrier is skipped by a part of it
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQB = g.nQT;
    const int sp = lb / nQB;                             // segment pair
    const int qb = lb - sp * nQB;                        // block of 128 QT queries
    const int h = lane >> 5, j = lane & 31;

    // this lane's segment (lane-half h walks segment 2 sp + h)
    const int s = 2 * sp + h;
    const bool seg_ok = s < g.S;
    // wave-uniform row counts of the two segments
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 ntile = ((len0 > len1 ? len0 : len1) + 15) / 16;
    const i64 nwin = (ntile + MX_WT - 1) / MX_WT;
    const i64 NG = (g.N + 15) >> 4;                                  // row groups in the image

    // ---- query side: LDS tables for the drain, B fragments, C = bias, slice cursors ----
    const int q0w = (qb * WPB + wave) * WQ;                      // first query of this wavefront
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
    i32x4 bq[QT][NM];
    f32x16 biasv[QT];
    MxDrain<NW, LW, QT, MX_WROWS, COMPACT> dr;                        // slice cursors, hit queue, record rings
    dr.init(mxlds, MxDrainLds{L.qcodes, L.qlabels, L.queue, L.rings, L.codes, L.labels}, wave, lane, qb, sp, a.cap, a.crow, a.probe,
            g.idx_base, g.L, cand);
    bool far[QT];                                                    // compact records hold 7-bit distances: a cut beyond 127 loses the bet
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        bool live = q < g.Q && seg_ok;
        int pop = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) pop += __builtin_popcount(q < g.Q ? qc[(i64)q * NW + w] : 0u);
#pragma unroll
        for (int m = 0; m < NM; ++m) bq[t][m] = *(const i32x4*)(qx + (((i64)(q0w / 32 + t) * NM + m) * 64 + lane) * 16);
        // past the query's last tie-collecting segment only rows strictly closer than the guess are taken
        int T = live ? a.T[q] - (s > a.sstar[q] ? 1 : 0) : -1;
        far[t] = COMPACT && NW >= 4 && T > 127;
        if (far[t]) { live = false; T = -1; }
        const float bias = (float)(pop - T - 1);         // dist + (-T - 1) < 0  <=>  dist <= T;  dead lane: never
#pragma unroll
        for (int r = 0; r < 16; ++r) biasv[t][r] = bias;
        dr.set_live(t, live);
    }

    // ---- window staging: global -> LDS, the four waves split the copy instructions ----
    const int ah = (j >> 2) & 1;                                     // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                           // its row inside that half's 16
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;                           // first row group of that segment
    auto stage_window = [&](const i64 win, const int buf) {
        u8* st = mxlds + buf * L.stage;
        // A fragments: MX_WT * NM chunks of 1 KiB, lane-linear in LDS
        for (int c = wave; c < MX_WT * NM; c += WPB) {
            const int T = c / NM, m = c - T * NM;
            i64 G = ag0 + win * MX_WT + T;
            G = G < NG ? G : NG - 1;                                 // past the end: any valid group (masked later)
            const u8* src = dbx + ((((G * NM + m) * 2 + h) * 16 + ar) * 16);
            HG_GLDS16(src, st + L.a + c * 1024);
        }
        // packed codes and labels of the window's rows, both halves: plain copies in 1 KiB pieces
        constexpr int CPH = (MX_WROWS * CB + 1023) / 1024, LPH = LW > 0 ? (MX_WROWS * LB + 1023) / 1024 : 0;
        for (int c = wave; c < 2 * (CPH + LPH); c += WPB) {
            const int hh = c & 1, k = c >> 1;
            const bool is_lab = k >= CPH;
            const int piece = is_lab ? k - CPH : k;
            const int rowb = is_lab ? LB : CB;
            const i64 seg_lo = hh ? lo1 : lo0;
            const i64 off = (seg_lo + win * MX_WROWS) * rowb + piece * 1024 + lane * 16;
            const u8* tab = is_lab ? (const u8*)dblab : (const u8*)db;
            const u8* src = tab + (off < g.N * rowb ? off : 0);      // rows past the table: anything (masked); the last
                                                                      // chunk may overhang the table by < 16 B (allocation slack)
            u8* dst = st + (is_lab ? L.labels : L.codes) + hh * MX_WROWS * rowb + piece * 1024;
            if (piece * 1024 + lane * 16 < MX_WROWS * rowb) HG_GLDS16(src, dst);
        }
    };

    const int scale1 = 0x7F7F7F7F;                                   // E8M0 block scales: 2^0
    auto issue = [&](const i32x4 (&af)[NM], const int t) -> f32x16 {
        f32x16 acc = biasv[t];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const i32x8 A = {af[m].x, af[m].y, af[m].z, af[m].w, 0, 0, 0, 0};
            const i32x8 B = {bq[t][m].x, bq[t][m].y, bq[t][m].z, bq[t][m].w, 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, scale1, 0, scale1);
        }
        return acc;
    };
    auto load_a = [&](i32x4 (&af)[NM], const u8* st, const int T) {
#pragma unroll
        for (int m = 0; m < NM; ++m) af[m] = *(const i32x4*)(st + L.a + ((T * NM + m) * 64 + lane) * 16);
    };

    if (nwin > 0) stage_window(0, 0);
    for (i64 win = 0; win < nwin; ++win) {
        const int buf = (int)(win & 1);
        // my copies of this window have landed (vmcnt), everybody's have and nobody still reads the other buffer (barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (win + 1 < nwin) stage_window(win + 1, buf ^ 1);
        const u8* st = mxlds + buf * L.stage;

        u32 m[QT][NWORD];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int w = 0; w < NWORD; ++w) m[t][w] = 0;
        // (The MFMA of step i + 1 used to be issued before the harvest of step i, a second accumulator set in flight: with
        // four wavefronts per SIMD the other waves fill the MFMA's latency anyway, and the 16 registers it held are worth
        // more to the drain -- 60 -> 32 B of scratch per lane, select 0.906 -> 0.894 ms at C2.)
        i32x4 acur[NM], anext[NM];
        load_a(acur, st, 0);
#pragma unroll
        for (int k = 0; k < MX_WT; ++k) {
            load_a(anext, st, k + 1 < MX_WT ? k + 1 : k);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const f32x16 acc = issue(acur, t);
                u32 mm = m[t][k >> 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(acc[r]), 31);
                asm volatile("" : "+v"(mm));                         // pin the chain here: pure ops would otherwise sink to the drain
                m[t][k >> 1] = mm;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int w = 0; w < NM; ++w) acur[w] = anext[w];
        }
        // rows past the end of the lane's segment (ragged last window, unpaired last segment) never count
        const i64 left = mylen - win * MX_WROWS;                     // valid rows of this lane in the window
        if (left < MX_WROWS) {
#pragma unroll
            for (int c = 0; c < NWORD; ++c) {
                const i64 v = left - 32 * c;                         // valid rows among the 32 of mask word c
                const u32 keep = v >= 32 ? 0xFFFFFFFFu : (v <= 0 ? 0u : ~(0xFFFFFFFFu >> (int)v));
#pragma unroll
                for (int t = 0; t < QT; ++t) m[t][c] &= keep;
            }
        }
        if (kProbes && (a.probe & 2)) {                                      // measurement probe: no drain
#pragma unroll
            for (int t = 0; t < QT; ++t) if (m[t][0] == 0x12345678u && m[t][NWORD - 1] == 0x1234567u) dr.flags |= 0x100u << t;
        } else {
            if constexpr (NWORD == 4) dr.drain_window(m, win, st);
            else dr.drain(m, 0, win, st);                            // a 64-row window is one half-window drain
        }
    }
    dr.finish();

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        if (seg_ok && q < g.Qpad) {
            const bool live = q < g.Q;
            a.sl_cnt[(i64)s * g.Qpad + q] = live ? dr.cnt[t] : 0u;
            if ((dr.lost(t) || far[t]) && live) a.fail[q] = 1u;
        }
    }
}

}  // namespace hg
