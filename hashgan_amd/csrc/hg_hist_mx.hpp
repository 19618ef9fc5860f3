// hashgan_amd -- the distance histogram (metric.py:13 reduced to counts) on the matrix cores.
//
// k_hist spends 2 NW + 1 vector ops per pair on xor + popcount before its LDS atomic.  The distance is an inner
// product (hg_select_mx.hpp): with C = popcount(q) an fp4 MFMA tile delivers 32 x 32 EXACT distances as floats, and a
// pair is left with a v_cvt_u32_f32, one address op and the same fire-and-forget ds_add_u32.  The LDS atomic rate
// (12-14 lanes per clock and CU) bounds both kernels; this one reaches it.
//
// Mapping: k_select_mx's -- block = segment pair x 256 queries, lane = (query j, lane-half h <-> segment 2 sp + h), 16
// rows per half and tile.  Both halves of a lane pair (j, j + 32) serve the SAME query, so they add into one column:
// the histograms come out per segment PAIR, [ceil(S/2)][NB][Qpad] -- the granularity the sampled pass has always used
// (its segments are two select segments long).  The two query tiles of a lane share a dword (16-bit halves) while a
// segment has fewer than 65536 rows.  A fragments come straight from the L2-resident image, one tile ahead; a sampling
// pass visits every stride-th tile of 16 rows.
#pragma once
#include "hg_kernels.hpp"
#include "hg_select_mx.hpp"

namespace hg {

// rows a sampling pass with tile stride `stride` visits (mirrors the kernel's loop)
inline i64 hist_mx_sampled_rows(const Geo& g, int stride) {
    i64 total = 0;
    for (int s = 0; s < g.S; ++s) {
        const i64 lo = (i64)s * g.L, len = lo + g.L < g.N ? g.L : g.N - lo;
        for (i64 T = 0; T * 16 < len; T += stride) total += len - T * 16 < 16 ? len - T * 16 : 16;
    }
    return total;
}

template <int NW, bool PACK16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_hist_mx(const u32* __restrict__ qc, const u8* __restrict__ qx, const u8* __restrict__ dbx, u32* __restrict__ hist, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 hlds[];
    constexpr int QT = 2, WQ = 32 * QT;
    constexpr int NM = (NW + 1) / 2;
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQB = g.nQT;
    const int sp = lb / nQB, qb = lb - sp * nQB;
    const int h = lane >> 5, j = lane & 31;
    const int NB = g.NB;
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 ntile = ((len0 > len1 ? len0 : len1) + 15) / 16;
    const i64 NG = (g.N + 15) >> 4;
    const int stride = g.hist_stride;

    // this wavefront's columns: [NB][32] dwords (PACK16: tile t in bits 16 t ..) or [QT][NB][32]
    u32* col = hlds + wave * (PACK16 ? 1 : QT) * NB * 32;
    for (int i = lane; i < (PACK16 ? 1 : QT) * NB * 32; i += 64) col[i] = 0u;

    const int q0w = (qb * WPB + wave) * WQ;
    i32x4 bq[QT][NM];
    f32x16 biasv[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        int pop = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) pop += __builtin_popcount(q < g.Q ? qc[(i64)q * NW + w] : 0u);
#pragma unroll
        for (int m = 0; m < NM; ++m) bq[t][m] = *(const i32x4*)(qx + (((i64)(q0w / 32 + t) * NM + m) * 64 + lane) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) biasv[t][r] = (float)pop;      // dist = pop(q) + sum_k x_k (1 - 2 q_k)
    }
    wave_lds_sync();

    const int ah = (j >> 2) & 1;                                     // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                           // its row inside that half's 16
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;
    auto chunk = [&](const i64 T, const int m) -> i32x4 {
        i64 G = ag0 + T;
        G = G < NG ? G : NG - 1;                                     // past the end: any valid group (masked below)
        return *(const i32x4*)(dbx + ((((G * NM + m) * 2 + h) * 16 + ar) * 16));
    };
    const int scale1 = 0x7F7F7F7F;
    i32x4 av[NM];
    if (ntile > 0) {
#pragma unroll
        for (int m = 0; m < NM; ++m) av[m] = chunk(0, m);
    }
    for (i64 T = 0; T < ntile; T += stride) {
        i32x4 an[NM];
        const i64 Tn = T + stride < ntile ? T + stride : T;
#pragma unroll
        for (int m = 0; m < NM; ++m) an[m] = chunk(Tn, m);
        const i64 left = mylen - T * 16;                             // valid rows of this lane in the tile
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            f32x16 acc = biasv[t];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const i32x8 A = {av[m].x, av[m].y, av[m].z, av[m].w, 0, 0, 0, 0};
                const i32x8 B = {bq[t][m].x, bq[t][m].y, bq[t][m].z, bq[t][m].w, 0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, scale1, 0, scale1);
            }
            u32* c0 = col + (PACK16 ? 0 : t * NB * 32) + j;
            const u32 inc = PACK16 ? (t ? 65536u : 1u) : 1u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const u32 d = (u32)acc[r];                           // exact: 0 .. b
                if ((i64)r < left) atomicAdd(c0 + d * 32, inc);
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) av[m] = an[m];
    }
    wave_lds_sync();
    // out: hist[sp][d][q], lane i -> query q0w + i (tile i >> 5, column i & 31)
    const int q = q0w + lane;
    if (q < g.Qpad) {
        u32* __restrict__ out = hist + (i64)sp * NB * g.Qpad + q;
        const int dn = g.hcap > 0 && g.hcap < NB ? g.hcap : NB;      // (the bet's sampled pass: the guess never reads beyond)
        for (int d = 0; d < dn; ++d) {
            const u32 v = PACK16 ? (col[d * 32 + j] >> (16 * h)) & 0xFFFFu : col[(h * NB + d) * 32 + j];
            out[(i64)d * g.Qpad] = v;
        }
    }
}

}  // namespace hg
