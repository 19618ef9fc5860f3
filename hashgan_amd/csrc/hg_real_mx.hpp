// hashgan_amd -- the real-valued select pass on the matrix cores.
//
// lib/metric.py:13 is a float32 GEMM, and gfx950 has float32-in MFMA: v_mfma_f32_32x32x2_f32 runs at the vector fma
// rate but takes its operands from VGPRs / LDS instead of the scalar cache that capped k_real_select at half its
// floor.  Measured on this part (tools/mfma_f32_probe.hip): one instruction computes, per output element,
//     acc' = fma(a[k+1], b[k+1], fma(a[k], b[k], acc))            -- bitwise an fmaf chain, k ascending,
// so a tile's 32 x 32 inner products over KP features are KP/2 chained instructions and equal ONE float32 fma chain
// from +0.0 over k = 0 .. KP-1: the order oracle/real_map.py restates.
//
// Mapping (k_select_mx's): D = A B, A rows = database rows, B columns = queries.  A lane holds column j = lane & 31 of
// D -- one query -- and 16 of the tile's 32 rows; lane-half h = lane >> 5 gets its 16 rows from segment 2 sp + h, so
// every lane walks ITS segment in index order and its records land in the (segment, query) slice.  The hit test is
// the sign of thr - ip (one v_sub + one v_alignbit per pair, noise next to 2 KP cycles of MFMA per 1024 pairs), and a
// hit's score IS the accumulator: the owning lane picks it out of its 16 registers and writes the sortable record
// {~mono(ip) | idx} itself -- no queue, no recomputation.
#pragma once
#include "hg_kernels.hpp"
#include "hg_real_kernels.hpp"
#include "hg_select_mx.hpp"

namespace hg {

constexpr int RMX_QT = 2;                // query tiles (of 32) per wavefront

// Database image in A-fragment order: groups of 16 rows; chunk (group G, m4, parity h, row r) = 16 bytes at
// (((G * (KP/8) + m4) * 2 + h) * 16 + r) * 16 holding features 8 m4 + 2 u + h, u = 0..3, of row 16 G + r.
// (rstride > 1: image row i is table row i * rstride -- the sample pass's image of every rstride-th row)
static __global__ __launch_bounds__(256) void k_expand_dbf(const float* __restrict__ dbf, float4* __restrict__ img, i64 N, i64 n16, int KP, i64 rstride) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    const int per_row = KP / 4;                              // chunks per row: (KP / 8) m4 x 2 parities
    if (i >= n16 * per_row) return;
    const i64 row = i / per_row;
    const int c = (int)(i - row * per_row), m4 = c >> 1, h = c & 1;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < N) {
        const float* f = dbf + row * rstride * KP + 8 * m4 + h;
        v.x = f[0]; v.y = f[2]; v.z = f[4]; v.w = f[6];
    }
    img[(((row >> 4) * (KP / 8) + m4) * 2 + h) * 16 + (row & 15)] = v;
}

template <int KP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, KP <= 64 ? 3 : 2)))
void k_real_select_mx(const float* __restrict__ qf, const u8* __restrict__ img, const RealSelArgs a,
                      u64* __restrict__ cand, const Geo g) {
    constexpr int QT = RMX_QT, WQ = 32 * QT;
    constexpr int NM4 = KP / 8;                              // 16-byte A chunks per lane and tile (4 MFMAs each)
    __shared__ uint4 stg[WPB][8 * 64];                       // per wavefront: the tile's records on their way to whole-line stores (no cut)

    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQB = g.nQT;
    const int sp = lb / nQB, qb = lb - sp * nQB;
    const int h = lane >> 5, j = lane & 31;
    const int s = 2 * sp + h;                                // this lane's segment
    const bool seg_ok = s < g.S;
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 ntile = ((len0 > len1 ? len0 : len1) + 15) / 16;
    const i64 NG = (g.N + 15) >> 4;

    // ---- queries: B fragments (feature 2 m + h of query j), thresholds, slice cursors ----
    const int q0w = (qb * WPB + wave) * WQ;
    float bq[QT][KP / 2];
    float thr[QT];
    u32 cnt[QT], room[QT], dropped[QT];
    bool alive[QT];                                          // the lane has a query and a segment
    u64* wp[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        const bool live = q < g.Q && seg_ok;
#pragma unroll
        for (int m = 0; m < KP / 2; ++m) bq[t][m] = q < g.Q ? qf[(i64)q * KP + 2 * m + h] : 0.0f;
        thr[t] = live ? a.thr[q] : __uint_as_float(0x7F800000u);      // +inf: nothing qualifies
        cnt[t] = 0; room[t] = live ? a.cap : 0u; dropped[t] = 0; alive[t] = live;
        wp[t] = cand + (i64)(q < g.Q ? q : 0) * a.crow + (i64)(seg_ok ? s : 0) * a.cap;
    }

    // ---- A fragments straight from the image into registers, one tile ahead: chunk m4 of the NEXT tile is requested
    // as soon as the last query tile has consumed chunk m4 of this one.  The four wavefronts of a block read the same
    // 8 KiB per tile (L2 hits after the first); nothing is shared through LDS, so nothing waits at a barrier.
    const int ah = (j >> 2) & 1;                             // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                   // its row inside that half's 16
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;
    auto chunk = [&](const i64 T, const int m4) -> float4 {
        i64 G = ag0 + T;
        G = G < NG ? G : NG - 1;                             // past the end: any valid group (masked later)
        return *(const float4*)(img + ((((G * NM4 + m4) * 2 + h) * 16 + ar) * 16));
    };
    float4 av[NM4];
    if (ntile > 0) {
#pragma unroll
        for (int m4 = 0; m4 < NM4; ++m4) av[m4] = chunk(0, m4);
    }
    for (i64 T = 0; T < ntile; ++T) {
        const i64 left = mylen - T * 16;                     // valid rows of this lane in the tile
        const u32 keep = left >= 16 ? 0xFFFFu : (left <= 0 ? 0u : (1u << (int)left) - 1u);
        const i64 Tn = T + 1 < ntile ? T + 1 : T;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int m4 = 0; m4 < NM4; ++m4) {
                const float4 x = av[m4];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, bq[t][4 * m4 + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, bq[t][4 * m4 + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, bq[t][4 * m4 + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, bq[t][4 * m4 + 3], acc, 0, 0, 0);
                if (t == QT - 1) av[m4] = chunk(Tn, m4);
            }
            // harvest: bit r <-> row 16 T + r of the lane's segment qualifies (ip > thr)
            u32 mask = 0;
#pragma unroll
            for (int r = 15; r >= 0; --r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(thr[t] - acc[r]), 31);
            mask &= keep;
            // Every row of the tile a record (no cut: R = N, or the exhaustive mode): a lane's sixteen records are 128 contiguous
            // bytes of its slice.  The wavefront turns them through LDS so that one store instruction writes eight whole 128-byte
            // runs (eight lanes per query) instead of 64 fragments of 16 bytes, one per query row: the L2 takes whole lines.
            // (The drain below -- sixteen rounds of row pick and 8-byte store -- was 3/4 of this kernel's instructions at the
            // CIFAR shape: 0.34 ms; the lane's own eight 16-byte stores: 0.20 ms.)
            const bool full = mask == 0xFFFFu && room[t] >= 16u && cnt[t] == (u32)(T * 16);
            const u32 idx0 = g.idx_base + (u32)((i64)s * g.L + T * 16);
            if (__all(full || !alive[t]) && __any(full)) {
                const u64 fm = __ballot(full);
                uint4* __restrict__ st = stg[wave];
#pragma unroll
                for (int c2 = 0; c2 < 8; ++c2) {
                    uint4 v;
                    v.x = idx0 + (u32)(2 * c2);      v.y = ~mono_key(acc[2 * c2] + 0.0f);
                    v.z = idx0 + (u32)(2 * c2) + 1u; v.w = ~mono_key(acc[2 * c2 + 1] + 0.0f);
                    st[c2 * 64 + (lane ^ c2)] = v;
                }
                wave_lds_sync();
                const int c2 = lane & 7;
#pragma unroll 1
                for (int i = 0; i < 8; ++i) {                // (not unrolled: eight 16-byte reads in flight cost 32 registers and the third wavefront per SIMD)
                    const int src = 8 * i + (lane >> 3);
                    const uint4 v = st[c2 * 64 + (src ^ c2)];
                    const int sq = q0w + t * 32 + (src & 31);
                    if ((fm >> src) & 1ull) {
                        u64* dst = cand + (i64)sq * a.crow + (i64)(2 * sp + (src >> 5)) * a.cap + T * 16;
                        ((uint4*)dst)[c2] = v;
                    }
                }
                wave_lds_sync();
                if (full) { cnt[t] += 16u; room[t] -= 16u; mask = 0u; }
            } else if (full) {
                uint4* __restrict__ w = (uint4*)(wp[t] + cnt[t]);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    uint4 v;
                    v.x = idx0 + (u32)r;      v.y = ~mono_key(acc[r] + 0.0f);
                    v.z = idx0 + (u32)r + 1u; v.w = ~mono_key(acc[r + 1] + 0.0f);
                    w[r >> 1] = v;
                }
                cnt[t] += 16u; room[t] -= 16u; mask = 0u;
            }
            // drain: the owning lane writes its hits itself, lowest row first (rounds = the busiest lane's hits, ~1-2)
            while (__any(mask != 0u)) {
                if (mask != 0u) {
                    const int r = __builtin_ctz(mask);
                    mask &= mask - 1u;
                    // acc[r] by a binary tree of selects on the bits of r (15 v_cndmask)
                    float v8[8], v4[4], v2[2];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v8[k] = (r & 8) ? acc[8 + k] : acc[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v4[k] = (r & 4) ? v8[4 + k] : v8[k];
#pragma unroll
                    for (int k = 0; k < 2; ++k) v2[k] = (r & 2) ? v4[2 + k] : v4[k];
                    const float ip = (r & 1) ? v2[1] : v2[0];
                    if (room[t]) {
                        const u32 idx = g.idx_base + (u32)((i64)s * g.L + T * 16 + r);
                        wp[t][cnt[t]] = ((u64)(~mono_key(ip + 0.0f)) << 32) | (u64)idx;
                        ++cnt[t];
                        --room[t];
                    } else {
                        ++dropped[t];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        if (seg_ok && q < g.Qpad) {
            const bool live = q < g.Q;
            a.sl_cnt[(i64)s * g.Qpad + q] = live ? cnt[t] : 0u;
            if (dropped[t] && live) a.fail[q] = 1u;
        }
    }
}

// Sample pass on the float32 MFMA (replaces k_real_sample up to 128 features): the inner products of every query with
// the M sampled rows, the same fma chains bit for bit, so the guessed cut does not change.  The sampled rows' image is
// cut into segments like a database (k_real_select_mx's mapping); a lane ends a tile with 16 consecutive samples of
// its query in its accumulator and stores them as they are: samp[q][mstride], 64 contiguous bytes per lane and tile.
template <int KP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, KP <= 64 ? 3 : 2)))
void k_real_sample_mx(const float* __restrict__ qf, const u8* __restrict__ img, float* __restrict__ samp, i64 mstride, const Geo g) {
    constexpr int QT = RMX_QT, WQ = 32 * QT;
    constexpr int NM4 = KP / 8;
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQB = g.nQT;
    const int sp = lb / nQB, qb = lb - sp * nQB;
    const int h = lane >> 5, j = lane & 31;
    const int s = 2 * sp + h;
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = lo0 >= g.N ? 0 : (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 ntile = ((len0 > len1 ? len0 : len1) + 15) / 16;
    const i64 NG = (g.N + 15) >> 4;
    const int q0w = (qb * WPB + wave) * WQ;
    float bq[QT][KP / 2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
#pragma unroll
        for (int m = 0; m < KP / 2; ++m) bq[t][m] = q < g.Q ? qf[(i64)q * KP + 2 * m + h] : 0.0f;
    }
    const int ah = (j >> 2) & 1;
    const int ar = (j & 3) + 4 * (j >> 3);
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;
    auto chunk = [&](const i64 T, const int m4) -> float4 {
        i64 G = ag0 + T;
        G = G < NG ? G : NG - 1;
        return *(const float4*)(img + ((((G * NM4 + m4) * 2 + h) * 16 + ar) * 16));
    };
    float4 av[NM4];
    if (ntile > 0) {
#pragma unroll
        for (int m4 = 0; m4 < NM4; ++m4) av[m4] = chunk(0, m4);
    }
    for (i64 T = 0; T < ntile; ++T) {
        const i64 left = mylen - T * 16;
        const i64 Tn = T + 1 < ntile ? T + 1 : T;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int m4 = 0; m4 < NM4; ++m4) {
                const float4 x = av[m4];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, bq[t][4 * m4 + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, bq[t][4 * m4 + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, bq[t][4 * m4 + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, bq[t][4 * m4 + 3], acc, 0, 0, 0);
                if (t == QT - 1) av[m4] = chunk(Tn, m4);
            }
            const int q = q0w + t * 32 + j;
            if (q < g.Q && left > 0) {
                float* out = samp + (i64)q * mstride + (i64)s * g.L + T * 16;       // 64-byte aligned: L and mstride are multiples of 16
                if (left >= 16) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        ((float4*)out)[r4] = float4{acc[4 * r4] + 0.0f, acc[4 * r4 + 1] + 0.0f, acc[4 * r4 + 2] + 0.0f, acc[4 * r4 + 3] + 0.0f};
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (r < left) out[r] = acc[r] + 0.0f;
                }
            }
        }
    }
}

}  // namespace hg
