// hashgan_amd -- the real-valued select pass as FILTER + RESCORE.
//
// lib/metric.py:13 is a float32 GEMM whose every element the ranking needs in ONE fixed arithmetic (the float32 fma
// chain, k ascending: oracle/real_map.py) -- but only for the ~R rows per query that can reach the top R.  The float32
// MFMA computes that chain for all Q x N pairs at the vector fma rate (k_real_select_mx: 12 ms at the C2 shape, 78 % of
// what the instruction allows).  Here the pair pass runs in bfloat16 on the matrix cores (16x the float32 rate) and
// only decides, with a rigorous error bound, which pairs CAN qualify; the survivors -- ~1.3 R per query -- get the exact
// chain from the float32 rows:
//
//   filter    approx = sum_k bf16(q_k) bf16(x_k), float32 accumulation.  With u = 2^-8 covering either rounding of the
//             conversions,  |approx - chain| <= sum_k |q_k x_k| (2u + u^2 + (K + 2) 2^-21)  <=  |q|_2 max_rows |x|_2 c
//             (Cauchy-Schwarz; the K 2^-21 term covers the float32 roundings of both accumulations, any order).
//             A pair is kept when approx > thr2[q] = thr[q] - eps[q] (k_real_thr2): a superset of {chain > thr[q]}.
//             The test is the sign of the accumulator itself: C = thr2, B = -bf16(q)  ->  acc = thr2 - approx.
//   rescore   k_real_rescore: every kept (query, row) gets the exact chain from the float32 tables; those with
//             chain > thr[q] become the same sortable record {~mono(ip) | idx} the exact kernels write, compacted to the
//             front of their slice, and the slice counts shrink accordingly -- from here on the records are exactly what
//             the exact kernels would have left.
//
// Mapping of the filter: k_real_select_mx's (lane = query column j and 16 rows of segment 2 sp + h per tile, records in
// index order into the (segment, query) slice, written by the owning lane).
#pragma once
#include "hg_kernels.hpp"
#include "hg_real_kernels.hpp"
#include "hg_select_mx.hpp"
#include <type_traits>

namespace hg {

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pack_bf16x2(float lo, float hi) {      // v_cvt_pk_bf16_f32 (round to nearest even)
    const f2 v = {lo, hi};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *(const u32*)&r;
}
__device__ __forceinline__ u32 pack_f16x2(float lo, float hi) {       // two v_cvt_f16_f32 (round to nearest even; beyond 65504: inf -- the callers keep such tables out)
    const f2 v = {lo, hi};
    const f16x2 r = __builtin_convertvector(v, f16x2);
    return *(const u32*)&r;
}
// HALF: the filter's 16-bit format is IEEE half (11 significant bits: the margin shrinks eightfold against bfloat16's 8) -- taken when
// every feature of the database is below 2^15 in magnitude (the host checks the largest row norm once per database; a query with a
// feature beyond that keeps every row, decided in the kernel); else bfloat16, whose range is float32's.
template <bool HALF> __device__ __forceinline__ u32 pack_h2(float lo, float hi) { return HALF ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }

// Database image in A-fragment order of v_mfma_f32_32x32x16_bf16: groups of 16 rows; chunk (group G, MFMA m, k-half
// hh, row r) = 16 bytes at (((G * (KP/16) + m) * 2 + hh) * 16 + r) * 16 holding bf16 features 16 m + 8 hh .. + 7 of
// row 16 G + r.
// (rstride > 1: image row i is table row i * rstride -- the sample pass's image of every rstride-th row, N = the sampled rows)
template <bool HALF>
static __global__ __launch_bounds__(256) void k_expand_dbf_bf16(const float* __restrict__ dbf, uint4* __restrict__ img, i64 N, i64 n16, int KP, i64 rstride) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    const int per_row = KP / 8;
    if (i >= n16 * per_row) return;
    const i64 row = i / per_row;
    const int c = (int)(i - row * per_row), m = c >> 1, hh = c & 1;
    uint4 v = {0u, 0u, 0u, 0u};
    if (row < N) {
        const float4* f = (const float4*)(dbf + row * rstride * KP + 16 * m + 8 * hh);
        const float4 a = f[0], b = f[1];
        v.x = pack_h2<HALF>(a.x, a.y); v.y = pack_h2<HALF>(a.z, a.w);
        v.z = pack_h2<HALF>(b.x, b.y); v.w = pack_h2<HALF>(b.z, b.w);
    }
    img[(((row >> 4) * (KP / 16) + m) * 2 + hh) * 16 + (row & 15)] = v;
}

// max over the rows of |x|_2^2 (float32 sum of squares; the caller inflates it), as float bits (non-negative floats
// order like unsigned integers).  out must be zeroed.
static __global__ __launch_bounds__(256) void k_row_norm_max(const float* __restrict__ dbf, i64 N, int KP, u32* __restrict__ out) {
    const i64 row = (i64)blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
    if (row < N) {
        const float4* f = (const float4*)(dbf + row * KP);
        for (int k = 0; k < KP / 4; ++k) { const float4 v = f[k]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    }
    u32 b = __float_as_uint(s);
    if (s != s) b = 0x7F800000u;                                  // NaN features: no bound holds; +inf keeps every pair
    for (int o = 32; o > 0; o >>= 1) { const u32 t = (u32)__shfl_xor((int)b, o); b = t > b ? t : b; }
    if ((threadIdx.x & 63) == 0 && b) atomicMax(out, b);
}

// thr2[q] = thr[q] - eps[q], rounded down (see the header).  One thread per query, double arithmetic.
// u: relative error of one conversion under either rounding (2^-8 bfloat16, 2^-10 half); eta: its absolute floor (half: 2^-14 --
// a subnormal half, or one the matrix pipe flushes to zero, is off by at most the smallest normal; bfloat16 has float32's exponents: 0):
//   |q^ x^ - q x| <= |q^| |x^ - x| + |x| |q^ - q|,  |x^ - x| <= u |x| + eta   =>
//   |approx - chain| <= (2u + u^2) sum |q x| + eta (1 + u) (|q|_1 + |x|_1) + K eta^2 + (K + 2) 2^-21 sum |q^ x^|
static __global__ __launch_bounds__(256) void k_real_thr2(const float* __restrict__ qf, const float* __restrict__ thr, const u32* __restrict__ xmax2,
                                                   float* __restrict__ thr2, int Q, int KP, const double u, const double eta) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    const float t = thr[q];
    if (!(t > -INFINITY && t < INFINITY)) { thr2[q] = t; return; }   // -inf: everything qualifies; +inf: nothing
    double qq = 0.0;
    for (int k = 0; k < KP; ++k) { const double v = (double)qf[(i64)q * KP + k]; qq += v * v; }
    const double xx = (double)__uint_as_float(*xmax2) * 1.0001;       // the float32 sum of squares, inflated
    const double c = 2.0 * u + u * u + (double)(KP + 2) * 4.76837158203125e-07;   // 2^-21
    const double eps = 1.0001 * c * sqrt(qq * xx) + (double)(KP + 2) * 4.76837158203125e-07 * fabs((double)t)
                     + 1.01 * eta * sqrt((double)KP) * (sqrt(qq) + sqrt(xx)) + (double)KP * eta * eta + 1e-30;
    const double want = (double)t - eps;
    float r = (float)want;
    if ((double)r > want) r = __uint_as_float(r > 0.0f ? __float_as_uint(r) - 1u : (r < 0.0f ? __float_as_uint(r) + 1u : 0x80000001u));
    thr2[q] = r;
}

__device__ __forceinline__ f32x16 real_filter_mfma(const bf16x8 a, const bf16x8 b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 real_filter_mfma(const f16x8 a, const f16x8 b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

constexpr int RB_WT = 4;                                     // row tiles per staged window
constexpr int real_bf_lds_bytes(int KP) { return 2 * RB_WT * (KP / 16) * 1024; }

template <int KP, int QT, bool HALF, bool FAR>          // QT query tiles (of 32) per wavefront: 2 up to 128 features, 1 beyond (B fragments live in registers);
                                                        // FAR: a wavefront's 64 record rows span 4 GB or more (every row a record of a huge database): 64-bit cursors
#ifndef HG_RB_WAVES
#define HG_RB_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KP <= 64 ? HG_RB_WAVES : 2, KP <= 64 ? HG_RB_WAVES : 2)))
void k_real_select_bf(const float* __restrict__ qf, const u8* __restrict__ img, const float* __restrict__ thr2, const RealSelArgs a,
                      u32* __restrict__ krows, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 blds[];
    constexpr int WQ = 32 * QT;
    constexpr int NM = KP / 16;                              // MFMAs (and 16-byte A chunks per lane) per tile

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

    // ---- queries: B fragments (-bf16 of features 16 m + 8 h .. + 7 of query j), cut, slice cursors ----
    const int q0w = (qb * WPB + wave) * WQ;
    typedef typename std::conditional<HALF, f16x8, bf16x8>::type hx8;
    hx8 bq[QT][NM];
    float cut[QT];
    u32 cnt[QT], room[QT], dropped[QT], woff[QT], wbeg[QT], wend[QT];
    u32 hold[QT];                                            // the first row number of a pair: two leave in one 8-byte store
    u32* wp[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        const bool live = q < g.Q && seg_ok;
        float big = 0.0f;                                            // HALF: the largest magnitude among the lane's share of the query's features
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            u32 w[4] = {0u, 0u, 0u, 0u};
            if (q < g.Q) {
                const float4* f = (const float4*)(qf + (i64)q * KP + 16 * m + 8 * h);
                const float4 x = f[0], y = f[1];
                w[0] = pack_h2<HALF>(-x.x, -x.y); w[1] = pack_h2<HALF>(-x.z, -x.w);
                w[2] = pack_h2<HALF>(-y.x, -y.y); w[3] = pack_h2<HALF>(-y.z, -y.w);
                if (HALF) {
                    big = fmaxf(big, fmaxf(fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))),
                                           fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)))));
                    // (fmaxf drops a NaN operand: a NaN feature is found by the sum instead -- NaN in, NaN out; inf - inf too, and
                    // an infinite feature is beyond 2^15 anyway)
                    const float sm = ((x.x + x.y) + (x.z + x.w)) + ((y.x + y.y) + (y.z + y.w));
                    if (sm != sm) big = __uint_as_float(0x7F800000u);
                }
            }
            bq[t][m] = *(const hx8*)w;
        }
        cut[t] = live ? thr2[q] : __uint_as_float(0x7F800000u);      // +inf: nothing qualifies
        if (HALF) {
            // a query feature half cannot hold (|q_k| >= 2^15, or NaN): no bound holds for this query -- B = 0 and C = -inf keep every
            // row of it (a superset, like everything this kernel keeps); both lane-halves of the column must agree
            const bool mine = !(big < 32768.0f);
            const bool theirs = __shfl_xor((int)mine, 32) != 0;     // (every lane asks, before any `||` can skip it)
            const bool wild = mine || theirs;
            if (wild) {
#pragma unroll
                for (int m = 0; m < NM; ++m) { const u32 z[4] = {0u, 0u, 0u, 0u}; bq[t][m] = *(const hx8*)z; }
                if (live) cut[t] = __uint_as_float(0xFF800000u);
            }
        }
        cnt[t] = 0; room[t] = live ? a.cap : 0u; dropped[t] = 0; hold[t] = 0;
        wp[t] = krows + (i64)(q < g.Q ? q : 0) * a.crow + (i64)(seg_ok ? s : 0) * a.cap;
        // !FAR: the lane's cursor is a 32-bit byte offset from the wavefront's first row of kept rows (the store takes a scalar base
        // and a vector offset: no 64-bit address arithmetic per hit), and a pair beyond the slice's capacity lands on its last pair
        // of slots -- the query is flagged and redone anyway -- so the hit costs no branch on the room left
        woff[t] = (u32)((((i64)(t * 32 + j)) * a.crow + (i64)(seg_ok ? s : 0) * a.cap) * 4);
        wbeg[t] = woff[t];
        wend[t] = woff[t] + (a.cap - 2u) * 4u;                    // (the slice's last pair of slots; cap is a multiple of 16)
    }
    const char* wbase = (const char*)(krows + (i64)q0w * a.crow);

    // ---- A fragments: windows of RB_WT tiles staged global -> LDS (k_select_mx's scheme), shared by the four
    // wavefronts of the block -- every wavefront copies a quarter of a window and reads all of it, lane-linear, so a
    // lane gets back exactly the 16-byte chunks the MFMA wants from it.  Double-buffered: one barrier per window.
    const int ah = (j >> 2) & 1;                             // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                   // its row inside that half's 16
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;
    const i64 nwin = (ntile + RB_WT - 1) / RB_WT;
    constexpr int STAGE = RB_WT * NM * 1024;
    auto stage_window = [&](const i64 win, const int buf) {
        for (int c = wave; c < RB_WT * NM; c += WPB) {
            const int T = c / NM, m = c - T * NM;
            i64 G = ag0 + win * RB_WT + T;
            G = G < NG ? G : NG - 1;                         // past the end: any valid group (masked later)
            HG_GLDS16(img + ((((G * NM + m) * 2 + h) * 16 + ar) * 16), blds + buf * STAGE + c * 1024);
        }
    };
    if (nwin > 0) stage_window(0, 0);
    for (i64 win = 0; win < nwin; ++win) {
        const int buf = (int)(win & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my copies have landed; after the barrier everybody's have
        __syncthreads();                                     // ... and nobody still reads the other buffer
        if (win + 1 < nwin) stage_window(win + 1, buf ^ 1);
        const u8* st = blds + buf * STAGE;
        const u32 rowbase = g.idx_base + (u32)((i64)s * g.L);      // (dead lanes: never used)
#pragma unroll 1
        for (int Tw = 0; Tw < RB_WT; Tw += 2) {              // two tiles (32 rows of the lane's segment) per drain
            const i64 T = win * RB_WT + Tw;
            if (T >= ntile) break;
            u32 mask[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) mask[t] = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int left = (int)mylen - (int)(T + half) * 16;    // valid rows of this lane in the tile (tiles past the end: none)
                const u32 keep = left >= 16 ? 0xFFFFu : (left <= 0 ? 0u : (1u << left) - 1u);
                hx8 av[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) av[m] = *(const hx8*)(st + (((Tw + half) * NM + m) * 64 + lane) * 16);
                // harvest: bit r <-> row 16 (T + half) + r of the lane's segment may qualify (thr2 - approx < 0)
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = cut[t];
#pragma unroll
                    for (int m = 0; m < NM; ++m) acc = real_filter_mfma(av[m], bq[t][m], acc);
                    u32 mm = 0;
#pragma unroll
                    for (int r = 15; r >= 0; --r) mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(acc[r]), 31);
                    mask[t] |= (mm & keep) << (16 * half);
                }
            }
            // drain: the owning lane writes the row numbers of its hits, lowest row first; the score comes later
            u32 any_mask = 0;
#pragma unroll
            for (int t = 0; t < QT; ++t) any_mask |= mask[t];
            const u32 row0 = rowbase + (u32)(T * 16);
            while (__any(any_mask != 0u)) {
                any_mask = 0;
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    if (mask[t] != 0u) {
                        const int r = __builtin_ctz(mask[t]);
                        mask[t] &= mask[t] - 1u;
                        // Row numbers are four bytes and leave two at a time: the first of a pair waits in a register.  (8-byte
                        // records one by one -- {row, 0}, the score filled in by the rescore -- were 70 M stores of 8 bytes into as
                        // many different lines per call at 10k x 1M: WRITE_SIZE 2.06 GB for 0.56 GB of records.)
                        const u32 row = row0 + (u32)r;
                        if (FAR) {
                            if (room[t]) {
                                if (cnt[t] & 1u) *(u64*)(wp[t] + (cnt[t] - 1u)) = (u64)hold[t] | ((u64)row << 32);
                                else hold[t] = row;
                                ++cnt[t];
                                --room[t];
                            } else {
                                ++dropped[t];
                            }
                        } else {
                            if (woff[t] & 4u) {
                                const u32 at = woff[t] - 4u < wend[t] ? woff[t] - 4u : wend[t];
                                *(u64*)(wbase + at) = (u64)hold[t] | ((u64)row << 32);
                            } else {
                                hold[t] = row;
                            }
                            woff[t] += 4u;
                        }
                    }
                    any_mask |= mask[t];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        if (seg_ok && q < g.Qpad) {
            const bool live = q < g.Q;
            if (!FAR) {
                const u32 hits = (woff[t] - wbeg[t]) >> 2;
                cnt[t] = hits < a.cap ? hits : a.cap;
                dropped[t] = hits - cnt[t];
            }
            if (live && (cnt[t] & 1u) && !dropped[t]) wp[t][cnt[t] - 1u] = hold[t];      // an odd count: the last row number on its own
            a.sl_cnt[(i64)s * g.Qpad + q] = live ? cnt[t] : 0u;
            if (dropped[t] && live) a.fail[q] = 1u;
        }
    }
}

// Sample pass in the filter's 16-bit arithmetic (round 6; k_real_sample_mx ran the exact float32 chains at the vector fma rate:
// 0.35 ms of a 5.4 ms call for scores that only place a cut).  The cut is the rank_s-th largest SAMPLED score and may be any
// number: the filter keeps every pair that can score above it and the rank stage verifies that R rows do (else the bet is lost
// and retried), so approximate sample scores -- off by the filter's own margin at most -- move it by a hair and change nothing
// that is returned.  k_real_sample_mx's mapping (lane = query column j and 16 rows per tile, 16 consecutive samples of its query
// stored as they lie in the accumulator) on k_real_select_bf's fragments: the sampled rows' image is k_expand_dbf_bf16's with a
// row stride, B = +q in half / bfloat16.  A query feature half cannot hold scores 0 everywhere (its cut is void: the filter keeps
// every row of that query anyway).
// OUT16: the scores leave as bfloat16 (round to nearest even: monotone, so the rank_s-th largest of the rounded scores is the
// rounded rank_s-th largest; half the bytes for the pass that is bound by them and for k_real_guess_lds, which reads them all)
template <int KP, bool HALF, bool OUT16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4)))
void k_real_sample_h(const float* __restrict__ qf, const u8* __restrict__ img, float* __restrict__ samp, i64 mstride, const Geo g) {
    constexpr int QT = 2, WQ = 32 * QT;
    constexpr int NM = KP / 16;
    typedef typename std::conditional<HALF, f16x8, bf16x8>::type hx8;
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
    hx8 bq[QT][NM];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        float big = 0.0f;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            u32 w[4] = {0u, 0u, 0u, 0u};
            if (q < g.Q) {
                const float4* f = (const float4*)(qf + (i64)q * KP + 16 * m + 8 * h);
                const float4 x = f[0], y = f[1];
                w[0] = pack_h2<HALF>(x.x, x.y); w[1] = pack_h2<HALF>(x.z, x.w);
                w[2] = pack_h2<HALF>(y.x, y.y); w[3] = pack_h2<HALF>(y.z, y.w);
                if (HALF) {
                    big = fmaxf(big, fmaxf(fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))),
                                           fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)))));
                    const float sm = ((x.x + x.y) + (x.z + x.w)) + ((y.x + y.y) + (y.z + y.w));
                    if (sm != sm) big = __uint_as_float(0x7F800000u);
                }
            }
            bq[t][m] = *(const hx8*)w;
        }
        if (HALF) {
            const bool mine = !(big < 32768.0f);
            const bool theirs = __shfl_xor((int)mine, 32) != 0;     // (every lane asks)
            if (mine || theirs) {
#pragma unroll
                for (int m = 0; m < NM; ++m) { const u32 z[4] = {0u, 0u, 0u, 0u}; bq[t][m] = *(const hx8*)z; }
            }
        }
    }
    const int ah = (j >> 2) & 1;
    const int ar = (j & 3) + 4 * (j >> 3);
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;
    auto chunk = [&](const i64 T, const int m) -> hx8 {
        i64 G = ag0 + T;
        G = G < NG ? G : NG - 1;                             // past the end: any valid group (never stored)
        return *(const hx8*)(img + ((((G * NM + m) * 2 + h) * 16 + ar) * 16));
    };
    hx8 av[NM];
    if (ntile > 0) {
#pragma unroll
        for (int m = 0; m < NM; ++m) av[m] = chunk(0, m);
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
            for (int m = 0; m < NM; ++m) {
                acc = real_filter_mfma(av[m], bq[t][m], acc);
                if (t == QT - 1) av[m] = chunk(Tn, m);
            }
            const int q = q0w + t * 32 + j;
            if (q < g.Q && left > 0) {
                if (OUT16) {
                    u16* out = (u16*)samp + (i64)q * mstride + (i64)s * g.L + T * 16;   // 32-byte aligned
                    if (left >= 16) {
#pragma unroll
                        for (int r8 = 0; r8 < 2; ++r8)
                            ((uint4*)out)[r8] = uint4{pack_bf16x2(acc[8 * r8] + 0.0f, acc[8 * r8 + 1] + 0.0f), pack_bf16x2(acc[8 * r8 + 2] + 0.0f, acc[8 * r8 + 3] + 0.0f),
                                                      pack_bf16x2(acc[8 * r8 + 4] + 0.0f, acc[8 * r8 + 5] + 0.0f), pack_bf16x2(acc[8 * r8 + 6] + 0.0f, acc[8 * r8 + 7] + 0.0f)};
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) if (r < left) out[r] = (u16)(pack_bf16x2(acc[r] + 0.0f, 0.0f) & 0xFFFFu);
                    }
                } else {
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
}

// A second, COUNTING sample (round 6).  The first sample places the cut `sigma` deviations of ITS count deep -- 64 expected hits: 5 sigma
// = 62 % more rows than R reach the filter's records, the rescore and the rank stage.  This pass runs the same 16-bit products over a
// sample four times as large and stores nothing: a score above the first cut lo[q] is counted into one of RC_BINS equal bins of
// [lo, lo + span) (the last bin takes everything beyond), in LDS, and flushed with global atomic adds.  k_real_guess2 then moves
// the cut up to the highest bin edge that still has `need` sampled scores at or above it (256 expected hits: 5 sigma = 31 %).  Any edge
// is a valid cut (the rank stage verifies that R rows score above it); a query whose lo is not a positive finite number is left alone.
constexpr int RC_BINS = 32;
__device__ __forceinline__ float rc_span(const float lo) { return 0.16f * lo + 1.0e-6f; }      // ~0.4 standard deviations of a query's scores when lo sits 2.4 deep

// LDS: the filter's double-buffered windows of RB_WT row tiles (one copy of the fragments for the block's four wavefronts: straight out of
// the L2 every wavefront fetched its own -- 1 GB per pass, the pass's time), then per wavefront 64 queries x 32 bins of 16-bit counts, two to
// a dword (a block meets at most 2 x L < 65 536 rows).
constexpr int real_count_lds_bytes(int KP) { return real_bf_lds_bytes(KP) + WPB * 64 * RC_BINS * 2; }

template <int KP, bool HALF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4)))
void k_real_sample_count(const float* __restrict__ qf, const u8* __restrict__ img, const float* __restrict__ thr, u32* __restrict__ hist, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 clds[];
    constexpr int QT = 2, WQ = 32 * QT;
    constexpr int NM = KP / 16;
    constexpr int STAGE = RB_WT * NM * 1024;
    typedef typename std::conditional<HALF, f16x8, bf16x8>::type hx8;
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32* lh = (u32*)(clds + 2 * STAGE) + wave * (WQ * RC_BINS / 2);        // [64 queries][16 dwords]: bin k of query x in half (k & 1) of dword x * 16 + k / 2
    const int nQB = g.nQT;
    const int sp = lb / nQB, qb = lb - sp * nQB;
    const int h = lane >> 5, j = lane & 31;
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = lo0 >= g.N ? 0 : (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 ntile = ((len0 > len1 ? len0 : len1) + 15) / 16;
    const i64 NG = (g.N + 15) >> 4;
    const int q0w = (qb * WPB + wave) * WQ;
    for (int e = lane; e < WQ * RC_BINS / 2; e += 64) lh[e] = 0u;
    hx8 bq[QT][NM];
    float nlo[QT], invw[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        float big = 0.0f;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            u32 w[4] = {0u, 0u, 0u, 0u};
            if (q < g.Q) {
                const float4* f = (const float4*)(qf + (i64)q * KP + 16 * m + 8 * h);
                const float4 x = f[0], y = f[1];
                w[0] = pack_h2<HALF>(-x.x, -x.y); w[1] = pack_h2<HALF>(-x.z, -x.w);        // B = -q (acc = lo - approx: the filter's sign test)
                w[2] = pack_h2<HALF>(-y.x, -y.y); w[3] = pack_h2<HALF>(-y.z, -y.w);
                if (HALF) {
                    big = fmaxf(big, fmaxf(fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))),
                                           fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)))));
                    const float sm = ((x.x + x.y) + (x.z + x.w)) + ((y.x + y.y) + (y.z + y.w));
                    if (sm != sm) big = __uint_as_float(0x7F800000u);
                }
            }
            bq[t][m] = *(const hx8*)w;
        }
        bool wild = false;
        if (HALF) {
            const bool mine = !(big < 32768.0f);
            const bool theirs = __shfl_xor((int)mine, 32) != 0;     // (every lane asks)
            wild = mine || theirs;
        }
        const float lo = q < g.Q ? thr[q] : 0.0f;
        const bool ok = q < g.Q && !wild && lo > 0.0f && lo < 3.0e38f;                 // (else: nothing is counted for this query, its cut stays)
        nlo[t] = ok ? lo : __uint_as_float(0x7F800000u);                                // C = lo: acc = lo - approx; +inf: never below zero
        invw[t] = ok ? -(float)RC_BINS / rc_span(lo) : 0.0f;
    }
    const int ah = (j >> 2) & 1;                             // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                   // its row inside that half's 16
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;
    const i64 nwin = (ntile + RB_WT - 1) / RB_WT;
    auto stage_window = [&](const i64 win, const int buf) {
        for (int c = wave; c < RB_WT * NM; c += WPB) {
            const int T = c / NM, m = c - T * NM;
            i64 G = ag0 + win * RB_WT + T;
            G = G < NG ? G : NG - 1;                         // past the end: any valid group (masked by `left`)
            HG_GLDS16(img + ((((G * NM + m) * 2 + h) * 16 + ar) * 16), clds + buf * STAGE + c * 1024);
        }
    };
    if (nwin > 0) stage_window(0, 0);
    for (i64 win = 0; win < nwin; ++win) {
        const int buf = (int)(win & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my copies have landed; after the barrier everybody's have
        __syncthreads();                                     // ... and nobody still reads the other buffer (nor zeroes its counters)
        if (win + 1 < nwin) stage_window(win + 1, buf ^ 1);
        const u8* st = clds + buf * STAGE;
#pragma unroll 1
        for (int Tw = 0; Tw < RB_WT; ++Tw) {
            const i64 T = win * RB_WT + Tw;
            if (T >= ntile) break;
            const int left = (int)(mylen - T * 16);
            const u32 keep = left >= 16 ? 0xFFFFu : (left <= 0 ? 0u : (1u << left) - 1u);
            hx8 av[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) av[m] = *(const hx8*)(st + ((Tw * NM + m) * 64 + lane) * 16);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = nlo[t];
#pragma unroll
                for (int m = 0; m < NM; ++m) acc = real_filter_mfma(av[m], bq[t][m], acc);
                u32 mask = 0;
#pragma unroll
                for (int r = 15; r >= 0; --r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(acc[r]), 31);
                mask &= keep;
                // a lane meets a score above the cut in one tile of four: a round or two per tile, the lane's row picked out of its sixteen
                // accumulators by four levels of bit-field inserts (sixteen tests and branches per tile -- some lane always has a hit --
                // were 3/4 of this pass's instructions; selects on the bits of r the compiler turns into an indexed trip through memory)
                u32* mine = lh + (t * 32 + j) * (RC_BINS / 2);
                while (__any(mask != 0u)) {
                    if (mask != 0u) {
                        const u32 r = (u32)__builtin_ctz(mask);
                        mask &= mask - 1u;
                        const u32 m3 = 0u - ((r >> 3) & 1u), m2 = 0u - ((r >> 2) & 1u), m1 = 0u - ((r >> 1) & 1u), m0 = 0u - (r & 1u);
                        u32 a8[8], a4[4], a2[2];
#pragma unroll
                        for (int k = 0; k < 8; ++k) a8[k] = (__float_as_uint(acc[8 + k]) & m3) | (__float_as_uint(acc[k]) & ~m3);
#pragma unroll
                        for (int k = 0; k < 4; ++k) a4[k] = (a8[4 + k] & m2) | (a8[k] & ~m2);
#pragma unroll
                        for (int k = 0; k < 2; ++k) a2[k] = (a4[2 + k] & m1) | (a4[k] & ~m1);
                        const float d = __uint_as_float((a2[1] & m0) | (a2[0] & ~m0));        // lo - approx < 0
                        const int bk0 = (int)(d * invw[t]);
                        const int bk = bk0 < RC_BINS - 1 ? bk0 : RC_BINS - 1;
                        atomicAdd(&mine[bk >> 1], 1u << (16 * (bk & 1)));
                    }
                }
            }
        }
    }
    __syncthreads();                                         // (a block without windows still zeroed its counters above)
    // this segment pair's counts, plain stores (k_real_guess2 adds the pairs up): [sp][q][bin] u32, unpacked on the way out
    u32* __restrict__ out = hist + ((i64)sp * g.Qpad + q0w) * RC_BINS;
    for (int e = lane; e < WQ * RC_BINS; e += 64)
        if (q0w + e / RC_BINS < g.Qpad) out[e] = (lh[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
}

// thr[q] <- the highest edge of k_real_sample_count's bins that still has `need` sampled scores at or above it (never below thr[q])
// (one wavefront-lane per (query, bin): lane k of a group of 32 sums bin k over the nSP segment pairs, the group scans from the top)
static __global__ __launch_bounds__(256) void k_real_guess2(const u32* __restrict__ hist, float* __restrict__ thr, const int Q, const i64 Qpad, const int nSP,
                                                     const u32 need) {
    const int q = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    const bool live = q < Q;
    u32 c = 0;
    if (live) for (int sp = 0; sp < nSP; ++sp) c += hist[((i64)sp * Qpad + q) * RC_BINS + k];
    // suffix sums inside the group of 32 lanes: cum[k] = c[k] + c[k + 1] + ... + c[31]
    u32 cum = c;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const u32 t = (u32)__shfl_down((int)cum, o, 32); if (k + o < 32) cum += t; }
    const u64 ok = __ballot(live && k >= 1 && cum >= need);
    const u32 mine = (u32)(ok >> (threadIdx.x & 32));          // this group's 32 flags
    if (!live || k != 0 || mine == 0u) return;
    const float lo = thr[q];
    if (!(lo > 0.0f && lo < 3.0e38f)) return;
    const float w = rc_span(lo) / (float)RC_BINS;
    const int edge = 31 - __builtin_clz(mine);                 // the highest bin that still has `need` scores at or above it
    {
        // a score in bin k is >= lo + k w up to the rounding of (score - lo) * (1 / w): a few units in the last place of slack
        const float e = lo + (float)edge * w;
        thr[q] = e - 4.0f * 1.1920929e-07f * fabsf(e) - w * 1.0e-3f;
    }
}

// Rescore: wavefront = (group of SG consecutive slices, query); lane = one kept row.  The query's features are
// wave-uniform (scalar loads); consecutive wavefronts take consecutive queries of the SAME segment group, whose rows
// (SG x real_segment_bytes) stay in the L2 while all queries pass.  The 64 rows of a round are fetched COALESCED, 32
// features at a time -- 8 lanes per row, 8 rows per load instruction, whole cache lines -- and turned through LDS so that
// every lane then walks its own row in feature order: one float32 fma chain, k ascending.  Rows lie 128 bytes apart in
// LDS, unpadded (8 KB per wavefront: five blocks per compute unit instead of the four that 144-byte rows allowed); the
// 16-byte piece p of row r sits in slot p ^ ((r >> 1) & 7), so the 16 lanes of a ds_read_b128 group (16 rows, one p)
// hit 16 different bank quads, and a row's 8 pieces written by 8 consecutive lanes fill its 128 bytes in some order.
constexpr int RS_ROWB = 128;                                 // bytes per staged row: 32 floats
constexpr int rescore_lds_bytes() { return WPB * 64 * RS_ROWB; }

template <int KPT, int SG>          // KPT: the (padded) feature count, 0 = taken at run time (kp_rt; beyond 128 features)
__global__ __launch_bounds__(256) void k_real_rescore(const float* __restrict__ qf, const float* __restrict__ dbf, const u32* sl_cnt,
                                                      const u32* __restrict__ krows, u64* __restrict__ cand, u32 cap, i64 crow, const float* __restrict__ thr,
                                                      u32* sl_cnt_out, u32* __restrict__ cnt_by_query, const u64* __restrict__ dblab, const u64* __restrict__ qlab,
                                                      const int embed_match, const int kp_rt, const Geo g) {      // (sl_cnt_out may be sl_cnt; cnt_by_query: the same counts [Q][S])
    const int KP = KPT ? KPT : kp_rt;
    extern __shared__ __attribute__((aligned(16))) u8 rlds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 unit = (i64)blockIdx.x * WPB + wave;
    const int nG = (g.S + SG - 1) / SG;
    if (unit >= (i64)nG * g.Q) return;
    const int sg = __builtin_amdgcn_readfirstlane((int)(unit / g.Q));
    const int q = __builtin_amdgcn_readfirstlane((int)(unit - (i64)sg * g.Q));
    const int s0 = sg * SG;
    u32 pre[SG + 1];
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < SG; ++k) pre[k + 1] = pre[k] + (s0 + k < g.S ? sl_cnt[(i64)(s0 + k) * g.Qpad + q] : 0u);
    const u32 total = pre[SG];
    const float* __restrict__ qrow = qf + (i64)q * KP;
    const float cutq = thr[q];
    u64* __restrict__ rows = cand + (i64)q * crow + (i64)s0 * cap;                  // the records it leaves, slice by slice
    const u32* __restrict__ kept_rows = krows + (i64)q * crow + (i64)s0 * cap;      // the row numbers the filter kept, the same slices
    u8* st = rlds + wave * 64 * RS_ROWB;
    u32 kept[SG];                                            // records of each slice that score above the cut, so far
#pragma unroll
    for (int x = 0; x < SG; ++x) kept[x] = 0;
    const u64 below = (1ull << lane) - 1ull;
    const int pr = lane >> 3, pp = lane & 7;                 // staging role: row 8 e + pr of the round, 16-byte piece pp
    const int rsw = (lane >> 1) & 7;                         // reading role: row `lane`, piece p from slot p ^ rsw
    // record i of the group's concatenated slices -> (slice k, offset): the row number of the NEXT round is requested before this
    // round's gathers, so a round is two dependent memory round trips (gathers, write-back) instead of three
    auto locate = [&](const u32 i, int& k, u32& off) {
        k = 0;
#pragma unroll
        for (int x = 1; x < SG; ++x) k += i >= pre[x] ? 1 : 0;
        off = i;
#pragma unroll
        for (int x = 1; x < SG; ++x) off = k == x ? i - pre[x] : off;
    };
    u32 idx_next = g.idx_base;
    {
        int k0; u32 off0;
        locate((u32)lane, k0, off0);
        if ((u32)lane < total) idx_next = kept_rows[(i64)k0 * cap + off0];
    }
    for (u32 base = 0; base < total; base += 64) {
        const u32 i = base + lane;
        const bool valid = i < total;
        int k; u32 off;
        locate(i, k, off);
        const u32 idx = idx_next;                                               // (idle lanes: any valid row)
        idx_next = g.idx_base;
        if (i + 64 < total) {
            int kn; u32 offn;
            locate(i + 64, kn, offn);
            idx_next = kept_rows[(i64)kn * cap + offn];
        }
        const u32 local = idx - g.idx_base;
        // the row's label words travel with its features (round 6): the record takes its label-match bit (metric.py:17-19) along as bit 31 of
        // the index half (a float table has fewer than 2^31 rows: run_real), and k_real_rank_lds no longer gathers 8 bytes per RANK out of the
        // label table -- 2.3 GB of fetches per call at 10k x 1M for 0.6 GB of records, its list phase a quarter of its time
        const u64 lab0 = dblab[(i64)local * g.LW];
        u32 ridx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ridx[e] = (u32)__shfl((int)local, 8 * e + pr);
        float acc = 0.0f;
#pragma unroll 1
        for (int c0 = 0; c0 < KP; c0 += 64) {                // 64 features per trip: both 32-feature pieces' loads in flight at once
            float4 gl[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    gl[u][e] = c0 + 32 * u + 4 * pp < KP ? *(const float4*)(dbf + (i64)ridx[e] * KP + c0 + 32 * u + 4 * pp)
                                                         : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cu = c0 + 32 * u;
                if (cu < KP) {                                // (wave-uniform)
#pragma unroll
                    for (int e = 0; e < 8; ++e) *(float4*)(st + (8 * e + pr) * RS_ROWB + ((pp ^ (pr >> 1) ^ (4 * (e & 1))) * 16)) = gl[u][e];
                    wave_lds_sync();
                    float4 v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = *(const float4*)(st + lane * RS_ROWB + ((e ^ rsw) * 16));
                    wave_lds_sync();
                    // KP is a multiple of 16: a piece is whole or half -- no per-feature guards, so the query's scalar loads batch
                    const float* __restrict__ qc = qrow + cu;
                    if (cu + 32 <= KP) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            acc = __builtin_fmaf(qc[4 * e + 0], v[e].x, acc);
                            acc = __builtin_fmaf(qc[4 * e + 1], v[e].y, acc);
                            acc = __builtin_fmaf(qc[4 * e + 2], v[e].z, acc);
                            acc = __builtin_fmaf(qc[4 * e + 3], v[e].w, acc);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc = __builtin_fmaf(qc[4 * e + 0], v[e].x, acc);
                            acc = __builtin_fmaf(qc[4 * e + 1], v[e].y, acc);
                            acc = __builtin_fmaf(qc[4 * e + 2], v[e].z, acc);
                            acc = __builtin_fmaf(qc[4 * e + 3], v[e].w, acc);
                        }
                    }
                }
            }
        }
        // Only rows that score above the cut stay (the filter's margin let others through): they move to the front of
        // their slice, in index order -- every read of this round is done, and a record never moves to the right.
        const float ipv = acc + 0.0f;
        const bool pass = valid && !(ipv <= cutq);
        const u64 bal = __ballot(pass);
        u64 mine = 0;
        u32 before = 0;
#pragma unroll
        for (int x = 0; x < SG; ++x) {
            const u64 m = __ballot(valid && k == x);
            if (k == x) { mine = m; before = kept[x]; }
            kept[x] += (u32)__popcll(bal & m);
        }
        if (pass) {
            u64 any = lab0 & qlab[(i64)q * g.LW];
            for (int w = 1; w < g.LW; ++w) any |= dblab[(i64)local * g.LW + w] & qlab[(i64)q * g.LW + w];
            // (embed_match = 0: the records go to kernels that order by the whole 64 bits -- every row a record, > 128 features)
            rows[(i64)k * cap + before + (u32)__popcll(bal & mine & below)] = ((u64)(~mono_key(ipv)) << 32) | (u64)(idx | (any && embed_match ? 0x80000000u : 0u));
        }
    }
#pragma unroll
    for (int x = 0; x < SG; ++x)
        if (lane == x && s0 + x < g.S) {
            sl_cnt_out[(i64)(s0 + x) * g.Qpad + q] = kept[x];
            cnt_by_query[(i64)q * g.S + s0 + x] = kept[x];       // the rank kernel reads a query's counts as one run (its block met S cache lines for S counts)
        }
}

// Sample pass for any feature count (k_real_sample keeps a query's features in registers: up to 128): wavefront =
// (64 sampled rows, 64 queries), lane = query.  16 rows at a time, 32 features at a time: the rows' pieces are staged
// coalesced in LDS and read back as broadcasts, the lane's query piece sits in registers; each of the 16 accumulators
// runs the float32 fma chain over k ascending, across the pieces.
static __global__ __launch_bounds__(256) void k_real_sample_any(const float* __restrict__ qf, const float* __restrict__ dbf, float* __restrict__ samp,
                                                         i64 M, i64 stride, const int KP, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u8 slds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 unit = (i64)blockIdx.x * WPB + wave;
    const i64 nTiles = (M + 63) / 64;
    if (unit >= nTiles * g.nQT) return;
    const i64 tile = unit / g.nQT;
    const int qt = (int)(unit - tile * g.nQT);
    const int q = qt * 64 + lane;
    const i64 j0 = tile * 64;
    const int nj = (int)(M - j0 < 64 ? M - j0 : 64);
    float* tl = (float*)(slds + wave * (64 * 65 * 4 + 16 * 128));       // [64][65] score tile, then [16][32] row pieces
    float* rowp = tl + 64 * 65;
    const float* __restrict__ qrow = qf + (i64)(q < g.Q ? q : 0) * KP;
    const int pr = lane >> 3, pp = lane & 7;                 // staging role: row 8 e + pr of the group, 16-byte piece pp
    for (int rg = 0; rg < 64; rg += 16) {
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        for (int c0 = 0; c0 < KP; c0 += 32) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jr = rg + 8 * e + pr;
                float4 v = {0.f, 0.f, 0.f, 0.f};
                if (jr < nj && c0 + 4 * pp < KP) v = *(const float4*)(dbf + (j0 + jr) * stride * KP + c0 + 4 * pp);
                *(float4*)(rowp + (8 * e + pr) * 32 + 4 * pp) = v;
            }
            float4 qv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] = c0 + 4 * e < KP ? *(const float4*)(qrow + c0 + 4 * e) : float4{0.f, 0.f, 0.f, 0.f};
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (c0 + 4 * e < KP) {
                        const float4 x = *(const float4*)(rowp + r * 32 + 4 * e);   // same address in every lane: a broadcast
                        acc[r] = __builtin_fmaf(qv[e].x, x.x, acc[r]);
                        acc[r] = __builtin_fmaf(qv[e].y, x.y, acc[r]);
                        acc[r] = __builtin_fmaf(qv[e].z, x.z, acc[r]);
                        acc[r] = __builtin_fmaf(qv[e].w, x.w, acc[r]);
                    }
                }
            }
            wave_lds_sync();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tl[(rg + r) * 65 + lane] = acc[r] + 0.0f;
    }
    wave_lds_sync();
    for (int qq = 0; qq < 64; ++qq) {                         // transposed write: lane <-> sample
        const int qo = qt * 64 + qq;
        if (qo < g.Q && lane < nj) samp[(i64)qo * M + j0 + lane] = tl[lane * 65 + qq];
    }
}

// Rank + finish in one kernel, one block of 1024 threads per query (replaces 4 x k_radix_pass + k_real_finish when a
// query's records fit the LDS): everything after the one coalesced copy of the records happens in LDS --
//   1  slice offsets (scan of the slice counts), records copied in index order: A[0 .. n)
//   2  radix select (11 + 11 + 10 bits of the key half) of K, the R-th smallest key, and of how many records with
//      key == K belong to the first R; the query is flagged if that record does not score above thr (see k_real_finish)
//   3  ordered compaction IN PLACE: the R chosen records, still in index order, A[0 .. R)
//   4  four stable counting passes over 16-bit positions P[0 .. R) by the bytes of the key (per-wave digit counters,
//      ballot matching inside a 64-record batch: k_radix_pass's scheme, but LDS to LDS)
//   5  the ranked list: idx and score of A[P[k]], and the label-match bit of every rank (k_match's gather).
// A query whose records exceed NA sets bit 1 of *err (the host then ranks with the global-memory passes); one whose
// cut was too high or whose slices overflowed sets bit 0 (a lost bet).
constexpr u32 RG_PILE = 96;                                 // a fine bucket this full is still ranked by counting (its records compare against each other: 96^2 per
                                                             // bucket at worst); beyond, the scores sit on a grid and the radix passes take over.  (24 until round 6: one
                                                             // bucket of 26 among a CIFAR-sized call's 10^4 groups sent the whole call to the radix passes, 1.8 -> 5.4 ms.)
constexpr int RK_SMAX = 4096;                                // slices per query the offsets array takes
constexpr int RK_RMAX = 6144;                                // ranked-list length the position arrays take
template <int NA> constexpr size_t real_rank_lds_bytes() { return (size_t)NA * 8 + (size_t)RK_RMAX * 4 + (RK_SMAX + 1) * 4; }

__device__ __forceinline__ u32 block_excl_scan_1024(const u32 v, u32* s_w, u32& total) {   // 1024 threads; s_w: 16 dwords
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    u32 wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const u32 x = s_w[w]; wbase += w < wave ? x : 0u; tot += x; }
    __syncthreads();
    total = tot;
    return wbase + inc - v;
}

template <int NA>
__global__ __launch_bounds__(1024) void k_real_rank_lds(const u64* __restrict__ cand, i64 crow, u32 cap, const u32* __restrict__ sl_cnt,
                                                        const u32* __restrict__ fail, const float* __restrict__ thr,
                                                        u32* __restrict__ out_idx, float* __restrict__ scores,
                                                        const u64* __restrict__ dblab, const u64* __restrict__ qlab, u64* __restrict__ mbits, i64 RW,
                                                        int* __restrict__ err, u32* __restrict__ qbad, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    u64* A = smem;                                           // [NA] records
    u16* P0 = (u16*)(A + NA);                                // [RK_RMAX] positions, ping
    u16* P1 = P0 + RK_RMAX;                                  // [RK_RMAX] pong
    u32* hw = (u32*)(P1 + RK_RMAX);                          // [16][256] per-wave digit counters / [2048] select histogram ...
    u32* off = hw;                                           // ... / [S + 1] slice offsets (step 1 only)
    __shared__ u32 s_w[16];
    __shared__ u32 s_prefix, s_need;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 R = (u32)g.R;
#ifdef HG_REAL_RANK_PROFILE          // phase timestamps overwrite the head of the query's ranked list (tools/real_rank_phase_profile.py)
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    u32 tks[8];
    int tkn = 0;
#define HG_RTK() do { tks[tkn++] = (u32)(__builtin_amdgcn_s_memtime() - tk0); } while (0)
#else
#define HG_RTK() do {} while (0)
#endif

    // ---- 1: offsets, copy ----
    u32 n = 0;
    for (int sb = 0; sb < g.S; sb += 1024) {
        const int s = sb + tid;
        const u32 c = s < g.S ? sl_cnt[(i64)q * g.S + s] : 0u;       // (query-major: k_real_rescore's cnt_by_query)
        u32 tot;
        const u32 ex = block_excl_scan_1024(c, s_w, tot);
        if (s < g.S) off[s] = n + ex;
        n += tot;
    }
    if (tid == 0) off[g.S] = n;
    __syncthreads();
    const bool over = n > (u32)NA;
    bool bad = fail[q] != 0u || over || n < R;
    if (!bad) {
        const u64* __restrict__ row = cand + (i64)q * crow;
        // 16 threads per slice, 64 slices per sweep -- and the loads of eight sweeps issued before the first record is
        // stored: one trip to memory per 512 slices instead of two per 64 (the usual slice holds ~20 records; the copy
        // was a chain of 16 dependent round trips, a quarter of the block's life at 10k x 1M)
        constexpr int CG = 8;
        for (int sb0 = 0; sb0 < g.S; sb0 += 64 * CG) {
            u64 v[CG][2];
            u32 o_[CG], c_[CG];
#pragma unroll
            for (int k = 0; k < CG; ++k) {
                const int s = sb0 + 64 * k + (tid >> 4);
                o_[k] = s < g.S ? off[s] : 0u;
                c_[k] = s < g.S ? off[s + 1] - o_[k] : 0u;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32 i = (u32)(tid & 15) + 16u * j;
                    v[k][j] = i < c_[k] ? row[(i64)s * cap + i] : 0ull;
                }
            }
#pragma unroll
            for (int k = 0; k < CG; ++k) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32 i = (u32)(tid & 15) + 16u * j;
                    if (i < c_[k]) A[o_[k] + i] = v[k][j];
                }
            }
#pragma unroll
            for (int k = 0; k < CG; ++k) {                                  // slices of more than 32 records
                const int s = sb0 + 64 * k + (tid >> 4);
                for (u32 i = 32u + (u32)(tid & 15); i < c_[k]; i += 16) A[o_[k] + i] = row[(i64)s * cap + i];
            }
        }
        __syncthreads();
        HG_RTK();                                             // 0: offsets + copy
        // ---- 2'-4': the usual case in one bucket pass over ALL n records.  4096 buckets of equal key width between the
        // smallest and the largest key; the bucket in which the cumulative count passes R is the boundary: every record of
        // the buckets up to it is scattered to its bucket's span (any order), then counts the records of its bucket that
        // precede it by (key, position) -- position order is index order, the tie order -- and takes that final rank if it is
        // below R.  One histogram instead of the select's three, the compaction and the ordering pass; keys that pile up
        // (scores on a grid) or a boundary bucket beyond the position arrays leave it to the steps below.
        const u16* Pfin = nullptr;
        bool ordered = false;
        {
            u32 kmin = 0xFFFFFFFFu, kmax = 0u;
            for (u32 i = tid; i < n; i += 1024) { const u32 k = (u32)(A[i] >> 32); kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const u32 v = (u32)__shfl_xor((int)kmin, o), w = (u32)__shfl_xor((int)kmax, o);
                kmin = v < kmin ? v : kmin; kmax = w > kmax ? w : kmax;
            }
            __syncthreads();                                  // (off[] in hw is no longer read)
            u32* wmax = (u32*)P1;                             // (the position arrays are idle until the scatter)
            if (lane == 0) { s_w[wave] = kmin; wmax[wave] = kmax; }
            for (int i = tid; i < 4096; i += 1024) hw[i] = 0u;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 16; ++w) { const u32 v = s_w[w], x = wmax[w]; kmin = v < kmin ? v : kmin; kmax = x > kmax ? x : kmax; }
            const u32 range = kmax - kmin;
            const int bsh = range < 4096u ? 0 : 32 - __builtin_clz(range) - 12;         // (key - kmin) >> bsh <= 4095
            for (u32 i = tid; i < n; i += 1024) atomicAdd(&hw[((u32)(A[i] >> 32) - kmin) >> bsh], 1u);
            __syncthreads();
            u32 c4[4], sum = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x) { c4[x] = hw[4 * tid + x]; sum += c4[x]; }
            u32 tot;
            const u32 ex = block_excl_scan_1024(sum, s_w, tot);
            if (ex < R && R <= ex + sum) {                    // this thread's bins hold the boundary
                u32 run = ex;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if (run < R && R <= run + c4[x]) { s_prefix = 4u * tid + x; s_need = run + c4[x]; }
                    run += c4[x];
                }
            }
            __syncthreads();
            const u32 bstar = s_prefix, cend = s_need;        // boundary bucket, records in the buckets up to it
            u32 big = cend > (u32)RK_RMAX ? 1u : 0u;
#pragma unroll
            for (int x = 0; x < 4; ++x) big |= (4u * tid + x <= bstar && c4[x] > RG_PILE) ? 1u : 0u;
            if (!__syncthreads_or((int)big)) {
                u32 run = ex;
#pragma unroll
                for (int x = 0; x < 4; ++x) { hw[4 * tid + x] = run; run += c4[x]; }
                __syncthreads();
                HG_RTK();                                     // 1: histogram + boundary
                for (u32 i = tid; i < n; i += 1024) {
                    const u32 bk = ((u32)(A[i] >> 32) - kmin) >> bsh;
                    if (bk <= bstar) P1[atomicAdd(&hw[bk], 1u)] = (u16)i;
                }
                __syncthreads();
                HG_RTK();                                     // 2: scatter
                for (u32 a = tid; a < cend; a += 1024) {
                    const u16 p = P1[a];
                    const u32 key = (u32)(A[p] >> 32);
                    const u32 bk = (key - kmin) >> bsh;
                    const u32 lo = bk ? hw[bk - 1] : 0u, hi = hw[bk];              // (after the scatter hw[b] is the END of bucket b)
                    const u64 kp = ((u64)key << 16) | p;
                    u32 before = 0;
                    for (u32 e = lo; e < hi; ++e) {
                        const u16 pq = P1[e];
                        const u64 kq = ((u64)(u32)(A[pq] >> 32) << 16) | pq;
                        before += kq < kp ? 1u : 0u;
                    }
                    if (lo + before < R) P0[lo + before] = p;
                }
                __syncthreads();
                HG_RTK();                                     // 3: order
                bad = !((u32)(A[P0[R - 1]] >> 32) < ~mono_key(thr[q]));             // the R-th record must score above thr
                Pfin = P0;
                ordered = true;
            }
        }
        if (!ordered) {
        // ---- 2: K = the R-th smallest key ----
        if (tid == 0) { s_prefix = 0u; s_need = R; }
        u32 mask = 0;
        const int shifts[3] = {21, 10, 0};
        const int widths[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            hw[tid] = 0u; hw[tid + 1024] = 0u;
            __syncthreads();
            const u32 prefix = s_prefix, need = s_need, bins = 1u << widths[pass];
            for (u32 i = tid; i < n; i += 1024) {
                const u32 k = (u32)(A[i] >> 32);
                if ((k & mask) == prefix) atomicAdd(&hw[(k >> shifts[pass]) & (bins - 1u)], 1u);
            }
            __syncthreads();
            const u32 c0 = 2u * tid < bins ? hw[2 * tid] : 0u, c1 = 2u * tid + 1u < bins ? hw[2 * tid + 1] : 0u;
            u32 tot;
            const u32 ex = block_excl_scan_1024(c0 + c1, s_w, tot);
            if (ex < need && need <= ex + c0) { s_prefix = prefix | ((2u * tid) << shifts[pass]); s_need = need - ex; }
            else if (ex + c0 < need && need <= ex + c0 + c1) { s_prefix = prefix | ((2u * tid + 1u) << shifts[pass]); s_need = need - ex - c0; }
            __syncthreads();
            mask |= (bins - 1u) << shifts[pass];
        }
        const u32 K = s_prefix, need_eq = s_need;
        HG_RTK();                                             // 1: select
        bad = !(K < ~mono_key(thr[q]));                      // the R-th record must score above thr
        if (!bad) {
            // ---- 3: the chosen R records, in index order, compacted in place (all reads, barrier, all writes) ----
            constexpr int CH = (NA + 1023) / 1024;
            const u32 chunk = (n + 1023u) / 1024u;
            const u32 lo = tid * chunk < n ? tid * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
            u64 mine[CH];
            u32 lt = 0, eq = 0;
#pragma unroll
            for (int x = 0; x < CH; ++x) {
                mine[x] = lo + x < hi ? A[lo + x] : ~0ull;
                const u32 k = (u32)(mine[x] >> 32);
                lt += lo + x < hi && k < K ? 1u : 0u; eq += lo + x < hi && k == K ? 1u : 0u;
            }
            u32 t0, t1;
            const u32 lt_ex = block_excl_scan_1024(lt, s_w, t0);     // (its barriers separate the reads above from the writes below)
            u32 e = block_excl_scan_1024(eq, s_w, t1);
            u32 pos = lt_ex + (e < need_eq ? e : need_eq);
#pragma unroll
            for (int x = 0; x < CH; ++x) {
                if (lo + x < hi) {
                    const u32 k = (u32)(mine[x] >> 32);
                    if (k < K) A[pos++] = mine[x];
                    else if (k == K) { if (e < need_eq) A[pos++] = mine[x]; ++e; }
                }
            }
            for (u32 i = tid; i < R; i += 1024) P0[i] = (u16)i;
            __syncthreads();
            HG_RTK();                                         // 2: compaction
            // ---- 4a: one bucket pass.  The R keys spread over a range whose size is known (K is the largest, a block
            // reduction finds the smallest): 4096 buckets of equal key width hold one or two records each on continuous scores
            // -- count, scan, scatter (any order inside a bucket), then every record counts the records of its bucket that precede
            // it by (key, position).  Position order is index order, so this is the stable order of the counting
            // passes below; they remain for keys that pile up (scores on a grid: thousands of equal keys in one bucket).
            u16* Pin = P0;
            u16* Pout = P1;
            bool bucketed = false;
            {
                u32 kmin = 0xFFFFFFFFu;
                for (u32 i = tid; i < R; i += 1024) { const u32 k = (u32)(A[i] >> 32); kmin = k < kmin ? k : kmin; }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const u32 v = (u32)__shfl_xor((int)kmin, o); kmin = v < kmin ? v : kmin; }
                if (lane == 0) s_w[wave] = kmin;
                for (int i = tid; i < 4096; i += 1024) hw[i] = 0u;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < 16; ++w) { const u32 v = s_w[w]; kmin = v < kmin ? v : kmin; }
                const u32 range = K - kmin;
                const int bsh = range < 4096u ? 0 : 32 - __builtin_clz(range) - 12;      // (key - kmin) >> bsh <= 4095
                for (u32 i = tid; i < R; i += 1024) atomicAdd(&hw[((u32)(A[i] >> 32) - kmin) >> bsh], 1u);
                __syncthreads();
                u32 c4[4], sum = 0, big = 0;
#pragma unroll
                for (int x = 0; x < 4; ++x) { c4[x] = hw[4 * tid + x]; sum += c4[x]; big |= c4[x] > RG_PILE ? 1u : 0u; }
                const bool piled = __syncthreads_or((int)big) != 0;
                if (!piled) {
                    u32 tot;
                    u32 run = block_excl_scan_1024(sum, s_w, tot);
#pragma unroll
                    for (int x = 0; x < 4; ++x) { hw[4 * tid + x] = run; run += c4[x]; }
                    __syncthreads();
                    for (u32 i = tid; i < R; i += 1024) {
                        const u32 pos = atomicAdd(&hw[((u32)(A[i] >> 32) - kmin) >> bsh], 1u);
                        P1[pos] = (u16)i;
                    }
                    __syncthreads();
                    // a record's place inside its bucket: how many of the bucket's records come before it by (key, position)
                    // -- every record counts for itself (the buckets hold a handful each; R / 1024 records per thread)
                    for (u32 a = tid; a < R; a += 1024) {
                        const u16 p = P1[a];
                        const u32 key = (u32)(A[p] >> 32);
                        const u32 bk = (key - kmin) >> bsh;
                        const u32 lo = bk ? hw[bk - 1] : 0u, hi = hw[bk];       // (after the scatter hw[b] is the END of bucket b)
                        const u64 kp = ((u64)key << 16) | p;
                        u32 before = 0;
                        for (u32 e = lo; e < hi; ++e) {
                            const u16 pq = P1[e];
                            const u64 kq = ((u64)(u32)(A[pq] >> 32) << 16) | pq;
                            before += kq < kp ? 1u : 0u;
                        }
                        P0[lo + before] = p;
                    }
                    __syncthreads();
                    Pin = P0; Pout = P1;
                    bucketed = true;
                }
            }
            // ---- 4b: stable counting passes by key byte ----
            const u32 wlo = (u32)((u64)R * wave / 16), whi = (u32)((u64)R * (wave + 1) / 16);
            const u64 below = (1ull << lane) - 1ull;
            for (int pass = 0; pass < 4 && !bucketed; ++pass) {
                const int sh = 8 * pass;
                for (int i = tid; i < 16 * 256; i += 1024) hw[i] = 0u;
                __syncthreads();
                u32* myh = hw + wave * 256;
                for (u32 i = wlo + lane; i < whi; i += 64) atomicAdd(&myh[((u32)(A[Pin[i]] >> 32) >> sh) & 0xFFu], 1u);
                __syncthreads();
                u32 dsum = 0;
                if (tid < 256) for (int w = 0; w < 16; ++w) dsum += hw[w * 256 + tid];
                // all R keys share this byte (the top R scores of a query usually share sign and exponent, often more): the
                // pass would move nothing
                if (__syncthreads_or(tid < 256 && dsum == R)) continue;
                u32 tot;
                u32 run = block_excl_scan_1024(tid < 256 ? dsum : 0u, s_w, tot);
                if (tid < 256) for (int w = 0; w < 16; ++w) { const u32 x = hw[w * 256 + tid]; hw[w * 256 + tid] = run; run += x; }
                __syncthreads();
                for (u32 base = wlo; base < whi; base += 64) {
                    const u32 i = base + lane;
                    const bool valid = i < whi;
                    const u32 p = valid ? Pin[i] : 0u;
                    const u32 d = ((u32)(A[p] >> 32) >> sh) & 0xFFu;
                    u64 peers = __ballot(valid);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool bit = (d >> k) & 1u;
                        const u64 m = __ballot(valid && bit);
                        peers &= bit ? m : ~m;
                    }
                    if (valid) {
                        const u32 rank = (u32)__popcll(peers & below), npeer = (u32)__popcll(peers);
                        const u32 start = myh[d];
                        Pout[start + rank] = (u16)p;
                        if (rank == npeer - 1) myh[d] = start + npeer;
                    }
                    wave_lds_sync();
                }
                __syncthreads();
                u16* t = Pin; Pin = Pout; Pout = t;
            }
            HG_RTK();                                         // 3: counting passes
            Pfin = Pin;
        }
        }   // !ordered
        if (!bad) {
            // ---- 5: the ranked list, and its label-match bits (metric.py:17-19; k_match's gather, one launch saved) ----
            (void)dblab; (void)qlab;
            for (u32 k = tid; k < (u32)RW * 64u; k += 1024) {
                bool m = false;
                if (k < R) {
                    const u64 rec = A[Pfin[k]];
                    if (out_idx) out_idx[(i64)q * g.R + k] = (u32)rec & 0x7FFFFFFFu;       // (null: hg_map_real -- match bits only)
                    if (scores) scores[(i64)q * g.R + k] = mono_inv(~(u32)(rec >> 32));
                    m = ((u32)rec >> 31) != 0u;               // the match bit k_real_rescore left in the record
                }
                const u64 word = __ballot(m);
                if (lane == 0) mbits[(i64)q * RW + (k >> 6)] = word;
            }
        }
    }
    if (tid == 0) { qbad[q] = bad ? 1u : 0u; if (bad) atomicOr(err, over ? 2 : 1); }
#ifdef HG_REAL_RANK_PROFILE
    HG_RTK();                                                 // 4: lists + match bits
    __syncthreads();
    if (tid == 0 && !bad) for (int k = 0; k < tkn; ++k) out_idx[(i64)q * g.R + k] = tks[k];
#endif
}

// ---- record lists beyond the LDS (R = N on a CIFAR-sized database: every row a record), two kernels instead of four radix
// passes.  k_real_group_split: a query's records are split by SCORE RANGE into groups of at most RG_CAP records (1024 coarse
// buckets of equal score width, consecutive buckets, cut where the cumulative count passes a multiple of RG_CAP - the largest bucket) and written group by group; k_real_group_sort: one
// block per (query, group) orders its group in LDS -- the bucket pass of k_real_rank_lds on 4096 fine score buckets,
// ranks by counting the bucket's predecessors, the record's own 64 bits (key, idx) being the order -- and writes it to its
// place in the sorted row.  Scores that pile up (a coarse bucket beyond RG_CAP, a fine one beyond RG_PILE, more than RG_MAXG
// groups) set bit 2 of *err: the host then runs the radix passes.
constexpr int RG_CAP = 6144;                                 // records per group: 48 KB + positions + counters, two blocks per CU
constexpr int RG_MAXG = 32;
constexpr int RG_COARSE = 1024;
#ifndef HG_RG_FLY
#define HG_RG_FLY 8
#endif
constexpr int RG_FLY = HG_RG_FLY;                            // 8-byte loads a thread of k_real_group_split has in flight (4: 0.269 ms at the CIFAR evaluation, 8: 0.258, 16: 0.257)
// score of a record key (= mono_inv(~key), spelt without a select: with the select form this compiler's instruction selection
// died in a float -> bucket computation)
__device__ __forceinline__ float rg_score(const u32 key) {
    const u32 k = ~key, m = (u32)((int)k >> 31);
    return __uint_as_float((k & m & 0x7FFFFFFFu) | (~k & ~m));
}
__device__ __forceinline__ u32 rg_bucket(const u32 key, const float smax, const float scale, const int nb) {
    // (monotone in the score whatever smax is: a score above it -- k_real_group_split places its buckets on a SAMPLE of the records --
    // lands in bucket 0, one far below in the last)
    const int bk = (int)((smax - rg_score(key)) * scale);
    return (u32)(bk < 0 ? 0 : bk < nb ? bk : nb - 1);
}
inline size_t real_group_split_lds(int S) { return ((size_t)S + 1 + RG_COARSE + RG_COARSE / 4 + 2 * (RG_MAXG + 1)) * 4; }

static __global__ __launch_bounds__(1024) void k_real_group_split(const u64* __restrict__ cand, i64 crow_in, u32 cap, const u32* __restrict__ sl_cnt,
                                                           const u32* __restrict__ fail, u32* __restrict__ tot, u64* __restrict__ grouped,
                                                           u32* __restrict__ gtab, i64 crow_out, int* __restrict__ err, const int maxg, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 gsl[];
    u32* off = gsl;                                          // [S + 1]
    u32* ch = off + g.S + 1;                                 // [RG_COARSE] counts
    u8* gmap = (u8*)(ch + RG_COARSE);                        // [RG_COARSE] bucket -> group
    u32* gstart = ch + RG_COARSE + RG_COARSE / 4;            // [RG_MAXG + 1]
    u32* gcur = gstart + RG_MAXG + 1;                        // [RG_MAXG + 1] scatter cursors
    __shared__ u32 s_w[16], s_x[16];
    __shared__ int s_ng;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32* __restrict__ gt = gtab + (i64)q * (RG_MAXG + 1);
    if (fail[q]) { if (tid == 0) tot[q] = 0xFFFFFFFFu; if (tid <= RG_MAXG) gt[tid] = 0u; return; }
    u32 n = 0;
    for (int sb = 0; sb < g.S; sb += 1024) {
        const int s = sb + tid;
        const u32 c = s < g.S ? sl_cnt[(i64)s * g.Qpad + q] : 0u;
        u32 t;
        const u32 ex = block_excl_scan_1024(c, s_w, t);
        if (s < g.S) off[s] = n + ex;
        n += t;
    }
    if (tid == 0) off[g.S] = n;
    if (tid < RG_COARSE) ch[tid] = 0u;
    __syncthreads();
    const u64* __restrict__ row = cand + (i64)q * crow_in;
    // Every row a record: the slices are whole segments, each full but the last -- the query's records are ONE dense run
    // and the block walks it 1024 records a step.  (Any other layout: slice by slice.)
    bool dense_ok = true;
    for (int s = tid; s < g.S; s += 1024) dense_ok &= off[s] == (u32)s * cap;
    const bool dense = __syncthreads_and((int)dense_ok) != 0;
    auto for_records = [&](const u32 every, auto&& body) {    // every: 1, or > 1 to visit 64 consecutive records (512 bytes) in every 64 * every (dense runs only)
        if (dense) {
            // four loads in flight per thread: one at a time, a walk was 53 dependent trips to memory at C1 (the body's LDS
            // atomic keeps the compiler from overlapping them itself)
            const u32 ne = every > 1u ? n / every : n;       // compact index a -> record (a / 64) * 64 * every + a % 64
            for (u32 a = tid; a < ne; a += 1024u * RG_FLY) {
                u64 r[RG_FLY];
#pragma unroll
                for (int k = 0; k < RG_FLY; ++k) {
                    const u32 ak = a + 1024u * k;
                    const u32 at = every > 1u ? (ak >> 6) * (64u * every) + (ak & 63u) : ak;
                    r[k] = ak < ne && at < n ? row[at] : 0ull;
                }
#pragma unroll
                for (int k = 0; k < RG_FLY; ++k) {
                    const u32 ak = a + 1024u * k;
                    const u32 at = every > 1u ? (ak >> 6) * (64u * every) + (ak & 63u) : ak;
                    if (ak < ne && at < n) body(r[k]);
                }
            }
        } else {
            for (int s = 0; s < g.S; ++s) {
                const u32 c = off[s + 1] - off[s];
                const u64* __restrict__ sl = row + (i64)s * cap;
                for (u32 i = tid; i < c; i += 1024) body(sl[i]);
            }
        }
    };
    // Where the buckets lie: the extremes of 64 consecutive records in every 512 are enough (a long dense run -- every row a record, in
    // index order: an eighth of the rows from all over the database, whole 512-byte pieces of the run); a score beyond them joins the first or the last
    // bucket (rg_bucket), the order of the buckets is the order of the scores either way, and a bucket that fills beyond what a group
    // holds is caught below as it always was.  One pass over 1/8 of the records instead of one over all (three passes -> 2 1/8).
    const u32 sample_every = dense && n >= 4u * 4096u ? 8u : 1u;
    u32 kmin = 0xFFFFFFFFu, kmax = 0u;
    for_records(sample_every, [&](const u64 rec) { const u32 k = (u32)(rec >> 32); kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; });
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u32 v = (u32)__shfl_xor((int)kmin, o), w = (u32)__shfl_xor((int)kmax, o);
        kmin = v < kmin ? v : kmin; kmax = w > kmax ? w : kmax;
    }
    if (lane == 0) { s_w[wave] = kmin; s_x[wave] = kmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) { const u32 v = s_w[w], x = s_x[w]; kmin = v < kmin ? v : kmin; kmax = x > kmax ? x : kmax; }
    const float smax = rg_score(kmin), width = smax - rg_score(kmax);
    const bool flat = !(width > 0.0f) || !(width < 3.0e38f);                    // one score for all rows, or not finite
    const float scale = (float)RG_COARSE / width;
    for_records(1u, [&](const u64 rec) { atomicAdd(&ch[rg_bucket((u32)(rec >> 32), smax, scale, RG_COARSE)], 1u); });
    __syncthreads();
    {   // consecutive coarse buckets -> groups of at most RG_CAP records, every bucket for itself: bucket b, whose records
        // start at cumulative count ex_b, goes to group ex_b / T with T = RG_CAP - (largest bucket) -- a group then spans less
        // than T + the largest bucket records, and with buckets of at most T no group number is skipped.  (A single thread
        // packing the buckets greedily held the other 1023 at a barrier for a third of the block's life.)
        const u32 cb = ch[tid];                              // (RG_COARSE == blockDim.x)
        u32 mx = cb;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const u32 v = (u32)__shfl_xor((int)mx, o); mx = v > mx ? v : mx; }
        if (lane == 0) s_x[wave] = mx;
        u32 t;
        const u32 ex = block_excl_scan_1024(cb, s_w, t);    // (its barriers publish s_x too)
#pragma unroll
        for (int w = 0; w < 16; ++w) { const u32 v = s_x[w]; mx = v > mx ? v : mx; }
        const bool bad = flat || n == 0 || mx > (u32)RG_CAP / 2u;
        const u32 T = bad ? 1u : (u32)RG_CAP - mx;
        const u32 gid = ex / T;
        u32 lastg = cb ? gid : 0u;                           // the last group that holds a record
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const u32 v = (u32)__shfl_xor((int)lastg, o); lastg = v > lastg ? v : lastg; }
        __syncthreads();                                     // (s_x is read above by everybody)
        if (lane == 0) s_x[wave] = lastg;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 16; ++w) { const u32 v = s_x[w]; lastg = v > lastg ? v : lastg; }
        const int ngroups = bad ? -1 : (int)lastg + 1;
        if (!bad && ngroups <= maxg) {
            gmap[tid] = (u8)gid;
            const u32 gprev = tid ? (ex - ch[tid - 1]) / T : 0xFFFFFFFFu;
            if (gid != gprev) gstart[gid] = ex;              // the group's first bucket
            if (tid == 0) gstart[ngroups] = n;
        }
        if (tid == 0) s_ng = bad || ngroups > maxg ? -1 : ngroups;
    }
    __syncthreads();
    const int ng = s_ng;
    if (ng < 0) { if (tid == 0) atomicOr(err, 4); if (tid <= RG_MAXG) gt[tid] = 0u; return; }
    if (tid <= RG_MAXG) { const u32 v = gstart[tid < ng ? tid : ng]; gcur[tid] = v; gt[tid] = v; }
    __syncthreads();
    u64* __restrict__ grp = grouped + (i64)q * crow_out;
    for_records(1u, [&](const u64 rec) { grp[atomicAdd(&gcur[gmap[rg_bucket((u32)(rec >> 32), smax, scale, RG_COARSE)]], 1u)] = rec; });
    if (tid == 0) tot[q] = n;
}

constexpr int RG_BMW = RG_CAP / 32 + 2;                     // dwords of match bits a group's ranks can touch
constexpr size_t real_group_sort_lds() { return (size_t)RG_CAP * 8 + (size_t)RG_CAP * 2 + 4096 * 4 + (size_t)RG_BMW * 4; }

// The ranked list leaves from here too (what k_real_finish and k_match do after the radix passes): a record's final rank is
// known the moment it has counted its bucket's predecessors -- idx and score go to the ranked lists, the label-match bit
// (metric.py:17-19) into the group's stretch of the query's bitmap (LDS, then OR-ed into the zeroed global words: the first
// and last word of a stretch are shared with the neighbouring groups).
struct GroupOut {
    u32* out_idx; float* scores; u32* mbits32; const u64* dblab; const u64* qlab; i64 RW; i64 R; int LW; u32 idx_base;
};

static __global__ __launch_bounds__(1024) void k_real_group_sort(const u64* __restrict__ grouped, const u32* __restrict__ gtab, const GroupOut o,
                                                          i64 crow, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) u64 gso[];
    u64* A = gso;                                            // [RG_CAP] the group's records
    u16* P = (u16*)(A + RG_CAP);                             // [RG_CAP] positions, grouped by fine bucket
    u32* hw = (u32*)(P + RG_CAP);                            // [4096] fine bucket counts -> starts -> ends
    u32* bm = hw + 4096;                                     // [RG_BMW] match bits of ranks lo .. hi, from dword lo / 32
    __shared__ u32 s_w[16], s_x[16];
    const int q = blockIdx.x, gi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 lo = gtab[(i64)q * (RG_MAXG + 1) + gi], hi = gtab[(i64)q * (RG_MAXG + 1) + gi + 1];
    const u32 m = hi - lo;
    if (m == 0) return;
    const u64* __restrict__ src = grouped + (i64)q * crow + lo;
    // A thread keeps its (at most RG_PER = 6) records in registers and asks for their label words the moment it has them: six gathers in
    // flight under the bucket passes.  (They used to be asked for one by one at the very end, each behind two dependent LDS reads: six
    // trips to memory in a row per thread, over half a block's life.)  Each record then finds its OWN rank -- its bucket's start + the
    // bucket's records before it -- and for hg_map_real only the records that match need one.
    constexpr int RG_PER = RG_CAP / 1024;
    static_assert(RG_CAP % 1024 == 0, "a whole number of records per thread");
    const u64* __restrict__ ql = o.qlab + (i64)q * o.LW;
    u64 myrec[RG_PER];
    u32 anym = 0;                                            // bit k: the thread's record k matches the query's labels (metric.py:17-19)
    u32 kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < RG_PER; ++k) {
        const u32 i = tid + 1024u * k;
        myrec[k] = i < m ? src[i] : 0xFFFFFFFFFFFFFFFFull;
    }
#pragma unroll
    for (int k = 0; k < RG_PER; ++k) {
        const u32 i = tid + 1024u * k;
        if (i < m) {
            A[i] = myrec[k];
            const u32 key = (u32)(myrec[k] >> 32);
            kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax;
            const u64* __restrict__ dl = o.dblab + (i64)((u32)myrec[k] - o.idx_base) * o.LW;
            u64 any = 0;
            for (int w = 0; w < o.LW; ++w) any |= dl[w] & ql[w];
            anym |= any ? 1u << k : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u32 v = (u32)__shfl_xor((int)kmin, o), w = (u32)__shfl_xor((int)kmax, o);
        kmin = v < kmin ? v : kmin; kmax = w > kmax ? w : kmax;
    }
    if (lane == 0) { s_w[wave] = kmin; s_x[wave] = kmax; }
    for (int i = tid; i < 4096 + RG_BMW; i += 1024) hw[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) { const u32 v = s_w[w], x = s_x[w]; kmin = v < kmin ? v : kmin; kmax = x > kmax ? x : kmax; }
    const float smax = rg_score(kmin), width = smax - rg_score(kmax);
    // (one score for the whole group: every record lands in bucket 0 and the pile check below decides)
    const float scale = width > 0.0f ? 4096.0f / width : 0.0f;
#pragma unroll
    for (int k = 0; k < RG_PER; ++k)
        if (tid + 1024u * k < m) atomicAdd(&hw[rg_bucket((u32)(myrec[k] >> 32), smax, scale, 4096)], 1u);
    __syncthreads();
    u32 c4[4], sum = 0, big = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x) { c4[x] = hw[4 * tid + x]; sum += c4[x]; big |= c4[x] > RG_PILE ? 1u : 0u; }
    if (__syncthreads_or((int)big)) { if (tid == 0) atomicOr(err, 4); return; }
    u32 t;
    u32 run = block_excl_scan_1024(sum, s_w, t);
#pragma unroll
    for (int x = 0; x < 4; ++x) { hw[4 * tid + x] = run; run += c4[x]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RG_PER; ++k) {
        const u32 i = tid + 1024u * k;
        // (a position takes 13 bits: bit 15 carries the record's match flag to whoever meets it by position)
        if (i < m) P[atomicAdd(&hw[rg_bucket((u32)(myrec[k] >> 32), smax, scale, 4096)], 1u)] = (u16)(i | (((anym >> k) & 1u) << 15));
    }
    __syncthreads();
    const u32 w0 = lo >> 5;
    if (o.out_idx == nullptr && o.scores == nullptr) {
        // hg_map_real: match bits only -- a thread ranks its own records, and only those that match
#pragma unroll
        for (int k = 0; k < RG_PER; ++k) {
            const u32 i = tid + 1024u * k;
            if (i < m && ((anym >> k) & 1u)) {
                const u64 rec = myrec[k];
                const u32 bk = rg_bucket((u32)(rec >> 32), smax, scale, 4096);
                const u32 b0 = bk ? hw[bk - 1] : 0u, b1 = hw[bk];    // (after the scatter hw[b] is the END of bucket b)
                u32 before = 0;
                for (u32 e = b0; e < b1; ++e) before += A[P[e] & 0x7FFFu] < rec ? 1u : 0u;
                const u32 rk = lo + b0 + before;                     // the record's rank in the query's list
                if ((i64)rk < o.R) atomicOr(&bm[(rk >> 5) - w0], 1u << (rk & 31));
            }
        }
    } else {
        // the ranked lists too: by position, so that a wavefront's ranks -- and its stores -- lie side by side
        for (u32 a = tid; a < m; a += 1024) {
            const u32 pa = P[a];
            const u64 rec = A[pa & 0x7FFFu];
            const u32 bk = rg_bucket((u32)(rec >> 32), smax, scale, 4096);
            const u32 b0 = bk ? hw[bk - 1] : 0u, b1 = hw[bk];
            u32 before = 0;
            for (u32 e = b0; e < b1; ++e) before += A[P[e] & 0x7FFFu] < rec ? 1u : 0u;
            const u32 rk = lo + b0 + before;
            if ((i64)rk < o.R) {
                if (o.out_idx) o.out_idx[(i64)q * o.R + rk] = (u32)rec;
                if (o.scores) o.scores[(i64)q * o.R + rk] = rg_score((u32)(rec >> 32));
                if (pa >> 15) atomicOr(&bm[(rk >> 5) - w0], 1u << (rk & 31));
            }
        }
    }
    __syncthreads();
    if (tid < RG_BMW) {
        const u32 v = bm[tid];
        if (v) atomicOr(&o.mbits32[(i64)q * 2 * o.RW + w0 + tid], v);
    }
}

// The guess with the sample in LDS (k_real_guess reads its M samples three times from global memory with 256 threads):
// 1024 threads copy the query's samples (as order-preserving keys) once, then the same 11 + 11 + 10 bit radix select of
// the rank_s-th largest runs out of LDS.  M <= RG_MMAX.
constexpr int RG_MMAX = 16384;
template <bool IN16>            // IN16: the samples are bfloat16 (k_real_sample_h<.., OUT16>): the float32 with those upper 16 bits
static __global__ __launch_bounds__(1024) void k_real_guess_lds(const float* __restrict__ samp, i64 M, i64 mstride, u32 rank_s,
                                                         float* __restrict__ thr) {
    __shared__ u32 keys[RG_MMAX];
    __shared__ u32 hist[2048];
    __shared__ u32 s_w[16];
    __shared__ u32 s_prefix, s_rank;
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ col = samp + (i64)q * mstride;
    if (rank_s > (u32)M) {
        if (tid == 0) thr[q] = __uint_as_float(0xFF800000u);  // -inf: everything qualifies
        return;
    }
    if (IN16) {
        const u16* __restrict__ col16 = (const u16*)samp + (i64)q * mstride;
        for (i64 i = tid; i < M; i += 1024) keys[i] = mono_key(__uint_as_float((u32)col16[i] << 16));
    } else {
        for (i64 i = tid; i < M; i += 1024) keys[i] = mono_key(col[i]);
    }
    if (tid == 0) { s_prefix = 0; s_rank = rank_s; }
    u32 mask = 0;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        hist[tid] = 0u; hist[tid + 1024] = 0u;
        __syncthreads();
        const u32 prefix = s_prefix, need = s_rank, bins = 1u << widths[pass];
        for (u32 i = tid; i < (u32)M; i += 1024) {
            const u32 k = keys[i];
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (bins - 1u)], 1u);
        }
        __syncthreads();
        // digits from the largest down: thread t speaks for digits bins - 1 - 2 t and bins - 2 - 2 t
        const u32 d0 = bins - 1u - 2u * tid, d1 = d0 - 1u;
        const u32 c0 = 2u * tid < bins ? hist[d0] : 0u, c1 = 2u * tid + 1u < bins ? hist[d1] : 0u;
        u32 tot;
        const u32 ex = block_excl_scan_1024(c0 + c1, s_w, tot);
        if (ex < need && need <= ex + c0) { s_prefix = prefix | (d0 << shifts[pass]); s_rank = need - ex; }
        else if (ex + c0 < need && need <= ex + c0 + c1) { s_prefix = prefix | (d1 << shifts[pass]); s_rank = need - ex - c0; }
        __syncthreads();
        mask |= (bins - 1u) << shifts[pass];
    }
    if (tid == 0) {
        const u32 k = s_prefix;                               // the key of the rank_s-th largest sample
        thr[q] = k ? mono_inv(k - 1u) : __uint_as_float(0xFF800000u);
    }
}

}  // namespace hg
