// hashgan_amd -- the distance histogram (metric.py:13 reduced to counts) with NO vector instruction per pair.
//
// k_hist_mx turns an fp4 MFMA tile into 32 x 32 exact distances as floats and still pays a v_cvt_u32_f32 and an address
// op per pair before the LDS add.  The integer matrix instruction can deliver the ADDRESS itself: with database bits as
// A = 8 x (x in {0,1}), query bits as B = 16 (1 - 2 q) (zero beyond the code) and
//     C = 128 pop(q) + 4 j + (byte offset of the wavefront's columns in LDS),
// v_mfma_i32_32x32x32_i8 leaves   acc = 128 dist + 4 j + base   -- the byte address of counter [dist][j] of the lane's
// query column -- and the pair costs one fire-and-forget ds_add_u32 whose address operand IS the accumulator register.
// What is left per pair is the LDS atomic (the bound of every histogram here: 12-14 lanes per clock and CU; giving the
// two lane-halves of a query columns of their own does not help -- the LDS it takes costs more in occupancy).
// One MFMA covers 32 code bits, so a tile takes NW of them (fp4: one per 64 bits) -- the matrix pipe has the room.
//
// Mapping, geometry and output: k_hist_mx's (block = segment pair x 256 queries, lane = (query j, lane-half h <-> segment
// 2 sp + h), both halves add into one column, histograms per segment PAIR, the two query tiles of a lane share a dword
// as 16-bit halves while a pair has fewer than 65536 visited rows).
#pragma once
#include "hg_kernels.hpp"
#include "hg_hist_mx.hpp"

namespace hg {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) u32 lds_u32;

// fire-and-forget LDS add whose address operand is a register holding the 32-bit LDS byte address (device pass only: on the
// host an LDS pointer has no 32-bit form and the cast would only earn a warning per instantiation)
__device__ __forceinline__ void lds_add_at(const u32 addr, const u32 inc) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add((lds_u32*)addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    (void)addr; (void)inc;
#endif
}

__device__ __forceinline__ uint4 bits16_to_bytes(const u32 bits, const u32 one, const u32 zero) {   // bit i -> byte i: `one` or `zero`
    u32 w[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        u32 v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v |= (((bits >> (4 * d + i)) & 1u) ? one : zero) << (8 * i);
        w[d] = v;
    }
    return uint4{w[0], w[1], w[2], w[3]};
}

// Database image in A-fragment order: groups of 16 rows; chunk (group G, word m, k-half hh, row r) = 16 bytes at
// (((G * NW + m) * 2 + hh) * 16 + r) * 16: byte i = 8 x bit (32 m + 16 hh + i) of row 16 G + r.  (A and B fragments of
// the instruction map a lane's 16 bytes to the same 16 values of k, so any fixed bit -> byte assignment works as long
// as the queries use the same one.)
static __global__ __launch_bounds__(256) void k_expand_db_i8(const u32* __restrict__ db, uint4* __restrict__ img, i64 N, i64 n16, int NW) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16 * NW * 2) return;
    const i64 row = i / (NW * 2);
    const int c = (int)(i - row * NW * 2), m = c >> 1, hh = c & 1;
    const u32 word = row < N ? db[row * NW + m] : 0u;
    img[(((row >> 4) * NW + m) * 2 + hh) * 16 + (row & 15)] = bits16_to_bytes((word >> (16 * hh)) & 0xFFFFu, 8u, 0u);
}

constexpr int hist_i8_cols(bool pack16) { return pack16 ? 1 : 2; }     // [NB][32] dword columns per wavefront

template <int NW, bool PACK16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_hist_i8(const u32* __restrict__ qc, const u8* __restrict__ dbx8, u32* __restrict__ hist, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 hlds[];
    constexpr int QT = 2, WQ = 32 * QT;
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQB = g.nQT;
    const int sp = lb / nQB, qb = lb - sp * nQB;
    const int h = lane >> 5, j = lane & 31;
    const int NB = g.NB, b = g.NB - 1;
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 ntile = ((len0 > len1 ? len0 : len1) + 15) / 16;
    const i64 NG = (g.N + 15) >> 4;
    const int stride = g.hist_stride;

    // this wavefront's columns: [NB][32] dwords (PACK16: tile t in bits 16 t ..) or [QT][NB][32]
    constexpr int NCOL = hist_i8_cols(PACK16);
    u32* col = hlds + wave * NCOL * NB * 32;
    for (int i = lane; i < NCOL * NB * 32; i += 64) col[i] = 0u;

    const int q0w = (qb * WPB + wave) * WQ;
    i32x4 bq[QT][NW];
    i32x16 biasv[QT];
    u32 inc[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        int pop = 0;
#pragma unroll
        for (int m = 0; m < NW; ++m) {
            const u32 word = q < g.Q ? qc[(i64)q * NW + m] : 0u;
            pop += __builtin_popcount(word);
            // bits beyond the code contribute nothing: B = 0 there (their A is 0 as well)
            const int first = 32 * m + 16 * h;
            const u32 valid = b - first >= 16 ? 0xFFFFu : (b - first <= 0 ? 0u : (1u << (b - first)) - 1u);
            const u32 bits = (word >> (16 * h)) & 0xFFFFu;
            const uint4 neg = bits16_to_bytes(bits & valid, 0xF0u, 0u);           // q = 1: -16
            const uint4 pos = bits16_to_bytes(~bits & valid, 0x10u, 0u);          // q = 0: +16
            bq[t][m] = i32x4{(int)(neg.x | pos.x), (int)(neg.y | pos.y), (int)(neg.z | pos.z), (int)(neg.w | pos.w)};
        }
        u32* c0 = col + (PACK16 ? 0 : t * NB * 32) + j;
        const int base = (int)(u32)(size_t)(lds_u32*)c0 + 128 * pop;            // byte address of counter [pop][j]
#pragma unroll
        for (int r = 0; r < 16; ++r) biasv[t][r] = base;
        inc[t] = PACK16 ? (t ? 65536u : 1u) : 1u;
    }
    wave_lds_sync();

    const int ah = (j >> 2) & 1;                                     // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                           // its row inside that half's 16
    const i64 ag0 = (ah ? lo1 : lo0) >> 4;
    auto chunk = [&](const i64 T, const int m) -> i32x4 {
        i64 G = ag0 + T;
        G = G < NG ? G : NG - 1;                                     // past the end: any valid group (masked below)
        return *(const i32x4*)(dbx8 + ((((G * NW + m) * 2 + h) * 16 + ar) * 16));
    };
    i32x4 av[NW];
    if (ntile > 0) {
#pragma unroll
        for (int m = 0; m < NW; ++m) av[m] = chunk(0, m);
    }
    for (i64 T = 0; T < ntile; T += stride) {
        i32x4 an[NW];
        const i64 Tn = T + stride < ntile ? T + stride : T;
#pragma unroll
        for (int m = 0; m < NW; ++m) an[m] = chunk(Tn, m);
        const i64 left = mylen - T * 16;                             // valid rows of this lane in the tile
        const bool whole = __all(left >= 16);                        // (wave-uniform)
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            i32x16 acc = biasv[t];
#pragma unroll
            for (int m = 0; m < NW; ++m) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[m], bq[t][m], acc, 0, 0, 0);
            if (whole) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    lds_add_at((u32)acc[r], inc[t]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((i64)r < left) lds_add_at((u32)acc[r], inc[t]);
            }
        }
#pragma unroll
        for (int m = 0; m < NW; ++m) av[m] = an[m];
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
