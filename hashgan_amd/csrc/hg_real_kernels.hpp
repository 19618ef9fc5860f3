// Real-valued feature ranking: what main.py actually feeds the
// metric -- float32 tanh outputs -- ranked by inner product (lib/metric.py:13-14), then the same
// label match and AP as the binary path.  Same decomposition as k_select: lane <-> query (its
// features in VGPRs), database rows wave-uniform through scalar loads, hit masks + drain.
//
// Arithmetic (fixed, restated exactly by oracle/real_map.py):
//   ip = (one float32 fma chain from +0.0 over k ascending) + 0.0
// -- the order of the matrix cores' float32 MFMA (hg_real_mx.hpp).  Ranking: ip descending, database index ascending.
// Sortable record: (~mono(ip) << 32) | idx, mono() = the usual order-preserving map of float
// bits to unsigned; ascending records = descending ip, ascending index.
#pragma once
#include "hg_kernels.hpp"

namespace hg {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 mono_key(float x) {           // order-preserving float -> uint
    const u32 b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float mono_inv(u32 k) {
    const u32 b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

template <int BP>   // BP = padded feature count / 2 (a multiple of 8)
__device__ __forceinline__ float ip_row(const f2 (&q)[BP], const f2* __restrict__ row) {
    // ONE float32 fma chain from +0.0 over k ascending: bitwise what the chained v_mfma_f32_32x32x2_f32 of
    // k_real_select_mx computes (hg_real_mx.hpp), so every kernel of this path yields the same scores
    float acc = 0.0f;
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        acc = __builtin_fmaf(q[p].x, row[p].x, acc);
        acc = __builtin_fmaf(q[p].y, row[p].y, acc);
    }
    return acc + 0.0f;
}

// ----------------------------------------------------------------------------
// R1  sample: inner products of every query with every stride-th database row, written
// samp[q][j] (j = sample number).  The wave stages 64 rows x 64 queries in LDS and writes
// the tile transposed, so each query's samples leave as 256-byte runs.
// ----------------------------------------------------------------------------
template <int BP>
__global__ __launch_bounds__(256) void k_real_sample(const float* __restrict__ qf, const float* __restrict__ dbf,
                                                     float* __restrict__ samp, i64 M, i64 stride, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 unit = (i64)blockIdx.x * WPB + wave;            // (sample tile of 64, query tile)
    const i64 nTiles = (M + 63) / 64;
    if (unit >= nTiles * g.nQT) return;
    const i64 tile = unit / g.nQT;
    const int qt = (int)(unit - tile * g.nQT);
    const int q = qt * 64 + lane;
    f2 qv[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) qv[p] = q < g.Q ? ((const f2*)(qf + (i64)q * 2 * BP))[p] : f2{0.0f, 0.0f};
    float* t = (float*)lds + wave * 64 * 65;                  // [64][65] padded tile
    const i64 j0 = tile * 64;
    const int nj = (int)(M - j0 < 64 ? M - j0 : 64);
    for (int j = 0; j < nj; ++j) {                            // wave-uniform rows
        const f2* __restrict__ row = (const f2*)(dbf + (j0 + j) * stride * 2 * BP);
        t[j * 65 + lane] = ip_row<BP>(qv, row);
    }
    wave_lds_sync();
    for (int qq = 0; qq < 64; ++qq) {                         // transposed write: lane <-> sample
        const int qo = qt * 64 + qq;
        if (qo < g.Q && lane < nj) samp[(i64)qo * M + j0 + lane] = t[lane * 65 + qq];
    }
}

// ----------------------------------------------------------------------------
// R2  guess: the rank_s-th largest of a query's M sample values (radix select on the
// order-preserving keys, 11 + 11 + 10 bits, LDS histograms), one block per query.  The select
// pass keeps ip > thr[q], thr = the next float below that value (so ip >= it qualifies).
// rank_s > M: everything qualifies (thr = -inf).
// ----------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_real_guess(const float* __restrict__ samp, i64 M, i64 mstride, u32 rank_s,
                                                    float* __restrict__ thr) {
    __shared__ u32 hist[2048];
    __shared__ u32 s_prefix, s_rank, s_wsum[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ col = samp + (i64)q * mstride;    // M samples, rows mstride apart
    if (rank_s > (u32)M) {
        if (tid == 0) thr[q] = __uint_as_float(0xFF800000u);  // -inf
        return;
    }
    if (tid == 0) { s_prefix = 0; s_rank = rank_s; }
    u32 mask = 0;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = tid; i < 2048; i += 256) hist[i] = 0;
        __syncthreads();
        const u32 prefix = s_prefix, bins = 1u << widths[pass];
        for (i64 i = tid; i < M; i += 256) {
            const u32 k = mono_key(col[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (bins - 1)], 1u);
        }
        __syncthreads();
        {   // the digit that holds the rank, counted from the largest digit down: 8 bins per thread, block scan of the sums
            const u32 need = s_rank;
            u32 c[8], sum = 0;
#pragma unroll
            for (int x = 0; x < 8; ++x) { const u32 d = bins - 1u - (8u * tid + x); c[x] = 8u * tid + x < bins ? hist[d] : 0u; sum += c[x]; }
            u32 inc = sum;
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
            if (lane == 63) s_wsum[wave] = inc;
            __syncthreads();
            u32 ex = inc - sum;
#pragma unroll
            for (int w = 0; w < 4; ++w) ex += w < wave ? s_wsum[w] : 0u;
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                if (ex < need && need <= ex + c[x]) { s_rank = need - ex; s_prefix = prefix | ((bins - 1u - (8u * tid + x)) << shifts[pass]); }
                ex += c[x];
            }
        }
        __syncthreads();
        mask |= ((1u << widths[pass]) - 1u) << shifts[pass];
    }
    if (tid == 0) {
        const float v = mono_inv(s_prefix);                   // the rank_s-th largest sample
        const u32 k = mono_key(v);
        thr[q] = k ? mono_inv(k - 1u) : __uint_as_float(0xFF800000u);
    }
}

// ----------------------------------------------------------------------------
// R3  select: every (query, row) with ip > thr[q] becomes a sortable record in the lane's
// slice of the query's record row, index order.  Hot loop per pair: BP packed fmas, one add
// (+ 0.0), thr - ip, v_alignbit -- the sign of (thr - ip) is the hit bit; drain as in k_select.
// ----------------------------------------------------------------------------
struct RealSelArgs {
    const float* thr;      // [Q]
    u32* sl_cnt;           // [S][Qpad]
    u32* fail;             // [Qpad]
    u32 cap;               // slice capacity
    i64 crow;              // record-row stride
};

// QPL queries per lane: the row stream through the scalar cache (2*BP*4 bytes per row per wave)
// is the bottleneck of this kernel (measured: ~3 B/clk/CU), so for short feature vectors every
// lane carries two queries and each scalar-loaded row is used for 128 pairs instead of 64.
template <int BP, int QPL>
__global__ __launch_bounds__(256) void k_real_select(const float* __restrict__ qf, const float* __restrict__ dbf,
                                                     const RealSelArgs a, u64* __restrict__ cand, const Geo g) {
    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 unit = (i64)lb * g.wpb + wave;
    if (unit >= g.nUnits) return;
    const int s = (int)(unit / g.nQT);                        // g.nQT counts tiles of 64*QPL queries here
    const int qt = (int)(unit - (i64)s * g.nQT);
    int q[QPL];
    bool live[QPL];
    f2 qv[QPL][BP];
    float thr[QPL];
    u64* wp[QPL];
    u64* wp0[QPL];
    u32 room[QPL], dropped[QPL];
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        q[u] = (qt * QPL + u) * 64 + lane;
        live[u] = q[u] < g.Q;
#pragma unroll
        for (int p = 0; p < BP; ++p) qv[u][p] = live[u] ? ((const f2*)(qf + (i64)q[u] * 2 * BP))[p] : f2{0.0f, 0.0f};
        thr[u] = live[u] ? a.thr[q[u]] : __uint_as_float(0x7F800000u);   // +inf: nothing qualifies
        wp[u] = cand + (i64)(live[u] ? q[u] : 0) * a.crow + (i64)s * a.cap;
        wp0[u] = wp[u];
        room[u] = a.cap;
        dropped[u] = 0;
    }
    const i64 lo = (i64)s * g.L;
    const i64 hi = lo + g.L < g.N ? lo + g.L : g.N;

    for (i64 n0 = lo; n0 < hi; n0 += 64) {
        const int cnt = (int)(hi - n0 < 64 ? hi - n0 : 64);
        u64 hm[QPL];
#pragma unroll
        for (int u = 0; u < QPL; ++u) hm[u] = 0;
        for (int j = 0; j < cnt; ++j) {                       // wave-uniform rows: scalar loads
            const f2* __restrict__ row = (const f2*)(dbf + (n0 + j) * 2 * BP);
#pragma unroll
            for (int u = 0; u < QPL; ++u) {
                const float ip = ip_row<BP>(qv[u], row);
                hm[u] = (hm[u] << 1) | (u64)(__float_as_uint(thr[u] - ip) >> 31);
            }
        }
#pragma unroll
        for (int u = 0; u < QPL; ++u) {                       // drain: bit cnt-1-j <-> row n0+j
            u64 m = hm[u];
            while (__any(m != 0ull)) {
                if (m != 0ull) {
                    const int k = 63 - __clzll((long long)m);
                    m ^= 1ull << k;
                    const i64 nr = n0 + (cnt - 1 - k);
                    const float ip = ip_row<BP>(qv[u], (const f2*)(dbf + nr * 2 * BP));   // per-lane re-read (L2)
                    if (room[u]) {
                        *wp[u] = ((u64)(~mono_key(ip)) << 32) | (u64)(g.idx_base + (u32)nr);
                        ++wp[u];
                        --room[u];
                    } else {
                        ++dropped[u];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        if (q[u] < g.Qpad) a.sl_cnt[(i64)s * g.Qpad + q[u]] = (u32)(wp[u] - wp0[u]);
        if (dropped[u] && live[u]) a.fail[q[u]] = 1u;
    }
}

// ----------------------------------------------------------------------------
// R4  one pass of a stable LSD radix sort over every query's records (8-bit digit of the key
// half), built on the same machinery as k_rank_fused: per-wave digit histograms, prefix, then a
// stable 64-wide placement (bit-sliced ballot match).  Four passes (shift 32, 40, 48, 56) sort
// the records by key; the index half never moves relative to equal keys because the records
// start in index order.  Pass 0 reads the slices k_real_select wrote and compacts them.
// ----------------------------------------------------------------------------
struct RadixArgs {
    const u32* sl_cnt;     // pass 0: [S][Qpad] slice counts
    const u32* fail;       // pass 0: overflow flags
    u32* tot;              // [Qpad] records per query (written by pass 0, read by later passes)
    u32 cap;
    i64 crow_in, crow_out;
    int first;             // pass 0
    int shift;
};

template <int NWAV>
__global__ __launch_bounds__(NWAV * 64) void k_radix_pass(const u64* __restrict__ in, u64* __restrict__ out,
                                                          const RadixArgs a, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    constexpr int NBIN = 256;
    constexpr int nthr = NWAV * 64;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32* hw = lds;                       // [NWAV][256]
    u32* tot = hw + NWAV * NBIN;         // [256]
    if (a.first && a.fail[q]) { if (tid == 0) a.tot[q] = 0xFFFFFFFFu; return; }   // lost: marked for the host
    for (int i = tid; i < (NWAV + 1) * NBIN; i += nthr) lds[i] = 0u;
    __syncthreads();
    const u64* __restrict__ row = in + (i64)q * a.crow_in;
    const u32 dense_tot = a.first ? 0u : a.tot[q];
    if (!a.first && dense_tot == 0xFFFFFFFFu) return;
    const int nsl = a.first ? g.S : (int)((dense_tot + 255) / 256);
    const u32 cap = a.first ? a.cap : 256u;
    const int s0 = (int)((i64)nsl * wave / NWAV), s1 = (int)((i64)nsl * (wave + 1) / NWAV);
    auto slice_cnt = [&](int s) -> u32 {
        if (s >= s1) return 0u;
        if (a.first) return a.sl_cnt[(i64)s * g.Qpad + q];
        const u32 left = dense_tot - (u32)s * 256u;
        return left < 256u ? left : 256u;
    };
    // phase 1: digit histogram of this wave's range
    u32* myh = hw + wave * NBIN;
    for (int s = s0; s < s1; ++s) {
        const u32 cnt = slice_cnt(s);
        for (u32 i = lane; i < cnt; i += 64) atomicAdd(&myh[(u32)(row[(i64)s * cap + i] >> a.shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (tid < NBIN) {
        u32 acc = 0;
        for (int w = 0; w < NWAV; ++w) acc += hw[w * NBIN + tid];
        tot[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        u32 cum = 0;
        for (int d = 0; d < NBIN; ++d) { const u32 c = tot[d]; tot[d] = cum; cum += c; }
        if (a.first) a.tot[q] = cum;
    }
    __syncthreads();
    if (tid < NBIN) {
        u32 run = tot[tid];
        for (int w = 0; w < NWAV; ++w) { const u32 h = hw[w * NBIN + tid]; hw[w * NBIN + tid] = run; run += h; }
    }
    __syncthreads();
    // phase 3: stable placement
    u32* pb = hw + wave * NBIN;
    u64* __restrict__ orow = out + (i64)q * a.crow_out;
    const u64 below = (1ull << lane) - 1ull;
    for (int s = s0; s < s1; ++s) {
        const u32 cnt = slice_cnt(s);
        for (u32 base = 0; base < cnt; base += 64) {
            const bool valid = base + lane < cnt;
            const u64 rec = valid ? row[(i64)s * cap + base + lane] : 0ull;
            const u32 d = (u32)(rec >> a.shift) & 0xFFu;
            u64 peers = __ballot(valid);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool bit = (d >> k) & 1u;
                const u64 m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
            if (valid) {
                const u32 rank = (u32)__popcll(peers & below), npeer = (u32)__popcll(peers);
                const u32 start = pb[d];
                orow[start + rank] = rec;
                if (rank == npeer - 1) pb[d] = start + npeer;
            }
            wave_lds_sync();
        }
    }
}

// R5  finish: the first R sorted records of every query -> ranked idx list and scores; a query with
// fewer than R records (guess too high) or an overflowed slice is flagged for the host.  thr (filter + rescore path,
// hg_real_bf.hpp): the records are a superset of the rows scoring above thr[q], so the first R are the top R of all rows
// only if the R-th still scores above it.
static __global__ __launch_bounds__(256) void k_real_finish(const u64* __restrict__ sorted, i64 crow, const u32* __restrict__ tot,
                                                     u32* __restrict__ out_idx, float* __restrict__ scores,
                                                     int* __restrict__ err, u32* __restrict__ qbad, int nKB, const float* __restrict__ thr,
                                                     const Geo g) {
    const int q = (int)(blockIdx.x / (u32)nKB);                  // nKB = ceil(R / 256) blocks per query
    const i64 k = (i64)(blockIdx.x - (u32)q * (u32)nKB) * 256 + threadIdx.x;
    const u32 n = tot[q];
    bool bad = n == 0xFFFFFFFFu || (i64)n < g.R;
    if (!bad && thr) bad = !(mono_inv(~(u32)(sorted[(i64)q * crow + g.R - 1] >> 32)) > thr[q]);
    if (k == 0) { qbad[q] = bad ? 1u : 0u; if (bad) atomicExch(err, 1); }
    if (bad || k >= g.R) return;
    const u64 rec = sorted[(i64)q * crow + k];
    out_idx[(i64)q * g.R + k] = thr ? (u32)rec & 0x7FFFFFFFu : (u32)rec;      // (filtered records carry their match bit as bit 31: k_real_rescore)
    if (scores) scores[(i64)q * g.R + k] = mono_inv(~(u32)(rec >> 32));
}

}  // namespace hg
