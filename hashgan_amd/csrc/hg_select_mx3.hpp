// hashgan_amd -- matrix-core select for codes of up to 64 bits with THREE rows per accumulator and a batched drain.
//
// k_select_mx leaves the vector ALU one op per (query, row) pair (the sign of an accumulator is one hit bit) and that
// op stream, not the matrix pipe, is what bounds its pair work.  An f32 accumulator has 23 payload bits; a distance of
// a <= 64-bit code against a cut T <= 63 needs 7:  T - dist + 64 lies in [0, 127] and its bit 6 says dist <= T
// (/root/reference/lib/metric.py:13-14: the inner product IS the ranking key).  So three 16-row tiles accumulate into
// the SAME 16 registers, tile f at A-scale 2^(7 f + s):
//     acc[r] = 2^23 + 2^s(r) * sum_{f = 0..2} 2^(7 f) * [T - dist(row(f, r)) + 64]
// exact in f32 (every partial sum is an integer in [2^23, 2^24): the fields never borrow, whatever the order), the
// MX block scale is per lane = per A row, so s(r) in {0, 1, 2} differs between registers, and
//     word = (acc[r] & K_s(r)) | word,      K_s = bits {6, 13, 20} << s
// gathers nine hit bits of three registers; two shift-merges per word fill seven sub-positions per field.  20 vector
// ops (24 as compiled) harvest 48 rows per lane (0.5 per pair) and 16 of them are plain v_and_b32 v, v, v -- the fast VOP2 form -- with
// the masks in registers (dead lanes simply hold K = 0).  Measured (tools/ubench_mx3.hip): 0.22 ms per 10^10 pairs
// against 0.34 for 16 v_alignbit per tile, 0 mismatches against xor + popcount on 1.2e7 pairs.
//
// Rows: a supertile = 48 consecutive rows of a segment.  Register r of tile f holds row m3_row(f, r) of it; harvested
// words A (registers 0..6), B (7..13), C (14, 15):  bit 6 + 7 f + r' of A <-> row 7 f + r', i.e. A >> 6 is the hit mask
// of rows 0..20 in row order, B >> 6 of rows 21..41, and C carries rows 42..47 at bits 6 + 7 f + r' (row 42 + 2 f + r').
// The database image dbx3 (k_expand_db3) is k_select_mx's fp4 image with that row permutation and -1.0 for a set bit,
// so the query image qx (+1 / -1 for a clear / set bit) is shared with the other matrix-core kernels.
//
// Drain (one-byte compact records only: hg_mx_drain.hpp explains rings and slices).  Per supertile and query tile every
// lane with a hit appends ONE 12-byte entry {A | query tag | lane-half | supertile | buffer, B | slice position & 15, C} to the
// wavefront's queue (ring buffer in LDS, slot = rank among the pushing lanes).  The emit works the queue off in batches
// of exactly 64 entries -- every lane busy -- and entries that do not fill a batch WAIT for the next window: the packed
// codes and labels the emit needs are triple-buffered, so an entry may be emitted one window late, and the
// owner-side flush of the 16-record rings lags one window accordingly (it flushes what was pushed before the window
// that just ended).  A block is eight wavefronts = one segment pair x 512 queries sharing windows of four supertiles (192
// rows per lane-half; two for 65..128 classes); ~140 entries per window and wavefront at C2.
// Bursts (a ring that could overflow: > 16 records of one slice pending) drain everything and, if one supertile alone
// still brings too many, the lane walks its own hits straight to global memory -- rare, slow, exact.
#pragma once
#include "hg_select_mx.hpp"

namespace hg {

constexpr int M3_QT = 2;                   // query tiles (of 32) per wavefront
constexpr int M3_ROWS = 48;                // rows per supertile and lane-half
// Supertiles per window (<= 4: two bits of a queue entry).  A window costs LDS -- 3 KiB of A fragments per supertile, twice,
// and the packed codes + labels of its rows three times -- and buys fewer barriers, flushes and window ends per row.
// Eight wavefronts share it: 79 KB per block with 4 supertiles and one label word (two blocks per CU, four wavefronts per
// SIMD); two label words (65..128 classes) get 2 supertiles (67 KB).  Measured at C2, 8 wavefronts per block: 2 supertiles
// 0.678 ms, 3: 0.672, 4: 0.662 (4 wavefronts per block, 2 supertiles: 0.688).
#ifndef HG_M3_WS
#define HG_M3_WS 4
#endif
__host__ __device__ constexpr int m3_ws(int LW) { return LW <= 1 ? HG_M3_WS : 2; }
constexpr int M3_WS_MAX = HG_M3_WS > 2 ? HG_M3_WS : 2;
constexpr int M3_QCAP = 128;               // queue entries per wavefront (ring buffer; a power of two)
constexpr int M3_RING = 16;                // records per slice ring
#ifndef HG_M3_WPB
#define HG_M3_WPB 8
#endif
#ifndef HG_M3_FLUSH
#define HG_M3_FLUSH 4                      // the owners flush their rings every this many supertiles (a multiple of the window)
#endif
#ifndef HG_M3_ILV
#define HG_M3_ILV 8                        // tile 1's MFMAs carry that many of tile 0's harvest ops between them (0: round 4's order, all six MFMAs first -- 0.634 vs 0.626 ms)
#endif
constexpr int M3_WPB = HG_M3_WPB;          // wavefronts per block: they share the staged window (4: 40 KB of LDS, four blocks per CU; 8: two)

// register r (0..15) of tile f (0..2) -> row of the 48-row supertile; the register's scale shift
__host__ __device__ constexpr int m3_row(int f, int r) { return r < 7 ? 7 * f + r : r < 14 ? 21 + 7 * f + (r - 7) : 42 + 2 * f + (r - 14); }
__host__ __device__ constexpr int m3_shift(int r) { return r < 7 ? r % 3 : r < 14 ? (r - 7) % 3 : r - 14; }
__host__ __device__ inline void m3_place(int rho, int& f, int& r) {     // the inverse: row of the supertile -> (tile, register)
    if (rho < 21) { f = rho / 7; r = rho % 7; }
    else if (rho < 42) { f = (rho - 21) / 7; r = 7 + (rho - 21) % 7; }
    else { f = (rho - 42) / 2; r = 14 + (rho - 42) % 2; }
}

// Database image: supertiles of 48 rows; chunk (supertile G, tile f, k-half kb, register r) = 16 bytes at
// (((G * 3 + f) * 2 + kb) * 16 + r) * 16 holding code word kb of row 48 G + m3_row(f, r) as 0.0 / -1.0 (fp4 0x0 / 0xA).
static __global__ __launch_bounds__(256) void k_expand_db3(const u32* __restrict__ db, uint4* __restrict__ dbx, i64 N, i64 n48, int NW) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n48 * 2) return;
    const i64 row = i >> 1;
    const int kb = (int)(i & 1);
    const u32 x = (row < N && kb < NW) ? db[row * NW + kb] : 0u;
    const i64 G = row / M3_ROWS;
    int f, r;
    m3_place((int)(row - G * M3_ROWS), f, r);
    uint4 e = expand_word(x, false);                                  // 0x2 per set bit
    e.x |= e.x << 2; e.y |= e.y << 2; e.z |= e.z << 2; e.w |= e.w << 2;   // 0xA = -1.0
    dbx[((G * 3 + f) * 2 + kb) * 16 + r] = e;
}

// (x << sh) | y in ONE op: the compiler prefers two shifts + v_or3 for a three-way merge
template <int SH> __device__ __forceinline__ u32 m3_lshl_or_t(const u32 x, const u32 y) {
    u32 d;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "n"(SH), "v"(y));
    return d;
}
#define m3_lshl_or(x, sh, y) m3_lshl_or_t<sh>(x, y)
// v_ffbl_b32: index of the lowest set bit, ~0 for 0
__device__ __forceinline__ u32 m3_ffbl(const u32 x) {
    u32 d;
    asm("v_ffbl_b32 %0, %1" : "=v"(d) : "v"(x));
    return d;
}
// (x & K) | y in one op
__device__ __forceinline__ u32 m3_and_or(const u32 x, const u32 k, const u32 y) {
    u32 d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "n"(k), "v"(y));
    return d;
}
// x * 0x421 for x < 2^22 (the C word's six hit bits -> contiguous): the full-rate 24-bit multiply (the compiler turns __umul24 by a
// constant back into the quarter-rate v_mul_lo_u32)
__device__ __forceinline__ u32 m3_mul24_421(const u32 x) {
    u32 d;
    asm("v_mul_u32_u24 %0, 0x421, %1" : "=v"(d) : "v"(x));
    return d;
}

struct Mx3Lds {                // byte offsets inside the block's dynamic LDS
    int a, abuf;               // A fragments: 2 buffers of abuf bytes
    int cl, clbuf, labels;     // packed codes + labels of a window's rows (both halves): 3 buffers of clbuf bytes; labels inside a buffer
    int qcodes, qlabels;       // the block's query tables
    int queue;                 // per-wave queues: [QCAP] entries of 12 bytes {A, B, C} (ONE address per entry; round 4 kept {A, B} and {C} in two arrays: 0.627 vs 0.615 ms)
    int rings;                 // per-wave slice rings
    int total;
};
__host__ __device__ inline Mx3Lds mx3_lds_layout(int NW, int LW) {
    Mx3Lds l;
    const int M3_WS = m3_ws(LW), M3_WROWS = M3_WS * M3_ROWS;
    l.a = 0;
    l.abuf = M3_WS * 3 * 1024;
    l.cl = 2 * l.abuf;
    l.labels = 2 * M3_WROWS * NW * 4;
    l.clbuf = (l.labels + 2 * M3_WROWS * LW * 8 + 15) & ~15;
    l.qcodes = l.cl + 3 * l.clbuf;
    l.qlabels = l.qcodes + M3_WPB * 64 * NW * 4;
    l.queue = l.qlabels + M3_WPB * 64 * LW * 8;
    l.rings = l.queue + M3_WPB * M3_QCAP * 12;
    l.total = l.rings + M3_WPB * 64 * M3_QT * M3_RING;
    return l;
}

template <int NW, int LW>
struct Mx3Drain {
    static constexpr int QT = M3_QT, CB = NW * 4, LB = LW * 8;
    static constexpr int M3_WS = m3_ws(LW), M3_WROWS = M3_WS * M3_ROWS;
    u8* lds;
    Mx3Lds L;
    u32 ring_base;                       // LDS address of the wavefront's first ring
    u32 q12_base;                        // this wavefront's queue: LDS address of its first 12-byte entry (kept opaque: ONE address per entry)
    u8* rings;                           // this wavefront's rings: slice (t, lane) at (t * 64 + lane) * M3_RING
    int wave, lane;
    u32 cap;                             // slice capacity (records), a multiple of 16
    u8* tb0;                             // the wavefront's first slice (t = 0, lane 0); tile t adds t * 32 * crow
    i64 crow;
    u32 lane_off;                        // byte offset of the lane's slices relative to that (the launcher keeps 64 * crow below 2^31)
    u32 cnt[QT];                         // records of slice (t, lane) pushed so far (may exceed cap: the surplus is dropped at the flush)
    u32 prev[QT];                        // ... pushed before the current window: those are in the rings for sure
    u32 flushed[QT];                     // ... written to global memory (a multiple of 8)
    u32 qhead, qfill, old;               // queue: first entry, entries, entries pushed before the current window (wave-uniform)
    int probe;

    __device__ __forceinline__ void init(u8* lds_, const Mx3Lds& L_, int wave_, int lane_, int qb, int sp, u32 cap_, i64 crow_, u8* cand8, int probe_) {
        lds = lds_; L = L_; wave = wave_; lane = lane_; cap = cap_; crow = crow_; probe = probe_;
        q12_base = (u32)(L.queue + wave * (M3_QCAP * 12));
        asm volatile("" : "+s"(q12_base));
        rings = lds + L.rings + wave * (64 * QT * M3_RING);
        ring_base = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)rings;
        const int h = lane >> 5, j = lane & 31;
        lane_off = (u32)j * (u32)crow + (u32)h * cap;
        tb0 = cand8 + (i64)(qb * M3_WPB + wave) * 64 * crow + (i64)(2 * sp) * cap;
        qhead = qfill = old = 0;
#pragma unroll
        for (int t = 0; t < QT; ++t) cnt[t] = prev[t] = flushed[t] = 0;
    }
    // ring of slice (t, lane): half * 64 + t * 32 + query-in-tile -- the low six bits are the tag a queue entry carries
    __device__ __forceinline__ int ring_index(const int t) const { return (lane >> 5) * 64 + t * 32 + (lane & 31); }
    __device__ __forceinline__ u8* slice(const int t) const { return tb0 + (i64)t * 32 * crow + lane_off; }

    // ---- owner side: completed 8-record pieces below limit[t] leave the ring with one aligned 8-byte store each ----
    // (a slice that is already full keeps advancing: its surplus pieces land on its last piece -- the query is flagged
    // as lost at the end of the kernel, what its slice holds no longer matters, only that the stores stay inside it)
    __device__ __forceinline__ void flush_to(const u32 (&limit)[QT]) {
        bool need = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) need |= limit[t] - flushed[t] >= 8u;
        while (__any(need)) {                                         // a second pass only if some slice had 16 pending
            need = false;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const u32 f = flushed[t];
                if (limit[t] - f >= 8u) {
                    const u8* ring = rings + ring_index(t) * M3_RING;
                    u8* tb = tb0 + (i64)t * 32 * crow;                // wave-uniform base; the lane's part fits 32 bits
                    *(u64*)(tb + (lane_off + min(f, cap - 8u))) = *(const u64*)(ring + (f & 8u));
                    flushed[t] = f + 8u;
                    need |= limit[t] - f >= 16u;
                }
            }
        }
        wave_lds_sync();                                              // ring reads done before an emit reuses the slots
    }

    // ---- emit: n <= 64 entries from the head of the queue, one per lane ----
    __device__ __forceinline__ void emit_batch(const u32 n) {
        wave_lds_sync();
        if ((u32)lane < n && !(kProbes && (probe & 8))) {
            const u32 i = (qhead + (u32)lane) & (M3_QCAP - 1);
            const u32* e = (const u32*)(lds + (q12_base + i * 12u));
            const u32 a = e[0], b = e[1], c = e[2];
            // entry: a = {query tag t * 32 + j : 6 | A : 21 | lane-half : 1 | supertile : 2 | buffer : 2}, b = {0 : 6 | B : 21 | position : 5}
            const u32 x = a & 63u, h = (a >> 27) & 1u, st = (a >> 28) & 3u, sel = a >> 30;
            u32 pos = b >> 27;                                        // slice position & 15 of the entry's first hit
            // flat hit mask of the supertile: bit P <-> row P.  b holds B at bits 6..26 and the position above them, nothing below:
            // b << 15 IS B's rows 0..10 at bits 21..31; the C word's six bits become contiguous through one 24-bit multiply
            u32 xlo = m3_lshl_or(b, 15, __builtin_amdgcn_ubfe(a, 6, 21));
            u32 xhi = ((m3_mul24_421(c) >> 6) & 0xFC00u) | __builtin_amdgcn_ubfe(b, 17, 10);
            const u32 ql = (u32)wave * 64u + x;                       // the entry's query, block-local
            u32 qcw[NW];
            u64 qlw[LW];
#pragma unroll
            for (int k = 0; k < NW; ++k) qcw[k] = ((const u32*)(lds + L.qcodes + ql * CB))[k];
#pragma unroll
            for (int k = 0; k < LW; ++k) qlw[k] = ((const u64*)(lds + L.qlabels + ql * LB))[k];
            const u32 ring = ring_base + (h * 64u + x) * M3_RING;     // LDS address (the block's dynamic LDS starts at 0), a multiple of 16
            // LDS byte offsets of the code / label words of the supertile's row 0 (buffer sel, lane-half h, supertile st)
            const u32 row0 = h * M3_WROWS + st * M3_ROWS;
            const u32 code0 = (u32)L.cl + sel * (u32)L.clbuf + row0 * CB;
            const u32 lab0 = (u32)L.cl + sel * (u32)L.clbuf + (u32)L.labels + row0 * LB;
            while (xlo | xhi) {
                const u32 P = min(m3_ffbl(xlo), m3_ffbl(xhi) | 32u);  // lowest set bit = earliest row (v_ffbl of 0 is ~0)
                const u32 lo1 = xlo - 1u;
                xhi &= xhi - (xlo == 0u ? 1u : 0u);
                xlo &= lo1;
                const u32* rp = (const u32*)(lds + (code0 + P * CB));
                u32 d = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k) d += __builtin_popcount(qcw[k] ^ rp[k]);
                const u64* lp = (const u64*)(lds + (lab0 + P * LB));
                u64 any = 0;
#pragma unroll
                for (int k = 0; k < LW; ++k) any |= lp[k] & qlw[k];
                if (!(kProbes && (probe & 4))) *(u8 __attribute__((address_space(3)))*)(uintptr_t)m3_and_or(pos, M3_RING - 1, ring) = make_rec8(d, any != 0);
                ++pos;
            }
        }
        wave_lds_sync();
        qhead = (qhead + n) & (M3_QCAP - 1);
        qfill -= n;
        old = old > n ? old - n : 0u;
    }
    __device__ __forceinline__ void emit_all() {
        while (qfill) emit_batch(qfill < 64u ? qfill : 64u);
    }

    // ---- rare: the lane writes the hits of one of its own supertile masks straight to global memory ----
    // (its ring's leftovers first, so the slice stays in index order; every record also passes through the ring, whose
    // last partial piece is then what a later flush expects)
    __device__ __forceinline__ void direct_walk(const int t, const u32 wa, const u32 wb, const u32 wc, const int st, const u32 sel) {
        const u8* ring_r = rings + ring_index(t) * M3_RING;
        u8* ring = rings + ring_index(t) * M3_RING;
        u8* out = slice(t);
        for (u32 p = flushed[t]; p < cnt[t]; ++p) if (p < cap) out[p] = ring_r[p & (M3_RING - 1)];
        const u32 a21 = (wa >> 6) & 0x1FFFFFu, b21 = (wb >> 6) & 0x1FFFFFu, c6 = ((wc * 0x421u) >> 16) & 0x3Fu;
        u64 x = (u64)a21 | ((u64)b21 << 21) | ((u64)c6 << 42);
        const int ql = wave * 64 + t * 32 + (lane & 31);
        u32 qcw[NW];
        u64 qlw[LW];
#pragma unroll
        for (int k = 0; k < NW; ++k) qcw[k] = ((const u32*)(lds + L.qcodes + ql * CB))[k];
#pragma unroll
        for (int k = 0; k < LW; ++k) qlw[k] = ((const u64*)(lds + L.qlabels + ql * LB))[k];
        const u8* clb = lds + L.cl + sel * L.clbuf;
        const u32 row0 = (u32)(lane >> 5) * M3_WROWS + (u32)st * M3_ROWS;
        u32 pos = cnt[t];
        while (x) {
            const u32 P = (u32)__builtin_ctzll(x);
            x &= x - 1ull;
            const u32* rp = (const u32*)(clb + (row0 + P) * CB);
            u32 d = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) d += __builtin_popcount(qcw[k] ^ rp[k]);
            const u64* lp = (const u64*)(clb + L.labels + (row0 + P) * LB);
            u64 any = 0;
#pragma unroll
            for (int k = 0; k < LW; ++k) any |= lp[k] & qlw[k];
            const u8 rec = make_rec8(d, any != 0);
            if (pos < cap) out[pos] = rec;
            ring[pos & (M3_RING - 1)] = rec;
            ++pos;
        }
        cnt[t] = pos;
        prev[t] = pos;
        flushed[t] = pos & ~7u;
    }

    // Rare: the queue cannot take this supertile's entries, or some slice would have more than M3_RING unflushed records.
    // Everything queued is emitted and flushed; slices that still cannot take their hits go the direct route and their
    // words are cleared.
    __device__ __forceinline__ void make_room(u32 (&w)[QT][3], const int st, const u32 sel) {
        emit_all();
#pragma unroll
        for (int t = 0; t < QT; ++t) prev[t] = cnt[t];
        flush_to(prev);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u32 want = cnt[t] + (u32)__builtin_popcount(w[t][0]) + (u32)__builtin_popcount(w[t][1]) + (u32)__builtin_popcount(w[t][2]);
            if (want - flushed[t] > (u32)M3_RING) {
                direct_walk(t, w[t][0], w[t][1], w[t][2], st, sel);
                w[t][0] = w[t][1] = w[t][2] = 0u;
            }
        }
        wave_lds_sync();
    }

    // The hit words of one supertile: w[t] = {A, B, C} of query tile t.  st = supertile of the window, sel = the
    // window's codes/labels buffer.
    __device__ __forceinline__ void push(u32 (&w)[QT][3], const int st, const u32 sel) {
        u32 any[QT], want[QT];
        u64 bal[QT];
        {
            bool over = false;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                want[t] = cnt[t] + (u32)__builtin_popcount(w[t][0]) + (u32)__builtin_popcount(w[t][1]) + (u32)__builtin_popcount(w[t][2]);
                over |= want[t] - flushed[t] > (u32)M3_RING;
            }
            if (__builtin_expect(__any(over) != 0, 0)) {              // rare: afterwards every ring takes what is left of the words
                make_room(w, st, sel);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    want[t] = cnt[t] + (u32)__builtin_popcount(w[t][0]) + (u32)__builtin_popcount(w[t][1]) + (u32)__builtin_popcount(w[t][2]);
            }
        }
        // (the hit flags and ballots have ONE definition, behind the rare branch: no second compare for the stores' exec mask)
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            any[t] = w[t][0] | w[t][1] | w[t][2];
            bal[t] = __ballot(any[t] != 0u);
        }
        u32 nz = 0;
#pragma unroll
        for (int t = 0; t < QT; ++t) nz += (u32)__builtin_popcountll(bal[t]);
        if (__builtin_expect(qfill + nz > (u32)M3_QCAP, 0)) {         // a full queue: work off whole batches (never wasted work);
            while (qfill >= 64u) emit_batch(64u);                     // a dense supertile (up to 128 entries) needs it empty
            if (qfill + nz > (u32)M3_QCAP) emit_batch(qfill);
        }
        const u32 desc = ((u32)st << 28) | (sel << 30);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u64 b = bal[t];
            const u32 slot = (qhead + qfill + __builtin_amdgcn_mbcnt_hi((u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((u32)b, 0u))) & (M3_QCAP - 1);
            if (__builtin_amdgcn_inverse_ballot_w64(b)) {             // (the ballot IS the exec mask: no second compare)
                const u32 ea = w[t][0] | ((u32)(lane & 31) | ((u32)t << 5) | ((u32)(lane >> 5) << 27)) | desc;
                const u32 eb = w[t][1] | (cnt[t] << 27);
                u32* e = (u32*)(lds + (q12_base + slot * 12u));
                e[0] = ea; e[1] = eb; e[2] = w[t][2];
            }
            cnt[t] = want[t];
            qfill += (u32)__builtin_popcountll(b);
        }
    }

    // End of a window: entries pushed before it must be emitted now (their codes/labels buffer is recycled next); of
    // this window's, whole batches only.  Then the owners flush what was pushed before this window.
    __device__ __forceinline__ void end_window(const bool do_flush) {
        while (qfill >= 64u) emit_batch(64u);
        if (old) emit_batch(qfill);
        old = qfill;
        if (do_flush) flush_to(prev);
#pragma unroll
        for (int t = 0; t < QT; ++t) prev[t] = cnt[t];
    }

    // End of the kernel: everything out; the last partial piece of a slice leaves as a whole 8-byte store (slots past
    // cnt are inside the slice's capacity, a multiple of 16).
    __device__ __forceinline__ void finish() {
        emit_all();
        flush_to(cnt);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u32 f = flushed[t];
            if (cnt[t] > f) {
                const u8* ring = rings + ring_index(t) * M3_RING;
                *(u64*)(slice(t) + min(f, cap - 8u)) = *(const u64*)(ring + (f & 8u));
            }
        }
    }
};

// Geo as set by the launcher: g.nQT = query blocks (of 64 M3_WPB queries) per segment pair, g.nBlk = blocks; g.L % 48 == 0.
template <int NW, int LW>
#ifndef HG_M3_WAVES
#define HG_M3_WAVES 4
#endif
__global__ __launch_bounds__(64 * M3_WPB) __attribute__((amdgpu_waves_per_eu(HG_M3_WAVES, HG_M3_WAVES)))
void k_select_mx3(const u32* __restrict__ qc, const u64* __restrict__ qlab, const u8* __restrict__ qx,
                  const u32* __restrict__ db, const u8* __restrict__ dbx, const u64* __restrict__ dblab,
                  const SelArgs a, u8* __restrict__ cand8, const Geo g) {
    static_assert(NW <= 2 && LW >= 1 && LW <= 2, "three 7-bit fields: codes of <= 64 bits; compact records: <= 128 classes");
    extern __shared__ __attribute__((aligned(1024))) u8 mxlds[];
    constexpr int QT = M3_QT, WQ = 32 * QT;
    constexpr int CB = NW * 4, LB = LW * 8;
    constexpr int M3_WS = m3_ws(LW), M3_WROWS = M3_WS * M3_ROWS;      // this label width's window
    const Mx3Lds L = mx3_lds_layout(NW, LW);

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
    const i64 minlen = len0 < len1 ? len0 : len1;
    const i64 nwin = ((len0 > len1 ? len0 : len1) + M3_WROWS - 1) / M3_WROWS;
    const i64 NG = (g.N + M3_ROWS - 1) / M3_ROWS;        // supertiles in the image

    // ---- query side: LDS tables for the emit, B fragments, C = the bias, harvest masks ----
    const int q0w = (qb * M3_WPB + wave) * WQ;               // first query of this wavefront
    {
        u32* qcl = (u32*)(mxlds + L.qcodes + wave * WQ * CB);
        for (int e = lane; e < WQ * NW; e += 64) {
            const i64 q = q0w + e / NW;
            qcl[e] = q < g.Q ? qc[q * NW + (e % NW)] : 0u;
        }
        u64* qll = (u64*)(mxlds + L.qlabels + wave * WQ * LB);
        for (int e = lane; e < WQ * LW; e += 64) {
            const i64 q = q0w + e / LW;
            qll[e] = q < g.Q ? qlab[q * LW + (e % LW)] : 0ull;
        }
    }
    i32x4 bq[QT];
    f32x16 cv[QT];
    u32 K[QT][3];
    bool far[QT];
    Mx3Drain<NW, LW> dr;
    dr.init(mxlds, L, wave, lane, qb, sp, a.cap, a.crow, cand8, a.probe);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        bool live = q < g.Q && seg_ok;
        int pop = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) pop += __builtin_popcount(q < g.Q ? qc[(i64)q * NW + w] : 0u);
        bq[t] = *(const i32x4*)(qx + ((i64)(q0w / 32 + t) * 64 + lane) * 16);
        // past the query's last tie-collecting segment only rows strictly closer than the guess are taken
        int T = live ? a.T[q] - (s > a.sstar[q] ? 1 : 0) : 0;
        far[t] = live && T > 63;                           // a 7-bit field holds T - dist + 64 only for T <= 63: such a query loses its bet
        if (T < 0 || T > 63) { live = false; T = 0; }
        const float base = (float)((T - pop + 64) * 16513);           // (1 + 2^7 + 2^14) * field
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[t][r] = 8388608.0f + base * (float)(1 << m3_shift(r));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u32 kk = live ? 0x102040u << k : 0u;           // a dead lane harvests nothing
            asm volatile("" : "+v"(kk));
            K[t][k] = kk;
        }
    }

    // ---- window staging: global -> LDS, the four waves split the copy instructions ----
    const int ah = (j >> 2) & 1;                                     // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                           // the accumulator register of that row
    const i64 ag0 = (ah ? lo1 : lo0) / M3_ROWS;                      // first supertile of that segment
    const int sa_sh = m3_shift(ar);
    const int scale_a = (127 + sa_sh) | ((134 + sa_sh) << 8) | ((141 + sa_sh) << 16);      // E8M0: tile f rides at 2^(7 f + s)
    const int scale_b = 0x7F7F7F7F;
    // The image is one linear array of 512-byte half-chunks: chunk c of window win is chunk number ch0 + win * WS * 3 + c of
    // the lane's segment.  The two segments of a pair walk the SAME number of windows, so the shorter one (the database's
    // ragged last segment, or none at all) runs past its rows -- into the next segment's, or past the image: the chunk
    // number is clamped to the image's last one (those rows are masked anyway; the image ends with a window of zero rows).
    const u32 ch_last = (u32)((NG + M3_WS_MAX) * 3 - 1);
    const u32 ch0 = (u32)(ag0 < NG ? ag0 : 0) * 3u;
    const u8* a_row = dbx + (h * 16 + ar) * 16;                      // the lane's 16 bytes inside a half-chunk pair
    const u32 lane16 = (u32)lane * 16u;
    auto stage_window = [&](const i64 win, const int abuf, const int clsel) {
        u8* sa = mxlds + L.a + abuf * L.abuf;
        u8* scl = mxlds + L.cl + clsel * L.clbuf;
#pragma unroll
        for (int k = 0; k < (M3_WS * 3 + M3_WPB - 1) / M3_WPB; ++k) {
            const int c = wave + k * M3_WPB;
            if (c < M3_WS * 3) {
                const u32 ch = min(ch0 + (u32)win * (u32)(M3_WS * 3) + (u32)c, ch_last);
                HG_GLDS16(a_row + (i64)ch * 512, sa + c * 1024);
            }
        }
        constexpr int CPH = (M3_WROWS * CB + 1023) / 1024, LPH = (M3_WROWS * LB + 1023) / 1024;
#pragma unroll
        for (int k = 0; k < (2 * (CPH + LPH) + M3_WPB - 1) / M3_WPB; ++k) {
            const int c = wave + k * M3_WPB;                            // wave-uniform: which table, half and piece
            if (c < 2 * (CPH + LPH)) {
                const int hh = c & 1, kk = c >> 1;
                const bool is_lab = kk >= CPH;
                const int piece = is_lab ? kk - CPH : kk;
                const int rowb = is_lab ? LB : CB;
                const i64 off = ((hh ? lo1 : lo0) + win * M3_WROWS) * rowb + piece * 1024;     // wave-uniform
                const i64 lim = g.N * rowb;
                const u8* tab = is_lab ? (const u8*)dblab : (const u8*)db;
                // rows past the table: anything (masked); the last chunk may overhang the table by < 16 B (allocation slack, see k_select_mx)
                u8* dst = scl + (is_lab ? L.labels : 0) + hh * M3_WROWS * rowb + piece * 1024;
                if (piece * 1024 + (int)lane16 < M3_WROWS * rowb) {
                    // (the lane's offset is made opaque here: hoisted out of the window loop, `table + lane offset` is a 64-bit value per
                    // table that lives across the whole kernel -- and, spilled, comes back behind an s_waitcnt vmcnt(0) that also
                    // waits for the A fragments just requested: 0.609 -> 0.644 ms when a refactoring made the allocator choose it)
                    u32 l16 = lane16;
                    asm volatile("" : "+v"(l16));
                    if (off + 1024 <= lim) HG_GLDS16(tab + off + l16, dst);
                    else HG_GLDS16(tab + (off + l16 < lim ? off + l16 : 0), dst);
                }
            }
        }
    };

    auto harvest = [&](const f32x16& acc, const int t, u32 (&w)[3]) {
#define HG_U(r) __float_as_uint(acc[r])
        const u32 a0 = (HG_U(2) & K[t][2]) | ((HG_U(1) & K[t][1]) | (HG_U(0) & K[t][0]));
        const u32 a1 = (HG_U(5) & K[t][2]) | ((HG_U(4) & K[t][1]) | (HG_U(3) & K[t][0]));
        const u32 a2 = HG_U(6) & K[t][0];
        w[0] = m3_lshl_or(a2, 6, m3_lshl_or(a1, 3, a0));
        const u32 b0 = (HG_U(9) & K[t][2]) | ((HG_U(8) & K[t][1]) | (HG_U(7) & K[t][0]));
        const u32 b1 = (HG_U(12) & K[t][2]) | ((HG_U(11) & K[t][1]) | (HG_U(10) & K[t][0]));
        const u32 b2 = HG_U(13) & K[t][0];
        w[1] = m3_lshl_or(b2, 6, m3_lshl_or(b1, 3, b0));
        w[2] = (HG_U(15) & K[t][1]) | (HG_U(14) & K[t][0]);
#undef HG_U
        asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]));      // pin here (pure ops would sink into the drain)
    };

    int clsel = 0;
    if (nwin > 0) stage_window(0, 0, 0);
    for (i64 win = 0; win < nwin; ++win) {
        const int abuf = (int)(win & 1);
        const int clnext = clsel == 2 ? 0 : clsel + 1;
        // my copies of this window have landed (vmcnt), everybody's have and nobody still reads the buffers about to be refilled (barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (win + 1 < nwin) stage_window(win + 1, abuf ^ 1, clnext);
        const u8* sa = mxlds + L.a + abuf * L.abuf;
#pragma unroll
        for (int st = 0; st < M3_WS; ++st) {
            // (requesting supertile st + 1's fragments while st's hits are pushed changes nothing -- 0.6092 vs 0.6090 ms: four
            // wavefronts per SIMD hide the LDS latency; round 5)
            i32x4 af[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) af[f] = *(const i32x4*)(sa + ((st * 3 + f) * 64 + lane) * 16);
            f32x16 acc[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) acc[t] = cv[t];
            u32 w[QT][3];
#if HG_M3_ILV
            // tile by tile: tile 1's three MFMAs are issued with tile 0's harvest between them (the matrix pipe works on one
            // tile while the vector ALU harvests the other; sched_group_barrier: 0x8 = MFMA, 0x2 = VALU)
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const i32x8 B = {bq[t].x, bq[t].y, bq[t].z, bq[t].w, 0, 0, 0, 0};
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const i32x8 A = {af[f].x, af[f].y, af[f].z, af[f].w, 0, 0, 0, 0};
                    acc[t] = f == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 0, scale_a, 0, scale_b)
                           : f == 1 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 1, scale_a, 0, scale_b)
                                    : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 2, scale_a, 0, scale_b);
                }
            }
            harvest(acc[0], 0, w[0]);
            harvest(acc[1], 1, w[1]);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, HG_M3_ILV, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, HG_M3_ILV, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);
#else
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const i32x8 A = {af[f].x, af[f].y, af[f].z, af[f].w, 0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const i32x8 B = {bq[t].x, bq[t].y, bq[t].z, bq[t].w, 0, 0, 0, 0};
                    acc[t] = f == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 0, scale_a, 0, scale_b)
                           : f == 1 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 1, scale_a, 0, scale_b)
                                    : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 2, scale_a, 0, scale_b);
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) harvest(acc[t], t, w[t]);
#endif
            // rows past the end of the lane's segment (ragged last window, unpaired last segment) never count
            const i64 base_row = win * M3_WROWS + st * M3_ROWS;
            if (minlen - base_row < M3_ROWS) {
                const i64 left = mylen - base_row;                   // valid rows of this lane in the supertile
                const int la = left < 0 ? 0 : left > 21 ? 21 : (int)left, lb2 = left < 21 ? 0 : left > 42 ? 21 : (int)left - 21;
                const u32 ka = ((1u << la) - 1u) << 6, kb = ((1u << lb2) - 1u) << 6;
                u32 kc = 0;
                for (int c = 0; c < 6; ++c) if (42 + c < left) kc |= 1u << (6 + 7 * (c >> 1) + (c & 1));
#pragma unroll
                for (int t = 0; t < QT; ++t) { w[t][0] &= ka; w[t][1] &= kb; w[t][2] &= kc; }
            }
            if (!(kProbes && (a.probe & 2))) dr.push(w, st, (u32)clsel);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the owners flush every fourth supertile (192 rows: ~1.2 records per slice at C2; every second one cost 4 % more)
        if (!(kProbes && (a.probe & 2))) dr.end_window(((win + 1) * M3_WS) % HG_M3_FLUSH == 0);
        clsel = clnext;
    }
    dr.finish();

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0w + t * 32 + j;
        if (seg_ok && q < g.Qpad) {
            const bool live = q < g.Q;
            a.sl_cnt[(i64)s * g.Qpad + q] = live ? (dr.cnt[t] < a.cap ? dr.cnt[t] : a.cap) : 0u;
            if ((dr.cnt[t] > a.cap || far[t]) && live) a.fail[q] = 1u;
        }
    }
}

}  // namespace hg
