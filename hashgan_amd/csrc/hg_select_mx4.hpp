// hashgan_amd -- matrix-core select for codes of 65..128 bits with TWO rows per accumulator and the batched drain.
//
// k_select_mx leaves the vector ALU one op per (query, row) pair for long codes too (the sign of an accumulator is one hit
// bit: 16 v_alignbit per 16-row tile).  k_select_mx3's packing carries over with wider fields: a distance of a <= 128-bit
// code against a cut T <= 127 needs 8 bits -- T - dist + 128 lies in [0, 255] and its bit 7 says dist <= T
// (/root/reference/lib/metric.py:13-14: the inner product IS the ranking key) -- so TWO 16-row tiles accumulate into the
// same 16 registers, tile f at A-scale 2^(8 f + s):
//     acc[r] = 2^23 + 2^s(r) * ( [T - dist(row(0, r)) + 128] + 2^8 [T - dist(row(1, r)) + 128] )
// exact in f32 (every partial sum is an integer in [2^23, 2^24): 2^s * (255 + 255 * 256) < 2^23 for s <= 7).  The MX block
// scale is per lane = per A row, so the shift s(r) = r mod 8 differs between the eight registers of a group and
//     word = OR_{r in group} (acc[r] & (0x8080 << s(r)))
// gathers sixteen hit bits: bit 7 + s <-> the tile-0 row of register s, bit 15 + s <-> its tile-1 row.  With
// row(f, r) = 16 (r / 8) + 8 f + (r mod 8), word A (registers 0..7) >> 7 is the hit mask of rows 0..15 of the 32-row
// supertile in row order and word B (registers 8..15) that of rows 16..31: 30 vector ops (16 v_and_b32 v, v, v with the
// masks in registers + 14 v_or) harvest 32 rows per lane where k_select_mx spends 32 v_alignbit (VOP3, twice the issue
// time each).  A code of 65..128 bits takes two MFMAs (K = 64 bits each) per tile and field, like k_select_mx.
//
// Image dbx4 (k_expand_db4): chunk (supertile G, field f, granule m, k-half kb, register r) = 16 bytes holding code word
// 2 m + kb of row 32 G + m4_row(f, r) as 0.0 / -1.0 (fp4 0x0 / 0xA): the query image qx (+1 / -1) is k_select_mx's.
//
// Drain: k_select_mx3's (hg_select_mx3.hpp), with two hit words per entry -- per supertile and query tile every lane with a
// hit appends ONE 8-byte entry {A | lane | tile | supertile | buffer, B | slice position & 15} to the wavefront's queue; the
// emit works the queue off in batches of 64, entries that do not fill a batch wait for the next window (packed codes and
// labels triple-buffered), one-byte records {match:1 | dist:7} leave through 16-record rings per (segment, query) slice.
// A block is eight wavefronts = one segment pair x 512 queries sharing windows of two supertiles (64 rows per half).
#pragma once
#include "hg_select_mx.hpp"
#include "hg_select_mx3.hpp"       // m3_lshl_or, the drain's conventions

namespace hg {

constexpr int M4_QT = 2;                   // query tiles (of 32) per wavefront
constexpr int M4_ROWS = 32;                // rows per supertile and lane-half
#ifndef HG_M4_WS
#define HG_M4_WS 2
#endif
constexpr int M4_WS = HG_M4_WS;            // supertiles per window (<= 4: two bits of a queue entry)
constexpr int M4_QCAP = 128;               // queue entries per wavefront (ring buffer; a power of two)
constexpr int M4_RING = 16;                // records per slice ring
constexpr int M4_WPB = 8;                  // wavefronts per block: they share the staged window
constexpr int M4_NM = 2;                   // MFMAs (64-bit granules) per tile: codes of 65..128 bits

// register r (0..15) of tile f (0..1) -> row of the 32-row supertile; the register's scale shift
__host__ __device__ constexpr int m4_row(int f, int r) { return 16 * (r >> 3) + 8 * f + (r & 7); }
__host__ __device__ constexpr int m4_shift(int r) { return r & 7; }
__host__ __device__ inline void m4_place(int rho, int& f, int& r) {     // the inverse: row of the supertile -> (tile, register)
    f = (rho >> 3) & 1;
    r = 8 * (rho >> 4) + (rho & 7);
}

// Database image: supertiles of 32 rows; chunk (supertile G, field f, granule m, k-half kb, register r) = 16 bytes at
// ((((G * 2 + f) * 2 + m) * 2 + kb) * 16 + r) * 16
static __global__ __launch_bounds__(256) void k_expand_db4(const u32* __restrict__ db, uint4* __restrict__ dbx, i64 N, i64 n32, int NW) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n32 * 4) return;
    const i64 row = i >> 2;
    const int wd = (int)(i & 3);                                      // code word of the row: granule m = wd / 2, k-half kb = wd % 2
    const u32 x = (row < N && wd < NW) ? db[row * NW + wd] : 0u;
    const i64 G = row / M4_ROWS;
    int f, r;
    m4_place((int)(row - G * M4_ROWS), f, r);
    uint4 e = expand_word(x, false);                                  // 0x2 per set bit
    e.x |= e.x << 2; e.y |= e.y << 2; e.z |= e.z << 2; e.w |= e.w << 2;   // 0xA = -1.0
    dbx[((((G * 2 + f) * 2 + (wd >> 1)) * 2 + (wd & 1)) * 16) + r] = e;
}

struct Mx4Lds {                // byte offsets inside the block's dynamic LDS
    int a, abuf;               // A fragments: 2 buffers of abuf bytes
    int cl, clbuf, labels;     // packed codes + labels of a window's rows (both halves): 3 buffers of clbuf bytes; labels inside a buffer
    int qcodes, qlabels;       // the block's query tables
    int queue;                 // per-wave queues: [QCAP] u64 {A, B}
    int rings;                 // per-wave slice rings
    int total;
};
__host__ __device__ inline Mx4Lds mx4_lds_layout(int NW, int LW) {
    Mx4Lds l;
    constexpr int WROWS = M4_WS * M4_ROWS;
    l.a = 0;
    l.abuf = M4_WS * 2 * M4_NM * 1024;
    l.cl = 2 * l.abuf;
    l.labels = 2 * WROWS * NW * 4;
    l.clbuf = (l.labels + 2 * WROWS * LW * 8 + 15) & ~15;
    l.qcodes = l.cl + 3 * l.clbuf;
    l.qlabels = l.qcodes + M4_WPB * 64 * NW * 4;
    l.queue = l.qlabels + M4_WPB * 64 * LW * 8;
    l.rings = l.queue + M4_WPB * M4_QCAP * 8;
    l.total = l.rings + M4_WPB * 64 * M4_QT * M4_RING;
    return l;
}

template <int NW, int LW>
struct Mx4Drain {
    static constexpr int QT = M4_QT, CB = NW * 4, LB = LW * 8;
    static constexpr int WROWS = M4_WS * M4_ROWS;
    u8* lds;
    Mx4Lds L;
    u64* qab;                            // this wavefront's queue
    u8* rings;                           // this wavefront's rings: slice (t, lane) at ring_index(t) * M4_RING
    u32 ring_base;                       // ... as an LDS address
    int wave, lane;
    u32 cap;                             // slice capacity (records), a multiple of 16
    u8* tb0;                             // the wavefront's first slice (t = 0, lane 0); tile t adds t * 32 * crow
    i64 crow;
    u32 lane_off;                        // byte offset of the lane's slices relative to that (the launcher keeps 64 * crow below 2^31)
    u32 cnt[QT];                         // records of slice (t, lane) pushed so far (may exceed cap: the surplus is dropped at the flush)
    u32 prev[QT];                        // ... pushed before the current window: those are in the rings for sure
    u32 flushed[QT];                     // ... written to global memory (a multiple of 8)
    u32 qhead, qfill, old;               // queue: first entry, entries, entries pushed before the current window (wave-uniform)

    __device__ __forceinline__ void init(u8* lds_, const Mx4Lds& L_, int wave_, int lane_, int qb, int sp, u32 cap_, i64 crow_, u8* cand8) {
        lds = lds_; L = L_; wave = wave_; lane = lane_; cap = cap_; crow = crow_;
        qab = (u64*)(lds + L.queue) + wave * M4_QCAP;
        rings = lds + L.rings + wave * (64 * QT * M4_RING);
        ring_base = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)rings;
        const int h = lane >> 5, j = lane & 31;
        lane_off = (u32)j * (u32)crow + (u32)h * cap;
        tb0 = cand8 + (i64)(qb * M4_WPB + wave) * 64 * crow + (i64)(2 * sp) * cap;
        qhead = qfill = old = 0;
#pragma unroll
        for (int t = 0; t < QT; ++t) cnt[t] = prev[t] = flushed[t] = 0;
    }
    // ring of slice (t, lane): half * 64 + t * 32 + query-in-tile -- the low six bits are the tag a queue entry carries
    __device__ __forceinline__ int ring_index(const int t) const { return (lane >> 5) * 64 + t * 32 + (lane & 31); }
    __device__ __forceinline__ u8* slice(const int t) const { return tb0 + (i64)t * 32 * crow + lane_off; }
    static __device__ __forceinline__ u32 flat(const u32 a, const u32 b) {       // {A, B} -> hit mask of the supertile, bit P <-> row P
        return ((a >> 7) & 0xFFFFu) | (((b >> 7) & 0xFFFFu) << 16);
    }

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
                    const u8* ring = rings + ring_index(t) * M4_RING;
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
        if ((u32)lane < n) {
            const u32 i = (qhead + (u32)lane) & (M4_QCAP - 1);
            const u64 ab = qab[i];
            const u32 a = (u32)ab, b = (u32)(ab >> 32);
            // entry: a = {query tag t * 32 + j : 6 | 0 | A : 16 | 0 : 4 | lane-half : 1 | supertile : 2 | buffer : 2}, b = {.. B : 16 .. | position : 5}
            const u32 qx = a & 63u, h = (a >> 27) & 1u, st = (a >> 28) & 3u, sel = a >> 30;
            u32 pos = b >> 27;                                        // slice position & 15 of the entry's first hit
            u32 x = flat(a, b);
            const u32 ql = (u32)wave * 64u + qx;                      // the entry's query, block-local
            u32 qcw[NW];
            u64 qlw[LW];
#pragma unroll
            for (int k = 0; k < NW; ++k) qcw[k] = ((const u32*)(lds + L.qcodes + ql * CB))[k];
#pragma unroll
            for (int k = 0; k < LW; ++k) qlw[k] = ((const u64*)(lds + L.qlabels + ql * LB))[k];
            const u32 ring = ring_base + (h * 64u + qx) * M4_RING;    // LDS address (the block's dynamic LDS starts at 0), a multiple of 16
            // LDS byte offsets of the code / label words of the supertile's row 0 (buffer sel, lane-half h, supertile st)
            const u32 row0 = h * WROWS + st * M4_ROWS;
            const u32 code0 = (u32)L.cl + sel * (u32)L.clbuf + row0 * CB;
            const u32 lab0 = (u32)L.cl + sel * (u32)L.clbuf + (u32)L.labels + row0 * LB;
            while (x) {
                const u32 P = (u32)__builtin_ctz(x);                  // lowest set bit = earliest row
                x &= x - 1u;
                const u32* rp = (const u32*)(lds + (code0 + P * CB));
                u32 d = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k) d += __builtin_popcount(qcw[k] ^ rp[k]);
                const u64* lp = (const u64*)(lds + (lab0 + P * LB));
                u64 any = 0;
#pragma unroll
                for (int k = 0; k < LW; ++k) any |= lp[k] & qlw[k];
                *(u8 __attribute__((address_space(3)))*)(uintptr_t)m3_and_or(pos, M4_RING - 1, ring) = make_rec8(d, any != 0);
                ++pos;
            }
        }
        wave_lds_sync();
        qhead = (qhead + n) & (M4_QCAP - 1);
        qfill -= n;
        old = old > n ? old - n : 0u;
    }
    __device__ __forceinline__ void emit_all() {
        while (qfill) emit_batch(qfill < 64u ? qfill : 64u);
    }

    // ---- rare: the lane writes the hits of one of its own supertile masks straight to global memory ----
    // (its ring's leftovers first, so the slice stays in index order; every record also passes through the ring, whose
    // last partial piece is then what a later flush expects)
    __device__ __forceinline__ void direct_walk(const int t, const u32 wa, const u32 wb, const int st, const u32 sel) {
        const u8* ring_r = rings + ring_index(t) * M4_RING;
        u8* ring = rings + ring_index(t) * M4_RING;
        u8* out = slice(t);
        for (u32 p = flushed[t]; p < cnt[t]; ++p) if (p < cap) out[p] = ring_r[p & (M4_RING - 1)];
        u32 x = flat(wa, wb);
        const int ql = wave * 64 + t * 32 + (lane & 31);
        u32 qcw[NW];
        u64 qlw[LW];
#pragma unroll
        for (int k = 0; k < NW; ++k) qcw[k] = ((const u32*)(lds + L.qcodes + ql * CB))[k];
#pragma unroll
        for (int k = 0; k < LW; ++k) qlw[k] = ((const u64*)(lds + L.qlabels + ql * LB))[k];
        const u8* clb = lds + L.cl + sel * L.clbuf;
        const u32 row0 = (u32)(lane >> 5) * WROWS + (u32)st * M4_ROWS;
        u32 pos = cnt[t];
        while (x) {
            const u32 P = (u32)__builtin_ctz(x);
            x &= x - 1u;
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
            ring[pos & (M4_RING - 1)] = rec;
            ++pos;
        }
        cnt[t] = pos;
        prev[t] = pos;
        flushed[t] = pos & ~7u;
    }

    // Rare: some slice would have more than M4_RING unflushed records.  Everything queued is emitted and flushed; slices
    // that still cannot take their hits go the direct route and their words are cleared.
    __device__ __forceinline__ void make_room(u32 (&w)[QT][2], const int st, const u32 sel) {
        emit_all();
#pragma unroll
        for (int t = 0; t < QT; ++t) prev[t] = cnt[t];
        flush_to(prev);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u32 want = cnt[t] + (u32)__builtin_popcount(w[t][0]) + (u32)__builtin_popcount(w[t][1]);
            if (want - flushed[t] > (u32)M4_RING) {
                direct_walk(t, w[t][0], w[t][1], st, sel);
                w[t][0] = w[t][1] = 0u;
            }
        }
        wave_lds_sync();
    }

    // The hit words of one supertile: w[t] = {A, B} of query tile t.  st = supertile of the window, sel = the window's
    // codes/labels buffer.
    __device__ __forceinline__ void push(u32 (&w)[QT][2], const int st, const u32 sel) {
        u32 any[QT], want[QT];
        u64 bal[QT];
        {
            bool over = false;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                want[t] = cnt[t] + (u32)__builtin_popcount(w[t][0]) + (u32)__builtin_popcount(w[t][1]);
                over |= want[t] - flushed[t] > (u32)M4_RING;
            }
            if (__builtin_expect(__any(over) != 0, 0)) {              // rare: afterwards every ring takes what is left of the words
                make_room(w, st, sel);
#pragma unroll
                for (int t = 0; t < QT; ++t) want[t] = cnt[t] + (u32)__builtin_popcount(w[t][0]) + (u32)__builtin_popcount(w[t][1]);
            }
        }
        // (the hit flags and ballots have ONE definition, behind the rare branch: no second compare for the stores' exec mask)
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            any[t] = w[t][0] | w[t][1];
            bal[t] = __ballot(any[t] != 0u);
        }
        u32 nz = 0;
#pragma unroll
        for (int t = 0; t < QT; ++t) nz += (u32)__builtin_popcountll(bal[t]);
        if (__builtin_expect(qfill + nz > (u32)M4_QCAP, 0)) {         // a full queue: work off whole batches (never wasted work);
            while (qfill >= 64u) emit_batch(64u);                     // a dense supertile (up to 128 entries) needs it empty
            if (qfill + nz > (u32)M4_QCAP) emit_batch(qfill);
        }
        const u32 desc = ((u32)st << 28) | (sel << 30);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u64 b = bal[t];
            const u32 slot = (qhead + qfill + __builtin_amdgcn_mbcnt_hi((u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((u32)b, 0u))) & (M4_QCAP - 1);
            if (__builtin_amdgcn_inverse_ballot_w64(b)) {             // (the ballot IS the exec mask: no second compare)
                // hit bits 7..22; the query tag t * 32 + j in 0..5; lane-half, supertile, buffer above
                const u32 ea = w[t][0] | ((u32)(lane & 31) | ((u32)t << 5) | ((u32)(lane >> 5) << 27)) | desc;
                const u32 eb = w[t][1] | (cnt[t] << 27);
                qab[slot] = ((u64)eb << 32) | ea;
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
                const u8* ring = rings + ring_index(t) * M4_RING;
                *(u64*)(slice(t) + min(f, cap - 8u)) = *(const u64*)(ring + (f & 8u));
            }
        }
    }
};

// Geo as set by the launcher: g.nQT = query blocks (of 64 M4_WPB queries) per segment pair, g.nBlk = blocks; g.L % 32 == 0.
#ifndef HG_M4_WAVES
#define HG_M4_WAVES 4
#endif
#ifndef HG_M4_SEQ
#define HG_M4_SEQ 1
#endif
template <int NW, int LW>
__global__ __launch_bounds__(64 * M4_WPB) __attribute__((amdgpu_waves_per_eu(HG_M4_WAVES, HG_M4_WAVES)))
void k_select_mx4(const u32* __restrict__ qc, const u64* __restrict__ qlab, const u8* __restrict__ qx,
                  const u32* __restrict__ db, const u8* __restrict__ dbx, const u64* __restrict__ dblab,
                  const SelArgs a, u8* __restrict__ cand8, const Geo g) {
    static_assert(NW >= 3 && NW <= 4 && LW >= 1 && LW <= 2, "two 8-bit fields: codes of 65..128 bits; compact records: <= 128 classes");
    extern __shared__ __attribute__((aligned(1024))) u8 mxlds[];
    constexpr int QT = M4_QT, WQ = 32 * QT;
    constexpr int CB = NW * 4, LB = LW * 8;
    constexpr int WROWS = M4_WS * M4_ROWS;
    constexpr int NM = M4_NM;
    const Mx4Lds L = mx4_lds_layout(NW, LW);

    const int lb = logical_block(g.nBlk);
    if (lb < 0) return;                                   // whole block: no barrier is skipped by a part of it
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQB = g.nQT;
    const int sp = lb / nQB;                             // segment pair
    const int qb = lb - sp * nQB;                        // block of 512 queries
    const int h = lane >> 5, j = lane & 31;

    const int s = 2 * sp + h;                            // this lane's segment
    const bool seg_ok = s < g.S;
    const i64 lo0 = (i64)(2 * sp) * g.L, lo1 = lo0 + g.L;
    const i64 len0 = (lo0 + g.L < g.N ? g.L : g.N - lo0);
    const i64 len1 = lo1 >= g.N ? 0 : (lo1 + g.L < g.N ? g.L : g.N - lo1);
    const i64 mylen = h ? len1 : len0;
    const i64 minlen = len0 < len1 ? len0 : len1;
    const i64 nwin = ((len0 > len1 ? len0 : len1) + WROWS - 1) / WROWS;
    const i64 NG = (g.N + M4_ROWS - 1) / M4_ROWS;        // supertiles in the image

    // ---- query side: LDS tables for the emit, B fragments, C = the bias, harvest masks ----
    const int q0w = (qb * M4_WPB + wave) * WQ;               // first query of this wavefront
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
    i32x4 bq[QT][NM];
    f32x16 cv[QT];
    u32 alive[QT];                                         // all ones / zero: a dead lane (query beyond Q, cut outside 0..127) harvests nothing
    bool far[QT];
    Mx4Drain<NW, LW> dr;
    dr.init(mxlds, L, wave, lane, qb, sp, a.cap, a.crow, cand8);
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
        int T = live ? a.T[q] - (s > a.sstar[q] ? 1 : 0) : 0;
        far[t] = live && T > 127;                          // an 8-bit field holds T - dist + 128 only for T <= 127: such a query loses its bet
        if (T < 0 || T > 127) { live = false; T = 0; }
        const float base = (float)((T - pop + 128) * 257);            // (1 + 2^8) * field
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[t][r] = 8388608.0f + base * (float)(1 << m4_shift(r));
        u32 al = live ? 0xFFFFFFFFu : 0u;
        asm volatile("" : "+v"(al));
        alive[t] = al;
    }
    u32 K[8];                                              // hit bits of a register with scale shift s: 7 + s and 15 + s
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        u32 kk = 0x8080u << k;
        asm volatile("" : "+v"(kk));                       // in a register: v_and_b32 v, v, v is the fast form
        K[k] = kk;
    }

    // ---- window staging: global -> LDS, the eight waves split the copy instructions ----
    const int ah = (j >> 2) & 1;                                     // lane-half (segment) that A row j feeds
    const int ar = (j & 3) + 4 * (j >> 3);                           // the accumulator register of that row
    const i64 ag0 = (ah ? lo1 : lo0) / M4_ROWS;                      // first supertile of that segment
    const int sa_sh = m4_shift(ar);
    const int scale_a = (127 + sa_sh) | ((135 + sa_sh) << 8);        // E8M0: tile f rides at 2^(8 f + s)
    const int scale_b = 0x7F7F7F7F;
    // The image is one linear array of 512-byte chunks, one per (supertile, field, granule): chunk c of window win is chunk number
    // ch0 + win * WS * 4 + c of the lane's segment; the shorter segment of a pair runs past its rows -- into the next
    // segment's, or past the image: the chunk number is clamped to the image's last one (those rows are masked anyway; the
    // image ends with a window of zero rows).
    constexpr int CPS = 2 * NM;                                      // chunks per supertile
    const u32 ch_last = (u32)((NG + M4_WS) * CPS - 1);
    const u32 ch0 = (u32)(ag0 < NG ? ag0 : 0) * (u32)CPS;
    const u8* a_row = dbx + (h * 16 + ar) * 16;                      // the lane's 16 bytes inside a half-chunk pair
    const u32 lane16 = (u32)lane * 16u;
    auto stage_window = [&](const i64 win, const int abuf, const int clsel) {
        u8* sa = mxlds + L.a + abuf * L.abuf;
        u8* scl = mxlds + L.cl + clsel * L.clbuf;
#pragma unroll
        for (int k = 0; k < (M4_WS * CPS + M4_WPB - 1) / M4_WPB; ++k) {
            const int c = wave + k * M4_WPB;
            if (c < M4_WS * CPS) {
                const u32 ch = min(ch0 + (u32)win * (u32)(M4_WS * CPS) + (u32)c, ch_last);
                HG_GLDS16(a_row + (i64)ch * 512, sa + c * 1024);      // 512 B of each of the two segments -> 1 KB of fragments
            }
        }
        constexpr int CPH = (WROWS * CB + 1023) / 1024, LPH = (WROWS * LB + 1023) / 1024;
#pragma unroll
        for (int k = 0; k < (2 * (CPH + LPH) + M4_WPB - 1) / M4_WPB; ++k) {
            const int c = wave + k * M4_WPB;                            // wave-uniform: which table, half and piece
            if (c < 2 * (CPH + LPH)) {
                const int hh = c & 1, kk = c >> 1;
                const bool is_lab = kk >= CPH;
                const int piece = is_lab ? kk - CPH : kk;
                const int rowb = is_lab ? LB : CB;
                const i64 off = ((hh ? lo1 : lo0) + win * WROWS) * rowb + piece * 1024;     // wave-uniform
                const i64 lim = g.N * rowb;
                const u8* tab = is_lab ? (const u8*)dblab : (const u8*)db;
                // rows past the table: anything (masked); the last chunk may overhang the table by < 16 B (allocation slack, see k_select_mx)
                u8* dst = scl + (is_lab ? L.labels : 0) + hh * WROWS * rowb + piece * 1024;
                if (piece * 1024 + (int)lane16 < WROWS * rowb) {
                    u32 l16 = lane16;                                // (opaque: see k_select_mx3's staging -- no hoisted 64-bit `table + lane offset` to spill)
                    asm volatile("" : "+v"(l16));
                    if (off + 1024 <= lim) HG_GLDS16(tab + off + l16, dst);
                    else HG_GLDS16(tab + (off + l16 < lim ? off + l16 : 0), dst);
                }
            }
        }
    };

    auto harvest = [&](const f32x16& acc, const int t, u32 (&w)[2]) {
#define HG_U(r) __float_as_uint(acc[r])
        const u32 a0 = ((HG_U(0) & K[0]) | (HG_U(1) & K[1])) | ((HG_U(2) & K[2]) | (HG_U(3) & K[3]));
        const u32 a1 = ((HG_U(4) & K[4]) | (HG_U(5) & K[5])) | ((HG_U(6) & K[6]) | (HG_U(7) & K[7]));
        w[0] = (a0 | a1) & alive[t];
        const u32 b0 = ((HG_U(8) & K[0]) | (HG_U(9) & K[1])) | ((HG_U(10) & K[2]) | (HG_U(11) & K[3]));
        const u32 b1 = ((HG_U(12) & K[4]) | (HG_U(13) & K[5])) | ((HG_U(14) & K[6]) | (HG_U(15) & K[7]));
        w[1] = (b0 | b1) & alive[t];
#undef HG_U
        asm volatile("" : "+v"(w[0]), "+v"(w[1]));                  // pin here (pure ops would sink into the drain)
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
        for (int st = 0; st < M4_WS; ++st) {
            u32 w[QT][2];
#if HG_M4_SEQ
            // one query tile at a time: 16 accumulator registers live instead of 32 (the kernel sits at the 128 of four
            // wavefronts per SIMD; the A fragments are read twice from LDS instead)
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x16 acc = cv[t];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const i32x4 af = *(const i32x4*)(sa + (((st * 2 + f) * NM + m) * 64 + lane) * 16);
                        const i32x8 A = {af.x, af.y, af.z, af.w, 0, 0, 0, 0};
                        const i32x8 B = {bq[t][m].x, bq[t][m].y, bq[t][m].z, bq[t][m].w, 0, 0, 0, 0};
                        acc = f == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, scale_a, 0, scale_b)
                                     : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 1, scale_a, 0, scale_b);
                    }
                }
                harvest(acc, t, w[t]);
            }
#else
            f32x16 acc[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) acc[t] = cv[t];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const i32x4 af = *(const i32x4*)(sa + (((st * 2 + f) * NM + m) * 64 + lane) * 16);
                    const i32x8 A = {af.x, af.y, af.z, af.w, 0, 0, 0, 0};
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const i32x8 B = {bq[t][m].x, bq[t][m].y, bq[t][m].z, bq[t][m].w, 0, 0, 0, 0};
                        acc[t] = f == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 0, scale_a, 0, scale_b)
                                        : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[t], 4, 4, 1, scale_a, 0, scale_b);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) harvest(acc[t], t, w[t]);
#endif
            // rows past the end of the lane's segment (ragged last window, unpaired last segment) never count
            const i64 base_row = win * WROWS + st * M4_ROWS;
            if (minlen - base_row < M4_ROWS) {
                const i64 left = mylen - base_row;                   // valid rows of this lane in the supertile
                const int la = left < 0 ? 0 : left > 16 ? 16 : (int)left, lb2 = left < 16 ? 0 : left > 32 ? 16 : (int)left - 16;
                const u32 ka = ((1u << la) - 1u) << 7, kb = ((1u << lb2) - 1u) << 7;
#pragma unroll
                for (int t = 0; t < QT; ++t) { w[t][0] &= ka; w[t][1] &= kb; }
            }
            dr.push(w, st, (u32)clsel);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the owners flush every fourth supertile (128 rows)
        dr.end_window(((win + 1) * M4_WS) % 4 == 0);
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
