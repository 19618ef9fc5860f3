// hashgan_amd -- the drain of the matrix-core select kernels (k_select_mx, k_select_mx2): how the rare hits
// (~0.7 % of the pairs at C2) leave a wavefront whose lanes each hold 32-row hit masks of their own
// (query, segment) slices.
//
//   push   branch-free, per (query tile t, 32-row mask word w): every lane whose word is non-zero appends
//          {word | slice position of its first hit | lane | t | w} (8 bytes) to the wavefront's queue in LDS
//          (slot = rank among the pushing lanes, one ballot) and advances its slice cursor by the word's
//          popcount -- positions are fixed here, in index order, so the queue order is free;
//   emit   64 queue entries at a time, every lane busy: walk the word's bits (usually one), exact distance and
//          label-match bit from the packed rows staged in LDS, one record per hit.
//
// Two record formats (metric.py:14,17-19 need, per ranked row, its distance, its index and the match bit):
//   lists wanted (hg_topr, staged lists, > 128 classes)   8-byte records {idx:32 | dist:8 | match:1}, stored
//          straight from the emit, one 8-byte store per record to 64 different cache lines per instruction --
//          3.7x write amplification at the HBM (profiles/r01_v7_pmc_traffic.json);
//   AP only (hg_map, the sharded bet)   the ranking stage needs no index: records in index order ARE the tie
//          order.  COMPACT records are ONE byte {match:1 | dist:7} and never leave the CU one by one: every slice
//          owns a 16-record ring in LDS, the emit writes record `pos` to ring slot pos & 15, and after the emit the
//          OWNER lane of a slice flushes each completed 8-record piece with one aligned 8-byte store.  1/8 of
//          the store instructions, 1/8 of the bytes, no partial-line rewrites.
//          Ring invariant: a slice never has more than 16 unflushed records; the checked paths flush after every
//          emit (< 8 pending at their start, a drain may add up to 16 - 7), the sparse-window path only on demand.  A
//          push that would break it (>= 10 hits of one query in <= 64 rows: clustered duplicates) first drains
//          what is pending, then routes that word's records directly to global memory (entry flag), after the
//          owner has written the ring's leftovers out byte by byte -- rare, slow, exact.
//          A distance needs 7 bits: queries whose cut T exceeds 127 (possible only for codes of > 128 bits ranked
//          against far-away data) are flagged as lost bets and rerun exactly.
#pragma once
#include "hg_kernels.hpp"

namespace hg {

constexpr u32 MX_POS_BITS = 17;         // slice positions in a queue entry: cap < 2^17
constexpr int MX_RING = 16;             // records per slice ring (compact mode)
constexpr int MX_PIECE = 8;             // records per flush

constexpr int mx_qcap(int QT, bool compact) { return compact ? 128 : 64 * 2 * QT; }   // queue entries per wavefront
constexpr int mx_ring_bytes(int QT, bool compact) { return compact ? 64 * QT * MX_RING : 0; }   // per wavefront

__device__ __forceinline__ u8 make_rec8(u32 d, bool m) { return (u8)(d | (m ? 0x80u : 0u)); }

struct MxDrainLds {                      // byte offsets inside the block's LDS
    int qcodes, qlabels;                 // [block queries][NW] u32, [block queries][LW] u64
    int queue, rings;                    // per-wave arrays (wave w at + w * per-wave size)
    int codes, labels;                   // inside a stage: packed codes / labels of the window's rows, both halves
};

template <int NW, int LW, int QT, int WROWS, bool COMPACT>
struct MxDrain {
    static constexpr int WQ = 32 * QT;
    static constexpr int CB = NW * 4, LB = LW * 8, LWA = LW > 0 ? LW : 1;
    static constexpr int QCAP = mx_qcap(QT, COMPACT);

    u8* lds;
    MxDrainLds L;
    u64* queue;                          // this wavefront's queue
    u8* rings;                           // this wavefront's rings: slice (t, lane) at (t * 64 + lane) * MX_RING
    int wave, lane, qb, sp;
    u32 cap;                             // slice capacity (records)
    i64 crow;                            // records per query row
    int probe;
    u32 idx_base;
    i64 segL;                            // rows per segment
    u8* cand8;                           // record rows as bytes (compact: 1 byte per record, else 8)
    u32 qfill;
    u32 cnt[QT];
    u32 flags;                           // bit t: slice (t, lane) is live; bit 8 + t: it lost records to a full slice
    // compact: records of the slice known to be in global memory.  Bits 31..3: flushed pieces * 8; bits 2..0: how
    // many records of the CURRENT piece are in global memory already (after a direct-routed word)
    u32 flushed[QT];
    u32 lane_off;                        // compact: byte offset of the lane's slices relative to the wavefront's first (t = 0, lane 0) slice
    i64 wave_base;                       // ... whose offset in cand8 this is (wave-uniform); tile t adds t * 32 * crow

    __device__ __forceinline__ void init(u8* lds_, const MxDrainLds& L_, int wave_, int lane_, int qb_, int sp_,
                                         u32 cap_, i64 crow_, int probe_, u32 idx_base_, i64 segL_, u64* cand) {
        lds = lds_; L = L_; wave = wave_; lane = lane_; qb = qb_; sp = sp_;
        cap = cap_; crow = crow_; probe = probe_; idx_base = idx_base_; segL = segL_;
        cand8 = (u8*)cand;
        queue = (u64*)(lds + L.queue) + wave * QCAP;
        rings = lds + L.rings + wave * mx_ring_bytes(QT, COMPACT);
        qfill = 0;
        flags = 0;
        const int h = lane >> 5, j = lane & 31;
#pragma unroll
        for (int t = 0; t < QT; ++t) { cnt[t] = 0; flushed[t] = 0; }
        lane_off = (u32)j * (u32)crow + (u32)h * cap;                    // the launcher keeps 64 * crow below 2^31
        wave_base = (i64)(qb * WPB + wave) * WQ * crow + (i64)(2 * sp) * cap;
    }
    __device__ __forceinline__ void set_live(const int t, const bool live) { flags |= live ? 1u << t : 0u; }
    __device__ __forceinline__ bool lost(const int t) const { return (flags >> (8 + t)) & 1u; }
    __device__ __forceinline__ u8* slice(const int t) const { return cand8 + wave_base + (i64)t * 32 * crow + lane_off; }

    // ---- compact mode: owner-side flushes ----
    __device__ __forceinline__ void flush_pieces(const int t) {      // completed 8-record pieces of slice (t, lane)
        u32 f = flushed[t];
        while (cnt[t] - (f & ~7u) >= (u32)MX_PIECE) {
            const u32 fl = f & ~7u, lo = f & 7u;
            const u8* ring = rings + (t * 64 + lane) * MX_RING;
            u8* out = slice(t) + fl;
            if (lo == 0) {
                *(u64*)out = *(const u64*)(ring + (fl & (MX_RING - 1)));
            } else {
                for (u32 p = lo; p < (u32)MX_PIECE; ++p) out[p] = ring[(fl + p) & (MX_RING - 1)];
            }
            f = fl + MX_PIECE;
        }
        flushed[t] = f;
    }
    __device__ __forceinline__ void flush_all_pieces() {
        bool any_need = false, odd = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            any_need |= cnt[t] - (flushed[t] & ~7u) >= (u32)MX_PIECE;
            odd |= (flushed[t] & 7u) != 0u;
        }
        if (!__any(any_need)) return;
        if (__builtin_expect(__any(odd) != 0, 0)) {                   // a direct-routed word left a piece half written: general form
#pragma unroll
            for (int t = 0; t < QT; ++t) flush_pieces(t);
            return;
        }
        // common case, straight line: whole pieces only, one 8-byte LDS read + one aligned 8-byte store each
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            u32 f = flushed[t];
            const u32 have = cnt[t] - f;
            const u8* ring = rings + (t * 64 + lane) * MX_RING;
            u8* tb = cand8 + wave_base + (i64)t * 32 * crow;          // wave-uniform base; the lane's part fits 32 bits
            if (have >= (u32)MX_PIECE) {
                *(u64*)(tb + (lane_off + f)) = *(const u64*)(ring + (f & 8u));
                if (have >= 2u * MX_PIECE) *(u64*)(tb + (lane_off + f + 8u)) = *(const u64*)(ring + ((f + 8u) & 8u));
                f += have & ~7u;
            }
            flushed[t] = f;
        }
    }
    // the ring's leftovers [flushed .. cnt) to global memory, byte by byte (rare path / end of the kernel)
    __device__ __forceinline__ void flush_tail(const int t) {
        const u32 f = flushed[t];
        const u32 fl = f & ~7u, lo = f & 7u;
        const u8* ring = rings + (t * 64 + lane) * MX_RING;
        u8* out = slice(t);
        for (u32 p = fl + lo; p < cnt[t]; ++p) out[p] = ring[p & (MX_RING - 1)];
    }

    // DIRECT: the queue may hold direct-routed entries (only drain_slow makes them); the common instantiation carries
    // none of their address arithmetic
    // FLUSH: write the completed pieces out right away (the paths that need < 8 pending records per slice at their start)
    template <bool DIRECT, bool FLUSH = true> __device__ __forceinline__ void emit(const i64 win, const u8* st) {
        wave_lds_sync();
        const u32 n = (kProbes && (probe & 8)) ? 0u : qfill;
        for (u32 i = lane; i < n; i += 64) {
            const u64 e = queue[i];
            u32 word = (u32)(e >> 32);
            const u32 desc = (u32)e;
            u32 pos = desc & ((1u << MX_POS_BITS) - 1u);
            const u32 src = (desc >> MX_POS_BITS) & 63u;
            const u32 t = (desc >> (MX_POS_BITS + 6)) & 3u, w = (desc >> (MX_POS_BITS + 8)) & 3u;
            const bool direct = COMPACT && DIRECT && ((desc >> (MX_POS_BITS + 10)) & 1u);
            const u32 hs = src >> 5;                                  // the source lane's half = segment
            const int ql = wave * WQ + (int)t * 32 + (int)(src & 31u);    // its query, block-local
            u32 qcw[NW];
            u64 qlw[LWA];
#pragma unroll
            for (int k = 0; k < NW; ++k) qcw[k] = ((const u32*)(lds + L.qcodes + ql * CB))[k];
#pragma unroll
            for (int k = 0; k < LWA; ++k) qlw[k] = LW > 0 ? ((const u64*)(lds + L.qlabels + ql * LB))[k] : 0ull;
            const i64 q = (i64)qb * (WPB * WQ) + ql;
            const i64 seg = 2 * sp + (int)hs;
            const i64 slice0 = COMPACT ? 0 : q * crow + seg * cap;   // first record of the slice (compact: computed on the rare direct route)
            u8* ring = rings + (t * 64 + src) * MX_RING;
            const u32 row0 = hs * WROWS + w * 32;                     // first row of the word in the stage tables
            const u32 idx0 = idx_base + (u32)(seg * segL + win * WROWS) + w * 32;
            while (word) {
                const int k = 31 - __builtin_clz(word);
                word ^= 1u << k;
                const u32 r = 31 - k;                                 // highest bit = earliest row
                const u32* rp = (const u32*)(st + L.codes + (row0 + r) * CB);
                u32 d = 0;
#pragma unroll
                for (int c = 0; c < NW; ++c) d += __builtin_popcount(qcw[c] ^ rp[c]);
                u64 any = 0;
                if (LW > 0) {
                    const u64* lp = (const u64*)(st + L.labels + (row0 + r) * LB);
#pragma unroll
                    for (int c = 0; c < LWA; ++c) any |= lp[c] & qlw[c];
                }
                if (!(kProbes && (probe & 4)) || d == 0x7fffffffu) {      // (the push trimmed the word to what the slice holds)
                    if (COMPACT) {
                        const u8 rec = make_rec8(d, any != 0);
                        if (DIRECT && __builtin_expect(direct, 0)) cand8[q * crow + seg * cap + pos] = rec;
                        else ring[pos & (MX_RING - 1)] = rec;
                    } else {
                        ((u64*)cand8)[slice0 + pos] = make_rec(idx0 + r, d, any != 0);
                    }
                }
                ++pos;
            }
        }
        wave_lds_sync();
        qfill = 0;
        if (COMPACT && FLUSH) {
            flush_all_pieces();
            wave_lds_sync();                                          // ring reads done before the next emit overwrites slots
        }
    }

    // the n earliest rows (highest bits) of a hit mask, n < popcount(word): what still fits a full slice
    __device__ __forceinline__ static u32 first_hits(const u32 word, u32 n) {
        u32 m = 0;
        while (n--) m |= 0x80000000u >> __builtin_clz(word ^ m);
        return m;
    }

    template <class T> __device__ __forceinline__ static T pick(const T (&arr)[QT], const int t) {   // arr[t], t not a constant
        T v = arr[0];
#pragma unroll
        for (int k = 1; k < QT; ++k) v = t == k ? arr[k] : v;
        return v;
    }
    template <class T> __device__ __forceinline__ static void put(T (&arr)[QT], const int t, const T v) {
#pragma unroll
        for (int k = 0; k < QT; ++k) arr[k] = t == k ? v : arr[k];
    }

    // One queue entry for the lane's mask word `word` (query tile t, word index w of the window), if it has hits.
    __device__ __forceinline__ void push(const u32 word, const u64 bal, const int t, const int w) {
        const u32 slot = qfill + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
        const u32 want = cnt[t] + (u32)__builtin_popcount(word);
        const u32 got = want < cap ? want : cap;                // the slice holds `cap` records; the rest is lost
        u32 kept = word;                                        // (dead lanes never hit: their bias keeps every accumulator >= 0)
        if (__builtin_expect(want != got, 0)) {                 // full slice: the word's first got - cnt hits only
            flags |= 0x100u << t;
            kept = first_hits(word, got - cnt[t]);
        }
        if (word != 0u)                                         // (an entry even when nothing is kept: the slot is counted)
            queue[slot] = ((u64)kept << 32) | (u64)(cnt[t] | ((u32)lane << MX_POS_BITS) | ((u32)t << (MX_POS_BITS + 6)) |
                                                    ((u32)w << (MX_POS_BITS + 8)));
        cnt[t] = got;
        qfill += (u32)__builtin_popcountll(bal);
    }

    // The hit masks of a window: m[t][w] = the lane's mask word w (32 rows) of query tile t.  Sparse windows -- the
    // case the kernel is tuned for, ~100 non-zero words per wavefront at C2 -- drain in one go: eight check-free pushes
    // and one emit.  What makes that safe is decided once, wave-uniformly: the queue takes the window's entries; no
    // slice fills up (nothing to trim); compact: every slice's ring takes its new records on top of what is pending.
    // Windows that do not fit drain in two halves, each with its own decision between the checked pushes and the
    // word-by-word form.  (Tried and dropped: a lane-private walk of the words for dense windows, R/N of several per
    // cent -- its registers spilled in the sparse path, 0.89 -> 1.38 ms at C2, and it was no faster where it ran.
    // Flushing pieces lazily -- only when a ring could not take the next window -- was tried: with
    // 128 slices per wavefront some ring is nearly always close to full, it saved nothing.)
    __device__ __forceinline__ void drain_window(const u32 (&m)[QT][4], const i64 win, const u8* st) {
        if (QT > 2) {                                                 // four query tiles: a window's entries rarely fit the queue
#pragma unroll 1
            for (int hw = 0; hw < 2; ++hw) {
                u32 wd[QT][2];
#pragma unroll
                for (int t = 0; t < QT; ++t) { wd[t][0] = hw ? m[t][2] : m[t][0]; wd[t][1] = hw ? m[t][3] : m[t][1]; }
                drain(wd, 2 * hw, win, st);
            }
            return;
        }
        u64 bal[QT][4];
        u32 nz = 0, want[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            want[t] = cnt[t];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                bal[t][w] = __ballot(m[t][w] != 0u);
                nz += (u32)__builtin_popcountll(bal[t][w]);
                want[t] += (u32)__builtin_popcount(m[t][w]);
            }
        }
        auto fits = [&]() {
            bool bad = false;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                bad |= want[t] > cap;
                if (COMPACT) bad |= want[t] - (flushed[t] & ~7u) > (u32)MX_RING;
            }
            return __any(bad) == 0;
        };
        if (__builtin_expect(!(nz <= (u32)QCAP && fits()), 0)) {
#pragma unroll 1
            for (int hw = 0; hw < 2; ++hw) {
                u32 wd[QT][2];
#pragma unroll
                for (int t = 0; t < QT; ++t) { wd[t][0] = hw ? m[t][2] : m[t][0]; wd[t][1] = hw ? m[t][3] : m[t][1]; }
                drain(wd, 2 * hw, win, st);
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            u32 pos = cnt[t];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 word = m[t][w];
                const u64 b = bal[t][w];
                const u32 slot = qfill + __builtin_amdgcn_mbcnt_hi((u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((u32)b, 0u));
                if (word != 0u)
                    queue[slot] = ((u64)word << 32) | (u64)(pos | ((u32)lane << MX_POS_BITS) | ((u32)t << (MX_POS_BITS + 6)) |
                                                            ((u32)w << (MX_POS_BITS + 8)));
                pos += (u32)__builtin_popcount(word);
                qfill += (u32)__builtin_popcountll(b);
            }
            cnt[t] = want[t];
        }
        emit<false>(win, st);
    }

    // Half a window: wd[t][i] = the lane's mask word w0 + i of query tile t.
    __device__ __forceinline__ void drain(const u32 (&wd)[QT][2], const int w0, const i64 win, const u8* st) {
        u64 bal[QT][2];
        u32 nz = 0;
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bal[t][i] = __ballot(wd[t][i] != 0u);
                nz += (u32)__builtin_popcountll(bal[t][i]);
            }
        bool slow = QCAP < 64 * QT * 2 && nz > (u32)QCAP;
        if (COMPACT) {
            bool over = false;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const u32 want = cnt[t] + (u32)__builtin_popcount(wd[t][0]) + (u32)__builtin_popcount(wd[t][1]);
                over |= (want < cap ? want : cap) - (flushed[t] & ~7u) > (u32)MX_RING;     // dead lanes: no hits, no change
            }
            slow |= __any(over) != 0;
        }
        if (__builtin_expect(!slow, 1)) {
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) push(wd[t][i], bal[t][i], t, w0 + i);
            emit<false>(win, st);
        } else {
            drain_slow(wd, w0, win, st);
        }
    }

    // Rare: dense windows (queue) or bursts of hits in one slice (ring).  One word at a time, each followed by its own
    // emit, so a slice's ring starts every step with < 8 records; a word that alone brings >= 10 is routed directly.
    __device__ __forceinline__ void drain_slow(const u32 (&wd)[QT][2], const int w0, const i64 win, const u8* st) {
#pragma unroll 1
        for (int k = 0; k < QT * 2; ++k) {
            const int t = k >> 1, i = k & 1;
            u32 w2[QT];
#pragma unroll
            for (int x = 0; x < QT; ++x) w2[x] = (wd[x][1] & (0u - (u32)i)) | (wd[x][0] & ((u32)i - 1u));   // wd[x][i] without an indexed (scratch) array
            const u32 word = pick(w2, t);
            const u64 bal = __ballot(word != 0u);
            const u32 c0 = pick(cnt, t);
            const u32 want = c0 + (u32)__builtin_popcount(word);
            const u32 got = want < cap ? want : cap;
            const u32 kept = want != got ? first_hits(word, got - c0) : word;
            bool direct = false;
            if (COMPACT) {
                const u32 f = pick(flushed, t);
                direct = got - (f & ~7u) > (u32)MX_RING;                 // >= 10 hits of its own: straight to global memory
                if (direct) {
                    const u8* ring = rings + (t * 64 + lane) * MX_RING;  // first the ring's < 8 leftovers, byte by byte
                    u8* out = cand8 + wave_base + (i64)t * 32 * crow + lane_off;
                    for (u32 p = f; p < c0; ++p) out[p] = ring[p & (MX_RING - 1)];     // f = pieces * 8 + records already out
                    put(flushed, t, got);                                 // [.. got) is, or will be by the direct entry, in global memory
                }
            }
            const u32 slot = qfill + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
            if (word != 0u)
                queue[slot] = ((u64)kept << 32) | (u64)(c0 | ((u32)lane << MX_POS_BITS) | ((u32)t << (MX_POS_BITS + 6)) |
                                                        ((u32)(w0 + i) << (MX_POS_BITS + 8)) | ((direct ? 1u : 0u) << (MX_POS_BITS + 10)));
            flags |= want != got ? 0x100u << t : 0u;
            put(cnt, t, got);
            qfill += (u32)__builtin_popcountll(bal);
            emit<true>(win, st);
        }
    }

    // end of the kernel: what is still in the rings
    __device__ __forceinline__ void finish() {
        if (!COMPACT) return;
        wave_lds_sync();
        flush_all_pieces();                                           // (lazy flushes: up to two pieces may be pending)
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u32 f = flushed[t];
            if (cnt[t] > (f & ~7u)) {
                if ((f & 7u) == 0) {          // the whole piece in one store: slots past cnt are inside the slice's capacity (a multiple of 16)
                    const u8* ring = rings + (t * 64 + lane) * MX_RING;
                    *(u64*)(slice(t) + f) = *(const u64*)(ring + (f & (MX_RING - 1)));
                } else {
                    flush_tail(t);
                }
            }
        }
    }
};

}  // namespace hg
