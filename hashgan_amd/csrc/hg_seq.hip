// libhashgan_amd.so -- the Hamming sequences: segment geometry, histogram -> plan -> select -> rank, in their staged
// (sharded) and one-shot forms (lib/metric.py:13-19 as a counting selection).
#include "hg_ctx.hpp"
#include <chrono>
#include "hg_select_mx.hpp"
#include "hg_select_mx3.hpp"
#include "hg_select_mx4.hpp"
#include "hg_hist_mx.hpp"
#include "hg_rank_cnt.hpp"
#include "hg_rank_lean.hpp"
#include "hg_rank_dense.hpp"

// Segment geometry of the pair passes: ~target_units wavefront-sized units.
constexpr i64 SAMPLE_RATIO = 2;      // the sampled pass works on segments this many times longer than the select pass's (= per segment pair)

void make_geometry(hg_ctx* c) {
    Geo& g = c->geo;
    g.Q = (int)c->Q;
    g.nQT = (int)((c->Q + 63) / 64);
    g.Qpad = g.nQT * 64;
    g.NW = c->NW; g.NB = c->NB; g.LW = c->LW;
    g.N = c->N; g.R = c->R; g.idx_base = c->idx_base;
    i64 S = (c->target_units + g.nQT - 1) / g.nQT;
    // few queries: more segments than fill the GPU twice only make every query's record row longer to walk
    // (Q = 64, N = 10M: 12500 slices per query cost the rank stage 3.4 ms; 2048 cost 0.1)
    if (S > c->opt_max_segments) S = c->opt_max_segments;
    const i64 maxS = (c->N + c->min_segment - 1) / c->min_segment;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    i64 L = (c->N + S - 1) / S;
    const i64 lq = (c->opt_select_packed >= 3 && c->NW <= 2) ? 96 : 32;   // k_select_mx walks segments in 32-row tiles, k_select_mx3 in 48-row supertiles
    L = (L + lq - 1) / lq * lq;
    if (L < lq) L = lq;
    S = (c->N + L - 1) / L;
    if (S < 1) S = 1;
    if (c->opt_enable && c->opt_select_mfma && S >= 4) {
        // k_select_mx runs (S / 2) x ceil(Q / 256 or 512) equal blocks, 4 or 2 resident per CU: pick the S near the
        // target that fills a whole number of such rounds, so the last round is not a nearly empty one
        const bool qt2 = c->NW <= 4;                                // mirrors launch_select_mx_t
        const bool mx3 = c->opt_select_packed == 3 && c->NW <= 4;   // k_select_mx3 / k_select_mx4: blocks of 8 wavefronts x 64 queries, two per CU
        const i64 qblk = mx3 ? 64 * M3_WPB : qt2 ? 256 : 512;
        const i64 nQB = (c->Q + qblk - 1) / qblk;
        const i64 slots = (i64)c->n_cu * (mx3 ? 16 / M3_WPB : qt2 ? 4 : 2);
        i64 k = (S / 2 * nQB + slots / 2) / slots;
        if (k < 1) k = 1;
        // k_rank_lean takes at most 256 slices per query.  A target beyond that (few queries: C3's 2100 ask for 496 segments)
        // is cut to the whole rounds that 256 segments fill, when that is at least one: C3 396 -> 198 segments, one round of
        // blocks instead of two, k_rank_lean instead of k_rank_cnt: 0.277 -> 0.200 ms per step
        if (mx3 && c->opt_rank_lean && S > 256 && 128 * nQB >= slots) {
            k = 128 * nQB / slots;
            L = (c->N + 255) / 256;
            L = (L + lq - 1) / lq * lq;
            S = (c->N + L - 1) / L;                // <= 256; the search below lands on the even count that fills k rounds
        }
        i64 S2 = 2 * (slots * k / nQB);
        if (S2 > maxS) S2 = maxS / 2 * 2;
        for (; S2 >= 4; S2 -= 2) {                 // rounding L up to 16 rows can drop segments: land on an even count
            i64 L2 = (c->N + S2 - 1) / S2;
            L2 = (L2 + lq - 1) / lq * lq;
            const i64 Sr = (c->N + L2 - 1) / L2;
            if (Sr * 4 < S * 3) break;             // too far from the target: keep the plain choice
            if (Sr % 2 == 0 && Sr * 4 <= S * 5 && (Sr / 2) * nQB <= slots * k) { S = Sr; L = L2; break; }
        }
    }
    g.S = (int)S; g.L = L;
    g.nUnits = (i64)g.S * g.nQT;
    g.hist_stride = 1;
    g.hcap = 0;
    g.wpb = WPB;
    g.nBlk = (int)((g.nUnits + WPB - 1) / WPB);
}

// The sampled pass only needs the shard total: use 2x longer segments so the per-segment
// histogram array (and its reduction) shrinks with the work.
Geo hist_geometry(const hg_ctx* c) {
    Geo g = c->geo;
    if (g.hist_stride > 1 || c->hist_pairs) {
        const i64 L = g.L * (c->hist_pairs ? 2 : SAMPLE_RATIO);
        g.L = L;
        g.S = (int)((g.N + L - 1) / L);
        g.nUnits = (i64)g.S * g.nQT;
    }
    return g;
}

// histogram on the matrix cores: blocks = (pair of segments) x (256 queries); stride in tiles of 16 rows
bool hist_mx_applies(const hg_ctx* c, int stride, bool pairs_ok) {
    if (!c->opt_hist_mfma || !c->opt_select_mfma || c->NW > 8 || c->is_sub) return false;
    {   // long segments (>= 65536 visited rows per pair) need one dword counter per query tile: with long codes the four
        // wavefronts' columns then exceed the CU's LDS -- the vector kernel, which shrinks its block, takes those
        const Geo& g = c->geo;
        const i64 tiles_per_half = ((g.L + 15) / 16 + stride - 1) / stride;
        const bool pack16 = 2 * tiles_per_half * 16 < 65536;
        if ((size_t)WPB * (pack16 ? 1 : 2) * g.NB * 32 * 4 > 160u * 1024u) return false;
    }
    return stride > 1 ? SAMPLE_RATIO == 2 : pairs_ok;
}

// The record pass of the current sequence: which kernel takes it (the launchers live in hg_pairs_valu.hip / hg_pairs_mx.hip).
// one-byte records {match, dist} (no index): only the matrix-core kernels of the bet produce them, and only when nobody wants the lists
static bool select_takes_mx(const hg_ctx* c) { return c->optimistic && c->opt_select_mfma && c->cap < (1u << MX_POS_BITS); }
static bool records_are_bytes(const hg_ctx* c) {
    return select_takes_mx(c) && c->opt_compact && !c->want_lists && c->LW <= 2 && c->cap % 16 == 0 && c->crow * 64 < (1ll << 31);
}
// the record rows of the coming select: Q x crow slots of one byte or eight.  (Until round 6 eight bytes were reserved either way:
// 2 GB at C2 for 0.25 GB of records -- and 14.9 GB once a class-sorted database had widened the slices, a hipMalloc that took
// between 0.4 ms and 3.5 s.)
static int reserve_records(hg_ctx* c) {
    return c->cand.reserve((size_t)c->geo.Q * (size_t)c->crow * (records_are_bytes(c) ? 1 : 8) + 64);
}

int launch_select(hg_ctx* c) {
    const int NW = c->NW;
    const int lw = c->LW <= 2 ? c->LW : 0;           // > 128 classes: match bits come from k_match
    const bool mx = select_takes_mx(c);
    c->rec8 = records_are_bytes(c);
    if (!c->optimistic && c->R * 4 >= c->n_total) { c->last_select = 2; return launch_select_dense(c, lw); }   // dense regime: most pairs are selected
    // three rows per accumulator + batched drain: codes of <= 64 bits, one-byte records (<= 128 classes).  (For <= 32 bits the
    // second k-half of every MFMA is empty, and it still beat round 2's two-rows-per-accumulator kernel: 0.69 vs 0.85 ms at b = 32.)
    if (NW <= 2 && c->opt_select_packed == 3 && c->rec8 && c->geo.L % M3_ROWS == 0 && (lw == 1 || lw == 2)) { c->last_select = 5; return launch_select_mx3(c, lw); }
    // codes of 65..128 bits: two rows per accumulator (8-bit fields) and the same drain
    if ((NW == 3 || NW == 4) && c->opt_select_packed == 3 && c->rec8 && c->geo.L % M4_ROWS == 0 && (lw == 1 || lw == 2)) { c->last_select = 6; return launch_select_mx4(c, lw); }
    if (mx) { c->last_select = 3; return launch_select_mx(c, lw); }
    c->last_select = 1;
    return launch_select_valu(c, lw, c->optimistic);
}

// rows k_hist visits with batch stride `stride` (mirrors its loop)
template <int NW> i64 sampled_rows_t(Geo g, int stride, int ratio) {
    g.hist_stride = stride;
    {
        const i64 L = g.L * ratio;                  // mirrors hist_geometry()
        g.L = L;
        g.S = (int)((g.N + L - 1) / L);
    }
    constexpr int B = Batch<NW>::rows;
    i64 total = 0;
    for (int s = 0; s < g.S; ++s) {
        const i64 lo = (i64)s * g.L, hi = lo + g.L < g.N ? lo + g.L : g.N;
        const i64 nb = (hi - lo) / B;
        total += (nb + stride - 1) / stride * B;
    }
    return total;
}
i64 sampled_rows(hg_ctx* c, int stride) {
    if (hist_mx_applies(c, stride, false)) return hist_mx_sampled_rows(c->geo, stride);
    switch (c->NW) {
        case 1: return sampled_rows_t<1>(c->geo, stride, (int)SAMPLE_RATIO);
        case 2: return sampled_rows_t<2>(c->geo, stride, (int)SAMPLE_RATIO);
        case 3: return sampled_rows_t<3>(c->geo, stride, (int)SAMPLE_RATIO);
        case 4: return sampled_rows_t<4>(c->geo, stride, (int)SAMPLE_RATIO);
        case 5: return sampled_rows_t<5>(c->geo, stride, (int)SAMPLE_RATIO);
        case 6: return sampled_rows_t<6>(c->geo, stride, (int)SAMPLE_RATIO);
        case 7: return sampled_rows_t<7>(c->geo, stride, (int)SAMPLE_RATIO);
        default: return sampled_rows_t<8>(c->geo, stride, (int)SAMPLE_RATIO);
    }
}


// =============================================================================
extern "C" {

static int do_hist(hg_ctx* c, int stride, bool reduce = true, bool pairs_ok = false, int hcap = 0) {
    make_geometry(c);
    c->geo.hist_stride = stride;
    const bool mx = hist_mx_applies(c, stride, pairs_ok);
    c->geo.hcap = mx && !reduce && stride > 1 ? hcap : 0;
    c->hist_pairs = mx && stride == 1;                 // full pass per segment pair: the plan's per-segment steps follow suit
    const Geo& g = c->geo;
    const size_t plane = (size_t)g.NB * g.Qpad * 4;
    HG_TRY(c->hist.reserve(plane * g.S));
    HG_TRY(c->hown.reserve(plane + TAIL_WORDS * 4));
    if (reduce) {   // tail of the exported histogram: [0] overflow flag, [1] rows this pass visited
        const u32 visited = (u32)(stride == 1 ? g.N : sampled_rows(c, stride));
        // written by a kernel: ordered with the kernels that read it, no pageable staging memory to keep alive
        hipLaunchKernelGGL(k_set_tail, dim3(1), dim3(64), 0, c->stream, (u32*)(c->hown.as<char>() + plane), visited);
        HG_TRY(c->check_launch("k_set_tail"));
    }
    HG_TRY(mx ? launch_hist_mx(c) : launch_hist(c));
    if (!reduce) { c->stage = ST_DB | ST_Q; return HG_OK; }      // the caller reads the per-segment histograms itself
    const Geo gh = hist_geometry(c);
    c->t_begin(KI_HIST_REDUCE);
    hipLaunchKernelGGL(k_hist_reduce, dim3(grid_for((i64)g.NB * g.Qpad)), dim3(256), 0, c->stream,
                       c->hist.as<u32>(), c->hown.as<u32>(), gh);
    c->t_end();
    HG_TRY(c->check_launch("k_hist_reduce"));
    c->stage = ST_DB | ST_Q | (stride == 1 ? ST_HIST : 0);
    return HG_OK;
}

int hg_hist(hg_ctx* c) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_hist", "hg_set_database + hg_set_queries"));
    HG_TRY(do_hist(c, 1));
    return c->stage_end();
}

int hg_hist_buffer(hg_ctx* c, void** dev_ptr, int64_t* nbytes) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_hist_buffer", "hg_hist / hg_sample_hist / hg_select_candidates"));
    if (!c->hown.p) return fail(HG_ERR_STATE, "hg_hist_buffer: no histogram computed yet");
    if (dev_ptr) *dev_ptr = c->hown.p;
    if (nbytes) *nbytes = ((int64_t)c->geo.NB * c->geo.Qpad + TAIL_WORDS) * 4;
    return HG_OK;
}

extern "C++" int set_R(hg_ctx* c, int64_t R, int G, int rank) {
    if (G < 1 || rank < 0 || rank >= G) return fail(HG_ERR_ARG, "rank %d of %d", rank, G);
    if (R < 1 || R > c->n_total)
        return fail(HG_ERR_ARG, "R=%lld outside 1..N (N=%lld rows in the database)", (long long)R, (long long)c->n_total);
    c->R = R; c->G = G; c->rank = rank;
    c->geo.R = R;
    c->RW = (R + 63) / 64;
    return HG_OK;
}

// k_plan on c->hown (full histogram, or the records' histogram in optimistic mode)
static int launch_plan(hg_ctx* c, const uint32_t* dev_hist_all) {
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->posbase.reserve((size_t)g.NB * qb));
    HG_TRY(c->t.reserve(qb)); HG_TRY(c->cnt_lt.reserve(qb)); HG_TRY(c->quota.reserve(qb));
    HG_TRY(c->tie_before.reserve(qb)); HG_TRY(c->n_lt.reserve(qb)); HG_TRY(c->err.reserve(16));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    Plan pl{c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->n_lt.as<u32>(),
            c->posbase.as<u32>(), c->err.as<int>()};
    c->t_begin(KI_PLAN);
    hipLaunchKernelGGL(k_plan, dim3(grid_for(g.Q)), dim3(256), 0, c->stream, c->hown.as<u32>(),
                       (const u32*)dev_hist_all, c->G, c->rank, pl, g);
    c->t_end();
    return c->check_launch("k_plan");
}

// exact plan: threshold from the full histogram, then the exact record-row layout
static int do_plan(hg_ctx* c, int64_t R, const uint32_t* dev_hist_all, int G, int rank) {
    if (G > 1 && !dev_hist_all) return fail(HG_ERR_ARG, "hg_plan: G > 1 needs the gathered histograms");
    HG_TRY(set_R(c, R, G, rank));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(launch_plan(c, dev_hist_all));
    HG_TRY(c->seglt.reserve((size_t)g.S * qb)); HG_TRY(c->segtie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_start.reserve((size_t)g.S * qb)); HG_TRY(c->sl_tie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->failq.reserve(qb));
    HG_TRY(c->sstar.reserve(qb));
    const Geo gp = hist_geometry(c);                   // per segment -- or per segment pair after k_hist_mx (then only sstar is used)
    const int ratio = (int)(gp.L / g.L);
    c->t_begin(KI_SEG_COUNTS);
    hipLaunchKernelGGL(k_seg_counts, dim3(grid_for((i64)gp.S * g.Qpad)), dim3(256), 0, c->stream, c->hist.as<u32>(),
                       c->t.as<int>(), c->seglt.as<u32>(), c->segtie.as<u32>(), gp);
    c->t_end();
    HG_TRY(c->check_launch("k_seg_counts"));
    c->t_begin(KI_SEG_LAYOUT);
    hipLaunchKernelGGL(k_seg_layout, dim3(grid_for(g.Qpad)), dim3(256), 0, c->stream, c->seglt.as<u32>(),
                       c->segtie.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->sl_start.as<u32>(),
                       c->sl_tie.as<u32>(), c->tot.as<u32>(), c->sstar.as<int>(), ratio, gp);
    c->t_end();
    HG_TRY(c->check_launch("k_seg_layout"));
    c->optimistic = false;
    c->crow = R;
    c->cap = 0;
    c->stage = ST_DB | ST_Q | ST_HIST | ST_PLAN;
    return HG_OK;
}

static int check_plan_flag(hg_ctx* c) {
    int err = 0;
    HG_TRY(read_plan_flag(c, &err));
    if (err) {
        c->stage = ST_DB | ST_Q | ST_HIST;
        return fail(HG_ERR_ARG, "R=%lld exceeds the rows present in the gathered histograms", (long long)c->R);
    }
    return HG_OK;
}

int hg_plan(hg_ctx* c, int64_t R, const uint32_t* dev_hist_all, int G, int rank) {
    HG_TRY(need(c, ST_HIST, "hg_plan", "hg_hist"));
    // a one-shot exact call may have left histograms per segment PAIR (k_hist_mx); the staged select lays its slices out
    // per segment
    if (c->hist_pairs) return fail(HG_ERR_STATE, "hg_plan called before hg_hist (the last histogram pass belonged to a one-shot call)");
    HG_TRY(do_plan(c, R, dev_hist_all, G, rank));
    // the device flag says "R exceeds the rows in the gathered histograms"; R <= n_total was checked on
    // the host already, so an unsynchronised caller loses nothing by skipping the read-back
    return c->stage_sync ? check_plan_flag(c) : HG_OK;
}

// k_rank_fused in one of its modes: 0 = histogram + plan + placement in one launch (single shard),
// 1 = histogram phase (several shards, before the exchange), 2 = placement phase (after k_plan).
// the dense regime through the byte matrix (hg_rank_dense.hpp): the counter columns always fit a block's LDS for codes of <= 126 bits
static bool rank_dense_fits(const hg_ctx* c, int64_t R) {
    (void)R;
    return c->opt_rank_dense && c->LW <= 2 && c->NW <= 4 && c->b <= 126 && c->N == c->n_total && !c->is_sub;
}

static void launch_dense_bytes(hg_ctx* c, u8* D, i64 Npad, int q0, int nq) {
    const int qper = 64;
    const dim3 grid((unsigned)(Npad / 1024), (unsigned)((nq + qper - 1) / qper));
    const u32 padbyte = (u32)c->b;                     // the distance of the rows past N (k_rank_dense: padrow)
#define HG_DENSE_BYTES(NW_, LW_)                                                                                                      \
    hipLaunchKernelGGL((k_dense_bytes<NW_, LW_>), grid, dim3(256), 0, c->stream, c->qc.as<u32>(), c->qlab.as<u64>(), c->db.as<u32>(), \
                       c->dblab.as<u64>(), D, c->N, Npad, q0, nq, qper, padbyte)
    const int key = c->NW * 2 + (c->LW <= 1 ? 0 : 1);
    switch (key) {
        case 2: HG_DENSE_BYTES(1, 1); break;
        case 3: HG_DENSE_BYTES(1, 2); break;
        case 4: HG_DENSE_BYTES(2, 1); break;
        case 5: HG_DENSE_BYTES(2, 2); break;
        case 6: HG_DENSE_BYTES(3, 1); break;
        case 7: HG_DENSE_BYTES(3, 2); break;
        case 8: HG_DENSE_BYTES(4, 1); break;
        default: HG_DENSE_BYTES(4, 2); break;
    }
#undef HG_DENSE_BYTES
}

static int launch_rank_dense(hg_ctx* c) {
    const Geo& g = c->geo;
    HG_TRY(c->err.reserve(16));
    HG_TRY(c->qbad.reserve((size_t)g.Qpad * 4));
    // Where the R-bit bitmap lives.  LDS while two blocks still share a CU, and the AP then leaves from the epilogue (C1: 0.12 ms
    // for ranking + AP).  A bitmap that leaves room for one block only: in global memory while the members are few (R < N/4:
    // the atomic ORs of the matching members cost ~3.6 ms per 10^9; two to four blocks per CU), else in LDS with one block per CU --
    // and k_ap afterwards either way (an epilogue on four wavefronts per CU is a latency chain: 62 chunks at R = 500k, 0.3 ms per query).
    // Q = 10k, N = 1M, b = 64, ranking + AP: R = 130k 7.4 ms global / 11.5 LDS; 200k 9.7 / 12.0; 500k 22.7 / 13.7.
    const int lds_cu = 160 * 1024;
    const int tot_lds = rank_dense_layout(g.NB, c->RW, false).total, tot_gbm = rank_dense_layout(g.NB, c->RW, true).total;
    const int blocks_lds = (c->RW * 8 + 4096 < lds_cu && tot_lds <= lds_cu) ? lds_cu / tot_lds : 0;
    // A bitmap beyond one block's LDS with R >= N/4: K blocks per query, each ranks everything and keeps its K-th of the ranks in LDS
    // (R = N = 1M: two blocks per query, 2 x 10 ms, against 37 ms of atomic ORs on a bitmap in global memory)
    i64 rw_part = 0;
    int kparts = 1;
    if (blocks_lds == 0 && g.R * 4 >= c->N && c->opt_rank_dense_gbm != 1) {
        const i64 room = lds_cu - tot_gbm - 4096;
        kparts = (int)((c->RW * 8 + room - 1) / room);
        if (kparts >= 2 && kparts <= 8) rw_part = (c->RW + kparts - 1) / kparts; else kparts = 1;
    }
    const bool gbm = rw_part ? false
                   : c->opt_rank_dense_gbm >= 0 ? c->opt_rank_dense_gbm != 0 || blocks_lds == 0 : (blocks_lds == 0 || (blocks_lds < 2 && g.R * 4 < c->N));
    const int total = gbm ? tot_gbm : rw_part ? rank_dense_layout(g.NB, rw_part, false).total : tot_lds;
    const i64 Npad = rank_dense_pieces(c->N) * RD_THREADS * 16;
    i64 qchunk = (c->opt_dense_budget_mb << 20) / Npad;
    if (qchunk < 1) qchunk = 1;
    if (qchunk > g.Q) qchunk = g.Q;
    // (the byte matrix is a budget, not a need: when the device cannot give that much, fewer queries per chunk do)
    for (;;) {
        const int rc = c->dbytes.reserve((size_t)qchunk * Npad);
        if (rc == HG_OK) break;
        if (qchunk == 1) return rc;
        (void)hipGetLastError();
        qchunk = (qchunk + 1) / 2;
    }
    bool use_recip = false;
    const bool fuse = !gbm && !rw_part && blocks_lds >= 2 && c->fuse_ap && c->opt_fuse_ap && !c->want_lists;
    if (fuse) HG_TRY(ensure_ap_tables(c, &use_recip));
    const bool fused = fuse && use_recip;
    if (gbm) HG_HIP(hipMemsetAsync(c->mbits.p, 0, (size_t)g.Q * c->RW * 8, c->stream));
    for (i64 q0 = 0; q0 < g.Q; q0 += qchunk) {
        const int nq = (int)(q0 + qchunk < g.Q ? qchunk : g.Q - q0);
        c->t_begin(KI_SELECT);
        launch_dense_bytes(c, c->dbytes.as<u8>(), Npad, (int)q0, nq);
        c->t_end();
        HG_TRY(c->check_launch("k_dense_bytes"));
        RankDenseArgs da{c->dbytes.as<u8>(), Npad, (int)q0, c->err.as<int>(), c->qbad.as<u32>(), c->RW,
                         fused ? c->shapes.as<ApShape>() : nullptr, fused ? c->ap_recip.as<double>() : nullptr, c->ap.as<double>(), c->rel.as<u32>(),
                         nullptr, nullptr, nullptr, 0u, 0, rw_part, nullptr, g.NB};
        c->t_begin(KI_RANK_FUSED);
#define HG_RANK_DENSE(LISTS_, GBM_)                                                                                                              \
    do {                                                                                                                                         \
        if (q0 == 0) HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rank_dense<LISTS_, GBM_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, total)); \
        hipLaunchKernelGGL((k_rank_dense<LISTS_, GBM_, false>), dim3(nq, kparts), dim3(RD_THREADS), (size_t)total, c->stream, da, c->out_idx.as<u32>(),  \
                           c->out_dist.as<u8>(), c->mbits.as<u32>(), g);                                                                         \
    } while (0)
        if (c->want_lists) { if (gbm) HG_RANK_DENSE(true, true); else HG_RANK_DENSE(true, false); }
        else { if (gbm) HG_RANK_DENSE(false, true); else HG_RANK_DENSE(false, false); }
#undef HG_RANK_DENSE
        c->t_end();
        HG_TRY(c->check_launch("k_rank_dense"));
    }
    c->last_rank = 7;
    c->ap_fused = fused;
    return HG_OK;
}

// k_rank_dense<slices> (hg_rank_dense.hpp) over the bet's one-byte records: the whole launch (long lists), or only the queries
// flagged in `only` (what k_rank_lean declined -- then always with the AP from the epilogue: a handful of blocks)
static int slices_rows(const hg_ctx* c) {
    // the bet's cut never exceeds b/2 + 1 -- enqueue_optimistic's sampled pass stops there --, so b/2 + 2 counter rows cover its records
    return (!c->exact_mx && c->geo.NB / 2 + 2 < c->geo.NB) ? c->geo.NB / 2 + 2 : c->geo.NB;
}
static bool rank_slices_fits(const hg_ctx* c) {
    return c->optimistic && c->rec8 && !c->want_lists && c->geo.S <= RD_THREADS && slices_rows(c) <= 126 &&
           rank_dense_layout(slices_rows(c) + 1, c->RW, false).total <= 160 * 1024;
}
static int launch_rank_slices(hg_ctx* c, const u32* only) {
    const Geo& g = c->geo;
    const int sl_rows = slices_rows(c);
    const int total = rank_dense_layout(sl_rows + 1, c->RW, false).total;
    const bool fuse = c->fuse_ap && c->opt_fuse_ap && (only || 160 * 1024 / total >= 2);
    bool use_recip = false;
    if (fuse) HG_TRY(ensure_ap_tables(c, &use_recip));
    const bool fused = fuse && use_recip;
    if (only && !fused) return fail(HG_ERR_STATE, "launch_rank_slices: the leftover form needs the AP tables");
    RankDenseArgs da{nullptr, 0, 0, c->err.as<int>(), c->qbad.as<u32>(), c->RW,
                     fused ? c->shapes.as<ApShape>() : nullptr, fused ? c->ap_recip.as<double>() : nullptr, c->ap.as<double>(), c->rel.as<u32>(),
                     c->cand.as<u8>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow, 0, only, sl_rows};
    HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rank_dense<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, total));
    c->t_begin(only ? KI_RANK_FUSED : KI_RANK_LDS);
    hipLaunchKernelGGL((k_rank_dense<false, false, true>), dim3(g.Q), dim3(RD_THREADS), (size_t)total, c->stream, da, c->out_idx.as<u32>(),
                       c->out_dist.as<u8>(), c->mbits.as<u32>(), g);
    c->t_end();
    HG_TRY(c->check_launch("k_rank_dense<slices>"));
    if (!only) { c->last_rank = 8; c->ap_fused = fused; }
    return HG_OK;
}

// The last fused step on this context had queries its rank kernel declined (one of C5's 10 000 spans more distances than
// k_rank_lean places): rank the flagged ones right behind it, in the same stream -- a launch of mostly returning blocks --
// instead of a second host round trip (k_rank_fused + k_ap + three downloads: 0.08 ms of C5's 1.27).
static int rank_leftovers_inline(hg_ctx* c, int mode, bool use_recip) {
    c->leftovers_inline = false;
    if (c->leftovers_expected && mode == 0 && c->opt_inline_leftovers && use_recip && rank_slices_fits(c)) {
        HG_TRY(launch_rank_slices(c, c->bigq.as<u32>()));
        c->leftovers_inline = true;
    }
    return HG_OK;
}

// Which LDS-resident rank kernel a bet's one-byte records will meet -- decided from what is known BEFORE the select runs (R, the
// slices' capacity, the segment count, options), because the select writes the interleaved record layout (SelArgs::il) only for
// the kernel that reads it, k_rank_lean (k_rank_fused, the general kernel behind every path, reads both layouts).
struct LeanPlan { bool ok; int nbc, psp; i64 rb; };
static LeanPlan rank_lean_plan(const hg_ctx* c, int mode) {
    const Geo& g = c->geo;
    LeanPlan l{false, 0, 1, 0};
    if (!(c->optimistic && c->opt_rank_cnt && c->opt_rank_lean && (mode == 0 || mode == 3) && c->rec8 && !c->want_lists &&
          g.S <= 256 && (c->cap & 15u) == 0 && c->cap <= 1024 && g.R <= 60000)) return l;
    const int nbc = rank_cnt_maxb(g.NB) + 2 < g.NB ? rank_cnt_maxb(g.NB) + 2 : 0;
    const int nbc_eff = nbc ? nbc : (g.NB < 128 ? g.NB : 128);
    // room for every piece of the row when that fits HG_RANK_WAVES blocks per CU; else for the usual list (2.2 R + padding)
    const double share = (double)c->N / (double)(c->n_total > 0 ? c->n_total : 1);
    const i64 all = (i64)g.S * c->cap;
    // pieces of a slice fetched up front: what it holds but for a 4-sigma exception, when the select keeps ~0.7 of the budgeted
    // mean (cap = mean + 6 sqrt(mean) + 16, inverted); at most what four loads per thread cover
    const double mb = std::pow(std::sqrt((double)c->cap > 7.0 ? (double)c->cap - 7.0 : 0.0) - 3.0, 2.0), est = 0.7 * mb;
    int psp = (int)std::ceil((est + 4.0 * std::sqrt(est) + 1.0) / 16.0);
    if (psp > (int)(c->cap >> 4)) psp = (int)(c->cap >> 4);
    if (psp > RL_MAX_PIECES / g.S) psp = RL_MAX_PIECES / g.S;
    if (psp < 1) psp = 1;
    const i64 room = ((160 * 1024 / HG_RANK_WAVES) & ~511ll) - rank_lean_layout(g.NB, c->RW, g.S, 0, nbc).total;
    i64 rb = all <= room ? all : room & ~15ll;
    const i64 least = ((i64)(2.2 * (double)c->R * share) + 256 + 16 * (i64)g.S + 15) & ~15ll;
    if (rb < least && least <= all) rb = least;
    if (rb > all) rb = all;
    if (rb > 16 * RL_MAX_PIECES) rb = 16 * RL_MAX_PIECES;      // a thread keeps at most four pieces of its query's list
    const RankLeanLds L = rank_lean_layout(g.NB, c->RW, g.S, (int)rb, nbc);
    l.nbc = nbc; l.psp = psp; l.rb = rb;
    l.ok = nbc_eff <= 63 && L.total <= 64 * 1024 && rb >= 16 && (rb >= least || rb == all);
    return l;
}
// leftovers_only: the second half of a fused step -- k_rank_cnt has run (with its AP epilogue) and flagged in bigq the queries
// it declined; rank just those with the general kernel
static int launch_rank(hg_ctx* c, int mode, int nbits, bool leftovers_only = false) {
    const Geo& g = c->geo;
    if (c->dense_rank && mode == 0) return launch_rank_dense(c);
    int nwav = (c->optimistic ? 3 * c->R : c->R) >= 16384 ? 16 : 4;   // k_rank_fused's wavefronts per query, by the records per query (~ 3R / R)
    if (leftovers_only && c->R >= 1024) nwav = 16;   // a handful of blocks (mode 0: no hwq): what counts is one block's latency
    const size_t fixed_words = (size_t)(nwav + 1) * g.NB + 8;
    const int bits_lds = (fixed_words + 2 * (size_t)c->RW) * 4 <= 64 * 1024;
    if (mode != 1 && !bits_lds && !leftovers_only) HG_HIP(hipMemsetAsync(c->mbits.p, 0, (size_t)g.Q * c->RW * 8, c->stream));
    HG_TRY(c->err.reserve(16));
    HG_TRY(c->qbad.reserve((size_t)g.Qpad * 4));
    if (mode != 0) HG_TRY(c->hwq.reserve((size_t)g.Q * nwav * g.NB * 4));
#ifdef HG_RANK_PROFILE
    HG_TRY(c->hwq.reserve((size_t)4096 * 16 * 4 + (size_t)g.Q * nwav * g.NB * 4));
#endif
    if (mode == 0 && !leftovers_only) {
        if (c->optimistic) { if (!c->err_zeroed) HG_HIP(hipMemsetAsync(c->err.p, 0, 8, c->stream)); }
        else HG_HIP(hipMemsetAsync(c->failq.p, 0, (size_t)g.Qpad * 4, c->stream));
        c->err_zeroed = false;
    }
    const u32* only = nullptr;
    bool counted = false;
    c->ap_fused = false;
    if (!leftovers_only) c->last_rank = 1;                // k_rank_fused unless one of the LDS-resident kernels takes the lists
    if (leftovers_only) { only = c->bigq.as<u32>(); counted = true; }
    const LeanPlan lp = rank_lean_plan(c, mode);
    if (!counted) {
        // the lean counting sort (k_rank_lean): the whole record row in one coalesced read, piecewise compaction, chunks in registers
        const int nbc = lp.nbc, psp = lp.psp;
        const i64 rb = lp.rb;
        const int* cut = nbc ? (c->exact_mx ? c->t.as<int>() : c->tguess.as<int>()) : nullptr;
        if (lp.ok) {
            const RankLeanLds L = rank_lean_layout(g.NB, c->RW, g.S, (int)rb, nbc);
            HG_TRY(c->bigq.reserve((size_t)g.Qpad * 4));
            const bool fuse = c->fuse_ap && c->opt_fuse_ap && mode == 0 && c->LW <= 2;
            bool use_recip = false;
            if (fuse) HG_TRY(ensure_ap_tables(c, &use_recip));
            RankLdsArgs la{c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->err.as<int>(), c->qbad.as<u32>(), c->bigq.as<u32>(),
                           c->cap, c->crow, 0, 1, c->RW, (int)rb, mode, c->hwq.as<u32>(), c->hown.as<u32>(),
                           c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->posbase.as<u32>(), nbc,
                           fuse ? c->shapes.as<ApShape>() : nullptr, fuse && use_recip ? c->ap_recip.as<double>() : nullptr,
                           c->ap.as<double>(), c->rel.as<u32>(), fuse ? c->err.as<u32>() + 1 : nullptr, cut, psp};
            c->t_begin(KI_RANK_LDS);
            hipLaunchKernelGGL(k_rank_lean, dim3(padded_grid(g.Q)), dim3(256), (size_t)L.total, c->stream, c->cand.as<u8>(), la, c->mbits.as<u32>(), g);
            c->t_end();
            HG_TRY(c->check_launch("k_rank_lean"));
            c->last_rank = 6;
            if (fuse) {
                c->ap_fused = true;
                HG_TRY(rank_leftovers_inline(c, mode, use_recip));
                return HG_OK;
            }
            only = c->bigq.as<u32>();                // k_rank_fused below ranks what this path declined
            counted = true;
        }
    }
    if (!counted && mode == 0 && g.R >= c->opt_rank_slices && c->opt_rank_slices > 0 && rank_slices_fits(c)) {
        // long lists of a bet (beyond k_rank_lean's LDS): k_rank_dense's two passes over the query's record slices, thread = part of a slice
        return launch_rank_slices(c, nullptr);
    }
    if (!counted && c->optimistic && c->opt_rank_cnt && (mode == 0 || mode == 3)) {
        // per-thread counting sort (k_rank_cnt): byte counters for every distance + a tile of the records, <= 64 KiB per
        // block; lists longer than a tile are ranked tile by tile
        const double share = (double)c->N / (double)(c->n_total > 0 ? c->n_total : 1);
        i64 r2 = (i64)(2.5 * (double)c->R * share) + 256;         // a tile of the records: the usual list (1.3 - 2 R) in one
        if (r2 < 4096) r2 = 4096;                                  // (small R: the guess's margin is relatively larger)
        r2 = r2 / 64 * 64;
        // every record of a bet lies within its query's cut (the guess; the exact threshold of the exact_mx sequence), and a
        // list reaching further down than the 16 (32) distances this kernel places leaves it anyway: byte counters for the
        // 18 (34) distances up to the cut are all it needs (round 3: b/2 + 2 of them -- 8.7 KB of LDS at b = 64, 16.9 KB at
        // b = 128); a query with a record below them goes to k_rank_fused
        const int nbc = rank_cnt_maxb(g.NB) + 2 < g.NB ? rank_cnt_maxb(g.NB) + 2 : 0;
        const int* cut = nbc ? (c->exact_mx ? c->t.as<int>() : c->tguess.as<int>()) : nullptr;
        RankCntLds L = rank_cnt_layout(g.NB, c->RW, g.S, (int)r2, c->want_lists ? 1 : 0, nbc);
        {   // one block more per CU when trimming the record tile by a few percent (never below 2.2 R) makes it fit
            const int per_cu = (int)(160 * 1024 / ((L.total + 511) & ~511));
            const i64 want = (160 * 1024 / (per_cu + 1)) & ~511ll;
            const i64 r3 = (r2 - (L.total - want)) / 64 * 64;
            // (the kernel is compiled for HG_RANK_WAVES wavefronts per SIMD = blocks per CU: more LDS room than that buys nothing)
            if (per_cu + 1 <= HG_RANK_WAVES && L.total > want && r3 >= (i64)(2.2 * (double)c->R * share) + 256 && r3 >= 4096 && !c->want_lists) {
                r2 = r3;
                L = rank_cnt_layout(g.NB, c->RW, g.S, (int)r2, 0, nbc);
            }
        }
        while (L.total > 64 * 1024 && r2 > 64) { r2 -= 64; L = rank_cnt_layout(g.NB, c->RW, g.S, (int)r2, c->want_lists ? 1 : 0, nbc); }
        if (L.total <= 64 * 1024 && r2 >= 4096) {
            HG_TRY(c->bigq.reserve((size_t)g.Qpad * 4));
            // hg_map's bet: the AP leaves with the ranking (the bitmap is in LDS), and the general kernel below is NOT launched --
            // the step's download carries the number of queries this kernel declined (err[1]); the host launches it only then
            const bool fuse = c->fuse_ap && c->opt_fuse_ap && mode == 0 && !c->want_lists && c->LW <= 2;
            bool use_recip = false;
            if (fuse) HG_TRY(ensure_ap_tables(c, &use_recip));
            RankLdsArgs la{c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->err.as<int>(), c->qbad.as<u32>(), c->bigq.as<u32>(),
                           c->cap, c->crow, c->want_lists ? 1 : 0, c->rec8 ? 1 : 0, c->RW, (int)r2, mode, c->hwq.as<u32>(), c->hown.as<u32>(),
                           c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->posbase.as<u32>(), nbc,
                           fuse ? c->shapes.as<ApShape>() : nullptr, fuse && use_recip ? c->ap_recip.as<double>() : nullptr,
                           c->ap.as<double>(), c->rel.as<u32>(), fuse ? c->err.as<u32>() + 1 : nullptr, cut};
            c->t_begin(KI_RANK_LDS);
            hipLaunchKernelGGL(k_rank_cnt, dim3(g.Q), dim3(256), (size_t)L.total, c->stream, c->cand.as<u64>(), la, c->out_idx.as<u32>(),
                               c->out_dist.as<u8>(), c->mbits.as<u32>(), g);
            c->t_end();
            HG_TRY(c->check_launch("k_rank_cnt"));
            c->last_rank = 3;
            if (fuse) { c->ap_fused = true; if (c->rec8) HG_TRY(rank_leftovers_inline(c, mode, use_recip)); return HG_OK; }
            only = c->bigq.as<u32>();                // k_rank_fused below ranks what this path declined
            counted = true;
        }
    }
    RankArgs ra{c->sl_cnt.as<u32>(), c->tot.as<u32>(), c->failq.as<u32>(), c->err.as<int>(), c->qbad.as<u32>(),
                mode, c->hwq.as<u32>(), c->hown.as<u32>(), c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(),
                c->tie_before.as<u32>(), c->posbase.as<u32>(),
                c->optimistic ? c->cap : 256u, c->crow, c->optimistic ? 0 : 1, c->want_lists ? 1 : 0, bits_lds, c->RW, only,
                c->direct_rank ? 1 : 0, c->rec8 ? 1 : 0, c->db.as<u32>(), c->dblab.as<u64>(), c->qc.as<u32>(), c->qlab.as<u64>()};
    const size_t lds_bytes = (fixed_words + (bits_lds ? 2 * (size_t)c->RW : 0)) * 4;
    c->t_begin(mode == 1 ? KI_CAND_HIST : KI_RANK_FUSED);
    if (nwav == 16)
        hipLaunchKernelGGL(k_rank_fused<16>, dim3(g.Q), dim3(1024), lds_bytes, c->stream, c->cand.as<u64>(), ra,
                           c->out_idx.as<u32>(), c->out_dist.as<u8>(), c->mbits.as<u32>(), nbits, g);
    else
        hipLaunchKernelGGL(k_rank_fused<4>, dim3(g.Q), dim3(256), lds_bytes, c->stream, c->cand.as<u64>(), ra,
                           c->out_idx.as<u32>(), c->out_dist.as<u8>(), c->mbits.as<u32>(), nbits, g);
    c->t_end();
    return c->check_launch("k_rank_fused");
}

// record pass + ordering (+ gather-based label match when labels are too wide for the record pass)
static int do_select(hg_ctx* c) {
    const Geo& g = c->geo;
    const size_t slots = (size_t)g.Q * g.R;
    if (c->LW > 2) c->want_lists = true;                  // k_match gathers through the idx list
    HG_TRY(reserve_records(c));
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(c->want_lists ? slots * 4 : 16));
    HG_TRY(c->out_dist.reserve(c->want_lists ? slots : 16));
    if (c->want_lists && c->G > 1) {  // slots of other shards stay IDX_NONE / 0xFF
        HG_HIP(hipMemsetAsync(c->out_idx.p, 0xFF, slots * 4, c->stream));
        HG_HIP(hipMemsetAsync(c->out_dist.p, 0xFF, slots, c->stream));
    }
    HG_TRY(launch_select(c));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    if (c->optimistic || c->G == 1) {
        // one block per query: verify (optimistic) + plan + order.  Exact single-shard rows hold
        // precisely the top R, so the same counting plan reproduces t and the bucket starts.
        HG_TRY(launch_rank(c, 0, nbits));
    } else {
        const size_t lds_words = (size_t)g.NB + 2 * (size_t)c->RW;
        const int bits_lds = WPB * lds_words * 4 <= 64 * 1024;
        if (!bits_lds) HG_HIP(hipMemsetAsync(c->mbits.p, 0, (size_t)g.Q * c->RW * 8, c->stream));
        OrdArgs oa{c->t.as<int>(), c->cnt_lt.as<u32>(), c->quota.as<u32>(), c->tie_before.as<u32>(), c->posbase.as<u32>(),
                   c->tot.as<u32>(), c->crow, c->want_lists ? 1 : 0, bits_lds, c->RW};
        c->t_begin(KI_ORDER);
        hipLaunchKernelGGL(k_order, dim3(grid_for(g.Q, WPB)), dim3(256),
                           (size_t)WPB * (g.NB + (bits_lds ? 2 * (size_t)c->RW : 0)) * 4, c->stream, c->cand.as<u64>(), oa,
                           c->out_idx.as<u32>(), c->out_dist.as<u8>(), c->mbits.as<u32>(), nbits, g);
        c->t_end();
        HG_TRY(c->check_launch("k_order"));
    }
    c->lists_valid = c->want_lists;
    c->stage = (c->stage & (ST_DB | ST_Q | ST_HIST | ST_PLAN)) | ST_PLAN | ST_SELECT;
    if (c->LW <= 2) c->stage |= ST_MATCH;                 // match bits came with the records
    else HG_TRY(do_match(c));
    return HG_OK;
}

int hg_select(hg_ctx* c) {
    HG_TRY(need(c, ST_PLAN, "hg_select", "hg_plan"));
    c->want_lists = c->staged_lists != 0;
    HG_TRY(do_select(c));
    return c->stage_end();
}

// ---- staged optimistic sequence (multi-shard): sample -> [gather] -> guess -> candidates ->
// [gather] -> rank.  Mirrors the one-shot bet, with the two histogram exchanges made explicit.
// Sampling stride of the bet, in row batches.  One row batch in 24: a fixed 4 % of a pass (with the
// matrix-core select the sampling pass is a visible share of the step; 16 -> 24 trades 0.05 ms of it for
// ~3 % more surplus records).  The
// guess's safety margin is relative to sqrt(sampled hits), so a small R only means relatively more
// surplus records (R = 100: ~3.5 R of them) -- still far cheaper than a full histogram pass.
// Capacity of a (segment, query) slice of the bet: the budgeted mean + 6 sigma, never more than the segment's rows, and
// the record rows of all queries together stay below 64 GB ("cap_boost" may ask for more than is sensible).
static u32 slice_capacity(const hg_ctx* c, double mean) {
    const Geo& g = c->geo;
    u32 cap = (u32)std::ceil(mean + 6.0 * std::sqrt(mean) + 16.0);
    cap = (cap + 15u) & ~15u;                      // a multiple of the compact records' ring (16) and flush piece (8)
    const u32 whole = (u32)((g.L + 15) & ~15ll);
    if (cap > whole) cap = whole;
    while (cap > 64u && (double)g.Q * (double)g.S * (double)cap * 8.0 > 64e9) cap = (cap / 2u + 15u) & ~15u;
    return cap;
}

static int auto_stride(hg_ctx* c, int64_t R) {
    (void)R;
    return c->opt_stride > 0 ? (int)c->opt_stride : 24;
}

int hg_bet_eligible(hg_ctx* c, int64_t R, int world, int* eligible) {
    if (!c || !eligible || world < 1) return fail(HG_ERR_ARG, "hg_bet_eligible: bad argument");
    // only quantities every rank shares: options, R, the size of the whole database, the world size -- and the count of
    // consecutive SHARDED bets lost, which only hg_merge_ranked / hg_rank / hg_bet_verdict touch, with a verdict that is
    // computed from gathered data and therefore the same on every rank (one-shot calls keep their own counter)
    const int stride = auto_stride(c, R);
    const i64 per_shard = c->n_total / world;
    *eligible = c->opt_enable && c->shard_bet_fail < 2 && stride >= 2 && R * 8 <= c->n_total && per_shard >= 65536;
    return HG_OK;
}

int hg_sample_hist(hg_ctx* c, int64_t R) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_sample_hist", "hg_set_database + hg_set_queries"));
    const int stride = auto_stride(c, R);
    if (stride < 2) return fail(HG_ERR_ARG, "hg_sample_hist: R=%lld is too small to sample for", (long long)R);
    if (c->beyond.p) HG_HIP(hipMemsetAsync(c->beyond.p, 0, 4, c->stream));      // (a new bet: "cut_beyond_planes" is about this one)
    HG_TRY(do_hist(c, stride));
    return c->stage_end();
}

int hg_guess(hg_ctx* c, int64_t R, const uint32_t* dev_hist_all, int G, int rank) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_guess", "hg_sample_hist"));
    if (G > 1 && !dev_hist_all) return fail(HG_ERR_ARG, "hg_guess: G > 1 needs the gathered sample histograms");
    HG_TRY(set_R(c, R, G, rank));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->tguess.reserve(qb));
    HG_TRY(c->sl_start.reserve((size_t)g.S * qb)); HG_TRY(c->sl_tie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->failq.reserve(qb));
    HG_HIP(hipMemsetAsync(c->failq.p, 0, qb, c->stream));
    c->t_begin(KI_GUESS);
    const Geo gh = hist_geometry(c);                   // the sampled pass ran on coarser segments
    HG_TRY(c->sstar.reserve(qb));
    hipLaunchKernelGGL(k_guess, dim3(grid_for(g.Q)), dim3(256), 0, c->stream, c->hown.as<u32>(), (const u32*)dev_hist_all, G,
                       rank, c->hist.as<u32>(), gh.S, (int)(gh.L / g.L), (double)c->opt_sigma, (i64)c->n_total,
                       c->tguess.as<int>(), c->sstar.as<int>(), g);
    c->t_end();
    HG_TRY(c->check_launch("k_guess"));
    // a guessed cut keeps at most ~2.6 R rows over ALL shards; a shard's share is proportional to its size,
    // with the same 6-sigma headroom per slice as the one-shot bet
    const double share = (double)c->N / (double)c->n_total;
    const double mean = 0.1 * (double)c->cand_budget_x10 * (double)c->cap_boost * (double)R * share / (double)g.S;
    u32 cap = slice_capacity(c, mean);
    c->optimistic = true;
    c->cap = cap;
    c->crow = (i64)g.S * cap;
    c->stage = ST_DB | ST_Q | ST_PLAN;
    return c->stage_end();
}

int hg_select_candidates(hg_ctx* c) {
    HG_TRY(need(c, ST_PLAN, "hg_select_candidates", "hg_guess"));
    if (!c->optimistic) return fail(HG_ERR_STATE, "hg_select_candidates: no guess in force (use hg_select after hg_plan)");
    const Geo& g = c->geo;
    c->want_lists = c->staged_lists != 0 || c->LW > 2;  // decides the record format (hg_rank places them)
    HG_TRY(reserve_records(c));
    HG_TRY(launch_select(c));
    const size_t plane = (size_t)g.NB * g.Qpad * 4;
    HG_HIP(hipMemsetAsync(c->hown.as<char>() + plane, 0, TAIL_WORDS * 4, c->stream));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    HG_TRY(launch_rank(c, 1, nbits));                // per-wave and shard histograms of the records
    return c->stage_end();
}

// Sharded bet, AP only: select, then rank THIS shard's records locally (rank kernels' mode 3).  What leaves the
// shard is its per-distance record counts (hg_hist_buffer) and its match bitmap in local rank order
// (hg_match_buffer); hg_merge_ranked stitches the global bitmap from the gathered pairs.  One exchange and one
// pass over the records fewer than hg_select_candidates + hg_rank.
int hg_select_ranked(hg_ctx* c) {
    HG_TRY(need(c, ST_PLAN, "hg_select_ranked", "hg_guess"));
    if (!c->optimistic) return fail(HG_ERR_STATE, "hg_select_ranked: no guess in force");
    const Geo& g = c->geo;
    // > 128 classes: the record pass leaves the match bit 0 (launch_select_nw instantiates LW = 0); the local bitmap
    // then comes from k_match gathering the labels through the LOCAL ranked index list, so that list is kept
    const bool wide = c->LW > 2;
    c->want_lists = wide;
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(reserve_records(c));
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(wide ? slots * 4 : 16)); HG_TRY(c->out_dist.reserve(wide ? slots : 16));
    if (wide) HG_HIP(hipMemsetAsync(c->out_idx.p, 0xFF, slots * 4, c->stream));    // slots past the shard's own records: IDX_NONE
    HG_TRY(c->err.reserve(16));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    HG_TRY(launch_select(c));
    const size_t plane = (size_t)g.NB * g.Qpad * 4;
    HG_HIP(hipMemsetAsync(c->hown.as<char>() + plane, 0, TAIL_WORDS * 4, c->stream));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    HG_TRY(launch_rank(c, 3, nbits));
    c->lists_valid = false;
    c->ranked_local = true;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT;
    if (wide) HG_TRY(do_match(c));                    // metric.py:17-19 through the local list
    c->want_lists = false;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_MATCH;     // hg_match_buffer hands out the LOCAL bitmap until the merge
    return c->stage_end();
}

static int merge_ranked_range(hg_ctx* c, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, i64 q0, i64 nq, const char* who,
                              const MergeSrc* routed = nullptr) {
    HG_TRY(need(c, ST_MATCH, who, "hg_select_ranked"));
    if (!c->ranked_local) return fail(HG_ERR_STATE, "%s: hg_select_ranked has not run", who);
    if (G < 1 || G > 64 || (G > 1 && (!dev_hist_all || !dev_bits_all)))
        return fail(HG_ERR_ARG, "%s: bad argument (1 <= G <= 64, gathered buffers for G > 1)", who);
    const Geo& g = c->geo;
    if (q0 < 0 || nq < 0 || q0 + nq > g.Q) return fail(HG_ERR_ARG, "%s: queries [%lld, %lld) of %d", who, (long long)q0, (long long)(q0 + nq), g.Q);
    HG_TRY(c->mbits2.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->qbad.reserve((size_t)g.Qpad * 4));
    const u32* hall = G > 1 || routed ? (const u32*)dev_hist_all : c->hown.as<u32>();
    const u64* ball = G > 1 || routed ? (const u64*)dev_bits_all : c->mbits.as<u64>();
    const size_t rows_lds = (size_t)WPB * G * c->RW * 8;       // the G local bitmap rows of a block's four queries
    const size_t cnt_lds = (size_t)WPB * G * g.NB * 4;         // their per-distance counts: at most 4 * 64 * 256 * 4 = 256 KiB ...
    const int use_lds = rows_lds + cnt_lds <= 64 * 1024;
    const size_t merge_lds = (use_lds ? rows_lds : 0) + cnt_lds;
    if (merge_lds > 160 * 1024)                                // ... which only many shards of long codes reach
        return fail(HG_ERR_ARG, "hg_merge_ranked: G=%d shards of %d-bit codes need %zu bytes of LDS per block (160 KiB available): "
                                "use the staged sequence (hg_select_candidates / hg_rank)", G, c->b, merge_lds);
    if (merge_lds > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_ranked), hipFuncAttributeMaxDynamicSharedMemorySize, (int)merge_lds));
    c->t_begin(KI_MERGE);
    if (nq > 0)
    {   // all-gathered whole tables by default; `routed`: the owner's blocks of an all-to-all (hg_merge_ap_owned)
        const MergeSrc whole{(i64)g.NB * g.Qpad + TAIL_WORDS, g.Qpad, (i64)g.NB * g.Qpad, (i64)g.Q * c->RW, 0};
        hipLaunchKernelGGL(k_merge_ranked, dim3(grid_for(nq, WPB)), dim3(256), merge_lds, c->stream, hall, ball, G,
                           c->RW, c->mbits2.as<u64>(), c->err.as<int>(), c->qbad.as<u32>(), use_lds, g, (int)q0, (int)(q0 + nq), routed ? *routed : whole);
    }
    c->t_end();
    HG_TRY(c->check_launch("k_merge_ranked"));
    std::swap(c->mbits, c->mbits2);                    // the global bitmap is what hg_ap and hg_get_match see
    c->ranked_local = false;
    c->G = G;
    return HG_OK;
}

int hg_merge_ranked(hg_ctx* c, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, int* bet_lost) {
    if (!bet_lost) return fail(HG_ERR_ARG, "hg_merge_ranked: null argument");
    HG_TRY(merge_ranked_range(c, dev_hist_all, dev_bits_all, G, 0, c ? c->geo.Q : 0, "hg_merge_ranked"));
    if (c->defer_verdict) {
        *bet_lost = -1;
        c->verdict_pending = true;
        c->verdict_known = false;
        c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
        return c->stage_end();
    }
    int flag = 0;
    HG_TRY(read_plan_flag(c, &flag));
    *bet_lost = flag;
    c->opt_runs++;
    if (flag) {
        c->opt_fallbacks++;
        c->shard_bet_fail++;
        c->stage = ST_DB | ST_Q;
        return HG_OK;
    }
    c->shard_bet_fail = 0;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
    return HG_OK;
}

// bookkeeping after the verdict of a staged bet is known
static int settle_bet(hg_ctx* c, int flag) {
    c->opt_runs++;
    if (flag) {
        c->opt_fallbacks++;
        c->shard_bet_fail++;
        c->stage = ST_DB | ST_Q;
        return HG_OK;
    }
    c->shard_bet_fail = 0;
    c->lists_valid = c->want_lists;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT;
    if (c->LW <= 2) c->stage |= ST_MATCH;
    else HG_TRY(do_match(c));
    return c->sync();
}

int hg_rank(hg_ctx* c, const uint32_t* dev_hist_all, int G, int rank, int* bet_lost) {
    HG_TRY(need(c, ST_PLAN, "hg_rank", "hg_select_candidates"));
    if (!c->optimistic) return fail(HG_ERR_STATE, "hg_rank: no guess in force");
    if (!bet_lost || G < 1 || (G > 1 && !dev_hist_all)) return fail(HG_ERR_ARG, "hg_rank: bad argument");
    const Geo& g = c->geo;
    c->G = G; c->rank = rank;
    c->want_lists = c->staged_lists != 0 || c->LW > 2;
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(c->want_lists ? slots * 4 : 16));
    HG_TRY(c->out_dist.reserve(c->want_lists ? slots : 16));
    if (c->want_lists && G > 1) {
        HG_HIP(hipMemsetAsync(c->out_idx.p, 0xFF, slots * 4, c->stream));
        HG_HIP(hipMemsetAsync(c->out_dist.p, 0xFF, slots, c->stream));
    }
    HG_TRY(launch_plan(c, dev_hist_all));
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    HG_TRY(launch_rank(c, 2, nbits));                // placement with the shared plan
    if (c->defer_verdict) {
        // the caller goes on as if the bet held (match bits, exchange, AP) and asks hg_bet_verdict at the end,
        // together with its final download: no host round trip in the middle of the step
        *bet_lost = -1;
        c->verdict_pending = true;
        c->verdict_known = false;
        c->lists_valid = c->want_lists;
        c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT;
        if (c->LW <= 2) c->stage |= ST_MATCH;
        else HG_TRY(do_match(c));
        return c->stage_end();
    }
    int flag = 0;
    HG_TRY(read_plan_flag(c, &flag));
    *bet_lost = flag;
    return settle_bet(c, flag);
}

int hg_bet_verdict(hg_ctx* c, int* bet_lost) {
    if (!c || !bet_lost) return fail(HG_ERR_ARG, "hg_bet_verdict: null argument");
    if (!c->verdict_pending) return fail(HG_ERR_STATE, "hg_bet_verdict: no deferred hg_rank outstanding");
    c->verdict_pending = false;
    int flag = 0;
    if (c->verdict_known) flag = c->verdict_flag;      // came over with hg_get_ap's download
    else HG_TRY(read_plan_flag(c, &flag));
    c->verdict_known = false;
    *bet_lost = flag;
    if (!flag) { c->opt_runs++; c->shard_bet_fail = 0; return HG_OK; }
    c->opt_runs++;
    c->opt_fallbacks++;
    c->shard_bet_fail++;
    c->lists_valid = false;
    c->stage = ST_DB | ST_Q;
    return HG_OK;
}

// The sharded bet with the per-query stages SPLIT over the ranks: after the all-gather of record counts and local bitmaps
// every rank merges and evaluates only its own queries [q0, q0 + nq) (instead of all Q on every rank), packs {AP, hits}
// and its verdict into `part`; one small all-gather later hg_unpack_parts gives every rank all of it.
int hg_merge_ap_part(hg_ctx* c, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, int64_t q0, int64_t nq,
                     int64_t width, void** dev_part, int64_t* nbytes) {
    if (!c || !dev_part || !nbytes || width < nq || width < 1) return fail(HG_ERR_ARG, "hg_merge_ap_part: bad argument");
    HG_TRY(merge_ranked_range(c, dev_hist_all, dev_bits_all, G, q0, nq, "hg_merge_ap_part"));
    const Geo& g = c->geo;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
    // AP of the merged rows only: k_ap's block index is the query
    {
        const i64 Qsave = c->geo.Q;
        HG_TRY(c->ap.reserve((size_t)Qsave * 8));
        HG_TRY(c->rel.reserve((size_t)Qsave * 4));
        HG_TRY(do_ap_range(c, q0, nq));
    }
    if (!(q0 == 0 && nq == g.Q)) c->stage &= ~(unsigned)ST_MATCH;   // the merged bitmap holds rows [q0, q0 + nq) only: hg_get_match has nothing whole to hand out
    const size_t pb = (size_t)(width + 1) * 16;
    HG_TRY(c->part.reserve(pb));
    hipLaunchKernelGGL(k_pack_part, dim3(grid_for(width + 1)), dim3(256), 0, c->stream, c->ap.as<double>(), c->rel.as<u32>(),
                       c->err.as<int>(), (i64)q0, (i64)nq, (i64)width, c->part.as<double>());
    HG_TRY(c->check_launch("k_pack_part"));
    (void)g;
    *dev_part = c->part.p;
    *nbytes = (int64_t)pb;
    return c->stage_end();
}

int hg_unpack_parts(hg_ctx* c, const void* dev_parts_all, int G, int64_t width, double* host_ap, int64_t* host_rel, int* bet_lost) {
    if (!c || !dev_parts_all || G < 1 || width < 1 || !host_ap || !host_rel || !bet_lost) return fail(HG_ERR_ARG, "hg_unpack_parts: bad argument");
    HG_TRY(c->use());
    const i64 Q = c->geo.Q;
    const size_t pb = (size_t)(width + 1) * 16, total = pb * (size_t)G;
    HG_TRY(ensure_pin(c, total));
    HG_HIP(hipMemcpyAsync(c->pin, dev_parts_all, total, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    int flag = 0;
    i64 done = 0;
    for (int r = 0; r < G; ++r) {
        const double* p = (const double*)((const char*)c->pin + (size_t)r * pb);
        const i64 n = (i64)p[2 * width + 1];
        if (n < 0 || n > width || done + n > Q) return fail(HG_ERR_ARG, "hg_unpack_parts: rank %d reports %lld queries (width %lld, %lld of %lld placed)",
                                                            r, (long long)n, (long long)width, (long long)done, (long long)Q);
        flag |= p[2 * width] != 0.0;
        for (i64 i = 0; i < n; ++i) { host_ap[done + i] = p[2 * i]; host_rel[done + i] = (int64_t)p[2 * i + 1]; }
        done += n;
    }
    if (done != Q) return fail(HG_ERR_ARG, "hg_unpack_parts: the parts cover %lld of %lld queries", (long long)done, (long long)Q);
    *bet_lost = flag;
    c->verdict_pending = false;
    c->verdict_known = false;
    c->opt_runs++;                                     // the verdict comes from gathered data: the same on every rank
    if (flag) {
        c->opt_fallbacks++;
        c->shard_bet_fail++;
        c->lists_valid = false;
        c->stage = ST_DB | ST_Q;
    } else {
        c->shard_bet_fail = 0;
    }
    return HG_OK;
}

// ---- the sharded bet with its exchanges routed by query owner (all-to-all instead of all-gather): see k_pack_sample_owner ----
static int owners_for(hg_ctx* c, int G, Owners* w, const char* who) {
    if (G < 1 || G > 64) return fail(HG_ERR_ARG, "%s: 1 <= G <= 64", who);
    *w = make_owners(c->geo.Q, G);
    return HG_OK;
}
static int sample_planes(const hg_ctx* c) { const int NB = c->geo.NB; return NB / 2 + 2 < NB ? NB / 2 + 2 : NB; }   // what a guess can need (enqueue_optimistic)

int hg_pack_sample_by_owner(hg_ctx* c, int G, void** dev_ptr, int64_t* nbytes_per_peer) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_pack_sample_by_owner", "hg_sample_hist"));
    if (!dev_ptr || !nbytes_per_peer || !c->hown.p) return fail(HG_ERR_ARG, "hg_pack_sample_by_owner: bad argument (hg_sample_hist first)");
    Owners w;
    HG_TRY(owners_for(c, G, &w, "hg_pack_sample_by_owner"));
    const Geo& g = c->geo;
    const int HC = sample_planes(c);
    const i64 blk = 4 + (i64)HC * w.width;
    HG_TRY(c->obuf[0].reserve((size_t)G * blk * 4));
    hipLaunchKernelGGL(k_pack_sample_owner, dim3(grid_for((i64)G * HC * w.width)), dim3(256), 0, c->stream, c->hown.as<u32>(), w, HC,
                       c->obuf[0].as<u32>(), g);
    HG_TRY(c->check_launch("k_pack_sample_owner"));
    *dev_ptr = c->obuf[0].p;
    *nbytes_per_peer = blk * 4;
    return c->stage_end();
}

int hg_guess_owned(hg_ctx* c, int64_t R, const void* dev_recv, int G, int rank, void** dev_ptr, int64_t* nbytes_per_peer) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_guess_owned", "hg_sample_hist"));
    if (!dev_recv || !dev_ptr || !nbytes_per_peer) return fail(HG_ERR_ARG, "hg_guess_owned: null argument");
    Owners w;
    HG_TRY(owners_for(c, G, &w, "hg_guess_owned"));
    HG_TRY(set_R(c, R, G, rank));
    const Geo& g = c->geo;
    HG_TRY(c->obuf[1].reserve((size_t)G * w.width * 16));
    c->t_begin(KI_GUESS);
    hipLaunchKernelGGL(k_guess_owner, dim3(grid_for(w.width)), dim3(256), 0, c->stream, (const u32*)dev_recv, w, sample_planes(c),
                       owner_nq(w, rank), (double)c->opt_sigma, (i64)c->n_total, c->obuf[1].as<u32>(), g);
    c->t_end();
    HG_TRY(c->check_launch("k_guess_owner"));
    *dev_ptr = c->obuf[1].p;
    *nbytes_per_peer = (int64_t)w.width * 16;
    return c->stage_end();
}

int hg_guess_finish(hg_ctx* c, int64_t R, const void* dev_answers, int G, int rank) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_guess_finish", "hg_sample_hist"));
    if (!dev_answers) return fail(HG_ERR_ARG, "hg_guess_finish: null argument");
    Owners w;
    HG_TRY(owners_for(c, G, &w, "hg_guess_finish"));
    HG_TRY(set_R(c, R, G, rank));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->tguess.reserve(qb));
    HG_TRY(c->sl_start.reserve((size_t)g.S * qb)); HG_TRY(c->sl_tie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->failq.reserve(qb));
    HG_TRY(c->sstar.reserve(qb));
    HG_HIP(hipMemsetAsync(c->failq.p, 0, qb, c->stream));
    HG_TRY(c->beyond.reserve(4));
    HG_HIP(hipMemsetAsync(c->beyond.p, 0, 4, c->stream));
    const Geo gh = hist_geometry(c);                   // the sampled pass ran on coarser segments
    c->t_begin(KI_GUESS);
    hipLaunchKernelGGL(k_guess_finish, dim3(grid_for(g.Q)), dim3(256), 0, c->stream, (const u32*)dev_answers, w, c->hist.as<u32>(), gh.S,
                       (int)(gh.L / g.L), c->tguess.as<int>(), c->sstar.as<int>(), c->beyond.as<u32>(), g);
    c->t_end();
    HG_TRY(c->check_launch("k_guess_finish"));
    // (the slices' budget: as hg_guess)
    const double share = (double)c->N / (double)c->n_total;
    const double mean = 0.1 * (double)c->cand_budget_x10 * (double)c->cap_boost * (double)R * share / (double)g.S;
    c->optimistic = true;
    c->cap = slice_capacity(c, mean);
    c->crow = (i64)g.S * c->cap;
    c->stage = ST_DB | ST_Q | ST_PLAN;
    return c->stage_end();
}

int hg_pack_ranked_by_owner(hg_ctx* c, int G, void** dev_ptr, int64_t* nbytes_per_peer) {
    HG_TRY(need(c, ST_MATCH, "hg_pack_ranked_by_owner", "hg_select_ranked"));
    if (!c->ranked_local) return fail(HG_ERR_STATE, "hg_pack_ranked_by_owner: hg_select_ranked has not run");
    if (!dev_ptr || !nbytes_per_peer) return fail(HG_ERR_ARG, "hg_pack_ranked_by_owner: null argument");
    Owners w;
    HG_TRY(owners_for(c, G, &w, "hg_pack_ranked_by_owner"));
    const Geo& g = c->geo;
    const i64 cw = ((i64)g.NB * w.width + 1) & ~1ll;
    const i64 blk32 = cw + TAIL_WORDS + 2 * (i64)w.width * c->RW;
    HG_TRY(c->obuf[0].reserve((size_t)G * blk32 * 4));
    const i64 items = (i64)G * (cw + TAIL_WORDS) + (i64)G * w.width * c->RW;
    hipLaunchKernelGGL(k_pack_ranked_owner, dim3(grid_for(items)), dim3(256), 0, c->stream, c->hown.as<u32>(), c->mbits.as<u64>(), w, c->RW, cw,
                       c->obuf[0].as<u32>(), g);
    HG_TRY(c->check_launch("k_pack_ranked_owner"));
    *dev_ptr = c->obuf[0].p;
    *nbytes_per_peer = blk32 * 4;
    return c->stage_end();
}

int hg_merge_ap_owned(hg_ctx* c, const void* dev_recv, int G, int rank, void** dev_part, int64_t* nbytes) {
    if (!c || !dev_recv || !dev_part || !nbytes) return fail(HG_ERR_ARG, "hg_merge_ap_owned: null argument");
    Owners w;
    HG_TRY(owners_for(c, G, &w, "hg_merge_ap_owned"));
    if (rank < 0 || rank >= G) return fail(HG_ERR_ARG, "hg_merge_ap_owned: rank %d of %d", rank, G);
    const Geo& g = c->geo;
    const i64 cw = ((i64)g.NB * w.width + 1) & ~1ll;
    const i64 blk32 = cw + TAIL_WORDS + 2 * (i64)w.width * c->RW;
    const i64 q0 = owner_q0(w, rank), nq = owner_nq(w, rank);
    const MergeSrc ms{blk32, w.width, cw, blk32 / 2, (int)q0};
    const u32* hall = (const u32*)dev_recv;
    const u64* ball = (const u64*)(hall + cw + TAIL_WORDS);
    HG_TRY(merge_ranked_range(c, hall, ball, G, q0, nq, "hg_merge_ap_owned", &ms));
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
    HG_TRY(c->ap.reserve((size_t)g.Q * 8));
    HG_TRY(c->rel.reserve((size_t)g.Q * 4));
    HG_TRY(do_ap_range(c, q0, nq));
    if (!(q0 == 0 && nq == g.Q)) c->stage &= ~(unsigned)ST_MATCH;
    const i64 width = w.width;
    const size_t pb = (size_t)(width + 1) * 16;
    HG_TRY(c->part.reserve(pb));
    hipLaunchKernelGGL(k_pack_part, dim3(grid_for(width + 1)), dim3(256), 0, c->stream, c->ap.as<double>(), c->rel.as<u32>(),
                       c->err.as<int>(), (i64)q0, (i64)nq, (i64)width, c->part.as<double>());
    HG_TRY(c->check_launch("k_pack_part"));
    *dev_part = c->part.p;
    *nbytes = (int64_t)pb;
    return c->stage_end();
}

// ---- one-shot forms: every stage enqueued back to back, one synchronisation ----
// Optimistic bet (single shard, R << N): instead of a full histogram pass, sample
// every stride-th row batch, guess the threshold a few sigma high, select a
// superset with it, then verify: the records' exact histogram must contain R rows
// and no slice may have overflowed.  The verified result is identical to the
// exact path's; a failed bet reruns the exact path.
static bool optimistic_eligible(hg_ctx* c, int64_t R, int* stride_out, u32* need_out) {
    if (!c->opt_enable || c->opt_consecutive_fail >= 2) return false;
    if (R * 8 > c->N || c->N < 65536) return false;
    make_geometry(c);
    const int stride = auto_stride(c, R);
    if (stride < 2) return false;
    const i64 sampled = sampled_rows(c, stride);
    const double fr = (double)R * (double)sampled / (double)c->N;   // expected sample count at the true cut
    const double need = fr + (double)c->opt_sigma * std::sqrt(fr) + 1.0;
    *stride_out = stride;
    *need_out = (u32)std::ceil(need);
    return true;
}

// R = N on one shard (the reference's CIFAR-10 setting): every row is a member of every ranked list, so nothing
// has to be selected or written down -- the ranking kernel walks the shard's rows directly, computing each row's
// distance and match bit from the codes and labels in both of its passes (counting, then stable placement).
static int enqueue_all_rows(hg_ctx* c, int64_t R) {
    make_geometry(c);
    HG_TRY(set_R(c, R, 1, 0));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(c->err.reserve(16)); HG_TRY(c->failq.reserve(qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->sl_cnt.reserve(qb));
    HG_TRY(c->cand.reserve(64));
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    HG_TRY(c->out_idx.reserve(c->want_lists ? slots * 4 : 16));
    HG_TRY(c->out_dist.reserve(c->want_lists ? slots : 16));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    c->optimistic = false;
    c->crow = R;
    c->cap = 0;
    int nbits = 1;
    while ((1 << nbits) < g.NB) ++nbits;
    c->direct_rank = true;
    c->rec8 = false;
    c->last_select = 0;                                // no record pass at all
    const int rc = launch_rank(c, 0, nbits);
    c->direct_rank = false;
    c->dense_rank = false;
    HG_TRY(rc);
    c->lists_valid = c->want_lists;
    c->stage = ST_DB | ST_Q | ST_PLAN | ST_SELECT | ST_MATCH;
    return HG_OK;
}

static int enqueue_exact(hg_ctx* c, int64_t R) {
    // N/8 < R <= N through the byte matrix (k_dense_bytes + k_rank_dense: R = N/2 of N = 1M 68.8 -> 16.3 ms, C1 0.35 -> 0.19 ms)
    if (R * 8 > c->N && c->opt_all_rows && rank_dense_fits(c, R)) {
        c->dense_rank = true;
        return enqueue_all_rows(c, R);
    }
    if (c->N == c->n_total && R == c->N && c->opt_all_rows && c->LW <= 2 && c->NW <= 8)
        return enqueue_all_rows(c, R);                 // one-shot calls are single-shard; codes / label sets the byte matrix does not take: k_rank_fused walks the rows
    HG_TRY(do_hist(c, 1));
    HG_TRY(do_plan(c, R, nullptr, 1, 0));
    return do_select(c);
}

// The exact sequence with its second pass on the matrix cores (one shard, R << N): full histogram -> plan (exact
// threshold t, and sstar = the last segment whose ties at t are still inside the quota) -> k_select_mx with T = t,
// fixed-capacity slices -> the bet's rank stage, which cuts the ties at the quota.  Nothing is guessed, so the only way
// this can fail is a slice overflowing its capacity (clustered rows): *err then, and the caller runs enqueue_exact.
static bool exact_mx_applies(const hg_ctx* c, int64_t R) {
    return c->opt_select_mfma && c->N == c->n_total && R * 8 <= c->N && c->N >= 65536 && !c->is_sub;
}
static int enqueue_exact_mx(hg_ctx* c, int64_t R) {
    HG_TRY(do_hist(c, 1, true, true));                 // per segment pair on the matrix cores where that applies
    HG_TRY(do_plan(c, R, nullptr, 1, 0));              // c->t, c->sstar; leaves optimistic = false, crow = R
    const Geo& g = c->geo;
    // in all the slices hold R records + the ties of one segment beyond the quota, but unevenly: segments up to sstar carry
    // ALL their rows at distance t (the cut bucket is typically the fullest), later ones none -- budget like the bet does
    const double mean = 0.1 * (double)c->cand_budget_x10 * (double)R / (double)g.S;
    u32 cap = (u32)std::ceil(mean + 6.0 * std::sqrt(mean) + 16.0);
    cap = (cap + 15u) & ~15u;
    c->optimistic = true;
    c->exact_mx = true;
    c->cap = cap;
    c->crow = (i64)g.S * cap;
    HG_HIP(hipMemsetAsync(c->failq.p, 0, (size_t)g.Qpad * 4, c->stream));
    const int rc = do_select(c);
    c->exact_mx = false;
    return rc;
}

static int enqueue_optimistic(hg_ctx* c, int64_t R, int stride, u32 need_cnt) {
    (void)need_cnt;
    // the guess stops at the cut, far below b/2 when R <= N/8 on any data whose queries resemble the database: the
    // sampled pass writes only those planes (130 -> 68 MB per launch at C2).  A cut beyond them reads as a thin
    // sample -- everything is taken, the slices overflow, the exact sequence answers.
    HG_TRY(do_hist(c, stride, false, false, c->NB / 2 + 2));
    HG_TRY(set_R(c, R, 1, 0));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->tguess.reserve(qb));
    HG_TRY(c->sl_start.reserve((size_t)g.S * qb)); HG_TRY(c->sl_tie.reserve((size_t)g.S * qb));
    HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->tot.reserve(qb)); HG_TRY(c->failq.reserve(qb));
    HG_TRY(c->err.reserve(16));
    HG_TRY(c->sstar.reserve(qb));
    // first bet on this database: the guess kernel also measures how the near rows crowd (err[2], err[3]) -- see below
    const bool probe = c->opt_crowd_probe && !c->crowd_probed && c->cap_boost == 1 && !c->capturing && !c->is_sub;
    u32* crowd = probe ? c->err.as<u32>() + 2 : nullptr;
    if (probe) HG_HIP(hipMemsetAsync(crowd, 0, 8, c->stream));
    c->t_begin(KI_GUESS);
    const Geo gh = hist_geometry(c);                   // the sampled pass ran on coarser segments
    {   // lanes per query by the number of sampled segments each has to sum
        const int ratio = (int)(gh.L / g.L);
        const u32 srows = (u32)sampled_rows(c, stride);
        double sigma = (double)c->opt_sigma;
        if (c->handicap_next && !c->capturing) { sigma = -(double)c->handicap_next; c->handicap_next = 0; }   // (test hook: this bet is meant to lose)
#define HG_GUESS(P)                                                                                                     \
        hipLaunchKernelGGL(k_guess_direct<P>, dim3(grid_for(g.Qpad, WPB * (64 / P))), dim3(256), 0, c->stream,          \
                           c->hist.as<u32>(), gh.S, ratio, sigma, (i64)c->n_total, srows,                \
                           c->tguess.as<int>(), c->sstar.as<int>(), c->failq.as<u32>(), c->err.as<int>(), crowd, g)
        // (more lanes per query shorten a lane's share of a plane but scatter a wavefront's loads over more lines: with 64
        // lanes for every query that the chip has room for, Q = 1000 went 0.068 -> 0.090 ms, C3 0.032 -> 0.061)
        if (gh.S <= 64) HG_GUESS(4);
        else if (gh.S <= 512) HG_GUESS(16);
        else HG_GUESS(64);
#undef HG_GUESS
    }
    c->t_end();
    HG_TRY(c->check_launch("k_guess_direct"));
    c->err_zeroed = true;                              // launch_rank need not clear the lost-bet flag again
    if (probe) {
        // One host round trip, once per database: sum over the queries of (fullest sampled segment) * segments / (their total)
        // is ~2 for rows in random order (the maximum of ~50 small Poisson counts) and ~the number of classes for a database
        // stored class by class, whose slices then need that many times the mean -- widen them NOW instead of losing the
        // first two bets (round 3: first call 6.6 ms at C2, two lost bets per new context).
        u32 cr[2] = {0u, 0u};
        HG_HIP(hipMemcpyAsync(cr, crowd, 8, hipMemcpyDeviceToHost, c->stream));
        HG_TRY(c->sync());
        c->crowd_probed = true;
        const double ratio = cr[1] ? (double)cr[0] * (double)gh.S / (double)cr[1] : 0.0;
        c->crowd_x100 = (i64)(ratio * 100.0);
        if (ratio > 4.0) {
            i64 boost = 2;
            while ((double)boost < ratio && boost < 64) boost *= 2;
            c->cap_boost = boost;
            c->cfg_epoch++;
        }
    }
    // slice capacity: a guessed cut typically keeps 1.3-3 R rows (the guess overshoots by at most one
    // distance bucket, and cumulative counts grow ~2x per bucket in the tail where the cut lies; clustered
    // codes grow faster) -- budget 4 R per query over the S segments plus 6 sigma per slice.  HBM is
    // plentiful (2.5 GB at C2); an overflow only costs the exact rerun.
    const double mean = 0.1 * (double)c->cand_budget_x10 * (double)c->cap_boost * (double)R / (double)g.S;
    u32 cap = slice_capacity(c, mean);
    c->optimistic = true;
    c->cap = cap;
    c->crow = (i64)g.S * cap;
    c->stage = ST_DB | ST_Q | ST_PLAN;
    return do_select(c);
}

static int enqueue_exact(hg_ctx* c, int64_t R);

// Lost bets are per query (a short superset, an overflowed slice).  When only a few queries lost,
// rerun just those through the exact sequence in a child context that borrows the database
// tables, and patch their results into place.  *handled = false: too many, caller reruns all.
static int rerun_lost_queries(hg_ctx* c, int64_t R, bool lists, bool with_ap, bool* handled) {
    *handled = false;
    const Geo g = c->geo;
    std::vector<u32> bad((size_t)g.Q);
    HG_HIP(hipMemcpyAsync(bad.data(), c->qbad.p, (size_t)g.Q * 4, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    std::vector<u32> lost;
    for (int q = 0; q < g.Q; ++q) if (bad[(size_t)q]) lost.push_back((u32)q);
    const i64 nF = (i64)lost.size();
    if (nF == 0 || nF * 8 > g.Q) return HG_OK;
    if (!c->sub) {
        c->sub = new hg_ctx();
        c->sub->is_sub = true;
        c->sub->device = c->device;
        c->sub->stream = c->stream;                  // same stream: ordered with the parent's work
    }
    hg_ctx* s = c->sub;
    s->N = c->N; s->b = c->b; s->C = c->C; s->n_total = c->n_total; s->NW = c->NW; s->NB = c->NB; s->LW = c->LW;
    s->idx_base = c->idx_base;
    s->target_units = c->target_units; s->min_segment = c->min_segment; s->opt_enable = 0;
    // a handful of queries: the per-segment bookkeeping (k_hist_reduce, k_seg_layout walk S segments per query) costs more than
    // the pair passes themselves -- 2048 segments: 0.8 ms of a 1 ms rerun; 256 keep every CU busy and cost 0.1
    s->opt_max_segments = 256;
    s->timing = 0;
    s->db.borrow(c->db);
    s->dblab.borrow(c->dblab);
    s->Q = nF;
    HG_TRY(c->flist.reserve((size_t)nF * 4));
    HG_HIP(hipMemcpyAsync(c->flist.p, lost.data(), (size_t)nF * 4, hipMemcpyHostToDevice, c->stream));
    HG_TRY(s->qc.reserve((size_t)nF * c->NW * 4 + 64 * 4));
    HG_TRY(s->qlab.reserve((size_t)nF * c->LW * 8));
    auto move = [&](const void* src, void* dst, i64 rowbytes, int gather) {
        hipLaunchKernelGGL(k_move_rows, dim3((unsigned)nF), dim3(256), 0, c->stream, (const u8*)src, (u8*)dst,
                           c->flist.as<u32>(), rowbytes, gather);
    };
    move(c->qc.p, s->qc.p, (i64)c->NW * 4, 1);
    move(c->qlab.p, s->qlab.p, (i64)c->LW * 8, 1);
    HG_TRY(c->check_launch("k_move_rows"));
    s->stage = ST_DB | ST_Q;
    s->want_lists = lists;
    HG_TRY(enqueue_exact(s, R));
    if (with_ap) HG_TRY(do_ap(s));
    move(s->mbits.p, c->mbits.p, c->RW * 8, 0);
    if (with_ap) {
        move(s->ap.p, c->ap.p, 8, 0);
        move(s->rel.p, c->rel.p, 4, 0);
    }
    if (lists) {
        move(s->out_idx.p, c->out_idx.p, R * 4, 0);
        move(s->out_dist.p, c->out_dist.p, R, 0);
    }
    HG_TRY(c->check_launch("k_move_rows"));
    HG_TRY(c->sync());                               // `lost` (the H2D source) must outlive the copy
    c->opt_requeried += nF;
    *handled = true;
    return HG_OK;
}

// The bet's sequence for hg_map, enqueued on the stream: sampled histogram -> guess -> select -> verify + order ->
// AP -> flag, AP and hit counts into pinned memory.  Pure enqueue (no synchronisation, no allocation once the
// buffers are warm), so it can run under stream capture.
// The verdict word, the APs and the hit counts into pinned memory behind everything enqueued so far (hg_get_ap then copies from there):
// one synchronisation for the whole call instead of a pageable 4-byte download, a wait, and hg_get_ap's own copies and wait.
// The step's results into pinned host memory by a KERNEL (the host block is device-addressable: stores go over the link as posted
// writes): a hipMemcpyAsync of these ~120 KB runs on a copy engine behind a cross-queue barrier and holds the stream ~20 us -- the
// gap between one step's last kernel and the next step's first (hg_map_begin) -- where this launch costs a few.
static __global__ __launch_bounds__(256) void k_copy_out(const uint4* __restrict__ src, uint4* __restrict__ dst, const u32 n16, const u32 tail_dwords) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i < tail_dwords) ((u32*)(dst + n16))[i] = ((const u32*)(src + n16))[i];
}

extern "C++" int stage_ap_download(hg_ctx* c, void* dst) {      // dst: a pinned block of its own (hg_map_begin), else (nullptr) the context's
    const size_t Q = (size_t)c->geo.Q;
    if (!dst) HG_TRY(ensure_pin(c, Q * 12 + 16));
    char* pb = (char*)(dst ? dst : c->pin);    // [flag: 16 B][ap Q x 8][rel Q x 4]
    const char* base = (const char*)c->outblk.p;
    if (base && c->outblk_q == (i64)Q && c->err.p == base && c->ap.p == base + 16 && c->rel.p == base + 16 + Q * 8) {
        const size_t bytes = 16 + Q * 12;      // the three are views of one block (ensure_out_block)
        static const bool by_engine = getenv("HG_COPY_ENGINE_DOWNLOAD") != nullptr;       // (A/B: the copy-engine form)
        if (by_engine || c->capturing) {
            HG_HIP(hipMemcpyAsync(pb, base, bytes, hipMemcpyDeviceToHost, c->stream));
            return HG_OK;
        }
        const u32 n16 = (u32)(bytes / 16), tail = (u32)((bytes % 16) / 4);
        hipLaunchKernelGGL(k_copy_out, dim3((n16 + 255) / 256 + (n16 % 256 == 0 && tail ? 1 : 0)), dim3(256), 0, c->stream, (const uint4*)base, (uint4*)pb, n16, tail);
        return c->check_launch("k_copy_out");
    }
    HG_HIP(hipMemcpyAsync(pb, c->err.p, 8, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(hipMemcpyAsync(pb + 16, c->ap.p, Q * 8, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(hipMemcpyAsync(pb + 16 + Q * 8, c->rel.p, Q * 4, hipMemcpyDeviceToHost, c->stream));
    return HG_OK;
}

static int enqueue_bet_with_ap(hg_ctx* c, int64_t R, int stride, u32 need_cnt, void* dst = nullptr) {
    c->t_step_begin();
    c->fuse_ap = true;
    const int rc = enqueue_optimistic(c, R, stride, need_cnt);
    c->fuse_ap = false;
    HG_TRY(rc);
    if (c->ap_fused) { c->ap_staged = false; c->stage |= ST_AP; }     // k_rank_cnt's epilogue left the APs (leftovers: run_oneshot)
    else HG_TRY(do_ap(c));
    HG_TRY(stage_ap_download(c, dst));         // [flag, leftover count: 16 B][ap Q x 8][rel Q x 4]
    c->t_step_end();
    return HG_OK;
}

// Second sighting of the same step (same tables, options, R, timing level; no buffer moved since): capture it.
static int capture_step(hg_ctx* c, int64_t R, int stride, u32 need_cnt) {
    c->drop_graph();
    const unsigned long long epoch0 = g_alloc_epoch.load();
    HG_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    c->capturing = true;
    const int rc = enqueue_bet_with_ap(c, R, stride, need_cnt);
    c->capturing = false;
    hipGraph_t gr = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &gr);
    if (rc != HG_OK || e != hipSuccess || !gr || g_alloc_epoch != epoch0) {
        if (gr) (void)hipGraphDestroy(gr);
        c->drop_graph();
        (void)hipGetLastError();
        if (rc != HG_OK) return rc;
        return fail(HG_ERR_HIP, "step capture failed: %s", e != hipSuccess ? hipGetErrorString(e) : "a buffer moved during capture");
    }
    hipGraphExec_t ex = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    if (e2 != hipSuccess) {
        (void)hipGraphDestroy(gr);
        c->drop_graph();
        return fail(HG_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
    }
    auto& sg = c->sg;
    sg.graph = gr; sg.exec = ex;
    sg.epoch = g_alloc_epoch; sg.cfg = c->cfg_epoch; sg.R = R; sg.timing = c->timing;
    sg.stage = c->stage; sg.optimistic = c->optimistic; sg.lists_valid = c->lists_valid; sg.cap = c->cap; sg.crow = c->crow;
    sg.RW = c->RW; sg.geo = c->geo; sg.ap_fused = c->ap_fused; sg.rec8 = c->rec8;
    c->graph_captures++;
    return HG_OK;
}

// After the synchronisation of a fused step (k_rank_cnt ranked AND evaluated): the queries it declined -- pin[1] of them,
// flagged in bigq; a list spanning more than 16 distances, a record beyond its counters -- get the general rank kernel and
// k_ap now, and the results come over again.  Rare by construction (none on any BASELINE workload), so the common step
// carries neither launch.
static int finish_leftovers(hg_ctx* c, int* flag) {
    if (!c->ap_fused) return HG_OK;
    const u32 nleft = ((const u32*)c->pin)[1];
    if ((nleft != 0) != c->leftovers_expected) { c->leftovers_expected = nleft != 0; c->cfg_epoch++; }
    if (!nleft) return HG_OK;
    c->opt_leftover += nleft;
    c->last_leftovers_inline = c->leftovers_inline;
    if (c->leftovers_inline) { c->leftovers_inline = false; return HG_OK; }     // k_rank_dense<slices> ranked them within the step
    const size_t Q = (size_t)c->geo.Q;
    int nbits = 1;
    while ((1 << nbits) < c->geo.NB) ++nbits;
    HG_TRY(launch_rank(c, 0, nbits, true));
    HG_TRY(do_ap_range(c, 0, c->geo.Q, c->bigq.as<u32>()));
    char* pb = (char*)c->pin;
    HG_HIP(hipMemcpyAsync(pb, c->err.p, 8, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(hipMemcpyAsync(pb + 16, c->ap.p, Q * 8, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(hipMemcpyAsync(pb + 16 + Q * 8, c->rel.p, Q * 4, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    *flag = *(const int*)c->pin;
    return HG_OK;
}

static int run_oneshot(hg_ctx* c, int64_t R, bool lists, bool with_ap) {
    c->real_lists = false;
    int stride = 0;
    u32 need_cnt = 0;
    if (R < 1 || R > c->n_total)
        return fail(HG_ERR_ARG, "R=%lld outside 1..N (N=%lld rows in the database)", (long long)R, (long long)c->n_total);
    c->want_lists = lists;
    if (with_ap && !c->is_sub) HG_TRY(ensure_out_block(c));
    const bool bet = optimistic_eligible(c, R, &stride, &need_cnt);
    int flag = 0;
    if (bet) {
        c->opt_runs++;
        if (with_ap) {
            HG_TRY(ensure_pin(c, (size_t)c->Q * 12 + 16));
            auto& sg = c->sg;
            bool launched = false;
            // event-record nodes inside a graph turned out slow and unreliable on ROCm 7.2 (a replayed step took 1.9 ms
            // instead of 1.55, elapsed times came back for one replay in twenty): with kernel timing on, steps stay eager
            if (c->opt_graph && !lists && !c->is_sub && c->timing == 0) {
                const bool same = sg.exec && sg.epoch == g_alloc_epoch && sg.cfg == c->cfg_epoch && sg.R == R && sg.timing == c->timing;
                const bool seen = sg.seen_epoch == g_alloc_epoch && sg.seen_cfg == c->cfg_epoch && sg.seen_R == R && sg.seen_timing == c->timing;
                if (!same && seen) {
                    if (capture_step(c, R, stride, need_cnt) != HG_OK) c->opt_graph = 0;      // not fatal: stay eager from now on
                }
                if (c->sg.exec && c->sg.epoch == g_alloc_epoch && c->sg.cfg == c->cfg_epoch && c->sg.R == R && c->sg.timing == c->timing) {
                    HG_HIP(hipGraphLaunch(sg.exec, c->stream));
                    HG_TRY(c->sync());
                    c->t_collect_graph();
                    // what the captured enqueue functions leave behind on the host side
                    HG_TRY(set_R(c, R, 1, 0));
                    c->geo = sg.geo; c->RW = sg.RW; c->stage = sg.stage; c->optimistic = sg.optimistic; c->lists_valid = sg.lists_valid;
                    c->cap = sg.cap; c->crow = sg.crow; c->err_zeroed = false; c->ap_fused = sg.ap_fused; c->rec8 = sg.rec8;
                    c->graph_replays++;
                    launched = true;
                }
            }
            if (!launched) {
                static const bool trace = getenv("HG_STEP_TRACE") != nullptr;        // debugging: where a slow step spent its time
                const auto tp0 = std::chrono::steady_clock::now();
                HG_TRY(enqueue_bet_with_ap(c, R, stride, need_cnt));
                const auto tp1 = std::chrono::steady_clock::now();
                HG_TRY(c->sync());
                if (trace) {
                    const auto tp2 = std::chrono::steady_clock::now();
                    const double e = std::chrono::duration<double, std::milli>(tp1 - tp0).count(), w = std::chrono::duration<double, std::milli>(tp2 - tp1).count();
                    if (e + w > 2.0) fprintf(stderr, "[hg] slow step: enqueue %.3f ms, wait %.3f ms (t_seq %lld, pending events %zu, pool %zu)\n", e, w, (long long)c->t_seq, c->pending.size(), c->pool.size());
                }
                sg.seen_epoch = g_alloc_epoch; sg.seen_cfg = c->cfg_epoch; sg.seen_R = R; sg.seen_timing = c->timing;
            }
            flag = *(const int*)c->pin;
            HG_TRY(finish_leftovers(c, &flag));
            c->ap_staged = flag == 0;
        } else {
            HG_TRY(enqueue_optimistic(c, R, stride, need_cnt));
            HG_TRY(read_plan_flag(c, &flag));
        }
        if (!flag) { c->opt_consecutive_fail = 0; return HG_OK; }
        bool handled = false;                      // some queries lost their bet
        HG_TRY(rerun_lost_queries(c, R, lists, with_ap, &handled));
        if (handled) { c->opt_consecutive_fail = 0; return HG_OK; }
        // many queries lost.  Before paying for the exact two-pass sequence (3x the bet at C2), bet once more with
        // twice the safety margin and twice the record budget -- the verification is what makes either bet exact.
        // Still lost: the hits crowd into few segments (a database stored class by class: ten classes put ten times the
        // mean into a query's slices), which no margin on the CUT cures -- escalate the slices' capacity (x8, x64, until a
        // slice would hold its whole segment) and remember what worked for the next calls on this database.
        if (c->opt_second_bet) {
            const i64 sigma0 = c->opt_sigma, budget0 = c->cand_budget_x10, boost0 = c->cap_boost;
            for (int attempt = 0; attempt < 3; ++attempt) {
                if (attempt > 0) {
                    if (c->cap >= (u32)((c->geo.L + 15) & ~15ll)) break;               // a slice already holds a segment
                    if ((double)c->geo.Q * (double)c->crow * 8.0 * 8.0 > 64e9) break;  // the record rows would not fit comfortably
                    c->cap_boost = c->cap_boost * 8 > 4096 ? 4096 : c->cap_boost * 8;
                }
                c->opt_sigma = 2 * sigma0 + 2;
                c->cand_budget_x10 = 2 * budget0;
                c->opt_rebets++;
                c->want_lists = lists;
                int rc;
                if (with_ap) {
                    rc = enqueue_bet_with_ap(c, R, stride, need_cnt);
                    if (rc == HG_OK) rc = c->sync();
                    flag = *(const int*)c->pin;
                    if (rc == HG_OK) rc = finish_leftovers(c, &flag);
                    c->ap_staged = rc == HG_OK && flag == 0;
                } else {
                    rc = enqueue_optimistic(c, R, stride, need_cnt);
                    if (rc == HG_OK) rc = read_plan_flag(c, &flag);
                }
                c->opt_sigma = sigma0;
                c->cand_budget_x10 = budget0;
                if (rc != HG_OK) { c->cap_boost = boost0; return rc; }
                // held with twice the budget of a first bet at this boost: the next call's first bet gets that budget
                // (a class-sorted database of tight clusters lost every first bet at x8 and won every second one)
                // -- only when a WIDENED attempt was the one that held: a held plain second bet (attempt 0) says the margin was
                // short this once, not that the slices are too small, and must not ratchet every later first bet's budget up
                auto keep = [&] {
                    if (attempt > 0 && c->cap_boost < 4096) c->cap_boost *= 2;
                    c->opt_consecutive_fail = 0;
                    c->cfg_epoch++;
                };
                if (!flag) { keep(); return HG_OK; }
                handled = false;
                HG_TRY(rerun_lost_queries(c, R, lists, with_ap, &handled));
                if (handled) { keep(); return HG_OK; }
            }
            c->cap_boost = boost0;                 // nothing helped: do not keep paying for big slices
        }
        c->opt_fallbacks++;                        // still too many: exact path for all
        c->opt_consecutive_fail++;
        c->want_lists = lists;
    }
    if (exact_mx_applies(c, R)) {
        c->want_lists = lists;
        c->t_step_begin();
        HG_TRY(enqueue_exact_mx(c, R));
        if (with_ap) {
            HG_TRY(do_ap(c));
            HG_TRY(stage_ap_download(c));
            c->t_step_end();
            HG_TRY(c->sync());
            flag = *(const int*)c->pin;
            c->ap_staged = flag == 0;
        } else {
            c->t_step_end();
            HG_TRY(read_plan_flag(c, &flag));
        }
        if (!flag) return HG_OK;
        c->want_lists = lists;                         // a slice overflowed: the vector-ALU select with exact-sized slices
    }
    c->t_step_begin();
    c->ap_fused = false;
    c->fuse_ap = with_ap && !lists;
    const int rce = enqueue_exact(c, R);
    c->fuse_ap = false;
    HG_TRY(rce);
    if (with_ap) {
        if (c->ap_fused) { c->ap_staged = false; c->stage |= ST_AP; }     // a rank kernel's epilogue left the APs
        else HG_TRY(do_ap(c));
    }
    c->ap_fused = false;                               // (no leftovers on this path: nothing for finish_leftovers)
    if (with_ap) {                                     // C1 (R = N through the byte matrix): 0.17 -> 0.14 ms per call
        HG_TRY(stage_ap_download(c));
        c->t_step_end();
        HG_TRY(c->sync());
        if (*(const int*)c->pin) {
            c->stage = ST_DB | ST_Q | ST_HIST;
            return fail(HG_ERR_ARG, "R=%lld exceeds the rows present in the gathered histograms", (long long)c->R);
        }
        c->ap_staged = true;
        return HG_OK;
    }
    c->t_step_end();
    return check_plan_flag(c);
}

int hg_topr(hg_ctx* c, int64_t R) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_topr", "hg_set_database + hg_set_queries"));
    return run_oneshot(c, R, true, false);
}

int hg_map(hg_ctx* c, int64_t R, double* host_ap, int64_t* host_rel) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_map", "hg_set_database + hg_set_queries"));
    const i64 fails0 = c->opt_fallbacks + c->opt_requeried + c->opt_rebets, left0 = c->opt_leftover;
    c->last_leftovers_inline = false;
    HG_TRY(run_oneshot(c, R, false, true));
    HG_TRY(hg_get_ap(c, host_ap, host_rel));
    // what hg_map_begin may enqueue without looking back: this very step, when it just won its bet outright
    // (queries the fused rank kernel declines are fine when the step ranks them itself within the stream: leftovers_inline)
    if (c->optimistic && c->opt_fallbacks + c->opt_requeried + c->opt_rebets == fails0 && (c->opt_leftover == left0 || c->last_leftovers_inline) && !c->is_sub) {
        c->map_warm_cfg = c->cfg_epoch; c->map_warm_epoch = g_alloc_epoch; c->map_warm_R = R;
    } else {
        c->map_warm_R = -1;
    }
    return HG_OK;
}

// hg_map in two halves, for a caller that evaluates batch after batch: hg_map_begin enqueues a step (kernels and the download of
// its verdict, APs and hit counts into a pinned block of its own) and returns; hg_map_end waits for the OLDEST step in flight and
// hands its results over.  Two steps may be in flight, so the GPU starts step i + 1 the moment step i ends -- the host's wake-up,
// its copies and its next enqueue (~40 us per step at C2) no longer sit between them.  A step is only enqueued blind when the
// last synchronous hg_map with the same tables, options and R won its bet outright; otherwise hg_map_begin runs the whole call
// itself (and keeps the results for hg_map_end).  A blind step that loses its bet is run again, synchronously, by hg_map_end:
// results are those of hg_map in every case.
int hg_map_begin(hg_ctx* c, int64_t R) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_map_begin", "hg_set_database + hg_set_queries"));
    if (c->ms_n == 2) return fail(HG_ERR_STATE, "hg_map_begin: two steps are in flight already (hg_map_end takes the oldest)");
    hg_ctx::MapSlot& m = c->mslot[(c->ms_head + c->ms_n) & 1];
    m.R = R; m.Q = c->Q;
    m.q_gen = c->q_gen; m.db_gen = c->db_gen;
    int stride = 0;
    u32 need_cnt = 0;
    bool blind = c->map_warm_R == R && c->map_warm_cfg == c->cfg_epoch && c->map_warm_epoch == g_alloc_epoch && !c->is_sub;
    if (blind) {
        c->real_lists = false;
        c->want_lists = false;
        HG_TRY(ensure_out_block(c));
        blind = optimistic_eligible(c, R, &stride, &need_cnt) && c->map_warm_epoch == g_alloc_epoch;
    }
    if (blind) {
        const size_t need_b = (size_t)c->Q * 12 + 16;
        if (m.cap < need_b) {
            pin_free(m.pin, m.cap);
            m.pin = nullptr; m.cap = 0;
            HG_HIP(pin_alloc(&m.pin, need_b, &m.cap));
        }
        if (!m.ev) HG_HIP(hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
        c->opt_runs++;
        HG_TRY(enqueue_bet_with_ap(c, R, stride, need_cnt, m.pin));
        HG_HIP(hipEventRecord(m.ev, c->stream));
        m.inline_ok = c->leftovers_inline;               // the step ranks what its fused kernel declines within the stream
        c->leftovers_inline = false;
        c->ap_staged = false;
        m.async = true;
        c->map_async_steps++;
    } else {
        m.ap.resize((size_t)c->Q); m.rel.resize((size_t)c->Q);
        HG_TRY(hg_map(c, R, m.ap.data(), m.rel.data()));
        m.async = false;
    }
    ++c->ms_n;
    return HG_OK;
}

int hg_map_end(hg_ctx* c, double* host_ap, int64_t* host_rel) {
    if (!c) return fail(HG_ERR_ARG, "hg_map_end: null context");
    HG_TRY(c->use());
    if (!c->ms_n) return fail(HG_ERR_STATE, "hg_map_end: no step in flight (hg_map_begin first)");
    hg_ctx::MapSlot& m = c->mslot[c->ms_head];
    c->ms_head ^= 1; --c->ms_n;                           // (taken off the queue whatever happens below)
    const size_t Q = (size_t)m.Q;
    if (!m.async) {
        if (host_ap) memcpy(host_ap, m.ap.data(), Q * 8);
        if (host_rel) memcpy(host_rel, m.rel.data(), Q * 8);
        return HG_OK;
    }
    HG_HIP(hipEventSynchronize(m.ev));
    const u32* w = (const u32*)m.pin;                      // [verdict][queries the fused rank kernel declined] ...
    if (w[0] == 0 && (w[1] == 0 || m.inline_ok)) {         // (won: the results are this step's whatever the tables hold by now)
        const char* pb = (const char*)m.pin;
        if (host_ap) memcpy(host_ap, pb + 16, Q * 8);
        if (host_rel) {
            const u32* r = (const u32*)(pb + 16 + Q * 8);
            for (size_t q = 0; q < Q; ++q) host_rel[q] = r[q];
        }
        return HG_OK;
    }
    // the bet was lost (or queries were left over): the synchronous call sorts that out -- reruns, deeper bets, wider slices -- on
    // the same tables; a younger step in flight keeps its own verdict in its own block
    c->map_warm_R = -1;
    if (m.q_gen != c->q_gen || m.db_gen != c->db_gen || (i64)Q != c->Q)
        return fail(HG_ERR_STATE, "hg_map_end: the %s replaced while a step that lost its bet was in flight: its tables are gone, "
                                  "load them again and call hg_map", m.db_gen != c->db_gen ? "database was" : "queries were");
    c->map_async_redone++;
    return hg_map(c, m.R, host_ap, host_rel);
}

}  // extern "C"

// hg_preload: the runtime loads a translation unit's code object when one of its kernels is first needed (milliseconds);
// asking for a kernel's attributes does that now
int preload_seq() {
    hipFuncAttributes a;
    HG_HIP(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_copy_out)));
    return HG_OK;
}
