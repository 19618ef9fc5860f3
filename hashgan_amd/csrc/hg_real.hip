// libhashgan_amd.so -- real-valued (float32 inner product) ranking, what main.py:157,164 feeds lib/metric.py:13-14:
// sampled cut -> bf16 filter on the matrix cores -> exact float32 rescoring -> LDS ranking.
#include "hg_ctx.hpp"
#include "hg_real_kernels.hpp"
#include "hg_real_mx.hpp"
#include "hg_real_bf.hpp"

constexpr double REAL_FIRST_SIGMA = 5.0;        // depth of the first cut in deviations of the sampled count: one query in ~3e5 loses it and is ranked again on its
                                              // own (real_requery_lost).  10k x 1M x 64, R = 5000: 6 -> 5.92 ms per call, 5 -> 5.81, 4 -> 5.86, 3.5 -> 6.01 (the
                                              // rescore's time follows its rounds, not its rows: the shallower cut mostly helps the rank stage)
constexpr i64 REAL_SAMPLE_HITS = 64;          // the real-valued bet samples so that this many of a query's top R rows are in the sample (tools/real_sample_sweep.py: 32 .. 256 measured)
constexpr i64 REAL_SECOND_SAMPLE = 4;         // the second, counting sample expects four times REAL_SAMPLE_HITS of a query's top R rows: 256
constexpr i64 REAL_FIRST_HITS_BRACKET = 24;   // ... and the first sample, when a second one follows, this many (10k x 1M x 64, R = 5000, call: 64 -> 4.86 ms, 48 -> 4.76, 32 -> 4.72,
                                              // 24 -> 4.68, 16 -> 4.65, 12 -> 4.70, 8 -> 4.80 -- below 16 the bins of the second sample no longer reach the cut; 6: lists beyond the LDS)
constexpr i64 REAL_SEG_BYTES = 512 * 1024;    // bytes of feature rows per segment of the real-valued pair passes

namespace {
template <int BP> int real_launch_sample(hg_ctx* c, i64 M, i64 stride) {
    const Geo& g = c->geo;
    const i64 units = (M + 63) / 64 * g.nQT;
    c->t_begin(KI_REAL_SAMPLE);
    hipLaunchKernelGGL(k_real_sample<BP>, dim3(grid_for(units, WPB)), dim3(256), (size_t)WPB * 64 * 65 * 4, c->stream,
                       c->qf.as<float>(), c->dbf.as<float>(), c->samp.as<float>(), M, stride, g);
    c->t_end();
    return c->check_launch("k_real_sample");
}
template <int BP, int QPL> int real_launch_select_q(hg_ctx* c) {
    Geo g = c->geo;
    g.nQT = (g.Q + 64 * QPL - 1) / (64 * QPL);
    g.nUnits = (i64)g.S * g.nQT;
    g.wpb = WPB;
    g.nBlk = (int)((g.nUnits + WPB - 1) / WPB);
    RealSelArgs a{c->thr.as<float>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow};
    c->t_begin(KI_REAL_SELECT);
    hipLaunchKernelGGL((k_real_select<BP, QPL>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qf.as<float>(),
                       c->dbf.as<float>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_real_select");
}
template <int BP> int real_launch_select(hg_ctx* c) {
    // queries per lane (see k_real_select): two while their features fit the register file comfortably
    return real_launch_select_q<BP, 1>(c);
}
// real-valued select on the matrix cores: blocks = (pair of segments) x (256 queries)
template <int KP> int real_launch_select_mx(hg_ctx* c) {
    if (!c->dbfx_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbfx.reserve((size_t)n16 * KP * 4));
        const i64 items = n16 * (KP / 4);
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_dbf, dim3(grid_for(items)), dim3(256), 0, c->stream, c->dbf.as<float>(), c->dbfx.as<float4>(),
                           (i64)c->N, n16, KP, (i64)1);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_dbf"));
        c->dbfx_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 32 * RMX_QT - 1) / (WPB * 32 * RMX_QT);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    RealSelArgs a{c->thr.as<float>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow};
    c->t_begin(KI_REAL_SELECT);
    hipLaunchKernelGGL((k_real_select_mx<KP>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream,
                       c->qf.as<float>(), c->dbfx.as<u8>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_real_select_mx");
}
// filter + rescore (hg_real_bf.hpp): bf16 pair pass with a rigorous margin, then the exact chain for the survivors
// the filter's 16-bit image of the database (and the choice between IEEE half and bfloat16), built on first use
int ensure_filter_image(hg_ctx* c) {
    const int KP = c->bpad;
    if (!c->dbfb_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbfb.reserve((size_t)n16 * KP * 2));
        HG_TRY(c->xmax2.reserve(4));
        HG_HIP(hipMemsetAsync(c->xmax2.p, 0, 4, c->stream));
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_row_norm_max, dim3(grid_for(c->N)), dim3(256), 0, c->stream, c->dbf.as<float>(), (i64)c->N, KP, c->xmax2.as<u32>());
        // Which 16-bit format the filter's image takes -- once per database, one 4-byte download: IEEE half (three more significant
        // bits: a margin an eighth of bfloat16's) when no feature can overflow it (every |x_k| <= the row's norm < 2^15), else bfloat16
        float xm = 0.0f;
        HG_HIP(hipMemcpyAsync(&xm, c->xmax2.p, 4, hipMemcpyDeviceToHost, c->stream));
        HG_TRY(c->sync());
        // (... and when the rows are not tiny either: half's error has an absolute floor -- subnormals, flushed or not -- that outgrows
        // the relative term once norms fall below ~0.1; bfloat16 has float32's exponents and no such floor)
        c->dbfb_half = c->opt_real_mfma == 2 && xm >= 1.0f && xm < 1073741824.0f;         // 1 <= largest row norm^2 < 2^30 (inf and the NaN marker fail the test)
        if (c->dbfb_half) hipLaunchKernelGGL(k_expand_dbf_bf16<true>, dim3(grid_for(n16 * (KP / 8))), dim3(256), 0, c->stream, c->dbf.as<float>(),
                                             c->dbfb.as<uint4>(), (i64)c->N, n16, KP, (i64)1);
        else hipLaunchKernelGGL(k_expand_dbf_bf16<false>, dim3(grid_for(n16 * (KP / 8))), dim3(256), 0, c->stream, c->dbf.as<float>(),
                                c->dbfb.as<uint4>(), (i64)c->N, n16, KP, (i64)1);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_dbf_bf16"));
        c->dbfb_valid = true;
    }
    return HG_OK;
}
template <int KP> int real_launch_select_bf(hg_ctx* c) {
    constexpr int QT = KP <= 128 ? 2 : 1;
    HG_TRY(ensure_filter_image(c));
    Geo g = c->geo;
    HG_TRY(c->thr2.reserve((size_t)g.Qpad * 4));
    c->t_begin(KI_REAL_GUESS);
    hipLaunchKernelGGL(k_real_thr2, dim3(grid_for(g.Q)), dim3(256), 0, c->stream, c->qf.as<float>(), c->thr.as<float>(),
                       c->xmax2.as<u32>(), c->thr2.as<float>(), g.Q, KP, c->dbfb_half ? 1.0 / 1024.0 : 1.0 / 256.0, c->dbfb_half ? 1.0 / 16384.0 : 0.0);
    c->t_end();
    HG_TRY(c->check_launch("k_real_thr2"));
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 32 * QT - 1) / (WPB * 32 * QT);
    Geo gs = g;
    gs.nQT = nQB;
    gs.nUnits = (i64)nSP * nQB;
    gs.wpb = WPB;
    gs.nBlk = (int)gs.nUnits;
    RealSelArgs a{c->thr.as<float>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow};
    HG_TRY(c->krows.reserve((size_t)g.Q * c->crow * 4));     // the kept rows' numbers: the filter writes, the rescore reads
    c->t_begin(KI_REAL_SELECT);
#define HG_FILTER(HALF_, FAR_)                                                                                                               \
    do {                                                                                                                                     \
        if (real_bf_lds_bytes(KP) > 64 * 1024)                                                                                               \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_select_bf<KP, QT, HALF_, FAR_>), hipFuncAttributeMaxDynamicSharedMemorySize, real_bf_lds_bytes(KP))); \
        hipLaunchKernelGGL((k_real_select_bf<KP, QT, HALF_, FAR_>), dim3(padded_grid(gs.nBlk)), dim3(256), real_bf_lds_bytes(KP), c->stream, \
                           c->qf.as<float>(), c->dbfb.as<u8>(), c->thr2.as<float>(), a, c->krows.as<u32>(), gs);                            \
    } while (0)
    // (a wavefront's 64 record rows within 4 GB: 32-bit cursors -- every bet; beyond, e.g. every row a record of a 10M-row database, 64-bit ones)
    // (a 32-bit cursor keeps counting past a full slice -- by up to a segment's rows -- so that the hits it dropped are known: the
    // furthest it can get is the wavefront's 64 record rows plus one segment)
    const bool far_rows = (64ull * (unsigned long long)c->crow + (unsigned long long)g.L) * 4ull >= (1ull << 32);
    if (c->dbfb_half) { if (far_rows) HG_FILTER(true, true); else HG_FILTER(true, false); }
    else { if (far_rows) HG_FILTER(false, true); else HG_FILTER(false, false); }
#undef HG_FILTER
    c->t_end();
    HG_TRY(c->check_launch("k_real_select_bf"));
    // slices per wavefront of the rescoring pass.  The kernel's time follows its ROUNDS of 64 rows and its wavefronts, hardly its rows
    // (10k x 1M x 64, R = 5000, 17 rows per slice: 1 slice 4.04 ms, 2 2.41, 3 2.06, 4 2.12, 5 2.05, 6 1.90, 7 1.92, 8 2.07, 11 2.20), so
    // the count that fills its rounds best: expected rounds of a wavefront (the rows a query is expected to keep -- the rank its cut was
    // guessed at, scaled back from the sample; every row without a cut -- over its slices, Poisson-ish) plus half a round of fixed cost,
    // a little more per round the more slices the lane-to-slice search walks, per slice
    const double per_slice = 1.03 * c->real_expect / (double)g.S;
#ifdef HG_RS_FORCE
    const int SG = HG_RS_FORCE;
    (void)per_slice;
#else
    int SG = 1;
    {
        double best = 1e300;
        for (const int sg : {1, 2, 3, 4, 6, 8}) {
            if (per_slice >= 64.0 && sg > 2) break;          // (slices of full rounds: nothing to pack, keep the rows' L2 footprint small)
            const double m = per_slice * sg, sd = std::sqrt(m > 1.0 ? m : 1.0);
            double rounds = 1.0;
            for (int k = 1; k <= 64; ++k) {
                const double p = 0.5 * std::erfc((64.0 * k - m) / sd * 0.7071067811865476);
                rounds += p;
                if (p < 1e-6) break;
            }
            const double cost = (0.5 + rounds * (1.0 + 0.03 * sg)) / sg;
            if (cost < best) { best = cost; SG = sg; }
        }
    }
#endif
    const i64 waves = (i64)((g.S + SG - 1) / SG) * g.Q;
    HG_TRY(c->cntq.reserve((size_t)g.Q * g.S * 4));
    c->t_begin(KI_REAL_RESCORE);
#define HG_RESCORE(sg)                                                                                                                   \
    case sg:                                                                                                                             \
        hipLaunchKernelGGL((k_real_rescore<(KP <= 128 ? KP : 0), sg>), dim3(grid_for(waves, WPB)), dim3(256), rescore_lds_bytes(), c->stream, c->qf.as<float>(),  \
                           c->dbf.as<float>(), c->sl_cnt.as<u32>(), c->krows.as<u32>(), c->cand.as<u64>(), c->cap, c->crow, c->thr.as<float>(), \
                           c->sl_cnt.as<u32>(), c->cntq.as<u32>(), c->dblab.as<u64>(), c->qlab.as<u64>(), c->real_no_cut ? 0 : 1, KP, g);                         \
        break;
#ifdef HG_RS_FORCE
    switch (SG) { HG_RESCORE(HG_RS_FORCE) }
#else
    switch (SG) { HG_RESCORE(8) HG_RESCORE(6) HG_RESCORE(4) HG_RESCORE(3) HG_RESCORE(2) HG_RESCORE(1) }
#endif
#undef HG_RESCORE
    c->t_end();
    c->real_filtered = true;
    return c->check_launch("k_real_rescore");
}
int real_select_bf(hg_ctx* c) {
    switch (c->bpad) {
        case 16: return real_launch_select_bf<16>(c);
        case 32: return real_launch_select_bf<32>(c);
        case 48: return real_launch_select_bf<48>(c);
        case 64: return real_launch_select_bf<64>(c);
        case 80: return real_launch_select_bf<80>(c);
        case 96: return real_launch_select_bf<96>(c);
        case 112: return real_launch_select_bf<112>(c);
        case 128: return real_launch_select_bf<128>(c);
        case 144: return real_launch_select_bf<144>(c);
        case 160: return real_launch_select_bf<160>(c);
        case 176: return real_launch_select_bf<176>(c);
        case 192: return real_launch_select_bf<192>(c);
        case 208: return real_launch_select_bf<208>(c);
        case 224: return real_launch_select_bf<224>(c);
        case 240: return real_launch_select_bf<240>(c);
        case 256: return real_launch_select_bf<256>(c);
        default: return fail(HG_ERR_ARG, "real-valued ranking supports up to 256 features (have %d)", c->b);
    }
}
int real_select_mx(hg_ctx* c) {
    switch (c->bpad) {
        case 16: return real_launch_select_mx<16>(c);
        case 32: return real_launch_select_mx<32>(c);
        case 48: return real_launch_select_mx<48>(c);
        case 64: return real_launch_select_mx<64>(c);
        case 80: return real_launch_select_mx<80>(c);
        case 96: return real_launch_select_mx<96>(c);
        case 112: return real_launch_select_mx<112>(c);
        case 128: return real_launch_select_mx<128>(c);
        default: return fail(HG_ERR_ARG, "real-valued ranking supports up to 128 features (have %d)", c->b);
    }
}

#define HG_DISPATCH_BP(fn, c, ...)                                  \
    switch ((c)->bpad / 2) {                                        \
        case 8: return fn<8>(c, ##__VA_ARGS__);                     \
        case 16: return fn<16>(c, ##__VA_ARGS__);                   \
        case 24: return fn<24>(c, ##__VA_ARGS__);                   \
        case 32: return fn<32>(c, ##__VA_ARGS__);                   \
        case 40: return fn<40>(c, ##__VA_ARGS__);                   \
        case 48: return fn<48>(c, ##__VA_ARGS__);                   \
        case 56: return fn<56>(c, ##__VA_ARGS__);                   \
        case 64: return fn<64>(c, ##__VA_ARGS__);                   \
        default: return fail(HG_ERR_ARG, "real-valued ranking supports up to 128 features (have %d)", (c)->b); \
    }
// sample pass on the float32 MFMA: image of the M sampled rows (rebuilt per call: a few MB), 16 segments
template <int KP> int real_launch_sample_mx(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    const i64 m16 = (M + 15) / 16 * 16;
    HG_TRY(c->sampx.reserve((size_t)m16 * KP * 4));
    c->t_begin(KI_REAL_SAMPLE);
    hipLaunchKernelGGL(k_expand_dbf, dim3(grid_for(m16 * (KP / 4))), dim3(256), 0, c->stream, c->dbf.as<float>(), c->sampx.as<float4>(),
                       M, m16, KP, stride);
    Geo g = c->geo;
    g.N = M;
    i64 L = (M + 31) / 32;                               // ~32 segments (16 pairs) of a multiple of 16 rows
    L = (L + 15) / 16 * 16;
    g.L = L;
    g.S = (int)((M + L - 1) / L);
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 32 * RMX_QT - 1) / (WPB * 32 * RMX_QT);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    hipLaunchKernelGGL((k_real_sample_mx<KP>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qf.as<float>(), c->sampx.as<u8>(),
                       c->samp.as<float>(), mstride, g);
    c->t_end();
    return c->check_launch("k_real_sample_mx");
}
// sample pass in the filter's 16-bit arithmetic (k_real_sample_h): the sampled rows' image rebuilt per call (1.6 MB at 10k x 1M), ~32 segments
template <int KP> int real_launch_sample_h(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    HG_TRY(ensure_filter_image(c));                      // (decides half / bfloat16 for this database)
    const i64 m16 = (M + 15) / 16 * 16;
    HG_TRY(c->sampx.reserve((size_t)m16 * KP * 2));
    c->t_begin(KI_REAL_SAMPLE);
    if (c->dbfb_half) hipLaunchKernelGGL(k_expand_dbf_bf16<true>, dim3(grid_for(m16 * (KP / 8))), dim3(256), 0, c->stream, c->dbf.as<float>(),
                                         c->sampx.as<uint4>(), M, m16, KP, stride);
    else hipLaunchKernelGGL(k_expand_dbf_bf16<false>, dim3(grid_for(m16 * (KP / 8))), dim3(256), 0, c->stream, c->dbf.as<float>(),
                            c->sampx.as<uint4>(), M, m16, KP, stride);
    Geo g = c->geo;
    g.N = M;
    i64 L = (M + 31) / 32;                               // ~32 segments (16 pairs) of a multiple of 16 rows
    L = (L + 15) / 16 * 16;
    g.L = L;
    g.S = (int)((M + L - 1) / L);
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 64 - 1) / (WPB * 64);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
#define HG_SAMPLE_H(HALF_, O16_) hipLaunchKernelGGL((k_real_sample_h<KP, HALF_, O16_>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qf.as<float>(), \
                                                     c->sampx.as<u8>(), c->samp.as<float>(), mstride, g)
    if (c->dbfb_half) { if (c->samp16) HG_SAMPLE_H(true, true); else HG_SAMPLE_H(true, false); }
    else { if (c->samp16) HG_SAMPLE_H(false, true); else HG_SAMPLE_H(false, false); }
#undef HG_SAMPLE_H
    c->t_end();
    return c->check_launch("k_real_sample_h");
}
int real_sample_h(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    switch (c->bpad) {
        case 16: return real_launch_sample_h<16>(c, M, stride, mstride);
        case 32: return real_launch_sample_h<32>(c, M, stride, mstride);
        case 48: return real_launch_sample_h<48>(c, M, stride, mstride);
        case 64: return real_launch_sample_h<64>(c, M, stride, mstride);
        case 80: return real_launch_sample_h<80>(c, M, stride, mstride);
        case 96: return real_launch_sample_h<96>(c, M, stride, mstride);
        case 112: return real_launch_sample_h<112>(c, M, stride, mstride);
        default: return real_launch_sample_h<128>(c, M, stride, mstride);
    }
}
// the second, counting sample (k_real_sample_count + k_real_guess2): thr[q] moves up to the deepest cut a four times larger sample supports
template <int KP> int real_launch_sample_count(hg_ctx* c, i64 M2, i64 stride2, u32 need2) {
    const Geo& g0 = c->geo;
    const i64 m16 = (M2 + 15) / 16 * 16;
    HG_TRY(c->sampx.reserve((size_t)m16 * KP * 2));     // (sized for this pass before the first one ran: real_attempt)
    c->t_begin(KI_REAL_SAMPLE);
    if (c->dbfb_half) hipLaunchKernelGGL(k_expand_dbf_bf16<true>, dim3(grid_for(m16 * (KP / 8))), dim3(256), 0, c->stream, c->dbf.as<float>(),
                                         c->sampx.as<uint4>(), M2, m16, KP, stride2);
    else hipLaunchKernelGGL(k_expand_dbf_bf16<false>, dim3(grid_for(m16 * (KP / 8))), dim3(256), 0, c->stream, c->dbf.as<float>(),
                            c->sampx.as<uint4>(), M2, m16, KP, stride2);
    Geo g = g0;
    g.N = M2;
    i64 L = (M2 + 63) / 64;                              // ~64 segments (32 pairs) of a multiple of 16 rows (16 .. 256 segments measured: 32 and up the same)
    L = (L + 15) / 16 * 16;
    g.L = L;
    g.S = (int)((M2 + L - 1) / L);
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + WPB * 64 - 1) / (WPB * 64);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    HG_TRY(c->hist2.reserve((size_t)nSP * g.Qpad * RC_BINS * 4 + (size_t)WPB * 64 * RC_BINS * 4));      // [segment pair][Qpad][bin], every word written (+ a block's overhang past Qpad)
    constexpr int lds = real_count_lds_bytes(KP);
    if (c->dbfb_half) {
        if (lds > 64 * 1024) HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_sample_count<KP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((k_real_sample_count<KP, true>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qf.as<float>(), c->sampx.as<u8>(),
                           c->thr.as<float>(), c->hist2.as<u32>(), g);
    } else {
        if (lds > 64 * 1024) HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_sample_count<KP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((k_real_sample_count<KP, false>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qf.as<float>(), c->sampx.as<u8>(),
                           c->thr.as<float>(), c->hist2.as<u32>(), g);
    }
    c->t_end();
    HG_TRY(c->check_launch("k_real_sample_count"));
    c->t_begin(KI_REAL_GUESS);
    hipLaunchKernelGGL(k_real_guess2, dim3(grid_for(g0.Q, 8)), dim3(256), 0, c->stream, c->hist2.as<u32>(), c->thr.as<float>(), g0.Q, (i64)g0.Qpad, nSP, need2);
    c->t_end();
    return c->check_launch("k_real_guess2");
}
int real_sample_count(hg_ctx* c, i64 M2, i64 stride2, u32 need2) {
    switch (c->bpad) {
        case 16: return real_launch_sample_count<16>(c, M2, stride2, need2);
        case 32: return real_launch_sample_count<32>(c, M2, stride2, need2);
        case 48: return real_launch_sample_count<48>(c, M2, stride2, need2);
        case 64: return real_launch_sample_count<64>(c, M2, stride2, need2);
        case 80: return real_launch_sample_count<80>(c, M2, stride2, need2);
        case 96: return real_launch_sample_count<96>(c, M2, stride2, need2);
        case 112: return real_launch_sample_count<112>(c, M2, stride2, need2);
        default: return real_launch_sample_count<128>(c, M2, stride2, need2);
    }
}
int real_sample_mx(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    switch (c->bpad) {
        case 16: return real_launch_sample_mx<16>(c, M, stride, mstride);
        case 32: return real_launch_sample_mx<32>(c, M, stride, mstride);
        case 48: return real_launch_sample_mx<48>(c, M, stride, mstride);
        case 64: return real_launch_sample_mx<64>(c, M, stride, mstride);
        case 80: return real_launch_sample_mx<80>(c, M, stride, mstride);
        case 96: return real_launch_sample_mx<96>(c, M, stride, mstride);
        case 112: return real_launch_sample_mx<112>(c, M, stride, mstride);
        default: return real_launch_sample_mx<128>(c, M, stride, mstride);
    }
}
int real_sample(hg_ctx* c, i64 M, i64 stride, i64 mstride) {
    // with the 16-bit filter behind it the sample runs in the same arithmetic (its scores only place the cut); "real_mfma" 1 keeps the exact chains
    if (c->bpad <= 128 && c->opt_real_mfma == 2 && c->opt_real_sample_h && c->geo.L % 16 == 0) return real_sample_h(c, M, stride, mstride);
    if (c->bpad <= 128 && c->opt_real_mfma) return real_sample_mx(c, M, stride, mstride);
    if (c->bpad > 128) {                                 // k_real_sample keeps the query in registers: the staged form beyond
        const Geo& g = c->geo;
        const i64 units = (M + 63) / 64 * g.nQT;
        const size_t lds = (size_t)WPB * (64 * 65 * 4 + 16 * 128);
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_sample_any), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->t_begin(KI_REAL_SAMPLE);
        hipLaunchKernelGGL(k_real_sample_any, dim3(grid_for(units, WPB)), dim3(256), lds, c->stream, c->qf.as<float>(), c->dbf.as<float>(),
                           c->samp.as<float>(), M, stride, c->bpad, g);
        c->t_end();
        return c->check_launch("k_real_sample_any");
    }
    (void)mstride;                                       // (the vector kernels write samp[q][M] densely: the caller passes mstride = M)
    HG_DISPATCH_BP(real_launch_sample, c, M, stride)
}
int real_select(hg_ctx* c) {
    c->real_filtered = false;
    // without a cut (every row a record: R = N, or after lost bets) a filter filters nothing and every pair would be rescored:
    // the exact float32 MFMA pass gives the scores at once (C1: 4.6 -> 4.0 ms per call)
    const bool filter = c->opt_real_mfma == 2 && !(c->real_no_cut && c->bpad <= 128);
    if ((filter || c->bpad > 128) && c->geo.L % 16 == 0) return real_select_bf(c);   // (the only pass for > 128 features)
    if (c->opt_real_mfma && c->geo.L % 16 == 0) return real_select_mx(c);
    HG_DISPATCH_BP(real_launch_select, c)
}


}  // namespace

extern "C" {

// ---- real-valued ranking: sample -> guess -> select -> 4-pass radix sort -> finish ----
// one attempt; *lost = some query came up short of R records or overflowed a slice (bet mode only)
static int real_attempt(hg_ctx* c, int64_t R, bool bet, double sigma, double budget, bool with_ap, int* lost) {
    ++c->real_attempts;
    c->real_expect = bet ? (double)R * (1.0 + sigma / std::sqrt((double)REAL_SAMPLE_HITS)) : (double)c->N;      // (a second sample lowers it: below)
    c->real_lds_ranked = 0;
    c->real_no_cut = !bet;
    make_geometry(c);
    {   // Float rows are 4*bpad bytes (32x a 64-bit code): keep a segment's rows within ~512 KB so the few
        // segments an XCD works on at a time stay in its 4 MiB L2 while all query tiles pass over them.
        Geo& gg = c->geo;
        i64 L = (i64)REAL_SEG_BYTES / ((i64)c->bpad * 4);
        L = L / 16 * 16;
        if (L < 64) L = 64;
        if (gg.L > L) {
            gg.L = L;
            gg.S = (int)((gg.N + L - 1) / L);
            gg.nUnits = (i64)gg.S * gg.nQT;
            gg.nBlk = (int)((gg.nUnits + WPB - 1) / WPB);
        }
    }
    if (!bet && c->bpad <= 128 && c->opt_real_mfma && c->opt_real_rounds > 0) {
        // Every row a record through k_real_select_mx: blocks = (pairs of segments) x (256 queries), four wavefronts each, up to
        // three resident per CU (two beyond 64 features).  The plain geometry gave the CIFAR evaluation (1000 x 54 000) 376 blocks
        // for 256 CUs -- half the CUs with two, half with one; cut the database so that the blocks fill whole rounds instead.
        Geo& gg = c->geo;
        const i64 nQB = (gg.Q + WPB * 32 * RMX_QT - 1) / (WPB * 32 * RMX_QT);
        const i64 per_cu = c->bpad <= 64 ? std::min<i64>(c->opt_real_rounds, 3) : std::min<i64>(c->opt_real_rounds, 2);
        const i64 slots = (i64)c->n_cu * per_cu;
        const i64 nSP0 = (gg.S + 1) / 2;
        i64 k = (nSP0 * nQB + slots - 1) / slots;          // rounds the plain geometry touches
        if (k < 1) k = 1;
        const i64 nSP = slots * k / nQB;
        if (nSP >= 1) {
            i64 L = (gg.N + 2 * nSP - 1) / (2 * nSP);
            L = (L + 15) / 16 * 16;
            if (L < 64) L = 64;
            if (L <= gg.L) {
                gg.L = L;
                gg.S = (int)((gg.N + L - 1) / L);
                gg.nUnits = (i64)gg.S * gg.nQT;
                gg.nBlk = (int)((gg.nUnits + WPB - 1) / WPB);
            }
        }
    }
    HG_TRY(set_R(c, R, 1, 0));
    const Geo& g = c->geo;
    const size_t qb = (size_t)g.Qpad * 4;
    HG_TRY(c->thr.reserve(qb)); HG_TRY(c->sl_cnt.reserve((size_t)g.S * qb)); HG_TRY(c->failq.reserve(qb));
    HG_TRY(c->tot.reserve(qb)); HG_TRY(c->err.reserve(16)); HG_TRY(c->qbad.reserve(qb));
    HG_HIP(hipMemsetAsync(c->failq.p, 0, qb, c->stream));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    if (bet) {
        // sample so that about 64 of a query's top R rows are in it; guess the cut `sigma` deviations deep.  With a second, counting sample
        // behind it (256 expected hits) the first one only has to bracket the cut from below: half the rows do (REAL_FIRST_HITS_BRACKET)
        const bool can16 = c->bpad <= 128 && c->opt_real_mfma == 2 && c->opt_real_sample_h && c->geo.L % 16 == 0;
        const i64 hits_b = REAL_FIRST_HITS_BRACKET;
        const i64 stride2 = (i64)((double)R / (double)(REAL_SAMPLE_HITS * REAL_SECOND_SAMPLE));
        i64 stride = (i64)((double)R / (double)hits_b);
        bool second = can16 && c->opt_real_second && stride2 >= 1 && stride >= 2 * stride2 && (c->N + stride - 1) / stride <= RG_MMAX;
        if (!second) stride = (i64)((double)R / (double)REAL_SAMPLE_HITS);
        if (stride < 1) stride = 1;
        const i64 M = (c->N + stride - 1) / stride;
        const double fr = (double)R * (double)M / (double)c->N;
        const u32 rank_s = (u32)std::ceil(fr + sigma * std::sqrt(fr)) + 1u;
        // samp[q][mstride]: the matrix-core sample pass stores 16 samples at a time (rows 64-byte aligned), the vector kernels M densely
        const i64 mstride = c->bpad <= 128 && c->opt_real_mfma ? (M + 15) / 16 * 16 : M;
        HG_TRY(c->samp.reserve((size_t)g.Q * mstride * 4));
        // 16-bit sample scores when both ends take them: k_real_sample_h writes, k_real_guess_lds reads
        c->samp16 = M <= RG_MMAX && can16;
        // ... and then the second, counting sample tightens the cut (k_real_sample_count)
        const i64 M2 = second ? (c->N + stride2 - 1) / stride2 : 0;
        if (second) HG_TRY(c->sampx.reserve((size_t)((M2 + 15) / 16 * 16) * c->bpad * 2));      // (before the first pass reads it: no move between the two)
        HG_TRY(real_sample(c, M, stride, mstride));
        c->t_begin(KI_REAL_GUESS);
        if (M <= RG_MMAX && c->samp16) hipLaunchKernelGGL(k_real_guess_lds<true>, dim3(g.Q), dim3(1024), 0, c->stream, c->samp.as<float>(), M, mstride, rank_s, c->thr.as<float>());
        else if (M <= RG_MMAX) hipLaunchKernelGGL(k_real_guess_lds<false>, dim3(g.Q), dim3(1024), 0, c->stream, c->samp.as<float>(), M, mstride, rank_s, c->thr.as<float>());
        else hipLaunchKernelGGL(k_real_guess, dim3(g.Q), dim3(256), 0, c->stream, c->samp.as<float>(), M, mstride, rank_s, c->thr.as<float>());
        c->t_end();
        HG_TRY(c->check_launch("k_real_guess"));
        if (second) {
            const double fr2 = (double)R * (double)M2 / (double)c->N;
            const u32 need2 = (u32)std::ceil(fr2 + sigma * std::sqrt(fr2)) + 1u;
            HG_TRY(real_sample_count(c, M2, stride2, need2));
            c->real_expect = (double)R * (1.0 + sigma / std::sqrt(fr2 > 1.0 ? fr2 : 1.0));
        }
        const double mean = budget * (double)R / (double)g.S;
        u32 cap = (u32)std::ceil(mean + 6.0 * std::sqrt(mean) + 16.0);
        cap = (cap + 15u) & ~15u;                         // a multiple of the compact records' ring (16) and flush piece (8)
        const u32 whole = (u32)((g.L + 15) & ~15ll);      // (a slice never needs more than its segment's rows)
        c->cap = cap < whole ? cap : whole;
    } else {
        // no bet: every row becomes a record (thr = -inf), slices are whole segments
        HG_HIP(hipMemsetD32Async((hipDeviceptr_t)c->thr.p, (int)0xFF800000u, (size_t)g.Q, c->stream));     // (a fill on the stream: no host vector, no wait)
        c->cap = (u32)g.L;
    }
    c->crow = (i64)g.S * c->cap;
    const size_t rows = (size_t)g.Q * c->crow * 8;
    // the record rows; the global-memory ranking passes (a query whose records exceed the LDS, the exhaustive mode) need two
    // more buffers of that size -- a widened bet (run_real) only goes as far as the rows alone stay moderate
    if (bet && rows > (size_t)64 << 30) { *lost = 1; return HG_OK; }
    if (!bet && rows * 3 > (size_t)200 << 30)
        return fail(HG_ERR_NOMEM, "real-valued ranking: %zu GB of records needed (Q=%d, %lld per query)", rows * 3 >> 30, g.Q, (long long)c->crow);
    HG_TRY(c->cand.reserve(rows));
    HG_TRY(real_select(c));
    const size_t slots = (size_t)g.Q * g.R;
    HG_TRY(c->mbits.reserve((size_t)g.Q * c->RW * 8));
    // hg_map_real wants match bits and APs: the kernels that rank in LDS skip the idx / score lists then (Q x R x 8 bytes of stores: 0.4 GB at
    // 10k x R = 5000 and at the CIFAR evaluation alike); the global-memory passes gather the labels THROUGH the idx list and always write it
    // -- and only they reserve the lists then (and the second sort buffer: 0.43 GB each at the CIFAR evaluation, where a recycled context's
    // first call at the shape meets hipMalloc for whatever the block cache cannot serve)
    const bool skip_lists = with_ap && !c->opt_real_map_lists && !c->is_sub;
    if (!skip_lists) { HG_TRY(c->out_idx.reserve(slots * 4)); HG_TRY(c->scores.reserve(slots * 4)); }
    c->real_lists_made = true;
    if (c->real_filtered && bet && c->opt_real_sort_lds && g.S <= RK_SMAX && R <= RK_RMAX) {
        // a query's records fit the LDS of one workgroup: copy + select + counting passes + ranked list in one kernel
        constexpr int NA = 14336;
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_rank_lds<NA>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)real_rank_lds_bytes<NA>()));
        c->t_begin(KI_RADIX);
        hipLaunchKernelGGL(k_real_rank_lds<NA>, dim3(g.Q), dim3(1024), real_rank_lds_bytes<NA>(), c->stream, c->cand.as<u64>(), c->crow, c->cap,
                           c->cntq.as<u32>(), c->failq.as<u32>(), c->thr.as<float>(), skip_lists ? nullptr : c->out_idx.as<u32>(), skip_lists ? nullptr : c->scores.as<float>(),
                           c->dblab.as<u64>(), c->qlab.as<u64>(), c->mbits.as<u64>(), c->RW, c->err.as<int>(), c->qbad.as<u32>(), g);
        c->t_end();
        HG_TRY(c->check_launch("k_real_rank_lds"));
        int flag = 0;
        if (with_ap) {
            // the usual case holds its bet: AP and the download of {verdict, AP, hit counts} ride behind the rank kernel and the call
            // synchronises ONCE (round 5: verdict, wait, AP, wait, two copies, wait); a lost bet's APs are simply not used
            c->stage = ST_DB | ST_Q | ST_SELECT | ST_MATCH;
            HG_TRY(do_ap(c));
            HG_TRY(stage_ap_download(c));
            HG_TRY(c->sync());
            flag = *(const int*)c->pin;
            c->stage = ST_DB | ST_Q | ST_SELECT;
        } else {
            HG_TRY(read_plan_flag(c, &flag));
        }
        if (!(flag & 2)) {
            c->real_lds_ranked = 1;
            c->real_lists_made = !skip_lists;
            *lost = flag & 1;
            c->stage = ST_DB | ST_Q | ST_SELECT;
            if (*lost) return HG_OK;
            c->stage |= ST_MATCH;                          // the rank kernel left the match bits too
            if (with_ap) { c->stage |= ST_AP; c->ap_staged = true; }
            return HG_OK;
        }
        c->stage = ST_DB | ST_Q | ST_SELECT;
        HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));    // some query's records exceed the LDS: the global-memory passes rank them all
    }
    if (rows * 3 > (size_t)200 << 30) {
        if (bet) { *lost = 1; return HG_OK; }
        return fail(HG_ERR_NOMEM, "real-valued ranking: %zu GB of records needed (Q=%d, %lld per query)", rows * 3 >> 30, g.Q, (long long)c->crow);
    }
    HG_TRY(c->sortA.reserve(rows));
    const int nwav = c->crow >= 16384 ? 16 : 4;
    const size_t lds = (size_t)(nwav + 1) * 256 * 4;
    const u64* in = c->cand.as<u64>();
    bool grouped = false;
    if (c->opt_real_groups && g.S <= 8192 && !bet && c->crow <= (i64)RG_MAXG * RG_CAP) {
        // every row a record (R = N on a CIFAR-sized database): split by score range into LDS-sized groups, order each group
        // in LDS (k_real_group_split / k_real_group_sort) -- two trips of the records through memory instead of the radix
        // passes' four, 3.1 -> 0.85 ms at C1; piled-up scores come back as bit 2 of the flag.  (A bet's list beyond the LDS --
        // 19 000 records in 489 short slices at R = 10 000 -- stays with the radix passes: 3.9 ms against 5.7 this way.)
        HG_TRY(c->gtab.reserve((size_t)g.Q * (RG_MAXG + 1) * 4));
        // (a group spans at least RG_CAP / 2 of cumulative count -- the largest bucket is at most RG_CAP / 2: at most 2 n / RG_CAP + 1 groups)
        const int maxg = (int)std::min<i64>(RG_MAXG, 2 * c->crow / RG_CAP + 2);
        // (per launch like everywhere else: the attribute is per DEVICE, and a process may hold contexts on several)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_real_group_sort), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)real_group_sort_lds()));
        c->t_begin(KI_RADIX);
        hipLaunchKernelGGL(k_real_group_split, dim3(g.Q), dim3(1024), real_group_split_lds(g.S), c->stream, c->cand.as<u64>(), c->crow, c->cap,
                           c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->tot.as<u32>(), c->sortA.as<u64>(), c->gtab.as<u32>(), c->crow, c->err.as<int>(), maxg, g);
        c->t_end();
        HG_TRY(c->check_launch("k_real_group_split"));
        // the sort writes the ranked lists and the match bits itself (k_real_finish and k_match are for the radix passes)
        HG_HIP(hipMemsetAsync(c->mbits.p, 0, (size_t)g.Q * c->RW * 8, c->stream));
        HG_HIP(hipMemsetAsync(c->qbad.p, 0, (size_t)g.Qpad * 4, c->stream));
        const GroupOut go{skip_lists ? nullptr : c->out_idx.as<u32>(), skip_lists ? nullptr : c->scores.as<float>(), c->mbits.as<u32>(), c->dblab.as<u64>(), c->qlab.as<u64>(), c->RW, g.R, g.LW, g.idx_base};
        c->t_begin(KI_RADIX);
        hipLaunchKernelGGL(k_real_group_sort, dim3(g.Q, maxg), dim3(1024), real_group_sort_lds(), c->stream, c->sortA.as<u64>(), c->gtab.as<u32>(),
                           go, c->crow, c->err.as<int>());
        c->t_end();
        HG_TRY(c->check_launch("k_real_group_sort"));
        int flag = 0;
        if (with_ap) {
            // the usual case has no pile: AP and the download of {flag, AP, hit counts} ride behind the sort and the call synchronises
            // ONCE (as above; a piled-up query's APs are simply not used -- the bitmaps they are read from are zeroed, defined memory)
            c->stage = ST_DB | ST_Q | ST_SELECT | ST_MATCH;
            HG_TRY(do_ap(c));
            HG_TRY(stage_ap_download(c));
            HG_TRY(c->sync());
            flag = *(const int*)c->pin;
            c->stage = ST_DB | ST_Q | ST_SELECT;
        } else {
            HG_TRY(read_plan_flag(c, &flag));
        }
        if (flag & 4) {
            HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
        } else {
            c->real_grouped = 1;
            c->real_lists_made = !skip_lists;
            c->stage = ST_DB | ST_Q | ST_SELECT | ST_MATCH;
            *lost = flag;
            if (with_ap) { c->stage |= ST_AP; c->ap_staged = flag == 0; }
            return HG_OK;
        }
    }
    c->real_grouped = 0;
    // the global-memory passes: two sort buffers, and the ranked lists whoever asked (k_match gathers the labels through them)
    HG_TRY(c->sortB.reserve(rows));
    HG_TRY(c->out_idx.reserve(slots * 4)); HG_TRY(c->scores.reserve(slots * 4));
    u64* bufs[2] = {c->sortA.as<u64>(), c->sortB.as<u64>()};
    for (int pass = 0; pass < 4 && !grouped; ++pass) {
        RadixArgs ra{c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->tot.as<u32>(), c->cap, c->crow, c->crow, pass == 0, 32 + 8 * pass};
        u64* out = bufs[pass & 1];
        c->t_begin(KI_RADIX);
        if (nwav == 16) hipLaunchKernelGGL(k_radix_pass<16>, dim3(g.Q), dim3(1024), lds, c->stream, in, out, ra, g);
        else hipLaunchKernelGGL(k_radix_pass<4>, dim3(g.Q), dim3(256), lds, c->stream, in, out, ra, g);
        c->t_end();
        HG_TRY(c->check_launch("k_radix_pass"));
        in = out;
    }
    c->t_begin(KI_REAL_FINISH);
    const i64 nKB = grid_for(g.R);
    if (nKB * g.Q > 0x7FFFFFFFll) return fail(HG_ERR_ARG, "real-valued ranking: Q*R too large for one launch");
    hipLaunchKernelGGL(k_real_finish, dim3((unsigned)(nKB * g.Q)), dim3(256), 0, c->stream, in, c->crow, c->tot.as<u32>(),
                       c->out_idx.as<u32>(), c->scores.as<float>(), c->err.as<int>(), c->qbad.as<u32>(), (int)nKB,
                       c->real_filtered ? c->thr.as<float>() : nullptr, g);
    c->t_end();
    HG_TRY(c->check_launch("k_real_finish"));
    c->stage = ST_DB | ST_Q | ST_SELECT;
    HG_TRY(do_match(c));                               // label gather through the ranked idx list
    if (with_ap) HG_TRY(do_ap(c));
    HG_TRY(read_plan_flag(c, lost));
    return HG_OK;
}

static int run_real(hg_ctx* c, int64_t R, bool with_ap);

// A few queries lost the first bet (their cut kept fewer than R rows, or their rows crowd into one slice): those alone run again,
// on a child context that borrows the database's tables, with the usual escalation; their lists and match bits go back into the
// parent's rows and the parent evaluates all queries.  The whole call is redone only when many queries lost (run_real).  Because a
// lost query now costs a fraction of a millisecond instead of the call, the first cut can sit shallower (REAL_FIRST_SIGMA).
static int real_requery_lost(hg_ctx* c, int64_t R, bool with_ap, bool* handled) {
    *handled = false;
    if (c->is_sub || !c->real_lds_ranked || !c->real_filtered) return HG_OK;
    const Geo g = c->geo;
    std::vector<u32> bad((size_t)g.Q);
    HG_HIP(hipMemcpyAsync(bad.data(), c->qbad.p, (size_t)g.Q * 4, hipMemcpyDeviceToHost, c->stream));
    HG_TRY(c->sync());
    std::vector<u32> lost;
    for (int q = 0; q < g.Q; ++q) if (bad[(size_t)q]) lost.push_back((u32)q);
    const i64 nF = (i64)lost.size();
    if (nF == 0 || nF * 16 > g.Q) return HG_OK;
    if (!c->sub) {
        c->sub = new hg_ctx();
        c->sub->is_sub = true;
        c->sub->device = c->device;
        c->sub->stream = c->stream;                  // same stream: ordered with the parent's work
    }
    hg_ctx* s = c->sub;
    s->N = c->N; s->b = c->b; s->C = c->C; s->n_total = c->n_total; s->NW = c->NW; s->NB = c->NB; s->LW = c->LW;
    s->idx_base = c->idx_base; s->n_cu = c->n_cu;
    s->target_units = c->target_units; s->min_segment = c->min_segment; s->opt_max_segments = c->opt_max_segments;
    s->timing = 0;
    s->bpad = c->bpad;
    s->opt_real_mfma = c->opt_real_mfma; s->opt_real_sort_lds = c->opt_real_sort_lds; s->opt_real_groups = c->opt_real_groups;
    s->real_cap_boost = c->real_cap_boost;
    s->db.borrow(c->db);
    s->dblab.borrow(c->dblab);
    s->dbf.borrow(c->dbf); s->dbf_resident = true;
    s->dbfb.borrow(c->dbfb); s->xmax2.borrow(c->xmax2); s->dbfb_valid = c->dbfb_valid; s->dbfb_half = c->dbfb_half;
    s->Q = nF;
    HG_TRY(c->flist.reserve((size_t)nF * 4));
    HG_HIP(hipMemcpyAsync(c->flist.p, lost.data(), (size_t)nF * 4, hipMemcpyHostToDevice, c->stream));
    HG_TRY(s->qf.reserve((size_t)nF * c->bpad * 4 + 256));
    HG_TRY(s->qlab.reserve((size_t)nF * c->LW * 8));
    auto move = [&](const void* src, void* dst, i64 rowbytes, int gather) {
        hipLaunchKernelGGL(k_move_rows, dim3((unsigned)nF), dim3(256), 0, c->stream, (const u8*)src, (u8*)dst,
                           c->flist.as<u32>(), rowbytes, gather);
    };
    move(c->qf.p, s->qf.p, (i64)c->bpad * 4, 1);
    move(c->qlab.p, s->qlab.p, (i64)c->LW * 8, 1);
    HG_TRY(c->check_launch("k_move_rows"));
    s->qf_resident = true;
    s->stage = ST_DB | ST_Q;
    HG_TRY(run_real(s, R, false));
    if (s->RW != c->RW) return fail(HG_ERR_HIP, "real-valued ranking: internal error, the requeried lists have another width");
    move(s->mbits.p, c->mbits.p, c->RW * 8, 0);
    if (c->real_lists_made) {
        move(s->out_idx.p, c->out_idx.p, R * 4, 0);
        move(s->scores.p, c->scores.p, R * 4, 0);
    }
    HG_TRY(c->check_launch("k_move_rows"));
    HG_HIP(hipMemsetAsync(c->err.p, 0, 4, c->stream));
    c->stage = ST_DB | ST_Q | ST_SELECT | ST_MATCH;
    if (with_ap) HG_TRY(do_ap(c));
    HG_TRY(c->sync());                               // `lost` (the H2D source) must outlive the copy
    c->real_requeried += nF;
    *handled = true;
    return HG_OK;
}

static int run_real(hg_ctx* c, int64_t R, bool with_ap) {
    if (!c->bpad || !c->dbf.p || !c->qf.p || !c->dbf_resident || !c->qf_resident)
        return fail(HG_ERR_STATE, "real-valued ranking needs the float features on the GPU: load them with hg_set_database_f32 / "
                                  "hg_set_queries_f32 (option keep_floats = 1 if the database is a +-1 code)");
    if (c->n_total != c->N) return fail(HG_ERR_STATE, "real-valued ranking is single-shard");
    if (R < 1 || R > c->N) return fail(HG_ERR_ARG, "R=%lld outside 1..N (N=%lld rows in the database)", (long long)R, (long long)c->N);
    if (c->N > 0x7FFFFFFFll) return fail(HG_ERR_ARG, "real-valued ranking takes up to 2^31 - 1 rows (have %lld)", (long long)c->N);   // (bit 31 of a record's index half is its match bit)
    int lost = 0;
    c->real_attempts = 0;
    if (with_ap && !c->is_sub) HG_TRY(ensure_out_block(c));      // verdict, APs and hit counts side by side: one download
    if (R * 8 <= c->N && c->N >= 65536) {              // bet on a sampled cut; retry once deeper, then give up betting
        const double boost0 = (double)c->real_cap_boost;
        HG_TRY(real_attempt(c, R, true, c->is_sub ? 6.0 : REAL_FIRST_SIGMA, 3.0 * boost0, with_ap, &lost));
        if (!lost) { c->real_lists = c->real_lists_made; return HG_OK; }
        bool handled = false;
        HG_TRY(real_requery_lost(c, R, with_ap, &handled));
        if (handled) { c->real_lists = c->real_lists_made; return HG_OK; }
        // a deeper cut with twice the budget; then -- features that follow the labels in a database stored class by class
        // put a query's top rows into a tenth of its slices -- eight and sixty-four times the slices' capacity, kept for
        // the next calls on this database (the exhaustive mode below writes EVERY pair down: 80 GB at 10k x 1M)
        for (int attempt = 0; attempt < 3; ++attempt) {
            if (attempt > 0) {
                if (c->cap >= (u32)((c->geo.L + 15) & ~15ll)) break;      // a slice already holds its segment
                c->real_cap_boost = c->real_cap_boost * 8 > 4096 ? 4096 : c->real_cap_boost * 8;
            }
            HG_TRY(real_attempt(c, R, true, 16.0, 6.0 * (double)c->real_cap_boost, with_ap, &lost));
            if (!lost) {
                if (attempt > 0 && c->real_cap_boost < 4096) c->real_cap_boost *= 2;     // (the budget that held: 6 = 2 x 3; a held plain retry changes nothing)
                c->real_lists = c->real_lists_made;
                return HG_OK;
            }
        }
        c->real_cap_boost = (i64)boost0;
    }
    HG_TRY(real_attempt(c, R, false, 0.0, 0.0, with_ap, &lost));
    if (lost) return fail(HG_ERR_HIP, "real-valued ranking: internal error, exhaustive pass came up short");
    c->real_lists = c->real_lists_made;
    return HG_OK;
}

int hg_topr_real(hg_ctx* c, int64_t R) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_topr_real", "hg_set_database_f32 + hg_set_queries_f32"));
    return run_real(c, R, false);
}

int hg_map_real(hg_ctx* c, int64_t R, double* host_ap, int64_t* host_rel) {
    HG_TRY(need(c, ST_DB | ST_Q, "hg_map_real", "hg_set_database_f32 + hg_set_queries_f32"));
    HG_TRY(run_real(c, R, true));
    return hg_get_ap(c, host_ap, host_rel);
}

}  // extern "C"

// hg_preload: the runtime loads a translation unit's code object when one of its kernels is first needed (milliseconds);
// asking for a kernel's attributes does that now
int preload_real() {
    hipFuncAttributes a;
    HG_HIP(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_real_thr2)));
    return HG_OK;
}
