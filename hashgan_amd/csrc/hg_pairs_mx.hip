// libhashgan_amd.so -- launchers of the matrix-core pair passes and of the images they read: k_select_mx3 / mx4
// (optimistic record pass), k_hist_mx / k_hist_i8 (histograms).
#include "hg_ctx.hpp"
#include "hg_mx_drain.hpp"
#include "hg_select_mx.hpp"
#include "hg_select_mx3.hpp"
#include "hg_select_mx4.hpp"
#include "hg_hist_mx.hpp"
#include "hg_hist_i8.hpp"

namespace {
}  // namespace

// the fp4 images of the codes in MFMA fragment order (k_select_mx, k_hist_mx; k_select_mx3 shares the query image), built on
// first use.  need_db = false: the query image only (k_select_mx3 has its own database image)
int ensure_mx_images(hg_ctx* c, const bool need_db) {
    const int NW = c->NW, NM = (NW + 1) / 2;
    if (need_db && !c->dbx_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbx.reserve((size_t)(n16 > 0 ? n16 : 16) * NM * 32));
        const i64 items = n16 * 2 * NM;
        c->t_begin(KI_PACK);
        if (items) hipLaunchKernelGGL(k_expand_db, dim3(grid_for(items)), dim3(256), 0, c->stream, c->db.as<u32>(),
                                      c->dbx.as<uint4>(), (i64)c->N, n16, NW, NM);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db"));
        c->dbx_valid = true;
    }
    if (!c->qx_valid) {
        const i64 qpad = ((i64)c->Q + 511) / 512 * 512;
        HG_TRY(c->qx.reserve((size_t)qpad * NM * 32));
        const i64 items = qpad * 2 * NM;
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_queries, dim3(grid_for(items)), dim3(256), 0, c->stream, c->qc.as<u32>(), c->qx.as<uint4>(),
                           (i64)c->Q, qpad, NW, NM);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_queries"));
        c->qx_valid = true;
    }
    return HG_OK;
}


namespace {
template <int NW> int launch_hist_mx_t(hg_ctx* c) {
    HG_TRY(ensure_mx_images(c, true));
    Geo g = c->geo;                                    // fine geometry; the kernel pairs the segments itself
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 255) / 256;
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    const i64 tiles_per_half = ((g.L + 15) / 16 + g.hist_stride - 1) / g.hist_stride;      // visited by one lane-half
    const i64 visited_per_pair = 2 * tiles_per_half * 16;
    const bool pack16 = visited_per_pair < 65536;
    const size_t lds = (size_t)WPB * (pack16 ? 1 : 2) * g.NB * 32 * 4;
    c->t_begin(KI_HIST);
    if (pack16) {
        hipLaunchKernelGGL((k_hist_mx<NW, true>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->qx.as<u8>(),
                           c->dbx.as<u8>(), c->hist.as<u32>(), g);
    } else {
        if (lds > 64 * 1024)
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist_mx<NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_hist_mx<NW, false>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->qx.as<u8>(),
                           c->dbx.as<u8>(), c->hist.as<u32>(), g);
    }
    c->t_end();
    return c->check_launch("k_hist_mx");
}

// codes of 33..64 bits, compact records: three rows per accumulator and the batched drain (k_select_mx3);
// blocks = (pair of segments) x (256 queries); the query image is k_select_mx's
template <int NW, int LW> int launch_select_mx3_t(hg_ctx* c) {
    HG_TRY(ensure_mx_images(c, false));
    if (!c->dbx3_valid) {
        const i64 n48 = (c->N + M3_ROWS - 1) / M3_ROWS * M3_ROWS + M3_WS_MAX * M3_ROWS;     // + one window of zero rows: the last segment's last window may run past the end
        HG_TRY(c->dbx3.reserve((size_t)(n48 > 0 ? n48 : M3_ROWS) * 32));
        const i64 items = n48 * 2;
        c->t_begin(KI_PACK);
        if (items) hipLaunchKernelGGL(k_expand_db3, dim3(grid_for(items)), dim3(256), 0, c->stream, c->db.as<u32>(),
                                      c->dbx3.as<uint4>(), (i64)c->N, n48, NW);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db3"));
        c->dbx3_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 64 * M3_WPB - 1) / (64 * M3_WPB);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = M3_WPB;
    g.nBlk = (int)g.nUnits;
    const Mx3Lds L = mx3_lds_layout(NW, LW);
    if (L.total > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_mx3<NW, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    SelArgs a{c->exact_mx ? c->t.as<int>() : c->tguess.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(),
              c->failq.as<u32>(), c->cap, c->crow, 1, c->sstar.as<int>(), (int)c->opt_probe};
    c->t_begin(KI_SELECT_MX);
    hipLaunchKernelGGL((k_select_mx3<NW, LW>), dim3(padded_grid(g.nBlk)), dim3(64 * M3_WPB), (size_t)L.total, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->qx.as<u8>(), c->db.as<u32>(), c->dbx3.as<u8>(), c->dblab.as<u64>(), a,
                       c->cand.as<u8>(), g);
    c->t_end();
    return c->check_launch("k_select_mx3");
}

// the same pass with the integer matrix instruction delivering the counter addresses (hg_hist_i8.hpp); codes of <= 128 bits
template <int NW> int launch_hist_i8_t(hg_ctx* c) {
    if (!c->dbx8_valid) {
        const i64 n16 = (c->N + 15) / 16 * 16;
        HG_TRY(c->dbx8.reserve((size_t)n16 * NW * 32));
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_db_i8, dim3(grid_for(n16 * NW * 2)), dim3(256), 0, c->stream, c->db.as<u32>(), c->dbx8.as<uint4>(),
                           (i64)c->N, n16, NW);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db_i8"));
        c->dbx8_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 255) / 256;
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    const i64 tiles_per_half = ((g.L + 15) / 16 + g.hist_stride - 1) / g.hist_stride;
    const bool pack16 = 2 * tiles_per_half * 16 < 65536;
    const size_t lds = (size_t)WPB * hist_i8_cols(pack16) * g.NB * 32 * 4;
    c->t_begin(KI_HIST);
    if (pack16) {
        if (lds > 64 * 1024)
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist_i8<NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_hist_i8<NW, true>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->dbx8.as<u8>(),
                           c->hist.as<u32>(), g);
    } else {
        if (lds > 64 * 1024)
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist_i8<NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_hist_i8<NW, false>), dim3(padded_grid(g.nBlk)), dim3(256), lds, c->stream, c->qc.as<u32>(), c->dbx8.as<u8>(),
                           c->hist.as<u32>(), g);
    }
    c->t_end();
    return c->check_launch("k_hist_i8");
}
}  // namespace

int launch_hist_mx(hg_ctx* c) {
    if (c->opt_hist_mfma == 2 && c->NW <= 4) {
        switch (c->NW) {
            case 1: return launch_hist_i8_t<1>(c);
            case 2: return launch_hist_i8_t<2>(c);
            case 3: return launch_hist_i8_t<3>(c);
            default: return launch_hist_i8_t<4>(c);
        }
    }
    HG_DISPATCH_NW(launch_hist_mx_t, c)
}


namespace {
// codes of 65..128 bits, compact records: two rows per accumulator and the batched drain (k_select_mx4);
// blocks = (pair of segments) x (512 queries); the query image is k_select_mx's
template <int NW, int LW> int launch_select_mx4_t(hg_ctx* c) {
    HG_TRY(ensure_mx_images(c, false));
    if (!c->dbx4_valid) {
        const i64 n32 = (c->N + M4_ROWS - 1) / M4_ROWS * M4_ROWS + M4_WS * M4_ROWS;     // + one window of zero rows: the last segment's last window may run past the end
        HG_TRY(c->dbx4.reserve((size_t)n32 * 64));
        const i64 items = n32 * 4;
        c->t_begin(KI_PACK);
        hipLaunchKernelGGL(k_expand_db4, dim3(grid_for(items)), dim3(256), 0, c->stream, c->db.as<u32>(), c->dbx4.as<uint4>(), (i64)c->N, n32, NW);
        c->t_end();
        HG_TRY(c->check_launch("k_expand_db4"));
        c->dbx4_valid = true;
    }
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + 64 * M4_WPB - 1) / (64 * M4_WPB);
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = M4_WPB;
    g.nBlk = (int)g.nUnits;
    const Mx4Lds L = mx4_lds_layout(NW, LW);
    if (L.total > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_mx4<NW, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    SelArgs a{c->exact_mx ? c->t.as<int>() : c->tguess.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(),
              c->failq.as<u32>(), c->cap, c->crow, 1, c->sstar.as<int>(), 0};
    c->t_begin(KI_SELECT_MX);
    hipLaunchKernelGGL((k_select_mx4<NW, LW>), dim3(padded_grid(g.nBlk)), dim3(64 * M4_WPB), (size_t)L.total, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->qx.as<u8>(), c->db.as<u32>(), c->dbx4.as<u8>(), c->dblab.as<u64>(), a,
                       c->cand.as<u8>(), g);
    c->t_end();
    return c->check_launch("k_select_mx4");
}
}  // namespace

int launch_select_mx4(hg_ctx* c, int lw) {           // codes of 65..128 bits, one-byte records, <= 128 classes
    if (c->NW < 3 || c->NW > 4 || lw < 1 || lw > 2) return fail(HG_ERR_ARG, "k_select_mx4 takes codes of 65..128 bits and 1..128 classes");
    if (c->NW == 3) return lw == 1 ? launch_select_mx4_t<3, 1>(c) : launch_select_mx4_t<3, 2>(c);
    return lw == 1 ? launch_select_mx4_t<4, 1>(c) : launch_select_mx4_t<4, 2>(c);
}

int launch_select_mx3(hg_ctx* c, int lw) {           // codes of <= 64 bits, one-byte records, <= 128 classes
    if (c->NW > 2 || lw < 1 || lw > 2) return fail(HG_ERR_ARG, "k_select_mx3 takes codes of <= 64 bits and 1..128 classes");
    if (c->NW == 1) return lw == 1 ? launch_select_mx3_t<1, 1>(c) : launch_select_mx3_t<1, 2>(c);
    return lw == 1 ? launch_select_mx3_t<2, 1>(c) : launch_select_mx3_t<2, 2>(c);
}

// hg_preload: the runtime loads a translation unit's code object when one of its kernels is first needed (milliseconds);
// asking for a kernel's attributes does that now
int preload_mx() {
    hipFuncAttributes a;
    HG_HIP(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_select_mx3<2, 1>)));
    return HG_OK;
}
