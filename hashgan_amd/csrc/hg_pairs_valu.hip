// libhashgan_amd.so -- launchers of the vector-ALU pair passes: k_hist (full / sampled histogram), k_select (exact and
// optimistic record pass), k_select_dense (R/N >= 1/4).  One instantiation per code length (and label width).
#include "hg_ctx.hpp"

namespace {
template <int NW> int launch_hist_t(hg_ctx* c) {
    Geo g = hist_geometry(c);
    // LDS: one u32 histogram column per lane: wpb * NB * 64 * 4 bytes (<= 160 KiB per workgroup)
    int wpb = WPB;
    while (wpb > 1 && (size_t)wpb * g.NB * 256 > 160u * 1024u) wpb >>= 1;
    g.wpb = wpb;
    g.nBlk = (int)((g.nUnits + wpb - 1) / wpb);
    const size_t lds = (size_t)wpb * g.NB * 256;
    if (lds > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    c->t_begin(KI_HIST);
    hipLaunchKernelGGL(k_hist<NW>, dim3(padded_grid(g.nBlk)), dim3(64 * wpb), lds, c->stream,
                       c->qc.as<u32>(), c->db.as<u32>(), c->hist.as<u32>(), g);
    c->t_end();
    return c->check_launch("k_hist");
}

template <int NW, int LW, bool OPT> int launch_select_t(hg_ctx* c) {
    const Geo& g = c->geo;
    SelArgs a{c->optimistic ? c->tguess.as<int>() : c->t.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(),
              c->sl_cnt.as<u32>(), c->failq.as<u32>(), c->cap, c->crow, c->optimistic ? 1 : 0, c->sstar.as<int>(), 0};
    c->t_begin(KI_SELECT);
    hipLaunchKernelGGL((k_select<NW, LW, OPT>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->db.as<u32>(), c->dblab.as<u64>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_select");
}

template <int NW, int LW> int launch_select_dense_t(hg_ctx* c) {
    const Geo& g = c->geo;
    SelArgs a{c->t.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(), c->failq.as<u32>(),
              c->cap, c->crow, 0, nullptr, 0};
    c->t_begin(KI_SELECT);
    hipLaunchKernelGGL((k_select_dense<NW, LW>), dim3(padded_grid(g.nBlk)), dim3(256), 0, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->db.as<u32>(), c->dblab.as<u64>(), a, c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_select_dense");
}


template <int NW> int select_valu_nw(hg_ctx* c, int lw, bool optimistic) {
    if (optimistic) {
        switch (lw) {
            case 1: return launch_select_t<NW, 1, true>(c);
            case 2: return launch_select_t<NW, 2, true>(c);
            default: return launch_select_t<NW, 0, true>(c);
        }
    }
    switch (lw) {
        case 1: return launch_select_t<NW, 1, false>(c);
        case 2: return launch_select_t<NW, 2, false>(c);
        default: return launch_select_t<NW, 0, false>(c);
    }
}
template <int NW> int select_dense_nw(hg_ctx* c, int lw) {
    switch (lw) {
        case 1: return launch_select_dense_t<NW, 1>(c);
        case 2: return launch_select_dense_t<NW, 2>(c);
        default: return launch_select_dense_t<NW, 0>(c);
    }
}
}  // namespace

int launch_hist(hg_ctx* c) { HG_DISPATCH_NW(launch_hist_t, c) }
int launch_select_valu(hg_ctx* c, int lw, bool optimistic) { HG_DISPATCH_NW(select_valu_nw, c, lw, optimistic) }
int launch_select_dense(hg_ctx* c, int lw) { HG_DISPATCH_NW(select_dense_nw, c, lw) }

// hg_preload: the runtime loads a translation unit's code object when one of its kernels is first needed (milliseconds);
// asking for a kernel's attributes does that now
int preload_valu() {
    hipFuncAttributes a;
    HG_HIP(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_hist<2>)));
    return HG_OK;
}
