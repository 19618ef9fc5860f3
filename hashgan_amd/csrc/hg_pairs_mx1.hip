// libhashgan_amd.so -- launcher of k_select_mx, the matrix-core record pass for every code length (one distance per
// accumulator; codes of <= 64 bits usually take k_select_mx3 / k_select_mx2 in hg_pairs_mx.hip instead).  A unit of its own
// because its 48 instantiations (code words x label words x record format) are the longest compile of the library.
#include "hg_ctx.hpp"
#include "hg_mx_drain.hpp"
#include "hg_select_mx.hpp"

namespace {
template <int NW, int LW, int QT, bool COMPACT> int launch_select_mx_q(hg_ctx* c);
template <int NW, int LW> int launch_select_mx_t(hg_ctx* c) {
    // two query tiles per wavefront; codes of up to 128 bits run 4 wavefronts per SIMD, longer codes need the registers of
    // the 2-waves-per-SIMD variant (B fragments: 4 per query tile and 64 bits) -- window lengths: mx_wt()
    constexpr int QT = mx_qt(NW);
    return c->rec8 ? launch_select_mx_q<NW, LW, QT, true>(c) : launch_select_mx_q<NW, LW, QT, false>(c);
}
template <int NW, int LW, int QT, bool COMPACT> int launch_select_mx_q(hg_ctx* c) {
    constexpr int QBLK = WPB * 32 * QT;                // queries per block
    HG_TRY(ensure_mx_images(c, true));
    Geo g = c->geo;
    const int nSP = (g.S + 1) / 2;
    const int nQB = (g.Q + QBLK - 1) / QBLK;           // query blocks
    g.nQT = nQB;
    g.nUnits = (i64)nSP * nQB;
    g.wpb = WPB;
    g.nBlk = (int)g.nUnits;
    const MxLds L = mx_lds_layout(NW, LW, QT, COMPACT);
    if (L.total > 64 * 1024)
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_mx<NW, LW, QT, COMPACT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    SelArgs a{c->exact_mx ? c->t.as<int>() : c->tguess.as<int>(), c->sl_start.as<u32>(), c->sl_tie.as<u32>(), c->sl_cnt.as<u32>(),
              c->failq.as<u32>(), c->cap, c->crow, 1, c->sstar.as<int>(), (int)c->opt_probe};
    c->t_begin(KI_SELECT_MX);
    hipLaunchKernelGGL((k_select_mx<NW, LW, QT, COMPACT>), dim3(padded_grid(g.nBlk)), dim3(256), (size_t)L.total, c->stream, c->qc.as<u32>(),
                       c->qlab.as<u64>(), c->qx.as<u8>(), c->db.as<u32>(), c->dbx.as<u8>(), c->dblab.as<u64>(), a,
                       c->cand.as<u64>(), g);
    c->t_end();
    return c->check_launch("k_select_mx");
}

}  // namespace

namespace {
template <int NW> int select_mx_nw(hg_ctx* c, int lw) {
    switch (lw) {
        case 1: return launch_select_mx_t<NW, 1>(c);
        case 2: return launch_select_mx_t<NW, 2>(c);
        default: return launch_select_mx_t<NW, 0>(c);
    }
}
}  // namespace

int launch_select_mx(hg_ctx* c, int lw) { HG_DISPATCH_NW(select_mx_nw, c, lw) }

// hg_preload: the runtime loads a translation unit's code object when one of its kernels is first needed (milliseconds);
// asking for a kernel's attributes does that now
int preload_mx1() {
    hipFuncAttributes a;
    HG_HIP(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_select_mx<2, 1, 2, true>)));
    return HG_OK;
}
