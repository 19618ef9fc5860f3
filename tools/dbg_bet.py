import sys, warnings
sys.path.insert(0, ".")
import numpy as np
from hashgan_amd import _native, metric
from oracle import hamming_map as O
rng = np.random.default_rng(7)
ctx = _native.Context(0)
for b, C, R in [(24, 5, 2000), (64, 70, 3333), (40, 130, 1000), (100, 10, 5000), (64, 10, 1), (32, 3, 7), (48, 10, 100)]:
    Q, N = 97, 70000 + int(rng.integers(0, 999))
    qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
    db = rng.integers(0, 2, (N, b), dtype=np.uint8)
    dl = (rng.random((N, C)) < 0.2).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.2).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, _, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
    ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
    ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
    for mf in (1, 0):
        ctx.set_option("select_mfma", mf)
        ap, rel = ctx.map(R)
        ok_ap = np.array_equal(ap, ap_ref, equal_nan=True)
        ctx.topr(R)
        idx, dist = ctx.get_topr()
        ok_i = np.array_equal(idx, idx_ref); ok_d = np.array_equal(dist, dist_ref)
        print("b=%d C=%d R=%d N=%d mfma=%d S=%d L=%d ap=%s idx=%s dist=%s bet=%d fb=%d" % (b, C, R, N, mf, ctx.get_stat("segments"), ctx.get_stat("segment_rows"), ok_ap, ok_i, ok_d, ctx.get_stat("last_optimistic"), ctx.get_stat("optimistic_fallbacks")), flush=True)
        if not ok_i:
            bad = np.where((idx != idx_ref).any(axis=1))[0]
            print("  bad queries", bad[:10], "of", len(bad))
            qq = bad[0]; pos = np.where(idx[qq] != idx_ref[qq])[0]
            print("  q", qq, "first bad pos", pos[:5], "got", idx[qq][pos[:5]], dist[qq][pos[:5]], "want", idx_ref[qq][pos[:5]], dist_ref[qq][pos[:5]])
        elif not ok_ap:
            bad = np.where(~((ap == ap_ref) | (np.isnan(ap) & np.isnan(ap_ref))))[0]
            print("  bad ap queries", bad[:10], ap[bad[:5]], ap_ref[bad[:5]])
