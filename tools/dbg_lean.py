#!/usr/bin/env python3
"""k_rank_lean against k_rank_cnt on a few shapes: APs must be identical; prints the paths taken."""
import sys, warnings
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

def run(Q, N, b, R, C=10, dup=None, seed=90):
    dl, _ = synth.onehot_labels(seed + 1, N, C)
    ql, _ = synth.onehot_labels(seed + 2, Q, C)
    db = synth.random_bits(seed + 3, N, b)
    qb = synth.random_bits(seed + 4, Q, b)
    if dup: db[dup[0]:dup[1]] = qb[dup[2]]
    res = {}
    for lean in (0, 1):
        ctx = _native.Context(0)
        ctx.set_option("rank_lean", lean)
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        ap, rel = ctx.map(R)
        st = {k: ctx.get_stat(k) for k in ("rank_variant", "optimistic_fallbacks", "optimistic_requeried", "optimistic_rebets", "rank_leftovers", "segments", "slice_capacity", "last_optimistic")}
        res[lean] = (ap.copy(), rel.copy(), st)
        ctx.close()
    a0, r0, s0 = res[0]; a1, r1, s1 = res[1]
    bad = np.flatnonzero(~((a0 == a1) | (np.isnan(a0) & np.isnan(a1))))
    print("Q=%d N=%d b=%d R=%d: equal=%s ndiff=%d first=%s\n   cnt  %s\n   lean %s" % (Q, N, b, R, len(bad) == 0, len(bad), bad[:8], s0, s1), flush=True)
    if len(bad): print("   ap0", a0[bad[:4]], "ap1", a1[bad[:4]], "rel", r0[bad[:4]], r1[bad[:4]])

run(256, 131072, 32, 4000, dup=(50000, 53000, 5))
run(256, 131072, 32, 4000)
run(2100, 190000, 48, 5000, C=81)
run(1000, 200000, 64, 1000)
run(10000, 1000000, 64, 5000)
run(4000, 500000, 128, 3000)
run(513, 70000, 17, 100)
