#!/usr/bin/env python3
"""Debug: the far-queries case through k_select_mx3 vs k_select_mx, bet and exact sequences."""
import sys, warnings
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, metric
import oracle.hamming_map as O

rng = np.random.default_rng(77)
Q, N, b, R, C = 70, 300000, 64, 2000, 6
proto = (rng.random(b) < 0.5).astype(np.uint8)
db = proto ^ (rng.random((N, b)) < 0.1).astype(np.uint8)
qb = (1 - proto) ^ (rng.random((Q, b)) < 0.1).astype(np.uint8)
dl = (rng.random((N, C)) < 0.3).astype(np.int8)
ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
for name, opts in (("bet_mx3", {"select_packed": 3}), ("bet_mx", {"select_packed": 1}), ("exact_mx3", {"select_packed": 3, "optimistic": 0}),
                   ("exact_mx", {"select_packed": 1, "optimistic": 0}), ("bet_mx3_nosecond", {"select_packed": 3, "second_bet": 0})):
    ctx = _native.Context(0)
    try:
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        ap, rel = ctx.map(R)
        bad = np.nonzero(~((ap == ap_ref) | (np.isnan(ap) & np.isnan(ap_ref))))[0]
        print(name, "mismatching queries:", len(bad), bad[:20], "runs", ctx.get_stat("optimistic_runs"), "fallbacks", ctx.get_stat("optimistic_fallbacks"),
              "requeried", ctx.get_stat("optimistic_requeried"), flush=True)
    finally:
        ctx.close()
