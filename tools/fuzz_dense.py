#!/usr/bin/env python3
"""Stress run on the GPU for round 4's rank kernels: the byte matrix (k_dense_bytes + k_rank_dense: N/8 < R <= N, bitmap in LDS
or global memory, queries in chunks), the slice-by-slice ranking of a bet's long lists (k_rank_dense<slices>, also as the
in-stream leftover kernel) and interleaved record rows -- each against the older sequences of the same library (k_hist +
k_select + k_rank_fused; k_rank_cnt), which the parity tests pin to the oracle at sizes it can follow."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

NEW = {"rank_dense": 1, "rank_slices": 7000, "inline_leftovers": 1, "rank_dense_gbm": -1, "dense_budget_mb": 16384, "optimistic": 1}

def one(seed):
    rng = np.random.default_rng(seed)
    b = int(rng.choice([16, 32, 48, 64, 64, 96, 100, 126]))
    N = int(rng.integers(60000, 300000))
    Q = int(rng.integers(1500, 6000))
    frac = float(rng.choice([0.03, 0.06, 0.1, 0.124, 0.126, 0.2, 0.26, 0.5, 0.9, 1.0]))
    R = max(1, min(N, int(N * frac)))
    C = int(rng.choice([3, 10, 81, 128]))
    dl, _ = synth.onehot_labels(seed * 3 + 1, N, C)
    ql, _ = synth.onehot_labels(seed * 3 + 2, Q, C)
    if rng.random() < 0.6:
        db = synth.planted_codes(seed, dl, b, 0.25); qb = synth.planted_codes(seed, ql, b, 0.25)
    else:
        db = (rng.random((N, b)) < 0.5).astype(np.uint8); qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    if rng.random() < 0.25:                      # duplicated neighbours: long runs of ties
        db = np.repeat(db[: N // 8 + 1], 8, axis=0)[:N]; dl = np.repeat(dl[: N // 8 + 1], 8, axis=0)[:N]
    if rng.random() < 0.3:                       # stored class by class
        order = np.argsort(dl.argmax(1), kind="stable")
        db, dl = np.ascontiguousarray(db[order]), np.ascontiguousarray(dl[order])
    ctx = _native.Context(0)
    try:
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        out, how = {}, {}
        variants = [("new", {}), ("new_again", {}), ("gbm", {"rank_dense_gbm": 1}), ("lds_chunks", {"rank_dense_gbm": 0, "dense_budget_mb": 64}),
                    ("no_inline", {"inline_leftovers": 0}),
                    ("old", {"rank_dense": 0, "rank_slices": 0, "inline_leftovers": 0}), ("old_exact", {"rank_dense": 0, "rank_slices": 0, "optimistic": 0})]
        for name, opts in variants:
            for k, v in NEW.items(): ctx.set_option(k, v)
            for k, v in opts.items(): ctx.set_option(k, v)
            out[name] = ctx.map(R)
            how[name] = (ctx.get_stat("rank_variant"), ctx.get_stat("last_optimistic"))
        ref = out["old_exact"]
        for name, (ap, rel) in out.items():
            if not (np.array_equal(ap, ref[0], equal_nan=True) and np.array_equal(rel, ref[1])):
                return "MISMATCH %s seed=%d b=%d N=%d Q=%d R=%d C=%d %s" % (name, seed, b, N, Q, R, C, how)
        return "ok seed=%d b=%d N=%d Q=%d R=%d C=%d rank %s" % (seed, b, N, Q, R, C, how["new"])
    finally:
        ctx.close()

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad, t = 0, time.time()
    for seed in range(s0, s0 + n):
        r = one(seed)
        if r.startswith("MISMATCH"): bad += 1
        print(r, flush=True)
    print("done: %d shapes, %d mismatches, %.0f s" % (n, bad, time.time() - t))
