#!/usr/bin/env python3
"""One-off stress run on the GPU: the one-shot bet (matrix-core select + drain, every record format) against the
exact two-pass sequence of the same library, over random shapes and hit densities.  Both are parity-tested against
the oracle at small sizes; this run looks for disagreements where the oracle is too slow to follow."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

def one(seed):
    rng = np.random.default_rng(seed)
    b = int(rng.choice([8, 16, 24, 32, 48, 64, 64, 64, 96, 100, 128, 160, 255]))
    N = int(rng.integers(20000, 400000))
    Q = int(rng.integers(1, 700))
    frac = float(rng.choice([0.0005, 0.002, 0.005, 0.02, 0.06, 0.12]))
    R = max(1, int(N * frac))
    C = int(rng.choice([3, 10, 40, 81, 130]))
    planted = rng.random() < 0.5
    dl, _ = synth.onehot_labels(seed * 3 + 1, N, C)
    ql, _ = synth.onehot_labels(seed * 3 + 2, Q, C)
    if planted:
        db = synth.planted_codes(seed, dl, b, 0.25); qb = synth.planted_codes(seed, ql, b, 0.25)
    else:
        db = (rng.random((N, b)) < 0.5).astype(np.uint8); qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    if rng.random() < 0.3:                       # bursts: duplicated neighbours
        db = np.repeat(db[: N // 8 + 1], 8, axis=0)[:N]; dl = np.repeat(dl[: N // 8 + 1], 8, axis=0)[:N]
    if rng.random() < 0.35:                      # stored class by class: a query's near rows crowd into few slices (cap_boost)
        order = np.argsort(dl.argmax(1), kind="stable")
        db, dl = np.ascontiguousarray(db[order]), np.ascontiguousarray(dl[order])
    ctx = _native.Context(0)
    try:
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        out = {}
        if "-v" in sys.argv: print("  b=%d N=%d Q=%d R=%d C=%d planted=%s" % (b, N, Q, R, C, planted), flush=True)
        for name, opts in (("bet", {}), ("bet_again", {}), ("bet8", {"compact_records": 0}), ("exact", {"optimistic": 0}), ("exact_valu", {"optimistic": 0, "hist_mfma": 0, "select_mfma": 0})):
            for k in ("compact_records", "optimistic", "hist_mfma", "select_mfma"):
                ctx.set_option(k, {"compact_records": 1, "optimistic": 1, "hist_mfma": 2, "select_mfma": 1}[k])
            for k, v in opts.items(): ctx.set_option(k, v)
            if "-v" in sys.argv: print("   ", name, flush=True)
            out[name] = ctx.map(R)
        ref = out["exact_valu"]
        for name, (ap, rel) in out.items():
            if not (np.array_equal(ap, ref[0], equal_nan=True) and np.array_equal(rel, ref[1])):
                return "MISMATCH %s seed=%d b=%d N=%d Q=%d R=%d C=%d planted=%s" % (name, seed, b, N, Q, R, C, planted)
        return "ok seed=%d b=%d N=%d Q=%d R=%d C=%d" % (seed, b, N, Q, R, C)
    finally:
        ctx.close()

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t = time.time()
    for seed in range(s0, s0 + n):
        if "-v" in sys.argv: print("seed", seed, flush=True)
        r = one(seed)
        if r.startswith("MISMATCH"): bad += 1; print(r, flush=True)
        elif seed % 10 == 0: print(r, flush=True)
    print("done: %d shapes, %d mismatches, %.0f s" % (n, bad, time.time() - t))
