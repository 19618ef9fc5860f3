#!/usr/bin/env python3
"""One-off stress run on the GPU: the Python surface (MAPs / MAP, resident databases, several objects alive at once)
on random inputs against the staged exact sequence of a separate context."""
import sys, time, types
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, metric, MAPs, MAP

def exact_map(qbits, dbits, ql, dl, R):
    ctx = _native.Context(0)
    try:
        for k, v in (("optimistic", 0), ("hist_mfma", 0), ("select_mfma", 0)): ctx.set_option(k, v)
        ctx.set_database(metric.pack_codes(dbits), metric.pack_labels(dl), dbits.shape[1], dl.shape[1])
        ctx.set_queries(metric.pack_codes(qbits), metric.pack_labels(ql))
        ap, rel = ctx.map(R)
        return float(np.mean(ap[rel != 0])) if (rel != 0).any() else float("nan")
    finally:
        ctx.close()

def one(seed, keep):
    rng = np.random.default_rng(seed)
    b = int(rng.choice([8, 16, 32, 48, 64, 64, 100, 128]))
    N = int(rng.integers(2000, 200000)); Q = int(rng.integers(1, 400))
    R = max(1, int(N * float(rng.choice([0.002, 0.01, 0.05, 0.3, 1.0]))))
    C = int(rng.choice([2, 10, 70, 150]))
    dbits = (rng.random((N, b)) < 0.5).astype(np.uint8); qbits = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    dl = (rng.random((N, C)) < 0.2).astype(np.int64); ql = (rng.random((Q, C)) < 0.2).astype(np.int64)
    want = exact_map(qbits, dbits, ql, dl, R)
    db = types.SimpleNamespace(output=(2.0 * dbits - 1.0).astype(np.float32), label=dl)
    qu = types.SimpleNamespace(output=(2.0 * qbits - 1.0).astype(np.float32), label=ql)
    m = MAPs(R)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = [m.get_maps_by_feature(db, qu)]
        m.set_database(db)
        got.append(m.get_maps_by_feature(None, qu))
        got.append(m.get_maps_by_feature(db, qu))
        got.append(MAP(qbits, dbits, ql, dl, R))
        got.append(MAP(qu.output, db.output, ql, dl, R))
    keep.append(m)                       # several objects alive: each owns its context
    if len(keep) > 3: keep.pop(0).close()
    for g in got:
        if not (g == want or (np.isnan(g) and np.isnan(want))):
            return "MISMATCH seed=%d b=%d N=%d Q=%d R=%d C=%d got=%s want=%r" % (seed, b, N, Q, R, C, got, want)
    return "ok seed=%d b=%d N=%d Q=%d R=%d C=%d" % (seed, b, N, Q, R, C)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0; t = time.time(); keep = []
    for seed in range(s0, s0 + n):
        if "-v" in sys.argv: print("seed", seed, flush=True)
        r = one(seed, keep)
        if r.startswith("MISMATCH"): bad += 1; print(r, flush=True)
        elif seed % 10 == 0: print(r, flush=True)
    print("done: %d shapes, %d mismatches, %.0f s" % (n, bad, time.time() - t))
