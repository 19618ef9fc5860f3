#!/usr/bin/env python3
"""One-off stress run on the GPU: precision/recall@k and precision within a Hamming radius against brute-force NumPy."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import extra_metrics as X
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for seed in range(n):
    rng = np.random.default_rng(seed)
    Q, N = int(rng.integers(1, 120)), int(rng.integers(50, 20000))
    b, C = int(rng.choice([8, 16, 31, 64, 100])), int(rng.choice([2, 6, 70, 130]))
    db = rng.integers(0, 2, (N, b), dtype=np.uint8)
    qb = db[rng.integers(0, N, Q)] ^ (rng.random((Q, b)) < 0.1).astype(np.uint8)
    dl = (rng.random((N, C)) < 0.2).astype(np.int8); ql = (rng.random((Q, C)) < 0.2).astype(np.int8)
    D = np.stack([(db != qb[i]).sum(1) for i in range(Q)])        # brute force
    rel = (ql.astype(np.int64) @ dl.astype(np.int64).T) > 0
    ks = sorted(set(int(k) for k in rng.integers(1, N + 1, 4)))
    relo = np.take_along_axis(rel, np.argsort(D, axis=1, kind="stable"), 1)
    p_ref = np.array([relo[:, :k].sum(1) / k for k in ks]).T.mean(0)
    tot = rel.sum(1); ok = tot > 0
    r_ref = np.array([relo[ok][:, :k].sum(1) / tot[ok] for k in ks]).T.mean(0) if ok.any() else np.full(len(ks), np.nan)
    p, r = X.precision_recall_at_k(qb, db, ql, dl, ks)
    good = np.allclose(p, p_ref, rtol=0, atol=1e-14) and np.allclose(r, r_ref, rtol=0, atol=1e-14, equal_nan=True)
    radius = int(rng.integers(0, max(1, b // 4)))
    inside = D <= radius; ball = inside.sum(1)
    ref = np.where(ball > 0, (inside & rel).sum(1) / np.maximum(ball, 1), 0.0).mean()
    got, balls = X.precision_within_radius(qb, db, ql, dl, radius)
    good = good and np.array_equal(balls, ball) and abs(got - ref) < 1e-14
    if not good:
        bad += 1; print("MISMATCH seed=%d Q=%d N=%d b=%d C=%d ks=%s radius=%d" % (seed, Q, N, b, C, ks, radius), p, p_ref, r, r_ref, got, ref, flush=True)
print("done: %d shapes, %d mismatches" % (n, bad))
