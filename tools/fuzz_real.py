#!/usr/bin/env python3
"""One-off stress run on the GPU: the real-valued ranking's pair passes against each other over random shapes --
bfloat16 filter + exact rescoring (LDS rank kernel and global-memory passes), float32 matrix-core pass, vector ALU."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native

REQUERIED = 0


def one(seed):
    rng = np.random.default_rng(seed)
    b = int(rng.choice([1, 7, 16, 20, 33, 48, 64, 64, 100, 128, 129, 200, 255]))
    N = int(rng.integers(3000, 200000))
    Q = int(rng.integers(1, 300))
    frac = float(rng.choice([0.002, 0.01, 0.03, 0.1, 0.5, 1.0]))
    R = min(N, max(1, int(N * frac)))
    if Q * R > 40_000_000: R = max(1, 40_000_000 // Q)
    C = int(rng.choice([3, 10, 70, 130]))
    kind = int(rng.integers(0, 4))
    if kind == 0: d, q = np.tanh(rng.standard_normal((N, b))), np.tanh(rng.standard_normal((Q, b)))
    elif kind == 1: d, q = rng.integers(-3, 4, (N, b)).astype(float), rng.integers(-3, 4, (Q, b)).astype(float)
    elif kind == 2: d, q = rng.standard_normal((N, b)) * 10.0 ** rng.integers(-2, 3, (N, 1)), rng.standard_normal((Q, b))
    else: d, q = (rng.random((N, b)) < 0.5).astype(float), (rng.random((Q, b)) < 0.5).astype(float)
    dbf, qf = d.astype(np.float32), q.astype(np.float32)
    dl = (rng.random((N, C)) < 0.15).astype(np.int64); ql = (rng.random((Q, C)) < 0.15).astype(np.int64)
    if rng.random() < 0.35:                      # features that follow a class, rows stored class by class: the top rows crowd into few slices
        cls, qcls = np.sort(rng.integers(0, 8, N)), rng.integers(0, 8, Q)
        proto = rng.standard_normal((8, b)).astype(np.float32) * np.float32(np.abs(dbf).mean() + 0.1)
        dbf, qf = (dbf + proto[cls]).astype(np.float32), (qf + proto[qcls]).astype(np.float32)
    elif rng.random() < 0.4 and N >= 70000 and Q >= 40:    # a few queries whose top rows are near-copies stored together: they alone lose
        for _ in range(int(rng.integers(1, 3))):            # their cut and are ranked again on their own (real_requery_lost)
            qi, at, n = int(rng.integers(0, Q)), int(rng.integers(0, N - 3000)), int(rng.integers(500, 3000))
            v = np.sign(rng.standard_normal(b)).astype(np.float32) * np.float32(np.abs(qf).mean() + 0.05)
            qf[qi] = v
            dbf[at:at + n] = (np.float32(0.8) * np.abs(dbf).mean() * np.sign(v) + np.float32(0.2) * dbf[at:at + n]).astype(np.float32)
    ctx = _native.Context(0)
    try:
        ctx.set_database_f32(dbf, dl); ctx.set_queries_f32(qf, ql)
        modes = [(2, 1), (2, 0)] + ([(1, 1), (0, 1)] if b <= 128 else [])
        out = {}
        for mode, lds in modes:
            ctx.set_option("real_mfma", mode); ctx.set_option("real_sort_lds", lds)
            idx, sc = ctx.topr_real(R)
            ap, rel = ctx.map_real(R)
            out[(mode, lds)] = (idx, sc.view(np.uint32), ap, rel)
        global REQUERIED
        REQUERIED += ctx.get_stat("real_requeried")
        ref = out[modes[-1]]
        for k, v in out.items():
            if not (np.array_equal(v[0], ref[0]) and np.array_equal(v[1], ref[1]) and np.array_equal(v[2], ref[2], equal_nan=True) and np.array_equal(v[3], ref[3])):
                return "MISMATCH %s seed=%d b=%d N=%d Q=%d R=%d C=%d kind=%d" % (k, seed, b, N, Q, R, C, kind)
        return "ok seed=%d b=%d N=%d Q=%d R=%d C=%d kind=%d" % (seed, b, N, Q, R, C, kind)
    finally:
        ctx.close()

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0; t = time.time()
    for seed in range(s0, s0 + n):
        if "-v" in sys.argv: print("seed", seed, flush=True)
        r = one(seed)
        if r.startswith("MISMATCH"): bad += 1; print(r, flush=True)
        elif seed % 10 == 0: print(r, flush=True)
    print("done: %d shapes, %d mismatches, %d queries ranked again on their own, %.0f s" % (n, bad, REQUERIED, time.time() - t))
