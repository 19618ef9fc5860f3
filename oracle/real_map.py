"""CPU oracle for the REAL-VALUED ranking of lib/metric.py.
TEST INFRASTRUCTURE ONLY -- same rules as oracle/hamming_map.py.

metric.py:13-14 ranks by float32 inner products, `np.argsort(-np.dot(q, db.T), 1)`.  A float32
GEMM's rounding depends on its summation order (OpenBLAS's is not the GPU's), so the build fixes
one order and this file restates it exactly:

    ip(q, d) = acc + 0.0,   acc = fma(q[k], d[k], acc) for k = 0, 1, ..., from +0.0            all in float32
               (features zero-padded to a multiple of 16)

-- ONE fma chain in feature order: measured on the MI355X (tools/mfma_f32_probe.hip), that is bit for bit what the
chained `v_mfma_f32_32x32x2_f32` instructions of the select kernel compute per (query, row) pair (the final `+ 0.0`
folds -0.0 into +0.0 so that equal values compare equal as bit patterns).  Ranking: ip descending, database
index ascending.  AP/mAP: the expressions of metric.py:17-24, via oracle.hamming_map.

Parity pin: on features whose products and partial sums are exactly representable in float32
(multiples of 1/64 in [-1, 1], up to the 255 features the loaders take) EVERY summation order gives the same value, so the
unmodified reference -- fed float64 copies plus the tie-breaking coordinate of
hamming_map.tie_free_features -- must agree with this oracle bit for bit: tests/golden/real_*.npz.
"""
import numpy as np
from oracle import hamming_map as H


def fma32(a, b, c):
    """Correctly rounded float32 fma(a, b, c) for float32 arrays, via float64 + an exact fix-up.
    a*b is exact in float64 (2 x 24-bit significands); s = fl64(p + c) may round, TwoSum recovers the
    error e exactly; float32(s) is then wrong only when s sits exactly on a float32 rounding
    boundary (a tie) while the true value s + e does not."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c64 = c.astype(np.float64)
    s = p + c64
    bb = s - p
    e = (p - (s - bb)) + (c64 - bb)              # TwoSum: p + c64 == s + e exactly
    r = s.astype(np.float32)
    need = (e != 0) & np.isfinite(s)
    if need.any():
        r64 = r.astype(np.float64)
        lo = np.nextafter(r, np.float32(-np.inf)).astype(np.float64)
        hi = np.nextafter(r, np.float32(np.inf)).astype(np.float64)
        # s was a tie between r and a neighbour iff |s - r| equals half the gap on that side
        tie_up = need & (s > r64) & ((s - r64) == (hi - s))
        tie_dn = need & (s < r64) & ((r64 - s) == (s - lo))
        r = np.where(tie_up & (e > 0), hi.astype(np.float32), r)      # truth is above the midpoint
        r = np.where(tie_dn & (e < 0), lo.astype(np.float32), r)      # truth is below the midpoint
        # ties rounded AWAY from r by float32(): s exactly midway, numpy chose the even neighbour r;
        # if the truth lies on the other side of the midpoint the other neighbour is right
        mid_hi = need & (s == (r64 + hi) / 2) & (e > 0)
        mid_lo = need & (s == (r64 + lo) / 2) & (e < 0)
        r = np.where(mid_hi, hi.astype(np.float32), r)
        r = np.where(mid_lo, lo.astype(np.float32), r)
    return r


def inner_products(qf, dbf):
    """float32 [Q, b] x float32 [N, b] -> float32 [Q, N] in the build's summation order."""
    qf = np.ascontiguousarray(qf, dtype=np.float32)
    dbf = np.ascontiguousarray(dbf, dtype=np.float32)
    Q, b = qf.shape
    N = dbf.shape[0]
    acc = np.zeros((Q, N), np.float32)
    for k in range(b):
        acc = fma32(np.broadcast_to(qf[:, k][:, None], (Q, N)), np.broadcast_to(dbf[:, k][None, :], (Q, N)), acc)
    return acc + np.float32(0.0)


def map_from_features(qf, dbf, qlabels, dblabels, R):
    """(mAP, ap [Q] with nan for skipped queries, idx int64 [Q, R], score float32 [Q, R])."""
    ips = inner_products(qf, dbf)
    Q, N = ips.shape
    if R > N:
        raise ValueError("R=%d exceeds database size N=%d" % (R, N))
    idx = np.empty((Q, R), np.int64)
    score = np.empty((Q, R), np.float32)
    ap = np.full(Q, np.nan)
    kept = []
    for i in range(Q):
        o = np.argsort(-ips[i], kind="stable")[:R]            # ip descending, index ascending
        idx[i], score[i] = o, ips[i, o]
        a, _ = H.average_precision(H.label_match(qlabels[i, :], dblabels[o, :]), R)
        if a is not None:
            ap[i] = a
            kept.append(a)
    return H.mean_ap(kept), ap, idx, score


def quantised_features(seed, n, b, levels=64):
    """Features on the grid {-1, ..., -1/levels, 0, 1/levels, ..., 1}: float32 GEMMs over them are exact."""
    from hashgan_amd import synth
    raw = synth.splitmix64(seed, n * b).reshape(n, b) % np.uint64(2 * levels + 1)
    return ((raw.astype(np.int64) - levels).astype(np.float32) / np.float32(levels))


def tie_free_real_features(x, is_query, n_db, levels=64):
    """float64 copy of grid features plus one coordinate that orders exact ties by database index
    (query rows +1, database row j gets -j*eps, eps far below the grid's smallest inner-product
    step 1/levels^2), so the UNMODIFIED reference's argsort has no ties to break."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    extra = np.empty((n, 1), np.float64)
    if is_query:
        extra[:] = 1.0
    else:
        eps = (1.0 / (levels * levels)) * 2.0 ** -(int(np.ceil(np.log2(max(n_db, 2)))) + 1)
        extra[:, 0] = -np.arange(n, dtype=np.float64) * eps
    return np.concatenate([x, extra], axis=1)
