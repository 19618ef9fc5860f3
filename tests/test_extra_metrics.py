"""precision/recall@k and precision within Hamming radius against brute-force NumPy."""
import numpy as np
import pytest
from oracle import hamming_map as O

pytestmark = pytest.mark.gpu


def test_extra_metrics_match_bruteforce():
    from hashgan_amd import extra_metrics as X
    rng = np.random.default_rng(4)
    Q, N, b, C = 60, 5000, 16, 6
    db = rng.integers(0, 2, (N, b), dtype=np.uint8)
    qb = db[rng.integers(0, N, Q)] ^ (rng.random((Q, b)) < 0.08).astype(np.uint8)
    dl = (rng.random((N, C)) < 0.3).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
    ql[0] = 0                                                    # a query without labels
    D = O.hamming_matrix(O.pack_bits(qb), O.pack_bits(db))
    rel = (ql.astype(np.int64) @ dl.astype(np.int64).T) > 0
    ks = [1, 10, 100, 1000]
    order = np.argsort(D, axis=1, kind="stable")
    relo = np.take_along_axis(rel, order, 1)
    p_ref = np.array([relo[:, :k].sum(1) / k for k in ks]).T.mean(0)
    tot = rel.sum(1)
    ok = tot > 0
    r_ref = np.array([relo[ok][:, :k].sum(1) / tot[ok] for k in ks]).T.mean(0)
    p, r = X.precision_recall_at_k(qb, db, ql, dl, ks)
    assert np.allclose(p, p_ref, rtol=0, atol=1e-15) and np.allclose(r, r_ref, rtol=0, atol=1e-15)
    for radius in (0, 2, 4):
        inside = D <= radius
        ball = inside.sum(1)
        ref = np.where(ball > 0, (inside & rel).sum(1) / np.maximum(ball, 1), 0.0).mean()
        got, balls = X.precision_within_radius(qb, db, ql, dl, radius)
        assert np.array_equal(balls, ball) and abs(got - ref) < 1e-15
