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


def test_map_between_extra_metrics_never_ranks_a_stale_database():
    """MAP on read-only arrays keeps the packed database resident; extra_metrics loads ANOTHER database into the same
    shared engine in between -- the next MAP on the first arrays must upload again, not rank against the other one."""
    from hashgan_amd import MAP_per_query, extra_metrics as X, metric
    rng = np.random.default_rng(11)
    Q, N, b, C = 40, 3000, 32, 5
    db1 = rng.integers(0, 2, (N, b), dtype=np.uint8)
    db2 = rng.integers(0, 2, (N, b), dtype=np.uint8)
    qb = db1[rng.integers(0, N, Q)] ^ (rng.random((Q, b)) < 0.1).astype(np.uint8)
    dl1 = np.eye(C, dtype=np.int8)[rng.integers(0, C, N)]
    dl2 = np.eye(C, dtype=np.int8)[rng.integers(0, C, N)]
    ql = np.eye(C, dtype=np.int8)[rng.integers(0, C, Q)]
    for a in (db1, dl1):
        a.flags.writeable = False
    m_ref, ap_ref, *_ = O.map_from_codes(qb, db1, ql, dl1, 500)
    m1, ap1, _ = MAP_per_query(qb, db1, ql, dl1, 500)
    assert metric._Shared.get(0).resident is not None
    X.precision_recall_at_k(qb, db2, ql, dl2, [10])
    assert metric._Shared.get(0).resident is None
    m2, ap2, _ = MAP_per_query(qb, db1, ql, dl1, 500)
    assert np.array_equal(ap1, ap_ref, equal_nan=True) and np.array_equal(ap2, ap_ref, equal_nan=True)
    assert m1 == m_ref == m2
