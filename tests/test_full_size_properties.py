"""BASELINE.json's configurations at FULL size on one GPU (C2: Q=10k, N=1M, b=64, R=5000; C3: Q=2.1k, N=190k,
b=48, 81 multi-hot classes; C5: b=128; C4: N=10M, all 10 000 queries), where the oracle would take hours: size-independent properties of the
HIP result plus a golden anchor on the first queries."""
import numpy as np
import pytest
from tests import cases
from oracle import hamming_map as O
from hashgan_amd import _native, metric

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c2_q64", "c2_iid_q64", "c3_nus_q64", "c5_b128_q32", "c4_n10m_q8"])
def test_full_size_properties(name):
    spec = dict(cases.CASES[name])
    spec.pop("q_take")
    cases.CASES["_full"] = spec
    try:
        c = cases.build_case("_full")
    finally:
        del cases.CASES["_full"]
    Q, N, R, b = c["qbits"].shape[0], c["dbbits"].shape[0], c["R"], c["b"]
    ctx = _native.Context(0)
    qw, dw = metric.pack_codes(c["qbits"]), metric.pack_codes(c["dbbits"])
    ctx.set_database(dw, metric.pack_labels(c["dblab"]), b, c["dblab"].shape[1])
    ctx.set_queries(qw, metric.pack_labels(c["qlab"]))
    ap, rel = ctx.map(R)                              # one-shot (sampled-threshold bet)
    assert ctx.get_stat("last_optimistic") == 1
    g = cases.load_golden(name)
    # anchor: the first queries are exactly the golden case
    k = g["ap"].shape[0]
    assert np.array_equal(ap[:k], g["ap"], equal_nan=True)
    # staged exact path: full histogram -> plan -> select; must agree with the one-shot result
    ctx.hist(); ctx.plan(R); ctx.select(); ctx.ap()
    ap2, rel2 = ctx.get_ap()
    assert np.array_equal(ap, ap2, equal_nan=True) and np.array_equal(rel, rel2)
    idx, dist = ctx.get_topr()
    # canonical order: (dist, idx) strictly increasing along every ranked list
    key = dist.astype(np.int64) * (1 << 32) + idx.astype(np.int64)
    assert (np.diff(key, axis=1) > 0).all()
    assert idx.max() < N
    # histogram: every query saw every row once
    h = ctx.get_hist().astype(np.int64)
    assert (h.sum(0) == N).all()
    # cut consistency: the list holds everything closer than its last distance
    t = dist[:, -1].astype(np.int64)
    closer = np.array([h[:t[q], q].sum() for q in range(Q)])
    upto = np.array([h[:t[q] + 1, q].sum() for q in range(Q)])
    assert (closer < R).all() and (upto >= R).all()
    assert ((dist < t[:, None]).sum(1) == closer).all()
    # distances are the Hamming distances of the listed rows (sampled queries)
    for q in range(0, Q, 997):
        d = np.bitwise_count(dw[idx[q].astype(np.int64)] ^ qw[q][None, :]).sum(1)
        assert np.array_equal(d, dist[q])
    # AP recomputed on the host from the device's match bits, NumPy expressions of metric.py:20-23
    m = ctx.get_match().astype(bool)
    for q in range(0, Q, 499):
        a, r = O.average_precision(m[q], R)
        assert r == rel[q] and (a == ap[q] or (a is None and np.isnan(ap[q])))
        # ... and the match bits themselves against the labels (metric.py:17-19)
        assert np.array_equal(m[q], O.label_match(c["qlab"][q], c["dblab"][idx[q].astype(np.int64)]))
    ctx.close()
