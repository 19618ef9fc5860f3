"""BASELINE.json's headline configuration at FULL size (C2: Q=10k, N=1M, b=64,
R=5000), where the oracle would take hours: size-independent properties of the
HIP result plus a golden anchor on the first 64 queries."""
import numpy as np
import pytest
from tests import cases
from oracle import hamming_map as O
from hashgan_amd import _native, metric

pytestmark = pytest.mark.gpu


def test_c2_full_properties():
    spec = dict(cases.CASES["c2_q64"])
    spec.pop("q_take")
    cases.CASES["_c2_full"] = spec
    try:
        c = cases.build_case("_c2_full")
    finally:
        del cases.CASES["_c2_full"]
    Q, N, R, b = c["qbits"].shape[0], c["dbbits"].shape[0], c["R"], c["b"]
    ctx = _native.Context(0)
    qw, dw = metric.pack_codes(c["qbits"]), metric.pack_codes(c["dbbits"])
    ctx.set_database(dw, metric.pack_labels(c["dblab"]), b, 10)
    ctx.set_queries(qw, metric.pack_labels(c["qlab"]))
    ap, rel = ctx.map(R)                              # one-shot (sampled-threshold bet)
    assert ctx.get_stat("last_optimistic") == 1
    g = cases.load_golden("c2_q64")
    # anchor: the first 64 queries are exactly the golden case
    assert np.array_equal(ap[:64], g["ap"], equal_nan=True)
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
        d = np.bitwise_count(dw[idx[q].astype(np.int64), 0] ^ qw[q, 0])
        assert np.array_equal(d, dist[q])
    # AP recomputed on the host from the device's match bits, NumPy expressions of metric.py:20-23
    m = ctx.get_match().astype(bool)
    for q in range(0, Q, 499):
        a, r = O.average_precision(m[q], R)
        assert r == rel[q] and (a == ap[q] or (a is None and np.isnan(ap[q])))
    ctx.close()
