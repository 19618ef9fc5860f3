"""The oracle against the golden fixtures (outputs of the UNMODIFIED reference,
tests/golden/make_golden.py): bit-exact per-query AP, mAP and -- for the small
cases -- the ranked index lists themselves.  Runs on CPU."""
import warnings
import numpy as np
import pytest
from tests import cases
from oracle import hamming_map as O

FAST = cases.SMALL + ["c3_nus_q64", "e_big_r", "c2_q64", "c2_iid_q64", "c5_b128_q32"]


def _check(name, case_cache, q_limit=None):
    c = case_cache(name)
    g = cases.load_golden(name)
    sl = slice(0, q_limit)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap, imatch, idx, dist = O.map_from_codes(c["qbits"][sl], c["dbbits"], c["qlab"][sl], c["dblab"], c["R"])
    assert np.array_equal(ap, g["ap"][sl], equal_nan=True), name
    if q_limit is None:
        assert (np.isnan(m) and np.isnan(g["map"])) or m == g["map"], name
    if "idx" in g:
        assert np.array_equal(idx, g["idx"][sl]), name
    # the order really is (distance asc, index asc)
    key = dist.astype(np.int64) * (1 << 32) + idx
    assert (np.diff(key, axis=1) > 0).all()


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference_golden(name, case_cache):
    _check(name, case_cache)


def test_oracle_c1_cifar_subset(case_cache):
    _check("c1_cifar_full", case_cache, q_limit=100)


def test_oracle_c4_n10m(case_cache):
    _check("c4_n10m_q8", case_cache, q_limit=4)


def test_r_exceeds_n_raises():
    q = np.zeros((2, 8), np.uint8)
    d = np.zeros((5, 8), np.uint8)
    with pytest.raises(ValueError):
        O.topr_from_codes(q, d, 6)


def test_pack_bits_layout():
    bits = np.zeros((1, 70), np.uint8)
    bits[0, 0] = 1
    bits[0, 63] = 1
    bits[0, 64] = 1
    bits[0, 69] = 1
    w = O.pack_bits(bits)
    assert w.shape == (1, 2)
    assert w[0, 0] == (1 | (1 << 63)) and w[0, 1] == (1 | (1 << 5))


def test_as_written_envelope():
    """metric.py as written (default unstable sort) on tied +-1 codes: its AP per
    query must lie inside the envelope spanned by best/worst tie order, and the
    canonical-order AP lies in the same envelope (class P2)."""
    from hashgan_amd import synth
    Q, N, b, R, C = 12, 4000, 16, 700, 5
    dl, _ = synth.onehot_labels(11, N, C)
    ql, _ = synth.onehot_labels(12, Q, C)
    db = synth.planted_codes(13, dl, b, 0.3)
    qb = synth.planted_codes(13, ql, b, 0.3)
    dbf = (db.astype(np.float32) * 2 - 1)
    qf = (qb.astype(np.float32) * 2 - 1)
    m_can, ap_can, *_ = O.map_from_codes(qb, db, ql, dl, R)
    D = O.hamming_matrix(O.pack_bits(qb), O.pack_bits(db))
    for i in range(Q):
        rel_all = (dl.astype(np.int64) @ ql[i].astype(np.int64)) > 0
        best = np.lexsort((~rel_all, D[i]))[:R]     # relevant first inside each tie group
        worst = np.lexsort((rel_all, D[i]))[:R]
        ap_hi, _ = O.average_precision(rel_all[best], R)
        ap_lo, _ = O.average_precision(rel_all[worst], R)
        one = O.reference_as_written(dbf, dl.astype(np.int64), qf[i:i + 1], ql[i:i + 1].astype(np.int64), R)
        assert ap_lo - 1e-12 <= one <= ap_hi + 1e-12
        assert ap_lo - 1e-12 <= ap_can[i] <= ap_hi + 1e-12
