"""Randomised small shapes through the whole HIP path against the oracle: odd Q/N/b/R/C, tiny
and degenerate sizes, heavy ties (few bits), high and low R/N, multi-hot and empty-label rows,
every segment geometry the engine may pick.  Bit-exact AP, ranked lists, match bits."""
import os
import warnings
import numpy as np
import pytest
from oracle import hamming_map as O
from hashgan_amd import _native, metric

pytestmark = pytest.mark.gpu


def _one(ctx, rng, Q, N, b, R, C, kind):
    if kind == "iid":
        db = rng.integers(0, 2, (N, b), dtype=np.uint8)
        qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
    elif kind == "fewvalues":                      # only a handful of distinct codes: giant tie groups
        pool = rng.integers(0, 2, (5, b), dtype=np.uint8)
        db = pool[rng.integers(0, 5, N)]
        qb = pool[rng.integers(0, 5, Q)]
    else:                                          # clustered around the queries
        qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
        db = qb[rng.integers(0, Q, N)] ^ (rng.random((N, b)) < 0.15).astype(np.uint8)
    dl = (rng.random((N, C)) < 0.3).astype(np.int8)   # multi-hot, some rows without any label
    ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m_ref, ap_ref, im_ref, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
    ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
    ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
    ctx.set_option("target_units", int(rng.choice([1, 7, 64, 1000, 16384, 200000])))
    ctx.set_option("min_segment", int(rng.choice([16, 64, 256])))
    tag = (Q, N, b, R, C, kind)
    ap, rel = ctx.map(R)                           # one-shot (bets when the shape allows)
    assert np.array_equal(ap, ap_ref, equal_nan=True), tag
    assert np.array_equal(rel, im_ref.sum(1)), tag
    ctx.topr(R)
    idx, dist = ctx.get_topr()
    assert np.array_equal(idx, idx_ref) and np.array_equal(dist, dist_ref), tag
    assert np.array_equal(ctx.get_match().astype(bool), im_ref), tag
    ctx.hist(); ctx.plan(R); ctx.select(); ctx.match(); ctx.ap()     # staged exact sequence
    ap2, rel2 = ctx.get_ap()
    assert np.array_equal(ap2, ap_ref, equal_nan=True), tag


def test_random_small_shapes():
    iters = int(os.environ.get("HG_RANDOM_ITERS", "60"))           # a longer campaign: HG_RANDOM_ITERS=1000
    rng = np.random.default_rng(20260928 + iters)
    ctx = _native.Context(0)
    try:
        for it in range(iters):
            b = int(rng.choice([1, 2, 5, 8, 16, 31, 32, 33, 48, 63, 64, 65, 96, 100, 128, 129, 200, 256]))
            Q = int(rng.choice([1, 2, 63, 64, 65, 100, 130]))
            N = int(rng.choice([1, 2, 15, 16, 17, 255, 256, 257, 1000, 4097]))
            R = int(rng.choice([1, max(1, N // 7), max(1, N // 2), N]))
            C = int(rng.choice([1, 3, 10, 64, 65, 81, 128, 129, 150]))
            _one(ctx, rng, Q, N, b, R, C, ["iid", "fewvalues", "clustered"][it % 3])
    finally:
        ctx.set_option("target_units", 16384)
        ctx.set_option("min_segment", 256)
        ctx.close()


def test_random_bet_shapes():
    """Shapes large enough for the sampled-threshold bet, with awkward R, b and label widths."""
    rng = np.random.default_rng(7)
    ctx = _native.Context(0)
    try:
        for b, C, R in [(24, 5, 2000), (64, 70, 3333), (40, 130, 1000), (100, 10, 5000), (64, 10, 1), (32, 3, 7), (48, 10, 100)]:
            Q, N = 97, 70000 + int(rng.integers(0, 999))
            qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
            db = rng.integers(0, 2, (N, b), dtype=np.uint8)
            dl = (rng.random((N, C)) < 0.2).astype(np.int8)
            ql = (rng.random((Q, C)) < 0.2).astype(np.int8)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, ap_ref, _, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
            ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
            ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
            r0 = ctx.get_stat("optimistic_runs")
            ap, rel = ctx.map(R)
            assert ctx.get_stat("optimistic_runs") == r0 + 1, (b, C, R)
            assert np.array_equal(ap, ap_ref, equal_nan=True), (b, C, R)
            ctx.topr(R)
            idx, dist = ctx.get_topr()
            assert np.array_equal(idx, idx_ref) and np.array_equal(dist, dist_ref), (b, C, R)
    finally:
        ctx.close()


def test_very_long_lists_use_the_global_bit_rows():
    """R beyond what a block's LDS bitmap holds (~0.5M slots): match bits go through global atomics
    on a zeroed row instead.  One shard and two virtual shards' worth of plan (staged, G = 1)."""
    rng = np.random.default_rng(11)
    Q, N, b, R, C = 3, 600000, 32, 530000, 4
    qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
    db = rng.integers(0, 2, (N, b), dtype=np.uint8)
    dl = (rng.random((N, C)) < 0.3).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.5).astype(np.int8)
    ql[:, 0] = 1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, im_ref, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        ctx.topr(R)
        idx, dist = ctx.get_topr()
        assert np.array_equal(idx, idx_ref) and np.array_equal(dist, dist_ref)
        assert np.array_equal(ctx.get_match().astype(bool), im_ref)
    finally:
        ctx.close()
