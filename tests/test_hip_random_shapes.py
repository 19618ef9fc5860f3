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
            b = int(rng.choice([1, 2, 5, 8, 16, 31, 32, 33, 48, 63, 64, 65, 96, 100, 128, 129, 200, 255]))
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
    """Shapes large enough for the sampled-threshold bet, with awkward R, b and label widths, through both
    select kernels of the bet: the matrix-core one (fp4 MFMA tiles, k_select_mx) and the vector-ALU one."""
    rng = np.random.default_rng(7)
    ctx = _native.Context(0)
    try:
        for b, C, R, Q in [(24, 5, 2000, 97), (64, 70, 3333, 97), (40, 130, 1000, 97), (100, 10, 5000, 97), (64, 10, 1, 97),
                           (32, 3, 7, 97), (48, 10, 100, 97), (255, 10, 700, 40), (33, 2, 500, 300), (160, 65, 900, 31),
                           (64, 10, 8000, 520)]:
            N = 70000 + int(rng.integers(0, 999))
            qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
            db = rng.integers(0, 2, (N, b), dtype=np.uint8)
            dl = (rng.random((N, C)) < 0.2).astype(np.int8)
            ql = (rng.random((Q, C)) < 0.2).astype(np.int8)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, ap_ref, _, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
            ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
            ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
            for mfma, packed in ((1, 3), (1, 1), (0, 1)):       # k_select_mx3 / mx4 (one-byte records) by the default rule, k_select_mx forced, vector ALU
                ctx.set_option("select_mfma", mfma)
                ctx.set_option("select_packed", packed)
                r0 = ctx.get_stat("optimistic_runs")
                ap, rel = ctx.map(R)
                assert ctx.get_stat("optimistic_runs") == r0 + 1, (b, C, R, mfma, packed)
                assert ctx.get_stat("last_optimistic") == 1, (b, C, R, mfma)
                assert np.array_equal(ap, ap_ref, equal_nan=True), (b, C, R, mfma)
                ctx.topr(R)
                idx, dist = ctx.get_topr()
                assert np.array_equal(idx, idx_ref) and np.array_equal(dist, dist_ref), (b, C, R, mfma)
    finally:
        ctx.close()


def test_mx_select_segment_shapes():
    """k_select_mx geometry corners: an unpaired last segment (odd segment count), a ragged last window,
    segments shorter than one window, under both matrix-core select kernels."""
    rng = np.random.default_rng(11)
    ctx = _native.Context(0)
    try:
        b, C, R, Q = 64, 10, 600, 70
        for N, units, minseg, qt in [(65536 + 77, 7 * 2, 256, 2), (65536 + 77, 7 * 2, 256, 4), (70001, 1000 * 2, 48, 2),
                                     (66000, 3 * 2, 256, 4), (80000, 16384, 256, 2)]:
            qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
            db = rng.integers(0, 2, (N, b), dtype=np.uint8)
            dl = (rng.random((N, C)) < 0.2).astype(np.int8)
            ql = (rng.random((Q, C)) < 0.2).astype(np.int8)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, ap_ref, _, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
            ctx.set_option("target_units", units)
            ctx.set_option("min_segment", minseg)
            ctx.set_option("select_packed", 3 if qt == 2 else 0)     # k_select_mx3 and k_select_mx both see these geometries
            ctx.set_option("optimistic", 1)
            ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
            ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
            ap, rel = ctx.map(R)
            key = (N, units, minseg, qt, ctx.get_stat("segments"))
            assert ctx.get_stat("last_optimistic") == 1, key
            assert np.array_equal(ap, ap_ref, equal_nan=True), key
            ctx.topr(R)
            idx, dist = ctx.get_topr()
            assert np.array_equal(idx, idx_ref) and np.array_equal(dist, dist_ref), key
    finally:
        ctx.close()


def test_fuzz_bet_shapes():
    """Random shapes in the bet's range (N >= 65536, R <= N / 8) through all three select kernels of the bet;
    HG_RANDOM_ITERS scales the number of shapes (default 6)."""
    iters = max(1, int(os.environ.get("HG_RANDOM_ITERS", "60")) // 10)
    rng = np.random.default_rng(int(os.environ.get("HG_RANDOM_SEED", "2024")))
    ctx = _native.Context(0)
    try:
        for it in range(iters):
            b = int(rng.choice([1, 7, 16, 31, 32, 33, 48, 63, 64, 65, 96, 128, 129, 200, 255]))
            C = int(rng.choice([1, 2, 10, 21, 64, 65, 128, 129]))
            N = int(rng.integers(65536, 140000))
            Q = int(rng.choice([1, 5, 31, 32, 33, 100, 257, 600]))
            R = int(rng.choice([1, 3, 50, 1000, N // 64, N // 9]))
            kind = ["iid", "fewvalues", "clustered"][it % 3]
            if kind == "iid":
                db = rng.integers(0, 2, (N, b), dtype=np.uint8)
                qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
            elif kind == "fewvalues":                                   # many exact duplicates: long tie runs at the cut
                proto = rng.integers(0, 2, (37, b), dtype=np.uint8)
                db = proto[rng.integers(0, 37, N)]
                qb = proto[rng.integers(0, 37, Q)]
            else:                                                      # noisy copies of a few centres
                cen = rng.integers(0, 2, (11, b), dtype=np.uint8)
                db = cen[rng.integers(0, 11, N)] ^ (rng.random((N, b)) < 0.08).astype(np.uint8)
                qb = cen[rng.integers(0, 11, Q)] ^ (rng.random((Q, b)) < 0.08).astype(np.uint8)
            dl = (rng.random((N, C)) < 0.3).astype(np.int8)
            ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, ap_ref, _, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
            ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
            ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
            for mfma, packed in ((1, 3), (1, 1), (0, 1)):
                ctx.set_option("select_mfma", mfma)
                ctx.set_option("select_packed", packed)
                ctx.set_option("optimistic", 1)
                key = (it, kind, b, C, N, Q, R, mfma, packed)
                ap, rel = ctx.map(R)
                assert np.array_equal(ap, ap_ref, equal_nan=True), key
                ctx.topr(R)
                idx, dist = ctx.get_topr()
                assert np.array_equal(idx, idx_ref) and np.array_equal(dist, dist_ref), key
    finally:
        ctx.close()
