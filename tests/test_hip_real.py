"""Real-valued feature ranking on the GPU (hg_map_real / hg_topr_real) against
the oracle's exact restatement (oracle/real_map.py) and the unmodified reference's goldens."""
import types
import warnings
import numpy as np
import pytest
from tests import cases
from oracle import real_map as RM
from hashgan_amd import _native

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


def _load(ctx, c):
    assert ctx.set_database_f32(c["dbf"], c["dblab"].astype(np.int64))[1] == 0
    assert ctx.set_queries_f32(c["qf"], c["qlab"].astype(np.int64))[1] == 0


@pytest.mark.parametrize("name", list(cases.REAL_CASES))
def test_grid_features_match_reference_golden(name, ctx):
    """Grid features: float arithmetic is exact, so the GPU must reproduce the UNMODIFIED reference."""
    c = cases.build_real_case(name)
    g = cases.load_golden(name)
    _load(ctx, c)
    ap, rel = ctx.map_real(c["R"])
    assert np.array_equal(ap, g["ap"], equal_nan=True), name
    assert np.mean(ap[rel != 0]) == g["map"]
    idx, score = ctx.topr_real(c["R"])
    if "idx" in g:
        assert np.array_equal(idx, g["idx"]), name
    exact = np.einsum("qb,qrb->qr", c["qf"].astype(np.float64), c["dbf"].astype(np.float64)[idx.astype(np.int64)])
    assert np.array_equal(score.astype(np.float64), exact)


@pytest.mark.parametrize("Q,N,b,R,C", [(40, 3000, 20, 700, 7), (65, 70000, 64, 2500, 10), (10, 2000, 128, 2000, 3),
                                       (3, 500, 1, 40, 2), (33, 5000, 33, 1200, 70), (9, 3000, 129, 500, 4),
                                       (70, 66000, 200, 1000, 5), (5, 1000, 255, 1000, 3)])
def test_generic_float_features_match_oracle_bit_for_bit(Q, N, b, R, C, ctx):
    """Arbitrary float32 features (tanh of Gaussians): the kernel's summation order is restated
    exactly by the oracle (fma32 chains), so scores, order and AP agree bit for bit."""
    rng = np.random.default_rng(Q * 7 + b)
    dbf = np.tanh(rng.standard_normal((N, b))).astype(np.float32)
    qf = np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
    dbf[N // 3] = dbf[N // 3 + 1]                                  # a duplicated row: exact tie, index order
    dl = (rng.random((N, C)) < 0.25).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.25).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql, dl, R)
    ctx.set_database_f32(dbf, dl.astype(np.int64))
    ctx.set_queries_f32(qf, ql.astype(np.int64))
    idx, score = ctx.topr_real(R)
    assert np.array_equal(score.view(np.uint32), score_ref.view(np.uint32))
    assert np.array_equal(idx, idx_ref)
    ap, rel = ctx.map_real(R)
    assert np.array_equal(ap, ap_ref, equal_nan=True)


@pytest.mark.parametrize("kind", ["tanh", "grid", "sorted"])
def test_every_row_ranked_group_by_group(kind):
    """R = N (the reference's CIFAR-10 setting) on real-valued features: every row is a record, far more than the LDS holds.
    Continuous scores are split by score range into LDS-sized groups and ordered group by group (k_real_group_split /
    k_real_group_sort, stat real_path bit 2); scores on a coarse grid pile up in the buckets and take the radix passes.  The
    oracle's lists and APs either way, bit for bit."""
    rng = np.random.default_rng(31)
    Q, N, b, C = 6, 30000, 32, 10
    R = N
    if kind == "tanh":
        dbf, qf = np.tanh(rng.standard_normal((N, b))).astype(np.float32), np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
    elif kind == "grid":
        dbf, qf = rng.integers(-1, 2, (N, b)).astype(np.float32), rng.integers(-1, 2, (Q, b)).astype(np.float32)
    dl = (rng.random((N, C)) < 0.2).astype(np.int64)
    ql = (rng.random((Q, C)) < 0.2).astype(np.int64)
    if kind == "sorted":
        # features that follow the class, rows stored class by class: k_real_group_split places its score buckets on an eighth of the
        # rows -- 512-byte pieces from all over the run, so every class is in the sample -- and scores beyond the sampled extremes join
        # the end buckets
        cls, qcls = np.sort(rng.integers(0, C, N)), rng.integers(0, C, Q)
        proto = rng.standard_normal((C, b)).astype(np.float32)
        dbf = np.tanh(proto[cls] + 0.5 * rng.standard_normal((N, b))).astype(np.float32)
        qf = np.tanh(proto[qcls] + 0.5 * rng.standard_normal((Q, b))).astype(np.float32)
        dl, ql = np.eye(C, dtype=np.int64)[cls], np.eye(C, dtype=np.int64)[qcls]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql.astype(np.int8), dl.astype(np.int8), R)
    c = _native.Context(0)
    try:
        c.set_database_f32(dbf, dl)
        c.set_queries_f32(qf, ql)
        idx, score = c.topr_real(R)
        assert ((c.get_stat("real_path") >> 2) & 1) == (0 if kind == "grid" else 1)
        assert np.array_equal(idx, idx_ref) and np.array_equal(score.view(np.uint32), score_ref.view(np.uint32))
        ap, rel = c.map_real(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        c.set_option("real_groups", 0)                               # the radix passes: the same lists
        idx2, _ = c.topr_real(R)
        assert ((c.get_stat("real_path") >> 2) & 1) == 0 and np.array_equal(idx2, idx_ref)
    finally:
        c.close()


@pytest.mark.parametrize("Q,N,b", [(70, 20011, 64), (33, 9000, 128), (100, 4999, 20)])
def test_every_row_a_record_leaves_in_whole_lines(Q, N, b):
    """No cut (R = N, metric.py:14 with MAP_R = DB_SIZE): k_real_select_mx turns a tile's records through LDS and stores whole
    128-byte runs; a wavefront with lanes past Q, the last segment's partial tile and a database that ends inside a segment pair
    take the lane-by-lane paths beside it.  The oracle's lists, scores and APs bit for bit -- on the geometry cut to whole rounds
    of blocks (option real_whole_rounds, default 3) and on the plain one."""
    rng = np.random.default_rng(Q + N)
    C, R = 10, N
    dbf, qf = np.tanh(rng.standard_normal((N, b))).astype(np.float32), np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
    dl = (rng.random((N, C)) < 0.2).astype(np.int64)
    ql = (rng.random((Q, C)) < 0.2).astype(np.int64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql.astype(np.int8), dl.astype(np.int8), R)
    c = _native.Context(0)
    try:
        c.set_database_f32(dbf, dl)
        c.set_queries_f32(qf, ql)
        for rounds in (3, 0, 1):
            c.set_option("real_whole_rounds", rounds)
            idx, score = c.topr_real(R)
            assert np.array_equal(idx, idx_ref) and np.array_equal(score.view(np.uint32), score_ref.view(np.uint32)), rounds
            ap, rel = c.map_real(R)
            assert np.array_equal(ap, ap_ref, equal_nan=True), rounds
        # few queries with long lists: k_ap took 512 threads per query above (option ap_wide, default 1); with 128 the same bits
        c.set_option("ap_wide", 0)
        ap, rel = c.map_real(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
    finally:
        c.close()


@pytest.mark.parametrize("Q,N,b,R", [(40, 70000, 64, 700), (12, 20000, 32, 20000)])
def test_map_real_skips_the_ranked_lists_unless_asked(Q, N, b, R):
    """hg_map_real wants label matches and APs (metric.py:17-23): the kernels that rank in LDS leave the idx / score lists
    unwritten then, as hg_map does for codes -- hg_get_topr_real after it is a state error that says so; with option
    real_map_lists = 1 the lists are there and equal hg_topr_real's.  The APs are the oracle's either way."""
    rng = np.random.default_rng(Q * N)
    C = 10
    dbf, qf = np.tanh(rng.standard_normal((N, b))).astype(np.float32), np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
    dl = (rng.random((N, C)) < 0.2).astype(np.int64)
    ql = (rng.random((Q, C)) < 0.2).astype(np.int64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql.astype(np.int8), dl.astype(np.int8), R)
    c = _native.Context(0)
    try:
        c.set_database_f32(dbf, dl)
        c.set_queries_f32(qf, ql)
        ap, rel = c.map_real(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        idx = np.empty((Q, R), np.uint32)
        score = np.empty((Q, R), np.float32)
        rc = c._lib.hg_get_topr_real(c._h, _native._ptr(idx), _native._ptr(score))
        assert rc == _native.HG_ERR_STATE and b"real_map_lists" in _native.load().hg_last_error()
        c.set_option("real_map_lists", 1)
        ap, rel = c.map_real(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        _native.check(c._lib.hg_get_topr_real(c._h, _native._ptr(idx), _native._ptr(score)))
        assert np.array_equal(idx, idx_ref) and np.array_equal(score.view(np.uint32), score_ref.view(np.uint32))
        idx2, score2 = c.topr_real(R)
        assert np.array_equal(idx2, idx_ref) and np.array_equal(score2.view(np.uint32), score_ref.view(np.uint32))
    finally:
        c.close()


def test_a_fine_bucket_of_26_records_stays_with_the_groups():
    """bench.py's CIFAR-shaped leg of round 6 (tanh features, Q = 1000, N = R = 54000, seed 0xD1): query 711's first group holds a
    fine score bucket of 26 records.  The pile guard (then 24) sent the WHOLE call to the four radix passes -- 5.4 ms per call
    instead of 1.8; a bucket of that size is simply ranked by counting (RG_PILE = 96).  The oracle's lists and APs, bit for bit,
    with the groups (stat real_path bit 2) and, for comparison, with the radix passes."""
    rng = np.random.default_rng(0xD1)
    N, Qall, b, C = 54000, 1000, 64, 10
    eye = np.eye(C, dtype=np.int64)
    dl, ql = eye[rng.integers(0, C, N)], eye[rng.integers(0, C, Qall)]
    dbf = np.tanh(rng.standard_normal((N, b), dtype=np.float32))
    qf = np.tanh(rng.standard_normal((Qall, b), dtype=np.float32))
    sl = slice(709, 714)
    qf, ql = np.ascontiguousarray(qf[sl]), np.ascontiguousarray(ql[sl])
    # (the premise, checked on the host: 4096 equal-width buckets over the scores of query 711's first ~6000 rows hold one of > 24)
    s = np.sort((qf[2:3] @ dbf.T)[0])[::-1][:5986]
    fb = np.minimum(((s[0] - s) * (np.float32(4096.0) / (s[0] - s[-1]))).astype(np.int64), 4095)
    assert np.bincount(fb).max() > 24
    R = N
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql.astype(np.int8), dl.astype(np.int8), R)
    c = _native.Context(0)
    try:
        c.set_database_f32(dbf, dl)
        c.set_queries_f32(qf, ql)
        idx, score = c.topr_real(R)
        assert (c.get_stat("real_path") >> 2) & 1, "the call left the groups for the radix passes"
        assert np.array_equal(idx, idx_ref) and np.array_equal(score.view(np.uint32), score_ref.view(np.uint32))
        ap, rel = c.map_real(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        c.set_option("real_groups", 0)
        idx2, _ = c.topr_real(R)
        assert ((c.get_stat("real_path") >> 2) & 1) == 0 and np.array_equal(idx2, idx_ref)
    finally:
        c.close()


def test_features_that_follow_the_labels_in_a_class_sorted_database():
    """What a trained network hands over when the database is stored class by class: a query's top rows all sit in its
    class's tenth of the segments.  Both sampled cuts lose on slice capacity, the slices are widened (real_cap_boost) and
    the third bet holds -- never the exhaustive mode, which writes every pair down (80 GB of records at 10k x 1M; a hard
    error before round 3).  The next call bets with the wide slices at once.  Bit for bit the oracle's lists and APs."""
    rng = np.random.default_rng(11)
    Q, N, b, R, C = 5, 200000, 64, 4000, 10
    proto = rng.standard_normal((C, b)).astype(np.float32)
    cls, qcls = np.sort(rng.integers(0, C, N)), rng.integers(0, C, Q)
    dbf = np.tanh(0.7 * proto[cls] + rng.standard_normal((N, b), dtype=np.float32)).astype(np.float32)
    qf = np.tanh(0.7 * proto[qcls] + rng.standard_normal((Q, b), dtype=np.float32)).astype(np.float32)
    eye = np.eye(C, dtype=np.int64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, eye[qcls], eye[cls], R)
    c = _native.Context(0)
    try:
        c.set_database_f32(dbf, eye[cls])
        c.set_queries_f32(qf, eye[qcls])
        ap, rel = c.map_real(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        assert c.get_stat("real_attempts") >= 3 and c.get_stat("real_cap_boost") > 1 and (c.get_stat("real_path") & 1) == 1
        idx, score = c.topr_real(R)
        assert c.get_stat("real_attempts") == 1                      # the widened slices are remembered
        assert np.array_equal(idx, idx_ref) and np.array_equal(score.view(np.uint32), score_ref.view(np.uint32))
        perm = rng.permutation(N)                                    # the same rows shuffled: a new database, ordinary slices
        c.set_database_f32(dbf[perm], eye[cls][perm])
        c.set_queries_f32(qf, eye[qcls])
        assert c.get_stat("real_cap_boost") == 1
        ap2, _ = c.map_real(R)
        assert c.get_stat("real_attempts") == 1 and c.get_stat("real_cap_boost") == 1
        assert np.array_equal(ap2, ap_ref, equal_nan=True)           # no ties among these scores: the same lists, the same APs
    finally:
        c.close()


def test_a_few_lost_queries_are_ranked_again_on_their_own():
    """Two of 240 queries have their whole top R in one stretch of the database (near-copies of themselves stored together): their
    slices overflow whatever the cut -- and those long rows crowd a few other queries' lists too --, the rest win their bet.
    Only the losers run again (a child context on the same tables, stat "real_requeried"), the call itself stays at one
    attempt, and every list and AP is the oracle's bit for bit."""
    rng = np.random.default_rng(17)
    Q, N, b, R, C = 240, 150000, 32, 1500, 7
    dbf = np.tanh(rng.standard_normal((N, b))).astype(np.float32)
    qf = np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
    for k, (q, at) in enumerate(((3, 20000), (29, 90000))):
        v = np.sign(rng.standard_normal(b)).astype(np.float32) * 0.9
        qf[q] = v
        dbf[at:at + 2500] = np.tanh(0.75 * np.sign(v) + 0.3 * rng.standard_normal((2500, b))).astype(np.float32)
    dl = (rng.random((N, C)) < 0.25).astype(np.int64)
    ql = (rng.random((Q, C)) < 0.25).astype(np.int64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql, dl, R)
    c = _native.Context(0)
    try:
        c.set_database_f32(dbf, dl)
        c.set_queries_f32(qf, ql)
        n0 = c.get_stat("real_requeried")
        ap, rel = c.map_real(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        n1 = c.get_stat("real_requeried") - n0
        assert c.get_stat("real_attempts") == 1 and 2 <= n1 <= 15, (c.get_stat("real_attempts"), n1)
        idx, score = c.topr_real(R)
        assert np.array_equal(idx, idx_ref) and np.array_equal(score.view(np.uint32), score_ref.view(np.uint32))
        assert c.get_stat("real_requeried") - n0 == 2 * n1
    finally:
        c.close()


def test_python_surface_ranks_real_features_like_the_reference():
    """MAPs(R).get_maps_by_feature on tanh-like features = the reference's own semantics."""
    from hashgan_amd import MAPs, MAP
    c = cases.build_real_case("real_multi")
    g = cases.load_golden("real_multi")
    database = types.SimpleNamespace(output=c["dbf"], label=c["dblab"].astype(np.int64))
    query = types.SimpleNamespace(output=c["qf"], label=c["qlab"].astype(np.int64))
    assert MAPs(c["R"]).get_maps_by_feature(database, query) == g["map"]
    with pytest.raises(ValueError):
        MAP(c["qf"], c["dbf"], c["qlab"], c["dblab"], c["R"])          # MAP() is the binary-code spelling


@pytest.mark.parametrize("head", ["tanh", "pm1"])
def test_float_table_upload_paths_of_the_python_surface(head):
    """A float table big enough for the pinned staging (>= 8 MB), 50 features (rows padded to 64 on the way).  Its first
    rows decide how it travels: real-valued from row 0 -- a second thread ships it while the codes are still packed; the
    first hundred rows exactly +-1 -- nobody knows before the census that the rest is real-valued, the floats follow the
    packing.  Either way MAPs must rank what np.dot ranks: the oracle's APs, bit for bit."""
    from hashgan_amd import MAPs
    rng = np.random.default_rng(77)
    Q, N, b, R, C = 6, 45000, 50, 900, 6
    dbf = np.tanh(rng.standard_normal((N, b))).astype(np.float32)
    if head == "pm1":
        dbf[:100] = np.where(rng.random((100, b)) < 0.5, -1.0, 1.0).astype(np.float32)
    qf = np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
    dl = (rng.random((N, C)) < 0.3).astype(np.int64)
    ql = (rng.random((Q, C)) < 0.3).astype(np.int64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql.astype(np.int8), dl.astype(np.int8), R)
    mp = MAPs(R)
    try:
        database = types.SimpleNamespace(output=dbf, label=dl)
        query = types.SimpleNamespace(output=qf, label=ql)
        for _ in range(2):                                            # the second call reuses the pinned buffers and the second stream
            assert mp.get_maps_by_feature(database, query) == m
    finally:
        mp.close()


@pytest.mark.parametrize("name", ["real_bits01", "real_ternary"])
def test_maps_ranks_non_pm1_codes_like_np_dot(name):
    """{0,1} bits and +-1 codes with zeros: np.dot (metric.py:13) does NOT rank them by Hamming distance, so the
    drop-in MAPs must not either (it once did).  Golden = the unmodified reference."""
    from hashgan_amd import MAPs, MAP
    c = cases.build_real_case(name)
    g = cases.load_golden(name)
    database = types.SimpleNamespace(output=c["dbf"], label=c["dblab"].astype(np.int64))
    query = types.SimpleNamespace(output=c["qf"], label=c["qlab"].astype(np.int64))
    m = MAPs(c["R"])
    assert m.get_maps_by_feature(database, query) == g["map"]
    m.close()
    if name == "real_bits01":       # the north-star spelling takes {0,1} as BITS: Hamming ranking, a different number
        from oracle import hamming_map as O
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mh, *_ = O.map_from_codes(c["qf"].astype(np.uint8), c["dbf"].astype(np.uint8), c["qlab"], c["dblab"], c["R"])
        assert MAP(c["qf"], c["dbf"], c["qlab"], c["dblab"], c["R"]) == mh != g["map"]
    else:                           # a ternary code is not a binary code
        with pytest.raises(ValueError):
            MAP(c["qf"], c["dbf"], c["qlab"], c["dblab"], c["R"])
        with pytest.raises(ValueError):   # spellings must agree between queries and database
            MAP((c["qf"] > 0).astype(np.float32), np.where(c["dbf"] > 0, 1.0, -1.0).astype(np.float32), c["qlab"], c["dblab"], c["R"])


def _adversarial(kind, rng, Q, N, b):
    if kind == "tanh":
        return np.tanh(rng.standard_normal((N, b))), np.tanh(rng.standard_normal((Q, b)))
    if kind == "wide":          # magnitudes over six decades, one dominant direction: many scores crowd the cut
        d = rng.standard_normal((N, b)) * 10.0 ** rng.integers(-3, 4, (N, 1))
        d[:, 0] += 50.0
        q = rng.standard_normal((Q, b)) * 10.0 ** rng.integers(-3, 4, (Q, 1))
        q[:, 0] += 50.0
        return d, q
    if kind == "bits":          # {0,1} features: integer scores, ties by the thousand, the cut falls inside a tie group
        return (rng.random((N, b)) < 0.5).astype(np.float64), (rng.random((Q, b)) < 0.5).astype(np.float64)
    if kind == "tiny":          # magnitudes around and below half's smallest normal (6.1e-5): half's absolute error floor would swamp the margin -- bfloat16 for this database
        d = rng.standard_normal((N, b)) * 10.0 ** rng.uniform(-7, -3, (N, 1))
        q = rng.standard_normal((Q, b)) * 10.0 ** rng.uniform(-6, -2, (Q, 1))
        return d, q
    if kind == "huge":          # database features beyond half's range: the filter must take bfloat16 for this database
        d = rng.standard_normal((N, b)) * 10.0 ** rng.integers(0, 7, (N, 1))
        return d, rng.standard_normal((Q, b))
    if kind == "wildq":         # an ordinary database, but every third query has a feature half cannot hold: it keeps every row (the slices overflow: deeper attempts)
        q = np.tanh(rng.standard_normal((Q, b)))
        q[::3, 5] = 1.0e5
        return np.tanh(rng.standard_normal((N, b))), q
    if kind == "dups":          # a few distinct rows repeated all over the database
        base = np.tanh(rng.standard_normal((37, b)))
        return base[rng.integers(0, 37, N)], np.tanh(rng.standard_normal((Q, b)))
    raise ValueError(kind)


@pytest.mark.parametrize("kind,Q,N,b,R", [("tanh", 70, 80000, 64, 3000), ("wide", 40, 70000, 48, 2000), ("bits", 33, 90000, 32, 5000),
                                          ("dups", 20, 66000, 16, 1500), ("tanh", 12, 131072, 128, 6000),
                                          ("tiny", 50, 80000, 64, 3000), ("huge", 30, 70000, 64, 2000), ("wildq", 30, 70000, 32, 1000)])
def test_filter_and_rescore_equals_the_exact_passes(kind, Q, N, b, R, ctx):
    """The sampled bet at sizes where it applies (N >= 65536, R <= N / 8): the bfloat16 filter + exact rescoring
    (hg_real_bf.hpp, real_mfma = 2), the float32 matrix-core pass (1) and the vector-ALU pass (0) must all deliver
    the oracle's ranked lists, scores and AP bit for bit -- whatever the features look like; so must the
    global-memory ranking passes (real_sort_lds = 0) that take over when a query's records exceed the LDS."""
    rng = np.random.default_rng(len(kind) * 1000 + b)
    d, q = _adversarial(kind, rng, Q, N, b)
    dbf, qf = d.astype(np.float32), q.astype(np.float32)
    C = 9
    dl = (rng.random((N, C)) < 0.2).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.2).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, ap_ref, idx_ref, score_ref = RM.map_from_features(qf, dbf, ql, dl, R)
    ctx.set_database_f32(dbf, dl.astype(np.int64))
    ctx.set_queries_f32(qf, ql.astype(np.int64))
    try:
        for mode, lds, half_sample, second in ((2, 1, 1, 1), (2, 1, 1, 0), (2, 1, 0, 1), (2, 0, 1, 1), (1, 1, 1, 1), (0, 1, 1, 1)):
            ctx.set_option("real_mfma", mode)
            ctx.set_option("real_sort_lds", lds)
            ctx.set_option("real_sample_half", half_sample)      # the cut from 16-bit sample scores (round 6) or from exact chains: the same lists
            ctx.set_option("real_second_sample", second)         # ... tightened by the second, counting sample, or not: the same lists
            idx, score = ctx.topr_real(R)
            assert np.array_equal(score.view(np.uint32), score_ref.view(np.uint32)), (kind, mode, lds, half_sample, second)
            assert np.array_equal(idx, idx_ref), (kind, mode, lds, half_sample, second)
            if kind not in ("bits", "wildq"):   # (a cut inside a tie group of thousands overflows the slices: deeper attempts, same lists)
                assert ctx.get_stat("real_attempts") == 1, (kind, mode, lds, half_sample, second)
            assert (ctx.get_stat("real_path") & 1) == (1 if mode == 2 else 0)
            if mode == 2:               # the filter's format: IEEE half unless a database feature could overflow it
                assert ((ctx.get_stat("real_path") >> 3) & 1) == (0 if kind in ("huge", "tiny") else 1), kind
            if ctx.get_stat("real_attempts") == 1 and kind not in ("bits", "wildq"):   # (tie groups can exceed the LDS: the global passes take over)
                assert ((ctx.get_stat("real_path") >> 1) & 1) == (1 if mode == 2 and lds and R <= 6144 else 0)
            ap, rel = ctx.map_real(R)
            assert np.array_equal(ap, ap_ref, equal_nan=True), (kind, mode, lds, half_sample, second)
    finally:
        ctx.set_option("real_mfma", 2)
        ctx.set_option("real_sort_lds", 1)
        ctx.set_option("real_sample_half", 1)
        ctx.set_option("real_second_sample", 1)


def test_real_valued_ranking_vs_np_dot_envelope(ctx):
    """What main.py:164 really hands over (tanh outputs) against lib/metric.py:13-14 as written -- float32 np.dot (OpenBLAS)
    -> np.argsort: the ranked lists differ only where float32 rounding reorders near-equal products.  The bounds are the
    ones tests/test_oracle_real.py holds the oracle to (measured: 1e-4 of the positions, 2e-8 in mAP)."""
    from tests.test_oracle_real import _tanh_case, REAL_ORDER_MISMATCH_MAX, REAL_MAP_DELTA_MAX
    from oracle import hamming_map as H
    from hashgan_amd import MAPs
    qf, dbf, ql, dl = _tanh_case()
    R = 1000
    ctx.set_database_f32(dbf, dl)
    ctx.set_queries_f32(qf, ql)
    idx, score = ctx.topr_real(R)
    ref_idx = np.argsort(-np.dot(qf, dbf.T), 1)[:, :R]
    frac = float(np.mean(idx.astype(np.int64) != ref_idx))
    assert frac <= REAL_ORDER_MISMATCH_MAX, frac
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m_ref = H.reference_as_written(dbf, dl, qf, ql, R)
    m = MAPs(R).get_maps_by_feature(types.SimpleNamespace(output=dbf, label=dl), types.SimpleNamespace(output=qf, label=ql))
    assert abs(m - m_ref) <= REAL_MAP_DELTA_MAX, (m, m_ref)
