"""HIP path vs oracle vs golden fixtures, through the C ABI (needs an MI355X).

Stage by stage, so a failure names the kernel:
  hist   -> per-query distance histograms           (k_hist, k_hist_reduce)
  topr   -> ranked idx / dist lists, canonical order (k_plan .. k_order)
  match  -> label-match bits                          (k_match)
  ap     -> float64 AP, bit-exact                     (k_ap)
"""
import warnings
import numpy as np
import pytest
from tests import cases
from oracle import hamming_map as O
from hashgan_amd import _native, metric

pytestmark = pytest.mark.gpu

STAGED = cases.SMALL + ["e_big_r", "c3_nus_q64"]
BIG = ["c2_q64", "c2_iid_q64", "c5_b128_q32", "c4_n10m_q8"]


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


def _load(ctx, c):
    ctx.set_database(metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"]), c["b"], c["dblab"].shape[1])
    ctx.set_queries(metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"]))


def _first_diff(a, b):
    bad = np.argwhere(a != b)
    return "first mismatch at %s: got %s want %s (%d mismatches)" % (bad[0], a[tuple(bad[0])], b[tuple(bad[0])], len(bad))


def _oracle(c):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return O.map_from_codes(c["qbits"], c["dbbits"], c["qlab"], c["dblab"], c["R"])


@pytest.mark.parametrize("name", STAGED)
def test_stages_match_oracle(name, ctx, case_cache):
    c = case_cache(name)
    g = cases.load_golden(name)
    m_ref, ap_ref, imatch_ref, idx_ref, dist_ref = _oracle(c)
    _load(ctx, c)
    # hist
    ctx.hist()
    h = ctx.get_hist()                                              # [b+1][Q]
    D = O.hamming_matrix(O.pack_bits(c["qbits"]), O.pack_bits(c["dbbits"]))
    h_ref = np.stack([np.bincount(D[i], minlength=c["b"] + 1) for i in range(D.shape[0])], axis=1)
    assert np.array_equal(h, h_ref), "hist: " + _first_diff(h, h_ref)
    # plan + select
    ctx.plan(c["R"])
    ctx.select()
    idx, dist = ctx.get_topr()
    assert np.array_equal(dist, dist_ref), "dist: " + _first_diff(dist.astype(np.int64), dist_ref)
    assert np.array_equal(idx, idx_ref), "idx: " + _first_diff(idx.astype(np.int64), idx_ref)
    # match
    ctx.match()
    m = ctx.get_match()
    assert np.array_equal(m.astype(bool), imatch_ref), "match: " + _first_diff(m.astype(bool), imatch_ref)
    # ap
    ctx.ap()
    ap, rel = ctx.get_ap()
    assert np.array_equal(rel, imatch_ref.sum(1))
    assert np.array_equal(ap, ap_ref, equal_nan=True), "ap: " + _first_diff(ap, ap_ref)
    assert np.array_equal(ap, g["ap"], equal_nan=True), "ap vs golden"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mm = metric.mean_over_hits(ap, rel)
    assert (np.isnan(mm) and np.isnan(g["map"])) or mm == g["map"]


@pytest.mark.parametrize("name", BIG)
def test_big_cases_match_golden(name, ctx, case_cache):
    c = case_cache(name)
    g = cases.load_golden(name)
    _load(ctx, c)
    nq = min(8, c["qbits"].shape[0])
    idx_ref, dist_ref = O.topr_from_codes(c["qbits"][:nq], c["dbbits"], c["R"])
    for optimistic in (1, 0):          # sampled-threshold bet (verified on device) and the exact two-pass path
        ctx.set_option("optimistic", optimistic)
        runs0 = ctx.get_stat("optimistic_runs")
        ap, rel = ctx.map(c["R"])
        assert ctx.get_stat("optimistic_runs") - runs0 == optimistic
        assert np.array_equal(ap, g["ap"], equal_nan=True), "ap vs golden: " + _first_diff(ap, g["ap"])
        assert metric.mean_over_hits(ap, rel) == g["map"]
        with pytest.raises(_native.HashganNativeError):
            ctx.get_topr()             # hg_map does not materialise the lists
        # ranked lists against the oracle on the first queries
        ctx.topr(c["R"])
        idx, dist = ctx.get_topr()
        assert np.array_equal(idx[:nq], idx_ref) and np.array_equal(dist[:nq], dist_ref)
    ctx.set_option("optimistic", 1)


def test_c1_cifar_full_golden(ctx, case_cache):
    """BASELINE config C1 at full size (Q=1000, N=54000, b=32, R=N) against the
    unmodified reference's per-query AP and mAP."""
    c = case_cache("c1_cifar_full")
    g = cases.load_golden("c1_cifar_full")
    _load(ctx, c)
    ap, rel = ctx.map(c["R"])
    assert np.array_equal(ap, g["ap"], equal_nan=True), _first_diff(ap, g["ap"])
    assert metric.mean_over_hits(ap, rel) == g["map"]


def test_python_surface_matches_golden(case_cache):
    """MAPs(R).get_maps_by_feature(database, query) and MAP(...) -- the drop-in calls."""
    import types
    from hashgan_amd import MAPs, MAP, calc_map
    for name in ["e_ragged", "e_b100", "e_some_skipped"]:
        c = case_cache(name)
        g = cases.load_golden(name)
        database = types.SimpleNamespace(output=c["dbbits"].astype(np.float32) * 2 - 1, label=c["dblab"].astype(np.int64))
        query = types.SimpleNamespace(output=c["qbits"].astype(np.float32) * 2 - 1, label=c["qlab"].astype(np.int64))
        assert MAPs(c["R"]).get_maps_by_feature(database, query) == g["map"]
        assert MAP(c["qbits"], c["dbbits"], c["qlab"], c["dblab"], c["R"]) == g["map"]
        assert calc_map is MAP
    c = case_cache("e_all_skipped")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.isnan(MAP(c["qbits"], c["dbbits"], c["qlab"], c["dblab"], c["R"]))
    with pytest.raises(ValueError):
        MAP(c["qbits"], c["dbbits"], c["qlab"], c["dblab"], c["dbbits"].shape[0] + 1)


def test_a_new_maps_object_per_evaluation_recycles_one_context(case_cache):
    """main.py:164 as written -- `MAPs(R).get_maps_by_feature(database, query)`, a NEW object per evaluation, never closed: fifty of
    them, alternating between two shapes and between +-1 codes and real-valued features, all draw the same pooled context (one
    created, the rest recycled), every result equals the golden / the first call's, and the device memory the pool holds stops
    growing after the first round of shapes (buffers only grow: what the largest evaluation needed is the floor)."""
    import gc
    import types
    from hashgan_amd import MAPs, pool_stats, release_engines
    ca, cb = case_cache("e_ragged"), case_cache("e_b100")
    ga, gb = cases.load_golden("e_ragged"), cases.load_golden("e_b100")

    def ns(c, k, real=False):
        out = c[k + "bits"].astype(np.float32) * 2 - 1
        if real:
            out = np.tanh(out * (1.0 + 0.25 * np.arange(out.shape[1], dtype=np.float32)))
        return types.SimpleNamespace(output=out, label=c[k + "lab"].astype(np.int64))
    release_engines()
    gc.collect()
    s0 = pool_stats()
    assert s0["contexts_idle"] == 0
    jobs = [(ca["R"], ns(ca, "db"), ns(ca, "q"), ga["map"]), (cb["R"], ns(cb, "db"), ns(cb, "q"), gb["map"]),
            (ca["R"], ns(ca, "db", True), ns(ca, "q", True), None)]
    floor = None
    first_real = None
    for i in range(51):
        R, db, q, want = jobs[i % 3]
        got = MAPs(R).get_maps_by_feature(db, q)           # the object dies with the statement
        if want is None:
            first_real = got if first_real is None else first_real
            want = first_real
        assert got == want, (i, got, want)
        st = pool_stats()
        assert st["contexts_idle"] == 1, st
        if i == 5:
            floor = st["idle_device_bytes"]
        if i > 5:                                          # (a few KB move with the result block's views: err / ap / rel are cut anew per Q)
            assert abs(st["idle_device_bytes"] - floor) < 65536, (i, st, floor)
    st = pool_stats()
    assert st["contexts_created"] - s0["contexts_created"] == 1 and st["contexts_recycled"] - s0["contexts_recycled"] == 50
    # a borrower that changed engine options does not hand its context back; one alive keeps its context to itself
    m = MAPs(ca["R"])
    m.get_maps_by_feature(jobs[0][1], jobs[0][2])
    m2 = MAPs(ca["R"])
    assert m2.get_maps_by_feature(jobs[0][1], jobs[0][2]) == ga["map"] and m2._eng is not m._eng
    m._eng.ctx.set_option("guess_sigma", 7)
    closed0 = pool_stats()["contexts_closed"]
    m.close()
    assert pool_stats()["contexts_closed"] == closed0 + 1
    m2.close()
    assert pool_stats()["contexts_idle"] == 1
    release_engines()
    assert pool_stats()["contexts_idle"] == 0 and pool_stats()["idle_device_bytes"] == 0


def test_block_cache_serves_new_contexts(case_cache):
    """hg_init / hg_destroy as a caller without the Python pool would use them (a context per evaluation): after the first context
    has come and gone, a new one takes its device blocks, pinned blocks and stream from the library's process-wide cache -- no
    hipMalloc, no hipHostMalloc, no hipStreamCreate (host-phase counters) -- and hg_release_cache hands everything back."""
    c = case_cache("c2_q64")
    g = cases.load_golden("c2_q64")

    def cycle():
        ctx = _native.Context(0)
        try:
            _load(ctx, c)
            ap, _ = ctx.map(c["R"])
            assert np.array_equal(ap, g["ap"], equal_nan=True)
            return {k: ctx.get_stat(k) for k in ("host_n_devmalloc", "host_n_hostmalloc", "host_n_stream", "cache_hits", "cache_misses")}
        finally:
            ctx.close()
    cycle()                                                # (whatever earlier tests left in the cache: this cycle completes it)
    first = cycle()
    for _ in range(4):
        st = cycle()
    for k in ("host_n_devmalloc", "host_n_hostmalloc", "host_n_stream", "cache_misses"):
        assert st[k] == first[k], (k, first, st)
    assert st["cache_hits"] > first["cache_hits"]
    probe = _native.Context(0)
    try:
        import os
        if not os.environ.get("HG_EFENCE"):                # (fenced debugging allocations bypass the cache; pinned blocks and streams do not)
            assert probe.get_stat("cache_device_bytes") > 0
        assert probe.get_stat("cache_pinned_bytes") > 0          # (the probe itself holds the cached stream)
        _native.release_cache()
        assert probe.get_stat("cache_device_bytes") == 0 and probe.get_stat("cache_pinned_bytes") == 0 and probe.get_stat("cache_streams") == 0
    finally:
        probe.close()


def test_maps_objects_are_independent_and_keep_a_resident_database(case_cache):
    """Two MAPs objects (own contexts) interleaved, from two threads; set_database / read-only arrays skip the
    re-upload (main.py:237-240 evaluates the same database again and again)."""
    import threading
    import types
    from hashgan_amd import MAPs
    ca, cb = case_cache("e_ragged"), case_cache("e_b100")
    ga, gb = cases.load_golden("e_ragged"), cases.load_golden("e_b100")

    def ns(c, k):
        return types.SimpleNamespace(output=c[k + "bits"].astype(np.float32) * 2 - 1, label=c[k + "lab"].astype(np.int64))
    ma, mb = MAPs(ca["R"]), MAPs(cb["R"])
    dba, dbb, qa, qb = ns(ca, "db"), ns(cb, "db"), ns(ca, "q"), ns(cb, "q")
    out = {}

    def work(key, m, db, q, n):
        out[key] = [m.get_maps_by_feature(db, q) for _ in range(n)]
    th = [threading.Thread(target=work, args=("a", ma, dba, qa, 3)), threading.Thread(target=work, args=("b", mb, dbb, qb, 3))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert out["a"] == [ga["map"]] * 3 and out["b"] == [gb["map"]] * 3
    # explicit resident database
    ma.set_database(dba)
    bytes_before = ma._eng.ctx.get_stat("device_bytes")
    assert ma.get_maps_by_feature(None, qa) == ga["map"]
    assert ma.get_maps_by_feature(dba, qa) == ga["map"]            # the same object: no re-upload either
    assert ma._eng.ctx.get_stat("device_bytes") == bytes_before
    # automatic reuse needs read-only arrays; a writable one is uploaded again (it may have changed in place)
    dbb.output.flags.writeable = False
    dbb.label.flags.writeable = False
    assert mb.get_maps_by_feature(dbb, qb) == gb["map"] and mb._resident[0] == "auto"
    held = mb._resident
    assert mb.get_maps_by_feature(dbb, qb) == gb["map"] and mb._resident is held
    with pytest.raises(ValueError):
        MAPs(5).get_maps_by_feature(None, qa)
    ma.close(); mb.close()


def test_matrix_core_histogram_is_exact(case_cache):
    """k_hist_mx (fp4 MFMA distances -> LDS counters, per segment pair): the full pass of the one-shot exact sequence must
    produce the oracle's histogram -- including ragged segment ends, an unpaired last segment and both counter layouts
    (16-bit halves shared by the two query tiles / one dword per tile for long segments)."""
    c = case_cache("c2_q64")
    Dref = None
    ctx = _native.Context(0)
    try:
        _load(ctx, c)
        ctx.set_option("optimistic", 0)
        g = cases.load_golden("c2_q64")
        for kernel in (2, 1):                                               # k_hist_i8 (addresses from the integer MFMA), k_hist_mx (fp4)
            ctx.set_option("hist_mfma", kernel)
            for units, maxseg in [(16384, 2048), (3, 2048), (16384, 7)]:    # S = 2048-ish, 3 long segments (dword counters), 7
                ctx.set_option("target_units", units)
                ctx.set_option("max_segments", maxseg)
                ap, rel = ctx.map(c["R"])
                assert np.array_equal(ap, g["ap"], equal_nan=True), (kernel, units, maxseg)
                h = ctx.get_hist().astype(np.int64)                  # [b + 1][Q] from the pass that just ran
                if Dref is None:
                    D = O.hamming_matrix(O.pack_bits(c["qbits"]), O.pack_bits(c["dbbits"]))
                    Dref = np.stack([np.bincount(D[i], minlength=c["b"] + 1) for i in range(D.shape[0])]).T
                assert np.array_equal(h, Dref), (kernel, units, maxseg)
                assert ctx.get_stat("optimistic_fallbacks") == 0
    finally:
        ctx.close()


@pytest.mark.parametrize("b", [1, 8, 17, 31, 32, 33, 48, 63, 65, 96, 100, 127, 128])
def test_histogram_kernels_agree_on_odd_code_lengths(b):
    """The matrix-core histogram kernels against the oracle for code lengths around the word boundaries (k_hist_i8 zeroes
    the B operand beyond the code; one MFMA per 32 bits), ragged N, exact sequence."""
    rng = np.random.default_rng(b)
    Q, N, R = 70, 5003, 400
    qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    db = (rng.random((N, b)) < 0.5).astype(np.uint8)
    lab = (rng.random((N, 5)) < 0.3).astype(np.uint8)
    ql = (rng.random((Q, 5)) < 0.3).astype(np.uint8)
    D = O.hamming_matrix(O.pack_bits(qb), O.pack_bits(db))
    Dref = np.stack([np.bincount(D[i], minlength=b + 1) for i in range(Q)]).T
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=lab, b=b))
        ctx.set_option("optimistic", 0)
        for kernel in (2, 1, 0):
            ctx.set_option("hist_mfma", kernel)
            ctx.map(R)
            assert np.array_equal(ctx.get_hist().astype(np.int64), Dref), (b, kernel)
    finally:
        ctx.close()


def test_bursts_of_hits_take_the_direct_route():
    """Near-duplicates stored next to each other: a query's hits arrive ten and more to a 64-row half window, more than a
    slice's 16-record ring takes between two flushes -- the compact-record drain must route such words straight to
    global memory (hg_mx_drain.hpp) and still deliver every record, in order.  Checked against the 8-byte-record
    kernels and the oracle."""
    from hashgan_amd import synth
    Q, N0, b, R, C, dup = 192, 8192, 64, 6000, 10, 24
    dl0, _ = synth.onehot_labels(51, N0, C)
    ql, _ = synth.onehot_labels(52, Q, C)
    db0 = synth.planted_codes(53, dl0, b, 0.3)
    qb = synth.planted_codes(53, ql, b, 0.3)
    # every row followed by dup - 1 copies that differ in at most one bit: N = 196608 rows, bursts of 24 near-equal rows
    db = np.repeat(db0, dup, axis=0)
    dl = np.repeat(dl0, dup, axis=0)
    flip = synth.splitmix64(54, db.shape[0]) % np.uint64(2 * b)
    rows = np.nonzero(flip < b)[0]
    db[rows, flip[rows].astype(np.int64)] ^= 1
    c = dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, R=R, b=b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb[:16], db, ql[:16], dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, c)
        ctx.set_option("step_graph", 0)
        out = {}
        for compact in (1, 0):
            ctx.set_option("compact_records", compact)
            ap, rel = ctx.map(R)
            out[compact] = ap
            assert np.array_equal(ap[:16], ap_ref, equal_nan=True), compact
        assert np.array_equal(out[0], out[1], equal_nan=True)
        assert ctx.get_stat("optimistic_runs") == 2 and ctx.get_stat("optimistic_fallbacks") == 0
        # long codes: k_select_mx4 (two rows per accumulator) feeding its copy of the drain
        rng = np.random.default_rng(55)
        ext = (rng.random((N0, 64)) < 0.5).astype(np.uint8)              # 64 more bits per distinct row, shared by its near-copies
        qext = (rng.random((Q, 64)) < 0.5).astype(np.uint8)
        c128 = dict(c, qbits=np.concatenate([qb, qext], 1), dbbits=np.concatenate([db, np.repeat(ext, dup, axis=0)], 1), b=128)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, ap_ref128, *_ = O.map_from_codes(c128["qbits"][:8], c128["dbbits"], ql[:8], dl, R)
        _load(ctx, c128)
        res = []
        for compact in (1, 0):
            ctx.set_option("compact_records", compact)
            res.append(ctx.map(R)[0])
            assert np.array_equal(res[-1][:8], ap_ref128, equal_nan=True), compact
            if compact:
                assert ctx.get_stat("select_variant") == 6
        assert np.array_equal(res[0], res[1], equal_nan=True)
        # and with short codes (the second k-half of every MFMA empty)
        c32 = dict(c, qbits=qb[:, :32].copy(), dbbits=db[:, :32].copy(), b=32)
        _load(ctx, c32)
        res = []
        for compact in (1, 0):
            ctx.set_option("compact_records", compact)
            res.append(ctx.map(R)[0])
        assert np.array_equal(res[0], res[1], equal_nan=True)
    finally:
        ctx.close()


def test_step_graph_replays_equal_eager_steps(case_cache):
    """hg_map captures its one-shot bet into a hipGraph on the second identical call and replays it afterwards:
    same AP bit for bit, with and without kernel timing, across a change of R and of the queries."""
    c = case_cache("c2_q64")
    g = cases.load_golden("c2_q64")
    ctx = _native.Context(0)
    try:
        _load(ctx, c)
        ctx.set_option("step_graph", 1)                  # opt-in (off by default)
        for timing in (0, 2, 1):
            ctx.timing_enable(timing)
            ctx.timing_reset()
            before = ctx.get_stat("graph_replays")
            for _ in range(4):
                ap, rel = ctx.map(c["R"])
                assert np.array_equal(ap, g["ap"], equal_nan=True), timing
            if timing:                                   # kernel timing keeps the steps eager (events per launch)
                assert ctx.get_stat("graph_replays") == before
                t = ctx.timing_read()
                assert t["k_select_mx"][1] == 4 and t["step_gpu_span"][1] == 4, t
                assert 0 < t["k_select_mx"][0] < t["step_gpu_span"][0], t
            else:
                assert ctx.get_stat("graph_replays") - before >= 2
        ctx.timing_enable(0)
        ctx.set_option("step_graph", 0)
        ap_eager, _ = ctx.map(c["R"] // 2)
        ctx.set_option("step_graph", 1)
        r0 = ctx.get_stat("graph_replays")
        for _ in range(3):
            ap2, _ = ctx.map(c["R"] // 2)                      # another R: a new capture, not the old graph
            assert np.array_equal(ap2, ap_eager, equal_nan=True)
        assert ctx.get_stat("graph_replays") > r0
        ctx.set_queries(metric.pack_codes(c["qbits"][::-1].copy()), metric.pack_labels(c["qlab"][::-1].copy()))
        for _ in range(3):
            ap3, _ = ctx.map(c["R"])
            assert np.array_equal(ap3, g["ap"][::-1], equal_nan=True)
    finally:
        ctx.close()


def test_map_in_two_halves_equals_map(case_cache):
    """hg_map_begin / hg_map_end: the same AP and hit counts as hg_map, bit for bit -- the first begin runs the call itself, the
    following ones enqueue blind with two steps in flight; a third begin is refused; a new query table of the same size keeps
    the licence to enqueue blind (batch after batch), one of another size makes the next begin synchronous again.  (A blind
    step that loses its bet: test_a_blind_step_that_loses_is_redone_on_its_own_tables.)"""
    c = case_cache("c2_q64")
    g = cases.load_golden("c2_q64")
    ctx = _native.Context(0)
    try:
        _load(ctx, c)
        R = c["R"]
        ap0, rel0 = ctx.map(R)
        assert np.array_equal(ap0, g["ap"], equal_nan=True)
        ctx.map_begin(R)
        ctx.map_begin(R)
        with pytest.raises(_native.HashganNativeError):
            ctx.map_begin(R)
        for k in range(6):
            ap, rel = ctx.map_end()
            assert np.array_equal(ap, ap0, equal_nan=True) and np.array_equal(rel, rel0), k
            ctx.map_begin(R)
        a1, r1 = ctx.map_end()
        a2, r2 = ctx.map_end()
        assert np.array_equal(a1, ap0, equal_nan=True) and np.array_equal(a2, ap0, equal_nan=True) and np.array_equal(r2, rel0)
        with pytest.raises(_native.HashganNativeError):
            ctx.map_end()
        if ctx.get_stat("last_optimistic"):
            assert ctx.get_stat("map_async_steps") >= 6 and ctx.get_stat("map_async_redone") == 0
        # other queries, as many as before: the next begin is blind too (a caller that hands over batch after batch keeps two
        # steps in flight) -- and the batch may be replaced while the step on the previous one is still in flight
        warm = bool(ctx.get_stat("last_optimistic"))
        ctx.set_queries(metric.pack_codes(c["qbits"][::-1].copy()), metric.pack_labels(c["qlab"][::-1].copy()))
        n0 = ctx.get_stat("map_async_steps")
        ctx.map_begin(R)
        assert ctx.get_stat("map_async_steps") == n0 + (1 if warm else 0)
        ctx.set_queries(metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"]))
        ctx.map_begin(R)
        ap, rel = ctx.map_end()
        assert np.array_equal(ap, g["ap"][::-1], equal_nan=True)
        ctx.set_queries(metric.pack_codes(c["qbits"][::-1].copy()), metric.pack_labels(c["qlab"][::-1].copy()))
        ctx.map_begin(R)
        ap, rel = ctx.map_end()
        assert np.array_equal(ap, g["ap"], equal_nan=True) and np.array_equal(rel, rel0)
        ap, rel = ctx.map_end()
        assert np.array_equal(ap, g["ap"][::-1], equal_nan=True)
        assert ctx.get_stat("map_async_redone") == 0
        # a table of another size: nothing is known about buffers of that size -- the next begin runs the call itself
        ctx.set_queries(metric.pack_codes(c["qbits"][:50].copy()), metric.pack_labels(c["qlab"][:50].copy()))
        n0 = ctx.get_stat("map_async_steps")
        ctx.map_begin(R)
        assert ctx.get_stat("map_async_steps") == n0
        ap, rel = ctx.map_end()
        assert np.array_equal(ap, g["ap"][:50], equal_nan=True)
        ctx.set_queries(metric.pack_codes(c["qbits"][::-1].copy()), metric.pack_labels(c["qlab"][::-1].copy()))
        ap, rel = ctx.map(R)
        # a step in flight while the queries are replaced by fewer: its results are its own (and of its own length)
        ctx.map_begin(R)
        ctx.set_queries(metric.pack_codes(c["qbits"][:40].copy()), metric.pack_labels(c["qlab"][:40].copy()))
        ap, rel = ctx.map_end()
        assert ap.shape[0] == c["qbits"].shape[0] and np.array_equal(ap, g["ap"][::-1], equal_nan=True)
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, g["ap"][:40], equal_nan=True)
        ctx.set_queries(metric.pack_codes(c["qbits"][::-1].copy()), metric.pack_labels(c["qlab"][::-1].copy()))
        # another R in flight next to the first
        ap_half, rel_half = ctx.map(R // 2)
        ctx.map_begin(R // 2)
        ctx.map_begin(R)
        a, r = ctx.map_end()
        assert np.array_equal(a, ap_half, equal_nan=True) and np.array_equal(r, rel_half)
        a, r = ctx.map_end()
        assert np.array_equal(a, g["ap"][::-1], equal_nan=True)
    finally:
        ctx.close()


def test_a_blind_step_that_loses_is_redone_on_its_own_tables(case_cache):
    """The branch of hg_map_end that identical inputs never reach: a step enqueued blind (hg_map_begin after a won hg_map) LOSES its
    bet -- forced by the test hook "handicap_next_bet", which puts the next guess far below the expected count without touching the
    configuration -- and hg_map_end runs it again on the same tables: stat map_async_redone == 1, results == the reference's golden.
    If the queries (same count!) or the database were replaced in between, the lost step's tables are gone: HG_ERR_STATE, never the
    new batch's results under the old step's name (include/hashgan_amd.h, hg_map_begin)."""
    c = case_cache("c2_q64")
    g = cases.load_golden("c2_q64")
    ctx = _native.Context(0)
    try:
        _load(ctx, c)
        R = c["R"]
        ap0, rel0 = ctx.map(R)
        assert np.array_equal(ap0, g["ap"], equal_nan=True)
        if not ctx.get_stat("last_optimistic"):
            pytest.skip("the bet does not apply to this shape")
        # 1. lost, tables untouched: redone
        ctx.set_option("handicap_next_bet", 12)
        n0 = ctx.get_stat("map_async_steps")
        ctx.map_begin(R)
        assert ctx.get_stat("map_async_steps") == n0 + 1, "the hook must not end the licence to enqueue blind"
        ap, rel = ctx.map_end()
        assert ctx.get_stat("map_async_redone") == 1
        assert np.array_equal(ap, g["ap"], equal_nan=True) and np.array_equal(rel, rel0)
        # (the redo was a synchronous hg_map that won: the next begin is blind again, and wins)
        ctx.map_begin(R)
        assert ctx.get_stat("map_async_steps") == n0 + 2
        ap, rel = ctx.map_end()
        assert ctx.get_stat("map_async_redone") == 1 and np.array_equal(ap, g["ap"], equal_nan=True)
        # 2. lost, and the queries replaced by ANOTHER table of the same count before the second half
        ctx.set_option("handicap_next_bet", 12)
        ctx.map_begin(R)
        ctx.set_queries(metric.pack_codes(c["qbits"][::-1].copy()), metric.pack_labels(c["qlab"][::-1].copy()))
        with pytest.raises(_native.HashganNativeError) as ei:
            ctx.map_end()
        assert ei.value.code == _native.HG_ERR_STATE and "queries were replaced" in str(ei.value)
        assert ctx.get_stat("map_async_redone") == 1
        ap, rel = ctx.map(R)                               # the context is fine: the tables it holds now
        assert np.array_equal(ap, g["ap"][::-1], equal_nan=True)
        # 3. lost, and the database reloaded
        ctx.set_option("handicap_next_bet", 12)
        ctx.map_begin(R)
        _load(ctx, c)
        with pytest.raises(_native.HashganNativeError) as ei:
            ctx.map_end()
        assert ei.value.code == _native.HG_ERR_STATE and "database was replaced" in str(ei.value)
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, g["ap"], equal_nan=True)
        # 4. a lost step with a younger one behind it: the redo answers the old step, the younger keeps its own block
        ctx.set_option("handicap_next_bet", 12)
        ctx.map_begin(R)
        ctx.map_begin(R)
        a1, r1 = ctx.map_end()
        a2, r2 = ctx.map_end()
        assert np.array_equal(a1, g["ap"], equal_nan=True) and np.array_equal(a2, g["ap"], equal_nan=True) and np.array_equal(r2, rel0)
        assert ctx.get_stat("map_async_redone") == 2
    finally:
        ctx.close()


def test_batch_after_batch_keeps_two_steps_in_flight(case_cache):
    """What a caller with a fresh query batch per step does (lib/metric.py once per batch): hg_set_queries -- which no longer waits
    for the stream -- then hg_map_begin, the previous batch's hg_map_end afterwards.  Every step after the first is enqueued blind
    and every batch's APs equal the golden's rows of that batch."""
    c = case_cache("c2_q64")
    g = cases.load_golden("c2_q64")
    Q = c["qbits"].shape[0]
    rng = np.random.default_rng(11)
    perms = [rng.permutation(Q) for _ in range(7)]
    packed = [(metric.pack_codes(c["qbits"][p]), metric.pack_labels(c["qlab"][p])) for p in perms]
    ctx = _native.Context(0)
    try:
        _load(ctx, c)
        R = c["R"]
        ctx.map(R)
        if not ctx.get_stat("last_optimistic"):
            pytest.skip("the bet does not apply to this shape")
        n0 = ctx.get_stat("map_async_steps")
        ctx.set_queries(*packed[0])
        ctx.map_begin(R)
        for i in range(1, len(perms)):
            ctx.set_queries(*packed[i])                    # while step i - 1 is in flight
            ctx.map_begin(R)
            ap, rel = ctx.map_end()
            assert np.array_equal(ap, g["ap"][perms[i - 1]], equal_nan=True), i
        ap, rel = ctx.map_end()
        assert np.array_equal(ap, g["ap"][perms[-1]], equal_nan=True)
        assert ctx.get_stat("map_async_steps") == n0 + len(perms) and ctx.get_stat("map_async_redone") == 0
    finally:
        ctx.close()


def test_device_pack_matches_host_pack(ctx):
    """The two ways float32 features / int64 labels become packed tables -- a pool of host threads before the upload
    (hg_host_pack.hpp, the default) and k_pack_sign_f32 / k_pack_labels_i64 on the GPU -- against the NumPy packing, bit
    for bit, including sign(0) -> 0, pad bits, odd word counts; and the census (non-binary entries, zeros, minus ones)."""
    rng = np.random.default_rng(5)
    try:
        for host_pack in (1, 0):
            ctx.set_option("host_pack", host_pack)
            for b, C in [(1, 1), (31, 10), (32, 64), (33, 65), (48, 81), (64, 10), (65, 128), (100, 3), (128, 200), (255, 7)]:
                n = 777
                x = rng.choice(np.array([-1.0, 1.0], np.float32), size=(n, b))
                lab = (rng.random((n, C)) < 0.2).astype(np.int64)
                bad = ctx.set_database_f32(x, lab)
                assert bad == (0, 0)
                assert ctx.census(0)[1] == 0 and ctx.census(0)[2] == int((x == -1).sum())
                assert ctx.census(0)[3] == (False if host_pack else True)          # a +-1 code: its floats stay on the host
                codes, labels = ctx.get_packed(0)
                ref = metric.pack_codes(x).view(np.uint32).reshape(n, -1)[:, :(b + 31) // 32]
                assert np.array_equal(codes, ref), (b, C, host_pack)
                assert np.array_equal(labels, metric.pack_labels(lab)), (b, C, host_pack)
                x01 = (x > 0).astype(np.float32)                       # {0,1} spelling packs to the same words
                assert ctx.set_database_f32(x01, lab) == (0, 0)
                assert np.array_equal(ctx.get_packed(0)[0], ref)
                assert ctx.census(0)[1] == int((x01 == 0).sum()) and ctx.census(0)[2] == 0
                assert ctx.census(0)[3] is True                                # not a +-1 code: np.dot would not rank it by Hamming distance
            x = rng.standard_normal((50, 16)).astype(np.float32)
            x[3, 3] = np.nan
            lab = np.zeros((50, 3), np.int64); lab[7, 1] = 2
            bad_c, bad_l = ctx.set_database_f32(x, lab)
            assert bad_c == 50 * 16 and bad_l == 1
            n = 300001                                                 # enough rows for several host threads, ragged split
            x = rng.choice(np.array([-1.0, 0.0, 1.0, 0.5], np.float32), size=(n, 48))
            lab = (rng.random((n, 10)) < 0.1).astype(np.int64)
            assert ctx.set_database_f32(x, lab) == (int((x == 0.5).sum()), 0)
            codes, labels = ctx.get_packed(0)
            assert np.array_equal(codes, metric.pack_codes(x).view(np.uint32).reshape(n, -1)[:, :2])
            assert np.array_equal(labels, metric.pack_labels(lab))
            assert ctx.census(0)[1] == int((x == 0).sum()) and ctx.census(0)[2] == int((x == -1).sum())
    finally:
        ctx.set_option("host_pack", 1)


def test_pm1_database_meets_real_valued_queries():
    """A +-1 database keeps its floats on the host; when queries turn out real-valued, MAPs must rank by inner product all
    the same (metric.py:13): the database's float table is then brought over."""
    import types
    from hashgan_amd import MAPs
    from oracle import real_map as RM
    rng = np.random.default_rng(11)
    N, Q, b, R, C = 5000, 30, 32, 700, 6
    dbf = rng.choice(np.array([-1.0, 1.0], np.float32), size=(N, b))
    qf = (np.round(np.tanh(rng.standard_normal((Q, b))) * 64) / 64).astype(np.float32)      # grid values: exact arithmetic
    dl = (rng.random((N, C)) < 0.3).astype(np.int64)
    ql = (rng.random((Q, C)) < 0.3).astype(np.int64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m_ref, *_ = RM.map_from_features(qf, dbf, ql, dl, R)
    m = MAPs(R)
    db = types.SimpleNamespace(output=dbf, label=dl)
    assert m.get_maps_by_feature(db, types.SimpleNamespace(output=np.where(qf > 0, 1.0, -1.0).astype(np.float32), label=ql)) is not None
    assert m._eng.ctx.census(0)[3] is False                     # +-1 on both sides: Hamming, no float table
    assert m.get_maps_by_feature(db, types.SimpleNamespace(output=qf, label=ql)) == m_ref
    assert m._eng.ctx.census(0)[3] is True
    m.close()


def test_python_surface_rejects_non_binary(case_cache):
    from hashgan_amd import MAP, MAPs
    import types
    c = case_cache("e_b8")
    soft = np.tanh((c["dbbits"].astype(np.float32) * 2 - 1) * 1.5)             # what HashGAN's heads emit
    qsoft = np.tanh((c["qbits"].astype(np.float32) * 2 - 1) * 1.5)
    with pytest.raises(ValueError):
        MAP(qsoft, soft, c["qlab"], c["dblab"], c["R"])
    with pytest.raises(ValueError):
        MAP(c["qbits"], c["dbbits"], c["qlab"] * 2, c["dblab"], c["R"])        # labels not {0,1}
    g = cases.load_golden("e_b8")
    database = types.SimpleNamespace(output=soft, label=c["dblab"].astype(np.int64))
    query = types.SimpleNamespace(output=qsoft, label=c["qlab"].astype(np.int64))
    assert MAPs(c["R"], binarize=True).get_maps_by_feature(database, query) == g["map"]


def test_deterministic_and_geometry_independent(ctx, case_cache):
    """Same lists whatever the segment geometry (unit count) and across repeats."""
    c = case_cache("e_ragged")
    _load(ctx, c)
    base = None
    for units, minseg in [(16384, 256), (7, 16), (100000, 16), (64, 4096), (16384, 256)]:
        ctx.set_option("target_units", units)
        ctx.set_option("min_segment", minseg)
        ctx.topr(c["R"])
        idx, dist = ctx.get_topr()
        if base is None:
            base = (idx.copy(), dist.copy())
        assert np.array_equal(idx, base[0]) and np.array_equal(dist, base[1]), (units, minseg)
    ctx.set_option("target_units", 16384)
    ctx.set_option("min_segment", 256)


def test_failed_bet_falls_back_to_exact(ctx):
    """The optimistic path is a bet, never an approximation: when the guessed threshold is too low
    (sigma 0, thin sample) or slices overflow (database sorted so that all near rows sit in one
    segment) the device flags the affected queries and they are rerun exactly -- all of them if
    many lost.  Results stay bit-exact either way."""
    from hashgan_amd import synth
    Q, N, b, R, C = 256, 131072, 32, 4000, 10
    dl, cls = synth.onehot_labels(71, N, C)
    ql, _ = synth.onehot_labels(72, Q, C)
    db = synth.planted_codes(73, dl, b, 0.2)
    qb = synth.planted_codes(73, ql, b, 0.2)
    order = np.argsort(cls, kind="stable")           # clustered by class: near rows are contiguous
    db, dl = db[order], dl[order]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m_ref, ap_ref, *_ = O.map_from_codes(qb[:32], db, ql[:32], dl, R)
    ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
    ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
    lost_some = 0
    for sigma in (6, 0):
        ctx.set_option("optimistic", 1)              # also clears the consecutive-failure latch
        ctx.set_option("guess_sigma", sigma)
        r0 = ctx.get_stat("optimistic_runs")
        f0, p0, b0 = ctx.get_stat("optimistic_fallbacks"), ctx.get_stat("optimistic_requeried"), ctx.get_stat("optimistic_rebets")
        for second in (1, 0):                        # with and without the second, wider bet before the exact sequence
            ctx.set_option("second_bet", second)
            ap, rel = ctx.map(R)
            assert np.array_equal(ap[:32], ap_ref, equal_nan=True), (sigma, second)
        assert ctx.get_stat("optimistic_runs") == r0 + 2
        lost_some += (ctx.get_stat("optimistic_fallbacks") - f0) + (ctx.get_stat("optimistic_requeried") - p0) + (
            ctx.get_stat("optimistic_rebets") - b0)
    assert lost_some > 0                             # the scenario really exercised a fallback
    ctx.set_option("second_bet", 1)
    ctx.set_option("guess_sigma", 6)
    ctx.set_option("optimistic", 1)


def test_single_lost_query_is_rerun_alone(ctx):
    """One query has 3000 exact duplicates of its code in one contiguous block of the database:
    its slices overflow there (and those of the few queries whose code is within their threshold
    of it); every other query's bet holds.  Only the lost queries are rerun (exactly) and patched
    in; AP and the ranked lists equal the oracle's for the lost query and for ordinary ones."""
    from hashgan_amd import synth
    Q, N, b, R, C = 256, 131072, 32, 4000, 10
    dl, _ = synth.onehot_labels(91, N, C)
    ql, _ = synth.onehot_labels(92, Q, C)
    db = synth.random_bits(93, N, b)
    qb = synth.random_bits(94, Q, b)
    db[50000:53000] = qb[5]
    probe = [4, 5, 6, 200]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, _, idx_ref, dist_ref = O.map_from_codes(qb[probe], db, ql[probe], dl, R)
    ctx.set_option("optimistic", 1)
    ctx.set_option("crowd_probe", 0)                 # (the first bet's crowding probe would widen the slices and nobody would lose)
    try:
        ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
        f0, p0 = ctx.get_stat("optimistic_fallbacks"), ctx.get_stat("optimistic_requeried")
        ap, rel = ctx.map(R)
        n_lost = ctx.get_stat("optimistic_requeried") - p0
        assert ctx.get_stat("optimistic_fallbacks") == f0 and 1 <= n_lost < Q // 8
        assert np.array_equal(ap[probe], ap_ref, equal_nan=True)
        ctx.topr(R)                                      # same thing with the lists materialised
        assert ctx.get_stat("optimistic_requeried") == p0 + 2 * n_lost
        idx, dist = ctx.get_topr()
        assert np.array_equal(idx[probe], idx_ref) and np.array_equal(dist[probe], dist_ref)
        ctx.ap()
        ap2, _ = ctx.get_ap()
        assert np.array_equal(ap2[probe], ap_ref, equal_nan=True)
    finally:
        ctx.set_option("crowd_probe", 1)


def test_trim_frees_work_buffers_and_keeps_tables(ctx, case_cache):
    c = case_cache("e_ragged")
    g = cases.load_golden("e_ragged")
    _load(ctx, c)
    ap, _ = ctx.map(c["R"])
    before = ctx.get_stat("device_bytes")
    ctx.trim()
    after = ctx.get_stat("device_bytes")
    assert after < before
    with pytest.raises(_native.HashganNativeError):
        ctx.get_ap()                       # results went with the buffers' stage
    ap2, _ = ctx.map(c["R"])               # tables are still loaded
    assert np.array_equal(ap2, g["ap"], equal_nan=True)


def test_state_and_argument_errors(ctx, case_cache):
    c = case_cache("e_b8")
    _load(ctx, c)
    with pytest.raises(_native.HashganNativeError) as e:
        ctx.plan(10)                       # before hist
    assert e.value.code == _native.HG_ERR_STATE
    ctx.hist()
    with pytest.raises(_native.HashganNativeError) as e:
        ctx.plan(c["dbbits"].shape[0] + 1)  # R > N
    assert e.value.code == _native.HG_ERR_ARG
    with pytest.raises(_native.HashganNativeError):
        ctx.set_option("no_such_option", 1)


def test_longest_code_and_largest_distance(ctx):
    """b = 255 (HG_MAX_BITS), with rows that are the exact complement of a query: distance 255, the largest a
    record or a ranked list has to hold; R = N ranks them (last).  A 256-bit code is refused: its complement
    would be 256 away, one more than the 8-bit distance fields carry."""
    rng = np.random.default_rng(3)
    b, N, Q, C = 255, 300, 5, 4
    db = rng.integers(0, 2, (N, b), dtype=np.uint8)
    qb = rng.integers(0, 2, (Q, b), dtype=np.uint8)
    db[7] = 1 - qb[0]
    db[150] = 1 - qb[3]
    dl = (rng.random((N, C)) < 0.4).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.4).astype(np.int8)
    ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
    ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
    for R in (N, N - 1, 50):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, ap_ref, _, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True), R
        ctx.topr(R)
        idx, dist = ctx.get_topr()
        assert np.array_equal(idx, idx_ref) and np.array_equal(dist, dist_ref), R
    wide = rng.integers(0, 2, (N, 256), dtype=np.uint8)
    with pytest.raises(_native.HashganNativeError) as e:
        ctx.set_database(metric.pack_codes(wide), metric.pack_labels(dl), 256, C)
    assert e.value.code == _native.HG_ERR_ARG


def test_queries_far_from_every_row_fall_back_exactly():
    """Queries on the other side of the code space (every distance > b/2): the cut lies beyond the distance planes the
    bet's sampled pass writes (hg_seq.hip::enqueue_optimistic), the guess reads a thin sample, the bet is lost and the
    exact sequence must answer -- same AP as the oracle."""
    rng = np.random.default_rng(77)
    Q, N, b, R, C = 70, 300000, 64, 2000, 6
    proto = (rng.random(b) < 0.5).astype(np.uint8)
    db = proto ^ (rng.random((N, b)) < 0.1).astype(np.uint8)
    qb = (1 - proto) ^ (rng.random((Q, b)) < 0.1).astype(np.uint8)
    dl = (rng.random((N, C)) < 0.3).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        assert ctx.get_stat("optimistic_runs") >= 1                  # the bet was placed ...
        assert ctx.get_stat("optimistic_fallbacks") + ctx.get_stat("optimistic_requeried") >= 1      # ... and lost
    finally:
        ctx.close()


def test_bet_equals_exact_over_random_shapes():
    """A slice of tools/fuzz_bet_vs_exact.py: random code lengths, sizes, hit densities (R/N 0.05 % .. 12 %), label widths
    (up to 130 classes: match bits gathered through the ranked list) and duplicated neighbours; the one-shot bet with
    compact and with 8-byte records and the matrix-core exact sequence must equal the vector-ALU exact sequence.
    Seeds 54..56 once ended in a memory fault: a query that lost its bet left stale words in its list row, and k_match
    followed them out of the label table."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_bet_vs_exact.py")
    spec = importlib.util.spec_from_file_location("fuzz_bet_vs_exact", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for seed in range(44, 64):
        r = fz.one(seed)
        assert r.startswith("ok"), r


def test_random_option_combinations_equal_the_exact_sequence():
    """A slice of tools/fuzz_options.py: random kernel choices, record formats, segment geometries and safety margins thin
    enough to lose bets, on random shapes; hg_map and hg_topr must equal the vector-ALU exact sequence.  Seed 796 once
    ended in a memory fault: the rerun of seven lost queries ranked its dense record runs with k_rank_fused, whose
    read-ahead went up to a chunk's capacity -- past the end of the last query's run, which is where the buffer ends."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_options.py")
    spec = importlib.util.spec_from_file_location("fuzz_options", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for seed in list(range(790, 800)) + [3, 17, 60]:
        r = fz.one(seed)
        assert r.startswith("ok"), r


def test_long_codes_in_long_segments_use_the_vector_histogram():
    """Codes of 200 bits in segments of > 32768 rows: the matrix-core histogram would need one dword counter per query tile
    and distance for four wavefronts -- more LDS than a CU has (the launch once failed with 'invalid argument');
    hist_mx_applies must hand such geometries to the vector kernel.  Bet and exact sequence against the oracle's AP."""
    rng = np.random.default_rng(200)
    Q, N, b, R, C = 40, 250000, 200, 300, 5
    db = (rng.random((N, b)) < 0.5).astype(np.uint8)
    qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    dl = (rng.random((N, C)) < 0.3).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        ctx.set_option("max_segments", 7)
        ctx.set_option("target_units", 64)
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        for optimistic in (1, 0):
            ctx.set_option("optimistic", optimistic)
            ap, rel = ctx.map(R)
            assert np.array_equal(ap, ap_ref, equal_nan=True), optimistic
    finally:
        ctx.close()


@pytest.mark.parametrize("R", [1, 7, 129, 5000, 8192, 8193, 20000])
def test_ap_through_reciprocals_equals_the_division(R):
    """k_ap's px[k] = cumsum / (k + 1) (metric.py:21) as three multiply-adds against a table of correctly rounded
    reciprocals: the same float64 bits as the division, and as the oracle's, across chunk boundaries of np.sum (8192)."""
    rng = np.random.default_rng(R)
    Q, N, b, C = 64, 70000, 32, 4
    db = (rng.random((N, b)) < 0.5).astype(np.uint8)
    qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    dl = (rng.random((N, C)) < 0.3).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        got = {}
        for v in (0, 1):
            ctx.set_option("ap_recip", v)
            got[v], _ = ctx.map(R)
        assert np.array_equal(got[0], got[1], equal_nan=True)
        assert np.array_equal(got[1], ap_ref, equal_nan=True)
    finally:
        ctx.close()


def test_failed_database_load_leaves_nothing_resident():
    """a load that fails on its labels must not leave MAPs ranking against a half-replaced database (or, with
    a wider code, reading past the query buffer).  Float labels that an int64 cast would turn into {0,1} are refused too."""
    import types
    rng = np.random.default_rng(3)
    N, Q, b, C = 5000, 20, 32, 4
    mk = lambda n, bb: types.SimpleNamespace(output=(rng.integers(0, 2, (n, bb)) * 2 - 1).astype(np.float32),
                                             label=np.eye(C, dtype=np.int64)[rng.integers(0, C, n)])
    good, q = mk(N, b), mk(Q, b)
    bad = mk(N, 2 * b)
    bad.label = bad.label.copy()
    bad.label[7, 1] = 3                                   # not an indicator matrix
    m = metric.MAPs(100)
    try:
        want = m.get_maps_by_feature(good, q)
        m.set_database(good)
        with pytest.raises(ValueError):
            m.get_maps_by_feature(bad, q)
        with pytest.raises(ValueError):                   # nothing is resident now: not the old database, not the bad one
            m.get_maps_by_feature(None, q)
        with pytest.raises(ValueError):
            m.set_database(bad)
        with pytest.raises(ValueError):
            m.get_maps_by_feature(None, q)
        assert m.get_maps_by_feature(good, q) == want     # and the object still works
        half = types.SimpleNamespace(output=good.output, label=good.label.astype(np.float64) * 0.5 + 0.25)   # 0.25 / 0.75
        with pytest.raises(ValueError):
            m.get_maps_by_feature(half, q)
    finally:
        m.close()


def test_function_spelling_reuses_a_read_only_database():
    """MAP / calc_map keep the packed database of the last call when handed the very same read-only arrays again (an
    evaluation loop over one database); writable arrays are uploaded again because they may have changed in place."""
    rng = np.random.default_rng(4)
    N, Q, b, C = 20000, 30, 48, 5
    db = rng.integers(0, 2, (N, b)).astype(np.float32)
    dl = np.eye(C, dtype=np.int64)[rng.integers(0, C, N)]
    qs = [rng.integers(0, 2, (Q, b)).astype(np.float32) for _ in range(2)]
    ql = np.eye(C, dtype=np.int64)[rng.integers(0, C, Q)]
    want = [metric.MAP(x, db, ql, dl, 500) for x in qs]   # writable: every call uploads
    db.flags.writeable = False
    dl.flags.writeable = False
    eng = metric._Shared.get(0)
    got = [metric.MAP(x, db, ql, dl, 500) for x in qs] + [metric.calc_map(qs[0], db, ql, dl, 500)]
    assert got[:2] == want and got[2] == want[0]
    assert eng.resident is not None and eng.resident[0] is db
    db2 = db.copy()
    db2[:, 0] = 1 - db2[:, 0]
    db2.flags.writeable = False
    metric.MAP(qs[0], db2, ql, dl, 500)                   # another database: reloaded
    assert eng.resident[0] is db2
    assert metric.MAP(qs[1], db, ql, dl, 500) == want[1]  # and back


@pytest.mark.parametrize("b,sigma", [(64, 12), (64, 40), (64, 64), (40, 40), (24, 64)])
def test_wild_guesses_never_change_the_result(b, sigma):
    """The bet's guess pushed far too high (guess_sigma): the three-distances-per-accumulator select then sees dense
    hits -- full queues, rings that overflow (the lanes' direct route), slices beyond their capacity -- and, at b = 64
    a thin sample, cuts beyond the 63 its 7-bit fields can hold (test_queries_far_from_every_row_fall_back_exactly).  Whatever it does, a lost bet
    must be noticed and the answer must be the oracle's."""
    rng = np.random.default_rng(500 + b + sigma)
    Q, N, R, C = 200, 150000, 700, 7
    db = (rng.random((N, b)) < 0.5).astype(np.uint8)
    qb = db[rng.integers(0, N, Q)] ^ (rng.random((Q, b)) < 0.2).astype(np.uint8)
    db[1000:1400] = db[1000]                                    # a run of identical rows: bursts of hits in one slice
    dl = (rng.random((N, C)) < 0.25).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.25).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        ctx.set_option("guess_sigma", sigma)
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        for _ in range(2):
            ap, rel = ctx.map(R)
            assert np.array_equal(ap, ap_ref, equal_nan=True)
    finally:
        ctx.close()


def test_sampled_kernel_timing_counts_every_nth_step(case_cache):
    """hg_timing level 1 with "timing_every" = 4: the select pass is bracketed by HIP events on every fourth one-shot step
    only (what bench.py's roofline averages over) -- and the answers do not depend on it."""
    c = case_cache("c2_q64")
    g = cases.load_golden("c2_q64")
    ctx = _native.Context(0)
    try:
        _load(ctx, c)
        ctx.map(c["R"])
        ctx.set_option("timing_every", 4)
        ctx.timing_enable(1)
        ctx.timing_reset()
        for _ in range(8):
            ap, rel = ctx.map(c["R"])
            assert np.array_equal(ap, g["ap"], equal_nan=True)
        t = ctx.timing_read()
        sel = [v for k, v in t.items() if k.startswith("k_select")]
        assert len(sel) == 1 and sel[0][1] == 2 and sel[0][0] > 0.0, t
        assert t["step_gpu_span"][1] == 2
        ctx.set_option("timing_every", 1)
        ctx.timing_enable(1)
        ctx.timing_reset()
        for _ in range(3):
            ctx.map(c["R"])
        assert [v[1] for k, v in ctx.timing_read().items() if k.startswith("k_select")] == [3]
    finally:
        ctx.timing_enable(False)
        ctx.close()


def test_a_database_stored_class_by_class_keeps_the_bet():
    """Rows sorted by label, codes that follow the labels: a query's near rows all sit in its class's tenth of the
    segments, ten times what slices sized for an even spread hold.  The first bet on a database measures that crowding in
    its guess kernel (k_guess_direct's probe) and widens the slices BEFORE selecting: no bet is lost.  With the probe off
    the first call loses the bet twice, widens (cap_boost) and wins, as it did until round 3.  Later calls on the same
    database bet with the wide slices at once; a new database starts over.  Every answer is the oracle's."""
    rng = np.random.default_rng(4242)
    Q, N, R, C, b = 64, 200000, 3000, 10, 64
    cls = np.sort(rng.integers(0, C, N))
    proto = (rng.random((C, b)) < 0.5).astype(np.uint8)
    db = proto[cls] ^ (rng.random((N, b)) < 0.25).astype(np.uint8)
    qcls = rng.integers(0, C, Q)
    qb = proto[qcls] ^ (rng.random((Q, b)) < 0.25).astype(np.uint8)
    dl = np.eye(C, dtype=np.int8)[cls]
    ql = np.eye(C, dtype=np.int8)[qcls]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        assert ctx.get_stat("last_optimistic") == 1 and ctx.get_stat("optimistic_fallbacks") == 0
        assert ctx.get_stat("cap_boost") >= 8 and ctx.get_stat("optimistic_rebets") == 0 and ctx.get_stat("crowding_x100") > 600
        ctx.set_option("crowd_probe", 0)                       # the adaptive way: lose, widen, remember
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        assert ctx.get_stat("last_optimistic") == 1 and ctx.get_stat("optimistic_fallbacks") == 0
        boost, rebets = ctx.get_stat("cap_boost"), ctx.get_stat("optimistic_rebets")
        assert boost > 1 and rebets >= 2
        ctx.set_option("crowd_probe", 1)
        ap, rel = ctx.map(R)                                   # the widened slices are remembered: no further lost bet
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        assert ctx.get_stat("optimistic_rebets") == rebets and ctx.get_stat("last_optimistic") == 1
        perm = rng.permutation(N)                              # the same rows shuffled: a new database, ordinary slices again
        _load(ctx, dict(qbits=qb, dbbits=db[perm], qlab=ql, dblab=dl[perm], b=b))
        assert ctx.get_stat("cap_boost") == 1
        ap2, _ = ctx.map(R)
        assert ctx.get_stat("cap_boost") == 1 and ctx.get_stat("optimistic_rebets") == rebets and ctx.get_stat("crowding_x100") < 400
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, ap_ref2, *_ = O.map_from_codes(qb, db[perm], ql, dl[perm], R)
        assert np.array_equal(ap2, ap_ref2, equal_nan=True)
    finally:
        ctx.close()


@pytest.mark.parametrize("b", [64, 40, 128, 72])
def test_one_lane_bursts_while_its_neighbours_stay_sparse(b):
    """Inside ONE wavefront of the matrix-core select: a few queries meet long runs of rows at distance 0 (their slices'
    rings overflow: the lane walks its hits to global memory itself, then returns to the rings), runs that straddle
    window and segment boundaries, every third row of a stretch -- while the other lanes' queries see ordinary sparse
    hits through the queue.  Same APs as the oracle, twice (the second call reuses every buffer)."""
    rng = np.random.default_rng(900 + b)
    Q, N, R, C = 130, 160000, 900, 5
    db = (rng.random((N, b)) < 0.5).astype(np.uint8)
    qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    db[5000:5600] = qb[3]                                  # 600 consecutive duplicates of query 3
    db[20000:21200:3] = qb[70]                             # every third row over 1200 rows
    db[47990:48110] = qb[64]                               # across a 48-row supertile / window boundary, first lane of the second tile
    db[N - 70:] = qb[129]                                  # the database's ragged end
    dl = (rng.random((N, C)) < 0.3).astype(np.int8)
    ql = (rng.random((Q, C)) < 0.3).astype(np.int8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        for _ in range(2):
            ap, rel = ctx.map(R)
            assert np.array_equal(ap, ap_ref, equal_nan=True)
        assert ctx.get_stat("optimistic_runs") >= 1
    finally:
        ctx.close()


@pytest.mark.parametrize("Q,N,b,R,C", [(40, 200000, 64, 100000, 10), (25, 70001, 33, 70001, 70), (60, 30000, 100, 9000, 3),
                                       (33, 150000, 16, 30001, 128), (70, 9000, 64, 8999, 5),
                                       (30, 300000, 126, 300000, 7), (24, 900000, 64, 800000, 10)])     # (bitmaps beyond one block's LDS: two blocks per query)
def test_dense_regime_ranks_the_rows_directly(Q, N, b, R, C):
    """R > N / 8 on one shard (up to the reference's own R = N, lib/metric.py:14,19 with MAP_R = DB_SIZE): the byte matrix
    (k_dense_bytes + k_rank_dense) ranks every row of every query -- counter columns per thread, a bitmap of R bits, ties by
    index.  AP, ranked indices and distances equal the oracle; no bet is placed, no histogram pass runs."""
    rng = np.random.default_rng(N + R)
    proto = (rng.random((C, b)) < 0.5).astype(np.uint8)
    lab_db, lab_q = rng.integers(0, C, N), rng.integers(0, C, Q)
    db = proto[lab_db] ^ (rng.random((N, b)) < 0.3).astype(np.uint8)
    qb = proto[lab_q] ^ (rng.random((Q, b)) < 0.3).astype(np.uint8)
    dl = np.eye(C, dtype=np.int8)[lab_db]
    ql = np.eye(C, dtype=np.int8)[lab_q]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, imatch_ref, idx_ref, dist_ref = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        ap3, _ = ctx.map(R)                               # the default: N/8 < R <= N through the byte matrix (k_dense_bytes + k_rank_dense)
        assert np.array_equal(ap3, ap_ref, equal_nan=True)
        assert ctx.get_stat("optimistic_runs") == 0
        assert ctx.get_stat("rank_variant") == 7
        ctx.topr(R)
        idx, dist = ctx.get_topr()
        assert np.array_equal(idx.astype(np.int64), idx_ref) and np.array_equal(dist.astype(np.int64), dist_ref)
        ctx.set_option("dense_budget_mb", 1)              # the byte matrix in several chunks of queries
        ap5, _ = ctx.map(R)
        assert np.array_equal(ap5, ap_ref, equal_nan=True)
        for gbm in (1, 0):                                # the R-bit bitmap in global memory (k_ap afterwards) / in LDS (AP from the epilogue)
            ctx.set_option("rank_dense_gbm", gbm)
            ap6, _ = ctx.map(R)
            assert np.array_equal(ap6, ap_ref, equal_nan=True)
            assert ctx.get_stat("rank_variant") == 7
            ctx.topr(R)
            idx, dist = ctx.get_topr()
            assert np.array_equal(idx.astype(np.int64), idx_ref) and np.array_equal(dist.astype(np.int64), dist_ref)
        ctx.set_option("rank_dense_gbm", -1)
        ctx.set_option("rank_dense", 0)                   # the older sequences give the same (R = N: k_rank_fused walks the rows itself)
        ap2, _ = ctx.map(R)
        assert np.array_equal(ap2, ap_ref, equal_nan=True)
        assert ctx.get_stat("rank_variant") != 7
        ctx.set_option("all_rows_shortcut", 0)            # ... and histogram -> plan -> select -> rank
        ap2, _ = ctx.map(R)
        assert np.array_equal(ap2, ap_ref, equal_nan=True)
        assert ctx.get_stat("rank_variant") != 7
    finally:
        ctx.close()


@pytest.mark.parametrize("Q,N,b,R,C,by_class", [(4200, 200000, 64, 20000, 10, False), (4100, 150000, 100, 15000, 100, False),
                                                (4300, 262144, 48, 30000, 12, True)])
def test_long_lists_of_a_bet_are_ranked_slice_by_slice(Q, N, b, R, C, by_class):
    """N/100 < R <= N/8: the bet's record lists no longer fit k_rank_lean's LDS; k_rank_dense<slices> ranks them in two passes
    with thread = part of a slice (private counter columns, one returning LDS add per record) instead of k_rank_cnt's tiles.
    Same APs as the oracle and as the tiled kernel; a database stored class by class loses its first bets (overflowing
    slices are flagged by the select and make this kernel leave the query) and still ends exact."""
    rng = np.random.default_rng(Q + N + R)
    proto = (rng.random((C, b)) < 0.5).astype(np.uint8)
    lab_db, lab_q = rng.integers(0, C, N), rng.integers(0, C, Q)
    if by_class:
        lab_db = np.sort(lab_db)
    db = proto[lab_db] ^ (rng.random((N, b)) < 0.3).astype(np.uint8)
    qb = proto[lab_q] ^ (rng.random((Q, b)) < 0.3).astype(np.uint8)
    dl = np.eye(C, dtype=np.int8)[lab_db]
    ql = np.eye(C, dtype=np.int8)[lab_q]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb, db, ql, dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        ap, rel = ctx.map(R)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        ap, rel = ctx.map(R)                              # (a second call: the widened slices of a class-sorted database are in force)
        assert np.array_equal(ap, ap_ref, equal_nan=True)
        if ctx.get_stat("last_optimistic") == 1:
            assert ctx.get_stat("rank_variant") == 8
        ctx.set_option("rank_slices", 0)
        ap2, _ = ctx.map(R)
        assert np.array_equal(ap2, ap_ref, equal_nan=True)
        assert ctx.get_stat("rank_variant") != 8
    finally:
        ctx.close()


def test_fused_step_hands_wide_lists_to_the_general_kernel(ctx):
    """hg_map's bet ranks AND evaluates in k_rank_cnt (the AP leaves from its epilogue) and launches the general rank
    kernel only when the step's download reports queries k_rank_cnt declined.  Two queries here have rows planted at
    EVERY distance 0..21 (30 each), so their top-2000 lists span more than the 16 distances k_rank_cnt places: they are
    the leftovers, ranked by k_rank_fused and evaluated by k_ap afterwards; all APs equal the oracle's, with the fused
    epilogue on and off."""
    from hashgan_amd import synth
    Q, N, b, R, C = 192, 100000, 64, 2000, 10              # (long lists)
    dl, _ = synth.onehot_labels(71, N, C)
    ql, _ = synth.onehot_labels(72, Q, C)
    db = synth.random_bits(73, N, b)
    qb = synth.random_bits(74, Q, b)
    rng = np.random.default_rng(75)
    for qi, base in ((7, 20000), (130, 60000)):
        k = 0
        for d in range(22):
            for _ in range(30):
                row = qb[qi].copy()
                row[rng.choice(b, d, replace=False)] ^= 1
                db[base + 7 * k] = row                       # spread over several segments
                k += 1
    probe = [0, 7, 8, 129, 130, 191]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb[probe], db, ql[probe], dl, R)
    ctx.set_option("optimistic", 1)
    ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
    ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
    try:
        l0, f0 = ctx.get_stat("rank_leftovers"), ctx.get_stat("optimistic_fallbacks")
        ap, rel = ctx.map(R)
        assert ctx.get_stat("last_optimistic") == 1 and ctx.get_stat("optimistic_fallbacks") == f0
        assert ctx.get_stat("ap_fused") == 0 and ctx.get_stat("rank_leftovers") - l0 >= 2      # (the leftover pass clears ap_fused)
        assert np.array_equal(ap[probe], ap_ref, equal_nan=True)
        # the next step expects leftovers and ranks them within its stream (k_rank_dense<slices> on the flagged queries): no second
        # round trip, the epilogue's APs stay in force, the same numbers
        ctx.set_option("max_segments", 200)               # (192 queries ask for more segments than that kernel's 256 slices)
        l1 = ctx.get_stat("rank_leftovers")
        ap1, rel1 = ctx.map(R)
        assert ctx.get_stat("ap_fused") == 1 and ctx.get_stat("rank_leftovers") - l1 >= 2
        assert np.array_equal(ap1, ap, equal_nan=True) and np.array_equal(rel1, rel)
        # ... and so do steps enqueued blind (hg_map_begin): leftovers ranked within the step are no reason to run it again
        n0, r0 = ctx.get_stat("map_async_steps"), ctx.get_stat("map_async_redone")
        ctx.map_begin(R)
        ctx.map_begin(R)
        for _ in range(2):
            apb, relb = ctx.map_end()
            assert np.array_equal(apb, ap, equal_nan=True) and np.array_equal(relb, rel)
        assert ctx.get_stat("map_async_steps") - n0 == 2 and ctx.get_stat("map_async_redone") == r0
        ctx.set_option("inline_leftovers", 0)
        ap1, rel1 = ctx.map(R)
        assert ctx.get_stat("ap_fused") == 0
        ctx.map_begin(R)                                  # leftovers the step does not rank itself: never blind
        apb, relb = ctx.map_end()
        assert np.array_equal(apb, ap, equal_nan=True) and ctx.get_stat("map_async_steps") - n0 == 2
        assert np.array_equal(ap1, ap, equal_nan=True) and np.array_equal(rel1, rel)
        ctx.set_option("inline_leftovers", 1)
        ctx.set_option("max_segments", 2048)
        ctx.set_option("fuse_ap", 0)
        ap2, rel2 = ctx.map(R)
        assert np.array_equal(ap2, ap, equal_nan=True) and np.array_equal(rel2, rel)
    finally:
        ctx.set_option("fuse_ap", 1)
    # an ordinary workload: every query evaluated in the epilogue, nothing left over
    db2 = synth.random_bits(76, N, b)
    ctx.set_database(metric.pack_codes(db2), metric.pack_labels(dl), b, C)
    ctx.set_queries(metric.pack_codes(qb), metric.pack_labels(ql))
    l1 = ctx.get_stat("rank_leftovers")
    ap3, _ = ctx.map(R)
    assert ctx.get_stat("ap_fused") == 1 and ctx.get_stat("rank_leftovers") == l1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap3_ref, *_ = O.map_from_codes(qb[probe], db2, ql[probe], dl, R)
    assert np.array_equal(ap3[probe], ap3_ref, equal_nan=True)


@pytest.mark.parametrize("b", [65, 80, 96, 97, 100, 127, 128])
def test_long_codes_take_the_packed_matrix_core_select(b):
    """Codes of 65..128 bits: hg_map's bet selects with k_select_mx4 (two rows per fp4 accumulator, 8-bit fields -- the
    harvested sign bits must be exactly dist <= T for every row of a 32-row supertile, both lane halves, ragged segment
    ends included).  APs equal the oracle's (the reference's canonical order) and the 8-byte-record kernel's."""
    from hashgan_amd import synth
    Q, N, R, C = 150, 140000 + b, 1100, 7
    dl, _ = synth.onehot_labels(500 + b, N, C)
    ql, _ = synth.onehot_labels(600 + b, Q, C)
    db = synth.planted_codes(700 + b, dl, b, 0.33)
    qb = synth.planted_codes(700 + b, ql, b, 0.33)
    db[N - 40:] = qb[149]                                   # a run of duplicates at the database's ragged end
    db[1000:1003] = qb[0]
    probe = [0, 1, 31, 32, 63, 64, 100, 149]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb[probe], db, ql[probe], dl, R)
    ctx = _native.Context(0)
    try:
        _load(ctx, dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, b=b))
        ap, rel = ctx.map(R)
        assert ctx.get_stat("select_variant") == 6 and ctx.get_stat("last_optimistic") == 1
        assert np.array_equal(ap[probe], ap_ref, equal_nan=True)
        ctx.set_option("compact_records", 0)                # k_select_mx, one distance per accumulator, 8-byte records
        ap2, rel2 = ctx.map(R)
        assert ctx.get_stat("select_variant") == 3
        assert np.array_equal(ap2, ap, equal_nan=True) and np.array_equal(rel2, rel)
        ctx.set_option("compact_records", 1)
        ctx.set_option("select_packed", 1)                  # k_select_mx with one-byte records
        ap3, _ = ctx.map(R)
        assert ctx.get_stat("select_variant") == 3 and np.array_equal(ap3, ap, equal_nan=True)
    finally:
        ctx.close()
