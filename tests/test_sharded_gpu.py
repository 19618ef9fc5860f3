"""Virtual shards on ONE GPU: G contexts, G threads, the real orchestration
(hashgan_amd.sharded.evaluate_shard) with an in-process communicator.  The
sharded result must equal the single-shard result bit for bit -- AP, mAP and the
merged ranked lists."""
import threading
import warnings
import numpy as np
import pytest
from tests import cases
from hashgan_amd import _native, metric, sharded

pytestmark = pytest.mark.gpu


def _run_virtual(c, G, gather_topr, defer=False, routed=True):
    N = c["dbbits"].shape[0]
    qw, ql = metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"])
    comms = sharded.LocalComm.create(G)
    results = [None] * G
    stats = [None] * G
    boosts = [None] * G
    errors = []

    def work(r):
        try:
            base, rows = sharded.shard_bounds(N, G)[r]
            ctx = _native.Context(0)
            ctx.set_database(metric.pack_codes(c["dbbits"][base:base + rows]), metric.pack_labels(c["dblab"][base:base + rows]),
                             c["b"], c["dblab"].shape[1], idx_base=base, n_total=N)
            ctx.set_queries(qw, ql)
            comms[r].ctx = ctx
            eng = sharded.HipShardEngine(ctx, want_lists=gather_topr)
            if defer:
                ctx.set_option("defer_verdict", 1)      # hg_rank does not wait; the verdict comes with the AP download
            results[r] = sharded.evaluate_shard(eng, comms[r], c["R"], gather_topr=gather_topr, route_by_owner=routed)
            stats[r] = (ctx.get_stat("optimistic_runs"), ctx.get_stat("optimistic_fallbacks"))
            boosts[r] = ctx.get_stat("cap_boost")
            ctx.close()
        except Exception as e:       # noqa: BLE001
            errors.append(e)
            try:
                comms[r]._s.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errors:
        raise errors[0]
    _run_virtual.last_stats = stats
    _run_virtual.last_boosts = boosts
    return results


@pytest.mark.parametrize("name,G", [("e_ragged", 2), ("e_ragged", 3), ("e_b100", 4), ("e_dups_alleq", 8),
                                    ("e_some_skipped", 2), ("e_r_eq_n", 5), ("c3_nus_q64", 8), ("c4_n10m_q8", 8)])
def test_virtual_shards_equal_single_shard(name, G, case_cache):
    c = case_cache(name)
    g = cases.load_golden(name)
    res = _run_virtual(c, G, gather_topr=True)
    ctx = _native.Context(0)
    ctx.set_database(metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"]), c["b"], c["dblab"].shape[1])
    ctx.set_queries(metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"]))
    ctx.set_option("optimistic", 0)
    ctx.topr(c["R"])
    idx1, dist1 = ctx.get_topr()
    ctx.close()
    for r in range(G):
        ap, rel, (idx, dist) = res[r]
        assert np.array_equal(ap, g["ap"], equal_nan=True), (name, G, r)
        assert np.array_equal(idx, idx1) and np.array_equal(dist, dist1), (name, G, r)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = sharded.mean_ap(*res[0][:2])
    assert (np.isnan(m) and np.isnan(g["map"])) or m == g["map"]


@pytest.mark.parametrize("name,G", [("c2_q64", 4), ("c2_q64", 8), ("c5_b128_q32", 2), ("c3_nus_q64", 2), ("c4_n10m_q8", 8)])
def test_virtual_shards_optimistic_sequence(name, G, case_cache):
    """Big enough for the sharded bet (sample -> guess -> candidates -> rank): must equal the golden
    AP of the unmodified reference, and really have taken the one-pass route."""
    c = case_cache(name)
    g = cases.load_golden(name)
    # routed: the bet's tables by all-to-all to the owner of their queries (the default); else the all-gather form
    combos = ((False, True), (False, False)) if name == "c4_n10m_q8" else ((False, True), (True, True), (False, False))
    for defer, routed in combos:
        res = _run_virtual(c, G, gather_topr=False, defer=defer, routed=routed)
        for r in range(G):
            ap, rel = res[r]
            assert np.array_equal(ap, g["ap"], equal_nan=True), (name, G, r, defer, routed)
        assert sharded.mean_ap(*res[0]) == g["map"]
        assert all(st == (1, 0) for st in _run_virtual.last_stats), _run_virtual.last_stats


def test_virtual_shards_lost_bet_is_consistent():
    """Database sorted by class: near rows pile up in a few segments of one shard, slices overflow
    there, every rank must see the bet as lost, widen its slices alike (cap_boost) and bet again -- same result."""
    from hashgan_amd import synth
    from oracle import hamming_map as O
    Q, N, b, R, C = 128, 262144, 32, 4000, 10
    dl, cls = synth.onehot_labels(81, N, C)
    ql, _ = synth.onehot_labels(82, Q, C)
    db = synth.planted_codes(83, dl, b, 0.2)
    qb = synth.planted_codes(83, ql, b, 0.2)
    order = np.argsort(cls, kind="stable")
    c = dict(qbits=qb, dbbits=db[order], qlab=ql, dblab=dl[order], R=R, b=b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb[:24], c["dbbits"], ql[:24], c["dblab"], R)
    for defer in (False, True):
        res = _run_virtual(c, 2, gather_topr=False, defer=defer)
        for r in range(2):
            assert np.array_equal(res[r][0][:24], ap_ref, equal_nan=True)
        st = _run_virtual.last_stats
        assert st[0] == st[1] == (2, 1), st                  # both ranks bet twice and lost the first, alike
        bo = _run_virtual.last_boosts
        assert bo == [8, 8], bo                              # the escalation ran in lockstep and its second bet held


def test_virtual_shards_cut_beyond_the_exchanged_planes_goes_exact_at_once():
    """Every row far from every query (distance ~0.8 b): the owner-routed guess finds no cut within the b/2 + 2 planes it is
    sent, the bet takes every row and overflows -- wider slices cannot help, so the ranks must NOT escalate cap_boost (stat
    "cut_beyond_planes") but run the exact sequence, alike, with the reference's result."""
    from hashgan_amd import synth
    from oracle import hamming_map as O
    Q, N, b, R, C = 64, 131072, 32, 2000, 10
    rng = np.random.default_rng(91)
    dl, _ = synth.onehot_labels(92, N, C)
    ql, _ = synth.onehot_labels(93, Q, C)
    db = (rng.random((N, b)) < 0.9).astype(np.int8)          # mostly ones
    qb = (rng.random((Q, b)) < 0.1).astype(np.int8)          # mostly zeros
    c = dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, R=R, b=b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb[:16], db, ql[:16], dl, R)
    res = _run_virtual(c, 2, gather_topr=False)
    for r in range(2):
        assert np.array_equal(res[r][0][:16], ap_ref, equal_nan=True)
    assert _run_virtual.last_boosts == [1, 1], _run_virtual.last_boosts      # nobody widened its slices


def test_virtual_shards_wide_labels_take_the_bet():
    """More than 128 classes: the record pass cannot carry the match bit, so the merged-ranking bet builds its local
    bitmaps with k_match through the local ranked lists (this path once produced all-zero bitmaps -> mAP nan)."""
    from hashgan_amd import synth
    from oracle import hamming_map as O
    Q, N, b, R, C = 96, 262144, 32, 3000, 150
    dl = synth.multihot_labels(91, N, C)
    ql = synth.multihot_labels(92, Q, C)
    db = synth.planted_codes(93, dl, b, 0.25)
    qb = synth.planted_codes(93, ql, b, 0.25)
    c = dict(qbits=qb, dbbits=db, qlab=ql, dblab=dl, R=R, b=b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ap_ref, *_ = O.map_from_codes(qb[:32], db, ql[:32], dl, R)
    assert np.isfinite(ap_ref).any()
    for G in (1, 2, 3):
        for defer in (False, True):
            res = _run_virtual(c, G, gather_topr=False, defer=defer)
            for r in range(G):
                assert np.array_equal(res[r][0][:32], ap_ref, equal_nan=True), (G, r, defer)
            assert all(st == (1, 0) for st in _run_virtual.last_stats), _run_virtual.last_stats


def _one_rank_rccl(c, gather_topr, async_stages, one_call=True):
    ctx = _native.Context(0)
    try:
        ctx.set_database(metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"]), c["b"], c["dblab"].shape[1])
        ctx.set_queries(metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"]))
        comm = sharded.init_rccl(ctx, rank=0, world=1)
        assert (comm.rank, comm.world) == (0, 1)
        eng = sharded.HipShardEngine(ctx, want_lists=gather_topr, async_stages=async_stages)
        out = sharded.evaluate_shard(eng, comm, c["R"], gather_topr=gather_topr, always_gather=True, one_call=one_call)
        stats = (ctx.get_stat("optimistic_runs"), ctx.get_stat("optimistic_fallbacks"))
        ctx.comm_destroy()
        return out, stats
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["c2_q64", "e_ragged", "c3_nus_q64"])
def test_one_rank_native_rccl_matches_golden(name, case_cache):
    """The collectives of every sequence through the library's own RCCL communicator (hg_comm_init / hg_allgather /
    hg_allgather_topr) -- one rank, because a box has one GPU; no torch in the process."""
    c = case_cache(name)
    g = cases.load_golden(name)
    for async_stages, one_call in ((False, True), (True, True), (True, False), (False, False)):
        # one_call: the owner-routed bet as ONE library call (hg_shard_step); else the same stages driven from Python, call by call
        (ap, rel), stats = _one_rank_rccl(c, False, async_stages, one_call)
        assert np.array_equal(ap, g["ap"], equal_nan=True), (name, async_stages, one_call)
        if name == "c2_q64":
            assert stats == (1, 0), stats                  # the merged-ranking bet ran, over real collectives
    (ap, rel, (idx, dist)), _ = _one_rank_rccl(c, True, False)
    assert np.array_equal(ap, g["ap"], equal_nan=True)
    if "idx" in g:
        assert np.array_equal(idx, g["idx"])


@pytest.mark.parametrize("name", ["c2_q64", "c3_nus_q64", "c5_b128_q32", "e_ragged"])
def test_shard_step_with_a_replica_world_of_one_is_the_one_gpu_result(name, case_cache):
    """hg_shard_step without a communicator: replica_world = 1 (this rank its own only peer -- the exchanges are device copies
    inside the library) is the whole sequence of the sharded bet on one GPU: equal to the reference's golden.  A shape the bet does
    not take reports None and enqueues nothing; replica worlds > 1 are a timing aid (tools/replica_shard_timing.py), the step
    must still complete on them."""
    c = case_cache(name)
    g = cases.load_golden(name)
    ctx = _native.Context(0)
    try:
        ctx.set_database(metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"]), c["b"], c["dblab"].shape[1])
        ctx.set_queries(metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"]))
        ap, rel, lost = ctx.shard_step(c["R"], replica_world=1)
        if name == "e_ragged":                             # N = 10 007: no bet on a database this small
            assert lost is None
        else:
            assert lost is False and np.array_equal(ap, g["ap"], equal_nan=True)
            ap2, rel2, lost2 = ctx.shard_step(c["R"], replica_world=1)     # and again: state left behind
            assert lost2 is False and np.array_equal(ap2, ap, equal_nan=True) and np.array_equal(rel2, rel)
            _, _, lost4 = ctx.shard_step(c["R"], replica_world=4)          # (peers that are copies of this rank: not the database's mAP)
            assert lost4 in (False, True, None)
            a3, r3 = ctx.map(c["R"])                                       # the context still serves the one-shot call
            assert np.array_equal(a3, g["ap"], equal_nan=True)
    finally:
        ctx.close()


def test_sharded_product_path_is_torch_free():
    """A fresh interpreter runs the sharded sequence over native RCCL; torch must never get imported."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from tests import cases\n"
        "from hashgan_amd import _native, metric, sharded\n"
        "c = cases.build_case('e_ragged'); g = cases.load_golden('e_ragged')\n"
        "ctx = _native.Context(0)\n"
        "ctx.set_database(metric.pack_codes(c['dbbits']), metric.pack_labels(c['dblab']), c['b'], c['dblab'].shape[1])\n"
        "ctx.set_queries(metric.pack_codes(c['qbits']), metric.pack_labels(c['qlab']))\n"
        "comm = sharded.init_rccl(ctx, rank=0, world=1)\n"
        "eng = sharded.HipShardEngine(ctx, async_stages=True)\n"
        "ap, rel = sharded.evaluate_shard(eng, comm, c['R'], always_gather=True)\n"
        "assert np.array_equal(ap, g['ap'], equal_nan=True)\n"
        "assert 'torch' not in sys.modules, sorted(m for m in sys.modules if m.startswith('torch'))[:5]\n"
        "print('TORCH_FREE_OK')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TORCH_FREE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("world,workload", [(2, "c3"), (3, "c3"), (8, "c3")])
def test_bench_multi_rank_control_flow(world, workload, tmp_path):
    """bench.py --gpus G as the driver launches it (one process per rank, RANK / WORLD_SIZE / LOCAL_RANK in the
    environment), dry-run on ONE GPU with the file communicator standing in for RCCL: rendezvous of the processes, shard
    bounds, the orchestration over real process boundaries, max-over-ranks clock, only rank 0 prints -- one JSON line
    whose AP matches the reference's golden.  (world 2: the merged-ranking bet, its tables routed by query owner; world 3
    and 8 -- the driver's full node: rendezvous of eight processes, shard_bounds(N, 8), eight slots -- shards too small for
    the bet, exact sequence.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29871", HG_BENCH_FILECOMM=str(tmp_path / "comm"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--workload", workload,
                                       "--steps", "2", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert all(outs[r][0].strip() == "" for r in range(1, world)), "only rank 0 prints"
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, outs[0][0][-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["unit"] == "queries/s"
    assert d["parity_vs_reference_golden"] is True
    assert abs(d["value"] - 2100 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # raw queries per second
    assert d["optimistic_runs"] == (3 if world == 2 else 0) and d["optimistic_fallbacks"] == 0
    assert "dry_run_not_a_measurement" in d and "cpu_baseline" not in d
    # the line explains its own scaling: the same workload on one GPU of this run, and both decompositions against it
    ss = d["strong_scaling"]
    assert ss["n_gpus"] == world and ss["one_gpu_parity_vs_reference_golden"] is True and ss["one_gpu_ms"] > 0
    for form in ("database_sharded", "query_split"):
        assert abs(ss[form]["speedup"] - ss["one_gpu_ms"] / ss[form]["ms_per_step"]) < 1e-3 * ss[form]["speedup"] + 1e-3
        assert abs(ss[form]["efficiency"] * world - ss[form]["speedup"]) < 1e-3 * ss[form]["speedup"] + 1e-3
    assert d["query_split"]["parity_vs_reference_golden"] is True


@pytest.mark.parametrize("G", [1, 3])
def test_query_split_equals_one_gpu(G):
    """evaluate_query_split: the whole database on every (virtual) rank, the queries split, 16 bytes per query gathered --
    equal to one context's hg_map bit for bit.  G = 1 over the library's RCCL communicator, G = 3 over threads."""
    c = cases.build_case("e_ragged")
    dw, dl = metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"])
    qw, ql = metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"])
    C = c["dblab"].shape[1]
    one = _native.Context(0)
    one.set_database(dw, dl, c["b"], C)
    one.set_queries(qw, ql)
    ap0, rel0 = one.map(c["R"])
    one.close()
    if G == 1:
        ctx = _native.Context(0)
        try:
            ctx.set_database(dw, dl, c["b"], C)
            comm = sharded.init_rccl(ctx, rank=0, world=1)
            ap, rel = sharded.evaluate_query_split(ctx, comm, qw, ql, c["R"])
            ctx.comm_destroy()
        finally:
            ctx.close()
        assert np.array_equal(ap, ap0, equal_nan=True) and np.array_equal(rel, rel0)
        return
    comms = sharded.LocalComm.create(G)
    results, errors = [None] * G, []

    def work(r):
        try:
            ctx = _native.Context(0)
            ctx.set_database(dw, dl, c["b"], C)
            comms[r].ctx = ctx
            results[r] = sharded.evaluate_query_split(ctx, comms[r], qw, ql, c["R"])
            ctx.close()
        except Exception as e:       # noqa: BLE001
            errors.append(e)
            comms[r]._s.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errors:
        raise errors[0]
    for ap, rel in results:
        assert np.array_equal(ap, ap0, equal_nan=True) and np.array_equal(rel, rel0)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment (how a driver may call it): the
    parent spawns the two rank processes itself and prints rank 0's single JSON line.  Dry run on one GPU, the file
    communicator standing in for RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HG_BENCH_FILECOMM"] = str(tmp_path / "comm")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c3", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "error" not in d
    assert d["parity_vs_reference_golden"] is True and d["optimistic_fallbacks"] == 0
    assert d["query_split"]["parity_vs_reference_golden"] is True and d["query_split"]["map"] == d["map"]


@pytest.mark.parametrize("G", [1, 3])
def test_real_valued_ranking_splits_queries_over_ranks(G):
    """evaluate_real_queries: every (virtual) rank holds the whole float table and ranks its share of the queries; the
    gathered per-query results equal one context's hg_map_real bit for bit.  G = 1 runs over the library's own RCCL
    communicator (all_gather_host -> hg_allgather), G = 3 over threads."""
    rng = np.random.default_rng(5)
    Q, N, b, R, C = 37, 70000, 48, 900, 6
    dbf = np.tanh(rng.standard_normal((N, b))).astype(np.float32)
    qf = np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
    dl = (rng.random((N, C)) < 0.2).astype(np.int64)
    ql = (rng.random((Q, C)) < 0.2).astype(np.int64)
    ql[3] = 0                                            # a query that matches nothing: skipped
    one = _native.Context(0)
    one.set_database_f32(dbf, dl)
    one.set_queries_f32(qf, ql)
    ap0, rel0 = one.map_real(R)
    one.close()
    if G == 1:
        ctx = _native.Context(0)
        try:
            ctx.set_database_f32(dbf, dl)
            comm = sharded.init_rccl(ctx, rank=0, world=1)
            ap, rel = sharded.evaluate_real_queries(ctx, comm, qf, ql, R)
            ctx.comm_destroy()
        finally:
            ctx.close()
        assert np.array_equal(ap, ap0, equal_nan=True) and np.array_equal(rel, rel0)
        return
    comms = sharded.LocalComm.create(G)
    results, errors = [None] * G, []

    def work(r):
        try:
            ctx = _native.Context(0)
            ctx.set_database_f32(dbf, dl)
            comms[r].ctx = ctx
            results[r] = sharded.evaluate_real_queries(ctx, comms[r], qf, ql, R)
            ctx.close()
        except Exception as e:       # noqa: BLE001
            errors.append(e)
            comms[r]._s.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errors:
        raise errors[0]
    for ap, rel in results:
        assert np.array_equal(ap, ap0, equal_nan=True) and np.array_equal(rel, rel0)
    assert rel0[3] == 0
