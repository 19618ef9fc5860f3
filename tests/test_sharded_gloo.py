"""The N > 1 path on CPU: hashgan_amd.sharded.evaluate_shard over a real
torch.distributed gloo group (world_size 2, two processes), with the NumPy
stand-in engine.  Checks the sharded result against the single-shard oracle."""
import os
import sys
import warnings
import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, out_dir, ranked=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests import cases
    from tests.numpy_shard_engine import NumpyShardEngine, NumpyRankedEngine, NumpyRoutedEngine
    from hashgan_amd import sharded
    from tests.torch_comm import TorchComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = cases.build_case(name)
    N = c["dbbits"].shape[0]
    base, rows = sharded.shard_bounds(N, world)[rank]
    cls = NumpyRoutedEngine if ranked == "routed" else NumpyRankedEngine if ranked else NumpyShardEngine
    eng = cls(c["qbits"], c["qlab"], c["dbbits"][base:base + rows], c["dblab"][base:base + rows], base)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ap, rel = sharded.evaluate_shard(eng, TorchComm(), c["R"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ap=ap, rel=rel)
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["e_b8", "e_some_skipped", "e_dups_alleq"])
def test_two_rank_gloo_matches_single_shard_oracle(name, tmp_path):
    from tests import cases
    from oracle import hamming_map as O
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    c = cases.build_case(name)
    g = cases.load_golden(name)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["ap"], r1["ap"], equal_nan=True)
    assert np.array_equal(r0["ap"], g["ap"], equal_nan=True)          # the unmodified reference's values


@pytest.mark.parametrize("name", ["e_b8", "e_dups_alleq"])
def test_two_rank_gloo_merged_local_rankings(name, tmp_path):
    """The one-exchange form of the bet (select_ranked / merge_ranked branch of evaluate_shard) over gloo."""
    from tests import cases
    port = 29950 + (os.getpid() % 40)
    mp.spawn(_worker, args=(2, port, name, str(tmp_path), True), nprocs=2, join=True)
    g = cases.load_golden(name)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["ap"], r1["ap"], equal_nan=True)
    assert np.array_equal(r0["ap"], g["ap"], equal_nan=True)


@pytest.mark.parametrize("name,world", [("e_b8", 2), ("e_some_skipped", 3), ("e_dups_alleq", 2)])
def test_gloo_owner_routed_exchanges(name, world, tmp_path):
    """The bet with its tables routed by query owner (evaluate_shard's all_to_all branch: sampled histograms -> owner's
    guess -> answers back -> record counts + local bitmaps -> the owner stitches and evaluates its queries) over a gloo
    group of 2 and 3 processes: every rank ends up with every query's AP, equal to the unmodified reference's golden."""
    from tests import cases
    port = 29100 + (os.getpid() % 150)
    mp.spawn(_worker, args=(world, port, name, str(tmp_path), "routed"), nprocs=world, join=True)
    g = cases.load_golden(name)
    rs = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(rs[0]["ap"], r["ap"], equal_nan=True) and np.array_equal(rs[0]["rel"], r["rel"])
    assert np.array_equal(rs[0]["ap"], g["ap"], equal_nan=True)


def test_shard_bounds():
    from hashgan_amd import sharded
    assert sharded.shard_bounds(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert sharded.shard_bounds(8, 8) == [(i, 1) for i in range(8)]


def _id_worker(rank, world, port, out_dir):
    """hashgan_amd.sharded.init_rccl's rendezvous without RCCL: rank 0 draws an id, the file carries it to the others."""
    sys.path.insert(0, ROOT)
    from hashgan_amd import _native, sharded
    os.environ["MASTER_PORT"] = str(port)

    class FakeCtx:                       # the three calls init_rccl makes on a context
        def comm_init(self, uid, r, w):
            self.uid, self.r, self.w = bytes(uid), r, w

        def comm_info(self):
            return self.r, self.w

        def barrier(self):               # the real one is an all-reduce; here: wait until every rank has written its result
            open(os.path.join(out_dir, "arrived%d" % self.r), "w").close()
            import time
            t0 = time.time()
            while not all(os.path.exists(os.path.join(out_dir, "arrived%d" % k)) for k in range(self.w)):
                assert time.time() - t0 < 60
                time.sleep(0.01)

    _native.comm_unique_id = lambda: bytes([(7 * i + os.getpid()) % 251 for i in range(_native.COMM_ID_BYTES)])
    ctx = FakeCtx()
    comm = sharded.init_rccl(ctx, rank, world, timeout=60)
    assert (comm.rank, comm.world) == (rank, world)
    with open(os.path.join(out_dir, "id%d" % rank), "wb") as f:
        f.write(ctx.uid)


def test_rccl_id_rendezvous_between_processes(tmp_path):
    """Three sibling processes (as a launcher starts them): every rank ends up with rank 0's 128 bytes, the
    rendezvous file is gone afterwards, and a stale file of an earlier launch is not mistaken for the id."""
    from hashgan_amd import sharded
    port = 31000 + (os.getpid() % 500)
    os.environ["MASTER_PORT"] = str(port)
    stale = sharded._id_file(3)                          # the name the children will use (address, port, world size)
    with open(stale, "wb") as f:
        f.write(b"\0" * 128)
    os.utime(stale, (1, 1))                              # ancient
    mp.spawn(_id_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    ids = [open(tmp_path / ("id%d" % r), "rb").read() for r in range(3)]
    assert ids[0] == ids[1] == ids[2] and len(ids[0]) == 128 and ids[0] != b"\0" * 128
    assert not os.path.exists(stale)


def _real_worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests import cases
    from oracle import real_map as RM
    from hashgan_amd import sharded
    from tests.torch_comm import TorchComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = cases.build_real_case(name)

    class OracleCtx:                     # stands in for a context holding the whole float table (test infrastructure)
        def set_queries_f32(self, x, lab):
            self.q, self.ql = x, lab

        def map_real(self, R):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, ap, *_ = RM.map_from_features(self.q, c["dbf"], self.ql, c["dblab"], R)
            rel = np.array([0 if np.isnan(a) else 1 for a in ap], np.int64)
            return np.nan_to_num(ap), rel

    ap, rel = sharded.evaluate_real_queries(OracleCtx(), TorchComm(), c["qf"], c["qlab"], c["R"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ap=ap, rel=rel)
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("real_multi", 2), ("real_bits01", 3)])
def test_gloo_query_split_of_the_real_valued_ranking(name, world, tmp_path):
    """evaluate_real_queries (queries split over the ranks, database replicated) over a gloo group: every rank ends
    up with every query's AP, in query order, equal to the unmodified reference's."""
    from tests import cases
    port = 29300 + (os.getpid() % 250)
    mp.spawn(_real_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    g = cases.load_golden(name)
    rs = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(rs[0]["ap"], r["ap"]) and np.array_equal(rs[0]["rel"], r["rel"])
    want = np.nan_to_num(g["ap"])
    assert np.array_equal(rs[0]["ap"], want)
    assert np.array_equal(rs[0]["rel"] != 0, ~np.isnan(g["ap"]))


def _qsplit_worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests import cases
    from oracle import hamming_map as O
    from hashgan_amd import sharded, metric
    from tests.torch_comm import TorchComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = cases.build_case(name)

    class OracleCtx:                     # stands in for a context holding the whole packed database (test infrastructure)
        def set_queries(self, codes, labels):
            self.qw, self.qlw = codes, labels

        def map(self, R):
            b, C = c["b"], c["dblab"].shape[1]
            unpack = lambda w, n: np.unpackbits(np.ascontiguousarray(w).view(np.uint8).reshape(w.shape[0], -1), axis=1, bitorder="little")[:, :n]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, ap, *_ = O.map_from_codes(unpack(self.qw, b), c["dbbits"], unpack(self.qlw, C).astype(np.int8), c["dblab"], R)
            rel = np.array([0 if np.isnan(a) else 1 for a in ap], np.int64)
            return ap, rel

    ap, rel = sharded.evaluate_query_split(OracleCtx(), TorchComm(), metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"]), c["R"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ap=ap, rel=rel)
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("e_some_skipped", 2), ("e_ragged", 3)])
def test_gloo_query_split_of_the_hamming_ranking(name, world, tmp_path):
    """evaluate_query_split (database replicated, queries split, no data-path collective) over a gloo group: every rank
    ends up with every query's AP, in query order, equal to the unmodified reference's golden."""
    from tests import cases
    port = 29600 + (os.getpid() % 250)
    mp.spawn(_qsplit_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    g = cases.load_golden(name)
    rs = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(rs[0]["ap"], r["ap"], equal_nan=True) and np.array_equal(rs[0]["rel"], r["rel"])
    assert np.array_equal(rs[0]["ap"], g["ap"], equal_nan=True)


# ---------------------------------------------------------------- the launch itself (no GPU needed)
_RENDEZVOUS = """import os, sys
sys.path.insert(0, %r)
from hashgan_amd import sharded
uid = sharded.exchange_id(int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), lambda: os.urandom(128), 128, timeout=60)
sys.stdout.write(uid.hex())
"""


def test_rccl_id_rendezvous_does_not_depend_on_the_launcher(tmp_path):
    """The id file is named from what all ranks of a launch share (MASTER_ADDR, MASTER_PORT, world size), not from a
    parent pid: ranks started through separate nested `bash -c` wrappers (every rank another parent) still meet, and every
    rank gets rank 0's 128 bytes."""
    import subprocess
    world = 3
    script = tmp_path / "rank.py"
    script.write_text(_RENDEZVOUS % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29877", TMPDIR=str(tmp_path))
        env.pop("HG_COMM_ID_FILE", None)
        procs.append(subprocess.Popen(["bash", "-c", "bash -c '%s %s'; exit $?" % (sys.executable, script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    assert len(outs[0][0]) == 256 and all(o[0] == outs[0][0] for o in outs)
    files = [f for f in os.listdir(tmp_path) if f.startswith("hashgan_amd_rccl_")]
    assert files and all((os.stat(tmp_path / f).st_mode & 0o077) == 0 for f in files)     # private to the user


def test_rccl_id_rendezvous_ignores_a_stale_file_and_times_out(tmp_path):
    """A leftover id file of an earlier (crashed) launch is older than this process: a rank > 0 never takes it, and
    without a rank 0 it gives up with TimeoutError instead of hanging."""
    import time
    from hashgan_amd import sharded
    path = str(tmp_path / "stale.id")
    with open(path, "wb") as f:
        f.write(b"\x01" * 128 + np.float64(time.time() - 1000.0).tobytes())
    os.utime(path, (time.time() - 1000.0, time.time() - 1000.0))
    t0 = time.time()
    with pytest.raises(TimeoutError):
        sharded.exchange_id(1, 2, None, 128, timeout=0.5, path=path)
    assert time.time() - t0 < 5.0
    # rank 0 replaces the leftover (never follows or reuses it)
    uid = sharded.exchange_id(0, 2, lambda: b"\x02" * 128, 128, path=path)
    assert uid == b"\x02" * 128 and sharded.exchange_id(1, 2, None, 128, timeout=5.0, path=path) == uid


def test_rccl_id_rendezvous_ignores_a_file_rank0_could_not_have_written(tmp_path):
    """Rank 0 creates the id file exclusively with mode 0600: a fresh file with any other mode, or a symlink to one, was put
    there by somebody else and is never read as the id."""
    import time
    from hashgan_amd import sharded
    planted = str(tmp_path / "planted.id")
    with open(planted, "wb") as f:
        f.write(b"\x03" * 128 + np.float64(time.time()).tobytes())
    os.chmod(planted, 0o644)
    with pytest.raises(TimeoutError):
        sharded.exchange_id(1, 2, None, 128, timeout=0.3, path=planted)
    os.chmod(planted, 0o600)
    link = str(tmp_path / "link.id")
    os.symlink(planted, link)
    with pytest.raises(TimeoutError):
        sharded.exchange_id(1, 2, None, 128, timeout=0.3, path=link)
    assert sharded.exchange_id(1, 2, None, 128, timeout=5.0, path=planted) == b"\x03" * 128     # mode 0600, own file, fresh: taken


def test_bench_without_a_launcher_reports_instead_of_hanging():
    """`python bench.py --gpus 2` with no WORLD_SIZE spawns its own ranks; on a box where they cannot run (no GPU here)
    the parent prints ONE JSON line with an `error` key and exits non-zero.  A --gpus / WORLD_SIZE mismatch likewise."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HG_BENCH_LAUNCH_TIMEOUT"] = "120"
    try:
        import ctypes
        gpu = ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(ctypes.c_int())) == 0
    except OSError:
        gpu = False
    if not gpu:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert r.returncode != 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-500:])
        d = json.loads(lines[0])
        assert "error" in d and d["value"] is None and d["n_gpus"] == 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode != 0 and "WORLD_SIZE" in d["error"]
