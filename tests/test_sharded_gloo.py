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
    from tests.numpy_shard_engine import NumpyShardEngine, NumpyRankedEngine
    from hashgan_amd import sharded
    from tests.torch_comm import TorchComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = cases.build_case(name)
    N = c["dbbits"].shape[0]
    base, rows = sharded.shard_bounds(N, world)[rank]
    cls = NumpyRankedEngine if ranked else NumpyShardEngine
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
    stale = sharded._id_file(3, str(port)).replace("_%d_" % os.getppid(), "_%d_" % os.getpid())   # the name the children will use
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
