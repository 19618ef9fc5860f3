"""Database-sharded evaluation: one shard per GPU, one process per GPU.

The database is split into contiguous index ranges (shard r owns rows
[base_r, base_r + N_r)); every rank holds all queries.  Exchange steps, all
all-gathers of small per-query tables over RCCL/xGMI through the library's own
C ABI (hg_comm_init / hg_allgather: no PyTorch in the process, collectives on the
context's stream, ordered with its kernels):

  1. per-shard distance histograms  uint32 [b+1][Qpad]   (2.6 MB at C2)
     -> every rank derives the SAME global threshold t, tie quota, and the
        global rank positions of its own rows (k_plan): no comparison-based
        merge is ever needed, shards interleave by (distance, shard, index).
  2. per-shard label-match bit rows, uint64 [Q][ceil(R/64)] (6.3 MB at C2)
     -> OR (global position space) or stitching (local rank order) -> AP.
  3. (the bet with locally ranked records) 16 bytes per query: every rank stitches and evaluates only its own
     share of the queries, the per-query APs and the verdict are gathered.
  The bet's tables (1: sampled histograms, 2: record counts + local bitmaps) travel by ALL-TO-ALL to the owner of
  their queries where engine and communicator can (hg_alltoall: `route_by_owner`, the default): a GPU then receives
  its own queries' share of every shard's table -- 10.6 MB per step at C4 on 8 GPUs -- instead of all of it (80 MB).

`gather_topr` additionally all-gathers the ranked (idx, dist) lists themselves
(the exchange BASELINE.json's north star names, hg_allgather_topr) for callers
that want them.

The real-valued (inner-product) ranking scales over GPUs by splitting the queries
instead (`evaluate_real_queries`): its float table is small enough to replicate.

The reference has no counterpart (lib/metric.py runs in one process); the
result is bit-identical to the single-GPU path, which the tests check with
virtual shards on one GPU (LocalComm) and, for the orchestration, with a NumPy
engine over a two-process gloo group on CPU (tests/torch_comm.py).
"""
import collections
import os
import threading
import time

import numpy as np

# A device allocation handed between an engine and a communicator: raw address + size.
DevBuf = collections.namedtuple("DevBuf", "ptr nbytes")


# ------------------------------------------------------------------ communicators
class RcclComm:
    """all_gather over the context's own RCCL communicator (hg_allgather).  Gathered buffers live in the
    context (four rotating slots: the orchestration below keeps at most three alive at a time)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.rank, self.world = ctx.comm_info()
        if self.world < 1:
            raise RuntimeError("the context has no communicator: call init_rccl(ctx) first")
        self._slot = 0
        self.routed_verified = None    # evaluate_shard: has the owner-routed (hg_alltoall) form matched the all-gather form here?

    def all_gather(self, buf):
        slot = self._slot
        self._slot = (slot + 1) % 4
        return DevBuf(self.ctx.allgather(slot, buf.ptr, buf.nbytes), buf.nbytes * self.world)

    def all_to_all(self, buf):
        """buf = [world] equal blocks, block r for rank r -> [world] blocks, block r from rank r (hg_alltoall)."""
        slot = self._slot
        self._slot = (slot + 1) % 4
        return DevBuf(self.ctx.alltoall(slot, buf.ptr, buf.nbytes // self.world), buf.nbytes)

    def barrier(self):
        self.ctx.barrier()

    def all_gather_host(self, arr):
        """all_gather of one small host array (same shape on every rank) -> [world, ...]: up to the GPU, hg_allgather, back."""
        arr = np.ascontiguousarray(arr)
        dev = self.ctx.scratch(3, arr.nbytes)
        self.ctx.memcpy_htod(dev, arr, arr.nbytes)
        g = self.all_gather(DevBuf(dev, arr.nbytes))
        out = np.empty((self.world,) + arr.shape, arr.dtype)
        self.ctx.memcpy_dtoh(out, g.ptr, g.nbytes)
        self.ctx.synchronize()
        return out

    def allreduce_max(self, x):
        """max over the ranks of one host number (a benchmark's step time)."""
        return self.ctx.allreduce_max(x)


def _id_file(world):
    """Where rank 0 publishes the RCCL id of this launch.  HG_COMM_ID_FILE names it outright (a launcher that spawns the
    ranks itself -- bench.py --gpus N -- passes a path inside a fresh private directory).  Otherwise the name comes from
    what EVERY rank of one launch shares whatever started it (torchrun, mpirun, shell wrappers -- never a parent pid):
    the rendezvous address and port, the world size, the launcher's run id if it set one, and the user."""
    explicit = os.environ.get("HG_COMM_ID_FILE")
    if explicit:
        return explicit
    tag = "%s_%s_%d_%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"), world,
                           os.environ.get("HG_COMM_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or "0")
    tag = "".join(ch if ch.isalnum() or ch in "._-" else "-" for ch in tag)
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "hashgan_amd_rccl_%d_%s.id" % (os.getuid(), tag))


def exchange_id(rank, world, make_id, nbytes, timeout=300.0, path=None):
    """Rank 0 draws an id (make_id() -> `nbytes` bytes) and publishes it; every rank returns the same bytes.
    The file is created exclusively with mode 0600 after removing any leftover of the same name, and carries rank 0's
    start time; a reader ignores files older than its own start (minus the launch skew it tolerates), so the id of a
    crashed earlier launch is never taken.  TimeoutError after `timeout` seconds -- no rank waits forever."""
    path = path or _id_file(world)
    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != nbytes:
            raise RuntimeError("id of %d bytes, expected %d" % (len(uid), nbytes))
        try:
            os.unlink(path)                        # a leftover (crashed launch, or planted): never reused, never followed
        except OSError:
            pass
        tmp = "%s.%d.tmp" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        try:
            os.write(fd, uid + np.float64(_PROCESS_START).tobytes())
        finally:
            os.close(fd)
        os.replace(tmp, path)                      # atomic: a reader sees no file or all of it
        return uid
    t0 = time.time()
    while True:
        data = _read_id_file(path, nbytes)
        if data is not None:
            return data
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d of %d: no RCCL id from rank 0 in %s after %.0f s (is rank 0 running? do all ranks "
                               "share MASTER_ADDR / MASTER_PORT, or HG_COMM_ID_FILE?)" % (rank, world, path, timeout))
        time.sleep(0.02)


def _read_id_file(path, nbytes):
    """The id rank 0 published, or None.  Only a regular file of THIS user with mode 0600 counts (what rank 0 creates:
    anything else at that path was planted or is debris), never through a symlink, and only if both its mtime and the
    start time rank 0 wrote into it are no older than this process minus the tolerated launch skew."""
    try:
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    except OSError:
        return None
    try:
        st = os.fstat(fd)
        import stat as _stat
        if not _stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o777) != 0o600:
            return None
        if st.st_mtime < _PROCESS_START - _LAUNCH_SKEW or st.st_size != nbytes + 8:
            return None
        data = os.read(fd, nbytes + 8)
    except OSError:
        return None
    finally:
        os.close(fd)
    if len(data) != nbytes + 8:
        return None
    if float(np.frombuffer(data[nbytes:], np.float64)[0]) < _PROCESS_START - _LAUNCH_SKEW:
        return None
    return data[:nbytes]


def init_rccl(ctx, rank=None, world=None, timeout=300.0):
    """Create the context's RCCL communicator from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT
    as `python -m torch.distributed.run` or any torchrun-like launcher sets them; one node).  Rank 0 draws the unique
    id (hg_comm_unique_id) and publishes its 128 bytes through exchange_id().  -> RcclComm"""
    from . import _native
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    path = _id_file(world)
    uid = exchange_id(rank, world, _native.comm_unique_id, _native.COMM_ID_BYTES, timeout, path)
    ctx.comm_init(uid, rank, world)
    comm = RcclComm(ctx)
    comm.barrier()                                 # everybody has read the id: rank 0 may remove the file
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
    return comm


_LAUNCH_SKEW = 120.0           # seconds one rank of a launch may start before another (fresh boxes page the image in for a minute
                               # or two, rank by rank).  A relaunch on the same port sooner than this after a CRASHED launch
                               # should set HG_COMM_NONCE, or HG_COMM_ID_FILE in a private directory as bench.py does
_PROCESS_START = time.time()


class LocalComm:
    """G virtual ranks inside one process (threads), each with its own context on the SAME GPU: the gather is G
    device-to-device copies into the rank's own scratch (hg_scratch / hg_memcpy_dtod).  Tests only."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.slots = [None] * world
            self.barrier = threading.Barrier(world)

    def __init__(self, shared, rank):
        self._s = shared
        self.rank = rank
        self.world = shared.world
        self.ctx = None                # set by the owner: the context this rank copies with
        self._slot = 0

    @classmethod
    def create(cls, world):
        sh = cls._Shared(world)
        return [cls(sh, r) for r in range(world)]

    def all_gather(self, buf):
        self._s.slots[self.rank] = buf          # the producing stage has synchronised (stage_sync = 1)
        self._s.barrier.wait()
        n = buf.nbytes
        slot = self._slot
        self._slot = (slot + 1) % 4
        base = self.ctx.scratch(slot, n * self.world)
        for r, src in enumerate(self._s.slots):
            assert src.nbytes == n
            self.ctx.memcpy_dtod(base + r * n, src.ptr, n)
        self._s.barrier.wait()                  # every rank has copied: the sources may be overwritten
        return DevBuf(base, n * self.world)

    def all_to_all(self, buf):
        self._s.slots[self.rank] = buf
        self._s.barrier.wait()
        n = buf.nbytes // self.world
        slot = self._slot
        self._slot = (slot + 1) % 4
        base = self.ctx.scratch(slot, buf.nbytes)
        for r, src in enumerate(self._s.slots):
            assert src.nbytes == buf.nbytes
            self.ctx.memcpy_dtod(base + r * n, src.ptr + self.rank * n, n)      # rank r's block for me
        self._s.barrier.wait()
        return DevBuf(base, buf.nbytes)

    def barrier(self):
        self._s.barrier.wait()

    def all_gather_host(self, arr):
        self._s.slots[self.rank] = np.ascontiguousarray(arr)
        self._s.barrier.wait()
        out = np.stack(self._s.slots)
        self._s.barrier.wait()
        return out


# ------------------------------------------------------------------ HIP shard engine
class HipShardEngine:
    """The staged C ABI of one context (one shard on one GPU), speaking raw device buffers."""

    def __init__(self, ctx, want_lists=False, async_stages=False):
        """async_stages: stop synchronising per stage -- kernels and RCCL collectives are ordered on the context's
        one stream, and the bet's verdict is read with the final download.  Use with RcclComm."""
        self.ctx = ctx
        ctx.set_option("staged_lists", 1 if want_lists else 0)
        if async_stages:
            ctx.set_option("stage_sync", 0)
            ctx.set_option("defer_verdict", 1)      # the bet's verdict is read with the final download

    @staticmethod
    def _ptr(buf):
        return buf.ptr if buf is not None else None

    def hist(self):
        self.ctx.hist()
        return DevBuf(*self.ctx.hist_buffer())

    def plan(self, R, gathered, world, rank):
        self.ctx.plan(R, self._ptr(gathered) if world > 1 else None, world, rank)

    def select_match(self):
        self.ctx.select()
        return self.match_bits()

    def match_bits(self):
        self.ctx.match()
        return DevBuf(*self.ctx.match_buffer())

    # -- optimistic sequence (one pass over the pairs) --------------------------------
    def bet_eligible(self, R, world):
        return self.ctx.bet_eligible(R, world)

    def widen_slices(self):
        """After a lost bet: eight times the slice capacity (hg_set_option "cap_boost"), twice at most; False -- and the
        ordinary capacity back -- when that has been tried, or when wider slices cannot help: the owner-routed guess met a
        query whose cut lies beyond the planes it exchanges (stat "cut_beyond_planes"; such a query takes every row).  The
        losses are the same on every rank, so is this."""
        boost = self.ctx.get_stat("cap_boost")
        if self.ctx.get_stat("cut_beyond_planes"):
            self.ctx.set_option("cap_boost", 1)
            return False
        if boost >= 64:
            self.ctx.set_option("cap_boost", 1)
            return False
        # (raising "cap_boost" also tells the library that the lost attempt is retried within the same call: it does not count
        # towards "two calls in a row lost their bets" -- hg_bet_eligible -- only the attempt a call gives up on does)
        self.ctx.set_option("cap_boost", boost * 8)
        return True

    def ranked_merge_ok(self, world):
        """hg_merge_ranked's limits (shard-independent): lane r <-> shard r, and the four queries of a block keep
        their [G][b+1] record counts in LDS."""
        return world <= 64 and 4 * world * (self.ctx.b + 1) * 4 <= 160 * 1024

    def sample_hist(self, R):
        self.ctx.sample_hist(R)
        return DevBuf(*self.ctx.hist_buffer())

    def guess(self, R, gathered, world, rank):
        self.ctx.guess(R, self._ptr(gathered), world, rank)

    def select_candidates(self):
        self.ctx.select_candidates()
        return DevBuf(*self.ctx.hist_buffer())

    def rank_candidates(self, gathered, world, rank):
        return self.ctx.rank(self._ptr(gathered), world, rank)

    def select_ranked(self):
        """Select with the shared guess and rank this shard's records; -> (record counts, local match bitmap)."""
        self.ctx.select_ranked()
        return DevBuf(*self.ctx.hist_buffer()), DevBuf(*self.ctx.match_buffer())

    def merge_ranked(self, gathered_hist, gathered_bits, world):
        return self.ctx.merge_ranked(self._ptr(gathered_hist), self._ptr(gathered_bits), world)

    def merge_ap_part(self, gathered_hist, gathered_bits, world, rank):
        """The merge and the AP of THIS rank's share of the queries only (shard_bounds(Q, world)[rank]) -> its part
        (a device buffer of the same size on every rank) for the all-gather: hg_merge_ap_part."""
        bounds = shard_bounds(self.ctx.Q, world)
        width = max(n for _, n in bounds)
        q0, nq = bounds[rank]
        return DevBuf(*self.ctx.merge_ap_part(self._ptr(gathered_hist), self._ptr(gathered_bits), world, q0, nq, width))

    def unpack_parts(self, gathered_parts, world):
        """-> (ap, rel, lost) of ALL queries from the gathered parts; synchronises (the step's one download)."""
        width = max(n for _, n in shard_bounds(self.ctx.Q, world))
        return self.ctx.unpack_parts(gathered_parts.ptr, world, width)

    # -- the exchanges routed by query owner (all-to-all): each returns the [world] blocks to send
    def pack_sample_by_owner(self, world):
        p, n = self.ctx.pack_sample_by_owner(world)
        return DevBuf(p, n * world)

    def guess_owned(self, R, received, world, rank):
        p, n = self.ctx.guess_owned(R, received.ptr, world, rank)
        return DevBuf(p, n * world)

    def guess_finish(self, R, answers, world, rank):
        self.ctx.guess_finish(R, answers.ptr, world, rank)

    def pack_ranked_by_owner(self, world):
        p, n = self.ctx.pack_ranked_by_owner(world)
        return DevBuf(p, n * world)

    def merge_ap_owned(self, received, world, rank):
        return DevBuf(*self.ctx.merge_ap_owned(received.ptr, world, rank))

    def shard_step(self, R):
        """The owner-routed bet as one library call over the context's own communicator -> (ap, rel, lost | None): hg_shard_step."""
        return self.ctx.shard_step(R)

    def verdict(self):
        """True if a deferred bet (rank_candidates returned None) turned out lost."""
        return self.ctx.bet_verdict()

    def finish(self, gathered_bits, world):
        if gathered_bits is not None:
            self.ctx.merge_match(gathered_bits.ptr, world)
        self.ctx.ap()
        return self.ctx.get_ap()                  # synchronises

    def topr_buffers(self):
        pi, pd, n = self.ctx.topr_buffers()
        return DevBuf(pi, n * 4), DevBuf(pd, n)

    def merge_topr(self, gathered_idx, gathered_dist, world):
        self.ctx.merge_topr(gathered_idx.ptr, gathered_dist.ptr, world)
        return self.ctx.get_topr()


# ------------------------------------------------------------------ orchestration
def evaluate_shard(engine, comm, R, gather_topr=False, always_gather=False, bet=True, route_by_owner=True, one_call=True):
    """Run one rank's part of the sharded evaluation.

    engine: HipShardEngine (or any object with the same five methods -- the CPU
    tests drive this very function with a NumPy engine over gloo).
    route_by_owner: the bet's exchanges as all-to-alls by query owner when engine and communicator can (else, or with
    False, the all-gather form: same results).  Over a real RCCL communicator with more than one rank the routed form has
    to EARN that default: the first call on a communicator runs the all-gather form (ncclAllGather only), then the routed
    form (grouped ncclSend / ncclRecv), compares the two per-query results bit for bit, and the ranks agree (one
    all-reduce) whether anyone saw an error or a difference -- only then do later calls route by owner
    (`comm.routed_verified`; HG_ROUTE_BY_OWNER=0 / 1 skips the check and forces the form).
    one_call: over an RcclComm the routed bet runs as ONE library call (hg_shard_step: every stage and exchange enqueued back to back);
    False keeps the same stages driven from here, call by call -- the form the tests hold the one-call form against.
    Returns (ap [Q] float64 with nan for skipped queries, rel [Q] int64) -- and
    (idx, dist) of the merged global top-R when gather_topr is set.
    """
    if (route_by_owner and bet and not gather_topr and isinstance(comm, RcclComm) and comm.world > 1
            and hasattr(engine, "merge_ap_owned")):
        forced = os.environ.get("HG_ROUTE_BY_OWNER")
        if forced in ("0", "1"):
            route_by_owner = forced == "1"
        elif comm.routed_verified is None:
            ref = _evaluate_shard(engine, comm, R, False, always_gather, bet, False, one_call)
            bad = 0.0
            try:
                got = _evaluate_shard(engine, comm, R, False, always_gather, bet, True, one_call)
                if not (np.array_equal(got[0], ref[0], equal_nan=True) and np.array_equal(got[1], ref[1])):
                    bad = 1.0
            except Exception:      # noqa: BLE001 -- an hg_alltoall failure on this rank: every rank must learn of it
                bad = 2.0
            comm.routed_verified = comm.allreduce_max(bad) == 0.0
            return ref
        else:
            route_by_owner = comm.routed_verified
    return _evaluate_shard(engine, comm, R, gather_topr, always_gather, bet, route_by_owner, one_call)


def _evaluate_shard(engine, comm, R, gather_topr, always_gather, bet, route_by_owner, one_call=True):
    multi = comm.world > 1 or always_gather          # always_gather: exercise the collectives even with one rank
    gather = (lambda t: comm.all_gather(t)) if multi else (lambda t: None)
    bits = None
    if (bet and not gather_topr and hasattr(engine, "select_ranked") and engine.ranked_merge_ok(comm.world)
            and engine.bet_eligible(R, comm.world)):
        # the bet with one record pass and one exchange after the guess: every shard ranks its own records, the
        # global bitmap is stitched from the gathered local ones (hg_merge_ranked)
        routed = route_by_owner and multi and hasattr(engine, "merge_ap_owned") and hasattr(comm, "all_to_all")
        while True:
            if routed and one_call and isinstance(comm, RcclComm) and hasattr(engine, "shard_step"):
                # the same owner-routed sequence as ONE library call: every stage and every RCCL exchange enqueued back to back
                # on the context's stream, one synchronisation at the final download (hg_shard_step)
                ap, rel, lost = engine.shard_step(R)
                if lost is False:
                    return ap, rel
                if lost is None or not (hasattr(engine, "widen_slices") and engine.widen_slices()):
                    break
                continue
            if routed:
                # every table goes only to the owner of its queries (all-to-all): the sampled histograms to the rank that
                # guesses for them, its answers back, the record counts + local bitmaps to the rank that stitches and
                # evaluates them -- a GPU receives its own queries' share of each table, not all of it
                engine.sample_hist(R)
                answers = engine.guess_owned(R, comm.all_to_all(engine.pack_sample_by_owner(comm.world)), comm.world, comm.rank)
                engine.guess_finish(R, comm.all_to_all(answers), comm.world, comm.rank)
                engine.select_ranked()
                part = engine.merge_ap_owned(comm.all_to_all(engine.pack_ranked_by_owner(comm.world)), comm.world, comm.rank)
                ap, rel, lost = engine.unpack_parts(comm.all_gather(part), comm.world)
                if not lost:
                    return ap, rel
                if not (hasattr(engine, "widen_slices") and engine.widen_slices()):
                    break
                continue
            engine.guess(R, gather(engine.sample_hist(R)), comm.world, comm.rank)
            h, b = engine.select_ranked()
            if multi and hasattr(engine, "merge_ap_part"):
                # the per-query stages split over the ranks: each merges and evaluates its own share of the queries, a third,
                # tiny all-gather (16 bytes per query) brings every rank all APs and the verdict (the same on every rank)
                part = engine.merge_ap_part(gather(h), gather(b), comm.world, comm.rank)
                ap, rel, lost = engine.unpack_parts(comm.all_gather(part), comm.world)
                if not lost:
                    return ap, rel
            else:
                lost = engine.merge_ranked(gather(h), gather(b), comm.world)
                if not lost:                              # held, or verdict deferred
                    ap, rel = engine.finish(None, comm.world)
                    if lost is not None or not engine.verdict():
                        return ap, rel
            # lost, on every rank alike.  A database stored class by class piles a query's near rows into a few slices of
            # one shard, however good the cut: widen the slices (x8, x64; remembered for the next call) before giving up
            if not (hasattr(engine, "widen_slices") and engine.widen_slices()):
                break
        bet = False                                   # exact sequence below
    if bet and hasattr(engine, "bet_eligible") and engine.bet_eligible(R, comm.world):
        # one pass over the pairs: sampled histograms -> shared guess -> candidate records ->
        # exact record histograms -> shared exact plan.  `lost` is the same on every rank.
        engine.guess(R, gather(engine.sample_hist(R)), comm.world, comm.rank)
        lost = engine.rank_candidates(gather(engine.select_candidates()), comm.world, comm.rank)
        deferred = lost is None                      # engine does not wait for the verdict: carry on as if the bet held
        if not lost:
            bits = engine.match_bits()
    else:
        deferred = False
    if bits is None:                                 # exact two-pass sequence
        engine.plan(R, gather(engine.hist()), comm.world, comm.rank)
        bits = engine.select_match()
    B = gather(bits)
    if deferred and not gather_topr:
        ap, rel = engine.finish(B, comm.world)
        if not engine.verdict():                     # the same on every rank (computed from gathered data)
            return ap, rel
        engine.plan(R, gather(engine.hist()), comm.world, comm.rank)      # lost after all: exact sequence
        B = gather(engine.select_match())
    elif deferred and engine.verdict():
        engine.plan(R, gather(engine.hist()), comm.world, comm.rank)
        B = gather(engine.select_match())
    lists = None
    if gather_topr:
        if multi and isinstance(comm, RcclComm):
            engine.ctx.allgather_topr()               # the north star's exchange, inside the library
            lists = engine.ctx.get_topr()
        elif multi:
            ti, td = engine.topr_buffers()
            lists = engine.merge_topr(comm.all_gather(ti), comm.all_gather(td), comm.world)
        else:
            lists = engine.ctx.get_topr()
    ap, rel = engine.finish(B, comm.world)
    return (ap, rel, lists) if gather_topr else (ap, rel)


def evaluate_real_queries(ctx, comm, q_feats, q_labels, R):
    """Real-valued (float32 inner-product) ranking on G GPUs.  This path splits the QUERIES, not the database: every
    rank holds the whole float table (256 MB at N = 1M x 64 features -- hg_set_database_f32 on each context before the
    call) and ranks its contiguous share of the queries against it, so nothing has to be merged; the only exchange is
    the all-gather of 16 bytes per query (AP, relevant count).  -> (ap[Q] float64, rel[Q] int64), identical on every
    rank and equal to the one-GPU hg_map_real bit for bit (a query's result does not depend on the others)."""
    q_feats, q_labels = np.asarray(q_feats), np.asarray(q_labels)
    Q = q_feats.shape[0]
    bounds = shard_bounds(Q, comm.world)
    lo, n = bounds[comm.rank]
    width = max(b[1] for b in bounds)
    mine = np.zeros((width, 2), np.float64)
    if n:
        ctx.set_queries_f32(np.ascontiguousarray(q_feats[lo:lo + n], np.float32), np.ascontiguousarray(q_labels[lo:lo + n], np.int64))
        ap, rel = ctx.map_real(R)
        mine[:n, 0] = ap
        mine[:n, 1] = rel                          # a count <= R: exact in a double
    every = comm.all_gather_host(mine)
    ap = np.concatenate([every[r, :bounds[r][1], 0] for r in range(comm.world)])
    rel = np.concatenate([every[r, :bounds[r][1], 1] for r in range(comm.world)]).astype(np.int64)
    return ap, rel


def evaluate_query_split(ctx, comm, q_code_words, q_label_words, R):
    """The OTHER decomposition of the path over G GPUs: queries are independent (lib/metric.py:16 loops over them), so every
    rank holds the WHOLE packed database (16 bytes per row: 160 MB at N = 10M, nothing for a 288 GB part -- hg_set_database on
    each context before the call) and evaluates its contiguous share of the queries with the one-GPU sequence (hg_map).  No
    data-path collective at all: the only exchange is the all-gather of 16 bytes per query (AP, hit count).
    -> (ap [Q] float64, rel [Q] int64), identical on every rank and equal to one GPU's hg_map bit for bit (a query's result
    does not depend on the other queries).  Database sharding (evaluate_shard) is what BASELINE.json's north star
    prescribes and what `bench.py --gpus G` reports as `value`; this form is reported beside it (`query_split`)."""
    q_code_words, q_label_words = np.asarray(q_code_words), np.asarray(q_label_words)
    Q = q_code_words.shape[0]
    bounds = shard_bounds(Q, comm.world)
    lo, n = bounds[comm.rank]
    width = max(b[1] for b in bounds)
    mine = np.zeros((width, 2), np.float64)
    if n:
        ctx.set_queries(np.ascontiguousarray(q_code_words[lo:lo + n]), np.ascontiguousarray(q_label_words[lo:lo + n]))
        ap, rel = ctx.map(R)
        mine[:n, 0] = ap
        mine[:n, 1] = rel                          # a count <= R: exact in a double
    every = comm.all_gather_host(mine)
    ap = np.concatenate([every[r, :bounds[r][1], 0] for r in range(comm.world)])
    rel = np.concatenate([every[r, :bounds[r][1], 1] for r in range(comm.world)]).astype(np.int64)
    return ap, rel


def shard_bounds(n_total, world):
    """Contiguous, near-equal index ranges: [(base, rows)] * world."""
    per, extra = divmod(int(n_total), int(world))
    out, base = [], 0
    for r in range(world):
        rows = per + (1 if r < extra else 0)
        out.append((base, rows))
        base += rows
    return out


def mean_ap(ap, rel):
    return np.mean(np.array(ap[rel != 0]))
