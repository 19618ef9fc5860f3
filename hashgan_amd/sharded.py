"""Database-sharded evaluation: one shard per GPU, one process per GPU.

The database is split into contiguous index ranges (shard r owns rows
[base_r, base_r + N_r)); every rank holds all queries.  Exchange steps, both
all-gathers of small per-query tables (RCCL over xGMI via torch.distributed
when the tensors live on the GPU, gloo on CPU tensors in the tests):

  1. per-shard distance histograms  uint32 [b+1][Qpad]   (2.6 MB at C2)
     -> every rank derives the SAME global threshold t, tie quota, and the
        global rank positions of its own rows (k_plan): no comparison-based
        merge is ever needed, shards interleave by (distance, shard, index).
  2. per-shard label-match bit rows in global position space, uint64
     [Q][ceil(R/64)] (6.3 MB at C2), disjoint between shards -> OR -> AP.

`gather_topr` additionally all-gathers the ranked (idx, dist) lists themselves
(the exchange BASELINE.json's north star names) for callers that want them.

The reference has no counterpart (lib/metric.py runs in one process); the
result is bit-identical to the single-GPU path, which the tests check with
virtual shards on one GPU.
"""
import threading

import numpy as np


# ------------------------------------------------------------------ communicators
class TorchComm:
    """all_gather over a torch.distributed process group (nccl = RCCL, or gloo)."""

    def __init__(self, group=None, host_sync=True):
        """host_sync=False: the engine runs on torch's current stream (HipShardEngine(share_stream=True)),
        collectives are stream-ordered with its kernels and nothing waits on the host."""
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.host_sync = host_sync
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather(self, t):
        import torch
        flat = t.contiguous().view(-1)
        if t.is_cuda and self._dist.get_backend(self.group) == "gloo":
            # functional fallback (dry runs of the multi-rank path on one GPU): stage through the host
            host = torch.empty(self.world * flat.numel(), dtype=t.dtype)
            self._dist.all_gather_into_tensor(host, flat.cpu(), group=self.group)
            return host.to(t.device).view((self.world,) + tuple(t.shape))
        out = torch.empty(self.world * flat.numel(), dtype=t.dtype, device=t.device)
        self._dist.all_gather_into_tensor(out, flat, group=self.group)
        if t.is_cuda and self.host_sync:
            torch.cuda.synchronize(t.device)       # the engine's own stream reads `out` next
        return out.view((self.world,) + tuple(t.shape))

    def barrier(self):
        self._dist.barrier(group=self.group)


class LocalComm:
    """G virtual ranks inside one process (threads): shards of one GPU in the tests."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.slots = [None] * world
            self.barrier = threading.Barrier(world)

    def __init__(self, shared, rank):
        self._s = shared
        self.rank = rank
        self.world = shared.world

    @classmethod
    def create(cls, world):
        sh = cls._Shared(world)
        return [cls(sh, r) for r in range(world)]

    def all_gather(self, t):
        import torch
        self._s.slots[self.rank] = t
        self._s.barrier.wait()
        out = torch.stack([x for x in self._s.slots])
        self._s.barrier.wait()
        return out

    def barrier(self):
        self._s.barrier.wait()


# ------------------------------------------------------------------ HIP shard engine
class _DevView:
    """Expose a raw device allocation to torch (zero copy) via __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _as_tensor(ptr, nbytes, device):
    import torch
    try:
        return torch.as_tensor(_DevView(ptr, nbytes), device=torch.device("cuda", device))
    except RuntimeError as e:
        raise RuntimeError("torch cannot see the GPU this context runs on.  Import torch (and let it initialise "
                           "CUDA/HIP) BEFORE creating the first hashgan_amd context: a process can host only one "
                           "HIP runtime, and torch bundles its own.  (%s)" % e) from e


class HipShardEngine:
    """The staged C ABI of one context (one shard on one GPU), speaking torch tensors."""

    def __init__(self, ctx, want_lists=False, share_stream=False):
        """share_stream: put the context on torch's current stream and stop synchronising per stage --
        use with TorchComm(host_sync=False)."""
        self.ctx = ctx
        ctx.set_option("staged_lists", 1 if want_lists else 0)
        self._keep = []                           # gathered tensors stay alive until the step's last kernel ran
        if share_stream:
            import torch
            ctx.set_stream(torch.cuda.current_stream(torch.device("cuda", ctx.device)).cuda_stream)
            ctx.set_option("stage_sync", 0)
            ctx.set_option("defer_verdict", 1)      # the bet's verdict is read with the final download

    def hist(self):
        self.ctx.hist()
        ptr, n = self.ctx.hist_buffer()
        return _as_tensor(ptr, n, self.ctx.device)

    def plan(self, R, gathered, world, rank):
        self._keep = [gathered]
        self.ctx.plan(R, gathered.data_ptr() if (world > 1 and gathered is not None) else None, world, rank)

    def select_match(self):
        self.ctx.select()
        return self.match_bits()

    def match_bits(self):
        self.ctx.match()
        ptr, n = self.ctx.match_buffer()
        return _as_tensor(ptr, n, self.ctx.device)

    # -- optimistic sequence (one pass over the pairs) --------------------------------
    def bet_eligible(self, R, world):
        return self.ctx.bet_eligible(R, world)

    def sample_hist(self, R):
        self.ctx.sample_hist(R)
        ptr, n = self.ctx.hist_buffer()
        return _as_tensor(ptr, n, self.ctx.device)

    def guess(self, R, gathered, world, rank):
        self._keep = [gathered]
        self.ctx.guess(R, gathered.data_ptr() if gathered is not None else None, world, rank)

    def select_candidates(self):
        self.ctx.select_candidates()
        ptr, n = self.ctx.hist_buffer()
        return _as_tensor(ptr, n, self.ctx.device)

    def rank_candidates(self, gathered, world, rank):
        self._keep.append(gathered)
        return self.ctx.rank(gathered.data_ptr() if gathered is not None else None, world, rank)

    def select_ranked(self):
        """Select with the shared guess and rank this shard's records; -> (record counts, local match bitmap)."""
        self.ctx.select_ranked()
        ph, nh = self.ctx.hist_buffer()
        pb, nb = self.ctx.match_buffer()
        return _as_tensor(ph, nh, self.ctx.device), _as_tensor(pb, nb, self.ctx.device)

    def merge_ranked(self, gathered_hist, gathered_bits, world):
        self._keep += [gathered_hist, gathered_bits]
        return self.ctx.merge_ranked(gathered_hist.data_ptr() if gathered_hist is not None else None,
                                     gathered_bits.data_ptr() if gathered_bits is not None else None, world)

    def verdict(self):
        """True if a deferred bet (rank_candidates returned None) turned out lost."""
        return self.ctx.bet_verdict()

    def finish(self, gathered_bits, world):
        if gathered_bits is not None:
            self._keep.append(gathered_bits)
            self.ctx.merge_match(gathered_bits.data_ptr(), world)
        self.ctx.ap()
        out = self.ctx.get_ap()                   # synchronises
        self._keep = []
        return out

    def topr_tensors(self):
        pi, pd, n = self.ctx.topr_buffers()
        return _as_tensor(pi, n * 4, self.ctx.device), _as_tensor(pd, n, self.ctx.device)

    def merge_topr(self, gathered_idx, gathered_dist, world):
        self.ctx.merge_topr(gathered_idx.data_ptr(), gathered_dist.data_ptr(), world)
        return self.ctx.get_topr()


# ------------------------------------------------------------------ orchestration
def evaluate_shard(engine, comm, R, gather_topr=False, always_gather=False, bet=True):
    """Run one rank's part of the sharded evaluation.

    engine: HipShardEngine (or any object with the same five methods -- the CPU
    tests drive this very function with a NumPy engine over gloo).
    Returns (ap [Q] float64 with nan for skipped queries, rel [Q] int64) -- and
    (idx, dist) of the merged global top-R when gather_topr is set.
    """
    multi = comm.world > 1 or always_gather          # always_gather: exercise the collectives even with one rank
    gather = (lambda t: comm.all_gather(t)) if multi else (lambda t: None)
    bits = None
    if (bet and not gather_topr and hasattr(engine, "select_ranked") and comm.world <= 64
            and engine.bet_eligible(R, comm.world)):
        # the bet with one record pass and one exchange after the guess: every shard ranks its own records, the
        # global bitmap is stitched from the gathered local ones (hg_merge_ranked)
        engine.guess(R, gather(engine.sample_hist(R)), comm.world, comm.rank)
        h, b = engine.select_ranked()
        lost = engine.merge_ranked(gather(h), gather(b), comm.world)
        if not lost:                                  # held, or verdict deferred
            ap, rel = engine.finish(None, comm.world)
            if lost is not None or not engine.verdict():
                return ap, rel
        bet = False                                   # lost (the same on every rank): exact sequence below
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
        ti, td = engine.topr_tensors()
        if multi:
            lists = engine.merge_topr(comm.all_gather(ti), comm.all_gather(td), comm.world)
        else:
            lists = engine.ctx.get_topr()
    ap, rel = engine.finish(B, comm.world)
    return (ap, rel, lists) if gather_topr else (ap, rel)


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
