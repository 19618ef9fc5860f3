#!/usr/bin/env python3
"""What ONE rank of a G-GPU run executes, alone on the GPU: the sharded sequence on a 10M / G-row shard of C4 with a
communicator whose all_gather hands back G copies of this rank's own buffer (device-to-device copies on the context's
stream stand in for the wire).  i.i.d. shards are statistically alike, so G copies of one shard's histograms / counts /
bitmaps cost the merge stages what the true gathered tables would -- the mAP is NOT the database's, the stage times are.
Prints wall time per step (one stream, no host waits between stages) and the per-kernel times.

    python tools/replica_shard_timing.py [G ...]      (each G through hg_shard_step; then G = 8 once more stage by stage from Python)
"""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, sharded, synth

N, Q, b, C, R, seed = 10_000_000, 10000, 64, 10, 5000, 0xC4


class ReplicaComm:
    def __init__(self, ctx, world):
        self.ctx, self.world, self.rank, self._slot = ctx, world, 0, 0

    def all_gather(self, buf):
        slot = self._slot
        self._slot = (slot + 1) % 4
        base = self.ctx.scratch(slot, buf.nbytes * self.world)
        for r in range(self.world):
            self.ctx.memcpy_dtod(base + r * buf.nbytes, buf.ptr, buf.nbytes)
        return sharded.DevBuf(base, buf.nbytes * self.world)

    def all_to_all(self, buf):
        """[world] equal blocks, block r for rank r -> [world] blocks, block r FROM rank r: every peer is a replica of this rank,
        so each of them sends what this rank would send to itself -- block `rank` of its own buffer."""
        slot = self._slot
        self._slot = (slot + 1) % 4
        per = buf.nbytes // self.world
        base = self.ctx.scratch(slot, buf.nbytes)
        for r in range(self.world):
            self.ctx.memcpy_dtod(base + r * per, buf.ptr + self.rank * per, per)
        return sharded.DevBuf(base, buf.nbytes)

    def barrier(self):
        self.ctx.synchronize()


def one(G, steps=10, opts=(), one_call=True):
    rows = N // G
    ctx = _native.Context(0)
    try:
        for k_, v_ in opts:
            ctx.set_option(k_, v_)
        ctx.set_database(synth.random_code_words(seed, rows, b), synth.onehot_label_words(seed * 3 + 1, rows, C), b, C, idx_base=0, n_total=N)
        ctx.set_queries(synth.random_code_words(seed + 7, Q, b), synth.onehot_label_words(seed * 3 + 2, Q, C))
        comm = ReplicaComm(ctx, G)
        eng = sharded.HipShardEngine(ctx, want_lists=False, async_stages=True)
        if one_call:                                  # hg_shard_step: the whole step one library call, the replica exchange inside the library
            def step():
                ap, rel, lost = ctx.shard_step(R, replica_world=G)
                assert lost is False, lost
        else:                                         # the same stages driven from Python (sharded.evaluate_shard), ReplicaComm's copies
            def step():
                sharded.evaluate_shard(eng, comm, R, always_gather=True)
        for _ in range(3):
            step()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ctx.timing_enable(2); ctx.timing_reset()
        for _ in range(3):
            step()
        ctx.synchronize()
        k = {n: round(v[0] / max(v[1], 1), 4) for n, v in ctx.timing_read().items()}
        ctx.timing_enable(False)
        print("%s G=%d  shard rows %d  step %.3f ms   kernels (ms per launch, timed separately): %s   [bets %d, lost %d, rank variant %d, segments %d]"
              % ("hg_shard_step" if one_call else "staged (Python)", G, rows, dt * 1e3, k, ctx.get_stat("optimistic_runs"), ctx.get_stat("optimistic_fallbacks"), ctx.get_stat("rank_variant"), ctx.get_stat("segments")), flush=True)
    finally:
        ctx.close()


if __name__ == "__main__":
    opts = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
    for G in ([int(x) for x in sys.argv[1:] if "=" not in x] or [1, 2, 4, 8]):
        one(G, opts=opts, one_call=True)
    one(8, opts=opts, one_call=False)
    print("# (exchanges routed by query owner: hg_alltoall stands in as G device copies of this rank's own block)")
