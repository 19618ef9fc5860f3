#!/usr/bin/env python3
"""Per-kernel times of the SHARDED sequence with G virtual shards on one GPU (what each rank of a G-GPU run executes,
minus the wire): C4-per-rank shape by default (N = 10M / G rows per shard, Q = 10k)."""
import sys, threading, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, sharded, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
Q, b, C, R, seed = 10000, 64, 10, 5000, 0xC4
qw = synth.random_code_words(seed + 7, Q, b)
ql = synth.onehot_label_words(seed * 3 + 2, Q, C)
comms = sharded.LocalComm.create(G)
out = [None] * G


def work(r):
    base, rows = sharded.shard_bounds(N, G)[r]
    ctx = _native.Context(0)
    ctx.set_database(synth.random_code_words(seed, rows, b, row_offset=base), synth.onehot_label_words(seed * 3 + 1, rows, C, row_offset=base),
                     b, C, idx_base=base, n_total=N)
    ctx.set_queries(qw, ql)
    comms[r].ctx = ctx
    eng = sharded.HipShardEngine(ctx, want_lists=False)
    for _ in range(2):
        sharded.evaluate_shard(eng, comms[r], R)
    ctx.timing_enable(2); ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        ap, rel = sharded.evaluate_shard(eng, comms[r], R)
    dt = (time.perf_counter() - t0) / 3
    out[r] = (dt, {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.timing_read().items()}, float(np.nanmean(ap)))
    ctx.close()


th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
[t.start() for t in th]; [t.join() for t in th]
print("G=%d N=%d: rank 0 wall %.3f ms per step (all %d shards share ONE GPU, so the wall time is not a per-rank figure)" % (G, N, out[0][0] * 1e3, G))
print("rank 0 kernels (ms per launch):", out[0][1])
print("mAP", out[0][2], "identical on all ranks:", all(o[2] == out[0][2] for o in out))
