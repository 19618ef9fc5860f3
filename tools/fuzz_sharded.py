#!/usr/bin/env python3
"""One-off stress run on ONE GPU: the sharded sequences (hashgan_amd.sharded.evaluate_shard over G virtual ranks, threads +
device-to-device copies standing in for RCCL) against the single-context result, over random shapes."""
import sys, time, threading
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric, sharded

def virtual(qw, ql, db, dl, b, C, N, G, R, gather_topr):
    comms = sharded.LocalComm.create(G)
    results, errors = [None] * G, []
    def work(r):
        try:
            base, rows = sharded.shard_bounds(N, G)[r]
            ctx = _native.Context(0)
            ctx.set_database(metric.pack_codes(db[base:base + rows]), metric.pack_labels(dl[base:base + rows]), b, C, idx_base=base, n_total=N)
            ctx.set_queries(qw, ql)
            comms[r].ctx = ctx
            eng = sharded.HipShardEngine(ctx, want_lists=gather_topr)
            results[r] = sharded.evaluate_shard(eng, comms[r], R, gather_topr=gather_topr)
            ctx.close()
        except Exception as e:       # noqa: BLE001
            errors.append(e)
            try: comms[r]._s.barrier.abort()
            except Exception: pass
    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]; [t.join() for t in th]
    if errors: raise errors[0]
    return results

def one(seed):
    rng = np.random.default_rng(seed)
    b = int(rng.choice([8, 16, 32, 48, 64, 64, 100, 128, 200]))
    N = int(rng.integers(5000, 300000)); Q = int(rng.integers(1, 400))
    frac = float(rng.choice([0.001, 0.005, 0.02, 0.1, 0.3])); R = max(1, int(N * frac))
    C = int(rng.choice([3, 10, 81, 130])); G = int(rng.choice([2, 3, 4, 8]))
    topr = rng.random() < 0.3
    dl, _ = synth.onehot_labels(seed * 3 + 1, N, C); ql, _ = synth.onehot_labels(seed * 3 + 2, Q, C)
    if rng.random() < 0.5: db = synth.planted_codes(seed, dl, b, 0.25); qb = synth.planted_codes(seed, ql, b, 0.25)
    else: db = (rng.random((N, b)) < 0.5).astype(np.uint8); qb = (rng.random((Q, b)) < 0.5).astype(np.uint8)
    one_ctx = _native.Context(0)
    try:
        one_ctx.set_database(metric.pack_codes(db), metric.pack_labels(dl), b, C)
        qw, qlw = metric.pack_codes(qb), metric.pack_labels(ql)
        one_ctx.set_queries(qw, qlw)
        one_ctx.set_option("optimistic", 0)
        ap0, rel0 = one_ctx.map(R)
        lists0 = None
        if topr:
            one_ctx.topr(R)
            lists0 = one_ctx.get_topr()
    finally:
        one_ctx.close()
    res = virtual(qw, qlw, db, dl, b, C, N, G, R, topr)
    for r in res:
        ok = np.array_equal(r[0], ap0, equal_nan=True) and np.array_equal(r[1], rel0)
        if ok and topr: ok = np.array_equal(r[2][0], lists0[0]) and np.array_equal(r[2][1], lists0[1])
        if not ok: return "MISMATCH seed=%d b=%d N=%d Q=%d R=%d C=%d G=%d topr=%s" % (seed, b, N, Q, R, C, G, topr)
    return "ok seed=%d b=%d N=%d Q=%d R=%d C=%d G=%d topr=%s" % (seed, b, N, Q, R, C, G, topr)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0; t = time.time()
    for seed in range(s0, s0 + n):
        if "-v" in sys.argv: print("seed", seed, flush=True)
        r = one(seed)
        if r.startswith("MISMATCH"): bad += 1; print(r, flush=True)
        elif seed % 10 == 0: print(r, flush=True)
    print("done: %d shapes, %d mismatches, %.0f s" % (n, bad, time.time() - t))
