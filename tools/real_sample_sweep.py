#!/usr/bin/env python3
"""Real-valued path at the C2 shape against the sample size of its bet (real_sample_hits)."""
import numpy as np, sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native
rng = np.random.default_rng(0)
Q, N, b, R = 10000, 1000000, 64, 5000
dbf = np.tanh(rng.standard_normal((N, b))).astype(np.float32); qf = np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
dl = np.zeros((N, 10), np.int64); dl[np.arange(N), rng.integers(0, 10, N)] = 1
ql = np.zeros((Q, 10), np.int64); ql[np.arange(Q), rng.integers(0, 10, Q)] = 1
ctx = _native.Context(0); ctx.set_database_f32(dbf, dl); ctx.set_queries_f32(qf, ql)
ref = None
for hits in [64, 96, 128, 192, 256, 384]:
    ctx.set_option("real_sample_hits", hits)
    for _ in range(2): ap, rel = ctx.map_real(R)
    if ref is None: ref = ap
    t = time.perf_counter()
    for _ in range(4): ctx.map_real(R)
    dt = (time.perf_counter() - t) / 4
    ctx.timing_enable(2); ctx.timing_reset()
    for _ in range(2): ctx.map_real(R)
    k = {n: round(v[0] / max(v[1], 1), 3) for n, v in ctx.timing_read().items()}; ctx.timing_enable(0)
    print("hits=%-4d %.2f ms  same AP %s  attempts %d  %s" % (hits, dt * 1e3, bool(np.array_equal(ap, ref, equal_nan=True)), ctx.get_stat("real_attempts"), k), flush=True)
ctx.close()
