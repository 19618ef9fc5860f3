#!/usr/bin/env python3
"""The real-valued call at the C2 shape under engine options: tools/real_ab.py [key=value ...] -> ms per call, per-kernel averages,
and the APs' equality with the unfused sequence (real_fused_scores=0)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native
rng = np.random.default_rng(0)
Q, N, b, R = 10000, 1000000, 64, 5000
dbf = np.tanh(rng.standard_normal((N, b))).astype(np.float32); qf = np.tanh(rng.standard_normal((Q, b))).astype(np.float32)
dl = np.zeros((N, 10), np.int64); dl[np.arange(N), rng.integers(0, 10, N)] = 1
ql = np.zeros((Q, 10), np.int64); ql[np.arange(Q), rng.integers(0, 10, Q)] = 1
variants = [[]] + [[a.split("=")] for a in sys.argv[1:]]
ref = None
for opts in variants:
    ctx = _native.Context(0)
    for k, v in opts: ctx.set_option(k, int(v))
    ctx.set_database_f32(dbf, dl); ctx.set_queries_f32(qf, ql)
    for _ in range(2): ap, rel = ctx.map_real(R)
    ctx.timing_enable(2); ctx.timing_reset()
    t = time.perf_counter()
    for _ in range(4): ap, rel = ctx.map_real(R)
    dt = (time.perf_counter() - t) / 4
    tm = ctx.timing_read()
    if ref is None: ref = ap.copy()
    print("%-28s %.2f ms per call  equal_to_first=%s attempts=%d lds_ranked=%d | %s" % (opts, dt * 1e3, bool(np.array_equal(ap, ref, equal_nan=True)),
          ctx.get_stat("real_attempts"), (ctx.get_stat("real_path") >> 1) & 1, {k: round(v[0] / max(v[1], 1), 3) for k, v in tm.items() if v[1]}), flush=True)
    ctx.close()
