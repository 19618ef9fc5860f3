#!/usr/bin/env python3
"""Fixed cost of one hg_map call: a problem so small that the kernels take ~0.1 ms in total."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric
Q, N, b, R = 64, 65536, 64, 100
dl, _ = synth.onehot_labels(1, N, 10)
ql, _ = synth.onehot_labels(2, Q, 10)
dw = synth.splitmix64(3, N).reshape(N, 1)
qw = synth.splitmix64(4, Q).reshape(Q, 1)
ctx = _native.Context(0)
ctx.set_database(dw, metric.pack_labels(dl), b, 10)
ctx.set_queries(qw, metric.pack_labels(ql))
for _ in range(20): ctx.map(R)
ctx.timing_enable(2); ctx.timing_reset()
for _ in range(50): ctx.map(R)
tm = ctx.timing_read()
ksum = sum(v[0] / max(v[1], 1) for v in tm.values())
ctx.timing_enable(0)
t = time.perf_counter()
for _ in range(500): ctx.map(R)
dt = (time.perf_counter() - t) / 500
print("per call %.1f us, kernels %.1f us, rest %.1f us; bet=%d" % (dt * 1e6, ksum * 1e3, dt * 1e6 - ksum * 1e3, ctx.get_stat("last_optimistic")))
print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in tm.items()})
