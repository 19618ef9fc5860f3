#!/usr/bin/env python3
"""Where the wall time of one bench step goes on the host side (C2): the hg_map call vs the Python around it."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from hashgan_amd import _native, metric
spec = bench.WORKLOADS["c2"]
c = bench.build_inputs(spec)
dw, dl = metric.pack_codes(c["dbbits"]), metric.pack_labels(c["dblab"])
qw, ql = metric.pack_codes(c["qbits"]), metric.pack_labels(c["qlab"])
ctx = _native.Context(0)
ctx.set_database(dw, dl, c["b"], spec["C"]); ctx.set_queries(qw, ql)
R = c["R"]
for _ in range(5): ctx.map(R)
n = 200
t_map = t_mean = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a0 = time.perf_counter()
    a, r = ctx.map(R)
    a1 = time.perf_counter()
    m = metric.mean_over_hits(a, r)
    a2 = time.perf_counter()
    t_map += a1 - a0; t_mean += a2 - a1
tot = (time.perf_counter() - t0) / n
print("step %.1f us: hg_map call %.1f us, mean_over_hits %.1f us, loop overhead %.1f us" % (tot * 1e6, t_map / n * 1e6, t_mean / n * 1e6, (tot - (t_map + t_mean) / n) * 1e6))
