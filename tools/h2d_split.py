#!/usr/bin/env python3
"""Where the drop-in call from host arrays spends its time at C2: hg_set_database_f32 (host pack + upload),
hg_set_queries_f32, hg_map -- and the packing alone for several thread counts."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from hashgan_amd import _native
spec = bench.WORKLOADS["c2"]
c = bench.build_inputs(spec)
dbf = c["dbbits"].astype(np.float32) * 2 - 1; dbl = c["dblab"].astype(np.int64)
qf = c["qbits"].astype(np.float32) * 2 - 1; ql = c["qlab"].astype(np.int64)
ctx = _native.Context(0)
ctx.set_option("keep_floats", 2)
R = spec["R"]
def once():
    t0 = time.perf_counter(); ctx.set_database_f32(dbf, dbl)
    t1 = time.perf_counter(); ctx.set_queries_f32(qf, ql)
    t2 = time.perf_counter(); ctx.map(R)
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for th in [0, 32, 64, 96, 128, 192]:
    ctx.set_option("pack_threads", th)
    for _ in range(3): once()
    ts = np.array([once() for _ in range(9)])
    m = np.median(ts, axis=0) * 1e3
    print("pack_threads=%-4d set_database_f32 %.2f ms  set_queries_f32 %.2f ms  map %.2f ms  total %.2f ms (min total %.2f)"
          % (th, m[0], m[1], m[2], m.sum(), ts.sum(axis=1).min() * 1e3), flush=True)
ctx.close()
