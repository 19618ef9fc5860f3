#!/usr/bin/env python3
"""hg_set_database_f32 with the float table kept (what the real-valued ranking needs): ms per call at 1M x 64, by pack_threads."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native
N, b, C = 1000000, 64, 10
rng = np.random.default_rng(1)
x = np.tanh(rng.standard_normal((N, b), dtype=np.float32)); lab = np.eye(C, dtype=np.int64)[rng.integers(0, C, N)]
ctx = _native.Context(0)
for keep in (0, 1):
    ctx.set_option("keep_floats", keep)
    for th in (0, 16, 32, 64, 128):
        ctx.set_option("pack_threads", th)
        for _ in range(2): ctx.set_database_f32(x, lab)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); ctx.set_database_f32(x, lab); ts.append(time.perf_counter() - t)
        print("keep_floats=%d pack_threads=%-3d  median %.2f ms  min %.2f ms" % (keep, th, np.median(ts) * 1e3, min(ts) * 1e3), flush=True)
ctx.close()
