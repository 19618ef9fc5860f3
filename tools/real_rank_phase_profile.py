#!/usr/bin/env python3
"""Phase timestamps of k_real_rank_lds (a build with -DHG_REAL_RANK_PROFILE=1, HG_LIBRARY=...: the kernel writes them over
the head of each ranked list): 100 MHz clock ticks from a block's start to the end of each phase, at the C2 shape."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native
Q, N, b, R, C = 10000, 1000000, 64, 5000, 10
rng = np.random.default_rng(3)
df = np.tanh(rng.standard_normal((N, b), dtype=np.float32)); qf = np.tanh(rng.standard_normal((Q, b), dtype=np.float32))
eye = np.eye(C, dtype=np.int64)
ctx = _native.Context(0)
ctx.set_option("keep_floats", 1)
ctx.set_database_f32(df, eye[rng.integers(0, C, N)]); ctx.set_queries_f32(qf, eye[rng.integers(0, C, Q)])
ctx.topr_real(R)
idx, _ = ctx.topr_real(R)
t = idx[:, :5].astype(np.float64)
names = ["offsets + copy", "radix select", "compaction", "counting passes", "lists + match bits"]
prev = np.zeros(Q)
print("phase               median ticks (cumulative)   median of the phase   (s_memtime: 100 MHz)")
for k, n in enumerate(names):
    print("%-20s %10.0f %24.0f" % (n, np.median(t[:, k]), np.median(t[:, k] - prev)))
    prev = t[:, k]
ctx.close()
