#!/usr/bin/env python3
"""Large-shape self-consistency: the matrix-core and the vector-ALU select must give identical AP."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric
Q, N, b, R = (int(x) for x in sys.argv[1:5])
dl, _ = synth.onehot_labels(1, N, 10)
ql, _ = synth.onehot_labels(2, Q, 10)
dw = synth.splitmix64(3, N).reshape(N, 1); qw = synth.splitmix64(4, Q).reshape(Q, 1)
if b < 64:
    m = np.uint64((1 << b) - 1); dw &= m; qw &= m
ctx = _native.Context(0)
ctx.set_database(dw, metric.pack_labels(dl), b, 10); ctx.set_queries(qw, metric.pack_labels(ql))
res = {}
for mf in (1, 0):
    ctx.set_option("select_mfma", mf); ctx.set_option("optimistic", 1)
    ctx.map(R)
    t = time.perf_counter(); ap, rel = ctx.map(R); dt = time.perf_counter() - t
    res[mf] = (ap, rel)
    print("select_mfma=%d: %.2f ms, bet=%d, fallbacks=%d, device MB %.0f" % (mf, dt * 1e3, ctx.get_stat("last_optimistic"), ctx.get_stat("optimistic_fallbacks"), ctx.get_stat("device_bytes") / 1e6), flush=True)
print("identical:", np.array_equal(res[1][0], res[0][0], equal_nan=True) and np.array_equal(res[1][1], res[0][1]))
