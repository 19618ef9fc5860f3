#!/usr/bin/env python3
"""Throughput of the binary path across shapes (i.i.d. random codes, one-hot labels, C=10):
ms per hg_map step with inputs resident in HBM, queries/s, pair rate, and the path taken."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

def run(Q, N, b, R, steps=5, opts=()):
    dl, _ = synth.onehot_labels(1, N, 10)
    ql, _ = synth.onehot_labels(2, Q, 10)
    dw = synth.splitmix64(3, N * ((b + 63) // 64)).reshape(N, -1)
    qw = synth.splitmix64(4, Q * ((b + 63) // 64)).reshape(Q, -1)
    if b % 64:
        m = np.uint64((1 << (b % 64)) - 1)
        dw[:, -1] &= m; qw[:, -1] &= m
    ctx = _native.Context(0)
    for k, v in opts: ctx.set_option(k, v)
    ctx.set_database(dw, metric.pack_labels(dl), b, 10)
    ctx.set_queries(qw, metric.pack_labels(ql))
    ctx.map(R); ctx.map(R)
    t = time.perf_counter()
    for _ in range(steps): ctx.map(R)
    dt = (time.perf_counter() - t) / steps
    path = "bet" if ctx.get_stat("last_optimistic") else "exact"
    ctx.timing_enable(2); ctx.timing_reset()
    for _ in range(3): ctx.map(R)
    tm = ctx.timing_read(); ctx.timing_enable(0)
    kern = " ".join("%s=%.3f" % (k.replace("k_", ""), v[0] / 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]) if k != "step_gpu_span" and v[0] / 3 >= 0.005)
    print("Q=%-6d N=%-9d b=%-3d R=%-7d %8.3f ms  %10.0f q/s  %6.2f Tpairs/s  %-5s S=%d | %s" % (Q, N, b, R, dt * 1e3, Q / dt, Q * N / dt / 1e12, path, ctx.get_stat("segments"), kern), flush=True)
    ctx.close()

if __name__ == "__main__":
    if len(sys.argv) > 1:          # shape_sweep.py Q N b R [opt=val ...]
        Q, N, b, R = map(int, sys.argv[1:5])
        run(Q, N, b, R, opts=[(kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[5:]])
        sys.exit(0)
    for (Q, N, b, R) in [(10000, 1000000, 64, 5000), (10000, 1000000, 64, 100), (10000, 1000000, 64, 50000), (10000, 1000000, 64, 500000),
                         (10000, 1000000, 32, 5000), (10000, 1000000, 48, 5000), (10000, 1000000, 128, 5000), (10000, 1000000, 255, 5000),
                         (1000, 1000000, 64, 5000), (50000, 1000000, 64, 5000), (10000, 100000, 64, 5000), (10000, 10000000, 64, 5000),
                         (1000, 54000, 32, 54000), (2100, 190000, 48, 5000), (64, 10000000, 64, 5000)]:
        run(Q, N, b, R)
