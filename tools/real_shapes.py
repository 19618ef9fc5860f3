#!/usr/bin/env python3
"""The real-valued (tanh) ranking at the reference's own shapes: ms per hg_map_real call with the tables resident.
usage: real_shapes.py"""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, metric

def run(name, Q, N, b, R, C, multi=False, steps=5):
    rng = np.random.default_rng(Q + N)
    proto = rng.standard_normal((C, b)).astype(np.float32)
    if multi:
        dl = (rng.random((N, C)) < 2.4 / C).astype(np.int64); ql = (rng.random((Q, C)) < 2.4 / C).astype(np.int64)
        df = np.tanh(0.5 * (dl @ proto).astype(np.float32) + rng.standard_normal((N, b), dtype=np.float32))
        qf = np.tanh(0.5 * (ql @ proto).astype(np.float32) + rng.standard_normal((Q, b), dtype=np.float32))
    else:
        cls, qcls = rng.integers(0, C, N), rng.integers(0, C, Q)
        dl, ql = np.eye(C, dtype=np.int64)[cls], np.eye(C, dtype=np.int64)[qcls]
        df = np.tanh(0.7 * proto[cls] + rng.standard_normal((N, b), dtype=np.float32))
        qf = np.tanh(0.7 * proto[qcls] + rng.standard_normal((Q, b), dtype=np.float32))
    ctx = _native.Context(0)
    try:
        ctx.set_option("keep_floats", 1)
        ctx.set_database_f32(df, dl); ctx.set_queries_f32(qf, ql)
        a, r = ctx.map_real(R); ctx.map_real(R)
        t = time.perf_counter()
        for _ in range(steps): ctx.map_real(R)
        dt = (time.perf_counter() - t) / steps
        ctx.timing_enable(2); ctx.timing_reset()
        for _ in range(2): ctx.map_real(R)
        tm = ctx.timing_read(); ctx.timing_enable(0)
        kern = " ".join("%s=%.3f" % (k.replace("k_", ""), v[0] / 2) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]) if k != "step_gpu_span" and v[0] / 2 >= 0.01)
        print("%-22s Q=%-6d N=%-8d b=%-3d R=%-6d %8.3f ms/call  attempts %d filtered %d lds_ranked %d  mAP %.4f | %s" % (name, Q, N, b, R, dt * 1e3,
              ctx.get_stat("real_attempts"), ctx.get_stat("real_path") & 1, (ctx.get_stat("real_path") >> 1) & 1, metric.mean_over_hits(a, r), kern), flush=True)
    finally:
        ctx.close()

if __name__ == "__main__":
    run("C1 cifar10 (R = N)", 1000, 54000, 32, 54000, 10)
    run("C3 nuswide", 2100, 190000, 48, 5000, 21, multi=True)
    run("imagenet100-like", 5000, 128000, 64, 1000, 100)
    run("C2 shape", 10000, 1000000, 64, 5000, 10)
