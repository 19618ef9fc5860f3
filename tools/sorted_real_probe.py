#!/usr/bin/env python3
"""The real-valued (tanh) ranking on features that follow the labels, rows in random order versus stored class by class.
usage: sorted_real_probe.py [Q N b R]"""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, metric

def run(tag, qf, ql, df, dl, R, steps=4):
    ctx = _native.Context(0)
    try:
        ctx.set_option("keep_floats", 1)
        ctx.set_database_f32(df, dl)
        ctx.set_queries_f32(qf, ql)
        a0, r0 = ctx.map_real(R)
        att0 = ctx.get_stat("real_attempts")
        ctx.map_real(R)
        t = time.perf_counter()
        for _ in range(steps): ctx.map_real(R)
        dt = (time.perf_counter() - t) / steps
        print("%-8s %8.3f ms/call  attempts first call %d, later %d  lds_ranked=%d boost=%d  mAP=%.6f" % (tag, dt * 1e3, att0, ctx.get_stat("real_attempts"),
              (ctx.get_stat("real_path") >> 1) & 1, ctx.get_stat("real_cap_boost"), metric.mean_over_hits(a0, r0)), flush=True)
        return a0
    finally:
        ctx.close()

if __name__ == "__main__":
    Q, N, b, R = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (10000, 1000000, 64, 5000)
    C = 10
    rng = np.random.default_rng(7)
    proto = rng.standard_normal((C, b)).astype(np.float32)
    cls, qcls = rng.integers(0, C, N), rng.integers(0, C, Q)
    feat = lambda c: np.tanh(0.7 * proto[c] + rng.standard_normal((len(c), b), dtype=np.float32))
    df, qf = feat(cls), feat(qcls)
    eye = np.eye(C, dtype=np.int64)
    run("shuffled", qf, eye[qcls], df, eye[cls], R)
    order = np.argsort(cls, kind="stable")
    run("sorted", qf, eye[qcls], np.ascontiguousarray(df[order]), eye[cls][order], R)
