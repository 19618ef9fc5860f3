#!/usr/bin/env python3
"""Which phase do the slow calls of the literal drop-in spend their extra time in?  K calls of MAPs(R).get_maps_by_feature(db, q) (a new object
per call) on one of the reference's shapes, every call's wall time and the host-phase timers' deltas (hg_get_stat host_us_*); prints the
distribution and the slowest calls with their phases beside the median call's.

    python tools/literal_outliers.py [nus|cifar|c2] [calls]
"""
import gc
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hashgan_amd import MAPs, _native  # noqa: E402

SHAPES = {"cifar": (1000, 54000, 64, 54000, 10, False, True), "nus": (5000, 168692, 64, 5000, 81, True, True), "c2": (10000, 1000000, 64, 5000, 10, False, False)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "nus"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    Q, N, b, R, C, multi, real = SHAPES[name]
    rng = np.random.default_rng(5)
    if multi:
        dl = (rng.random((N, C)) < 0.03).astype(np.int64); dl[np.arange(N), rng.integers(0, C, N)] = 1
        ql = (rng.random((Q, C)) < 0.03).astype(np.int64); ql[np.arange(Q), rng.integers(0, C, Q)] = 1
    else:
        eye = np.eye(C, dtype=np.int64)
        dl, ql = eye[rng.integers(0, C, N)], eye[rng.integers(0, C, Q)]
    f = np.tanh if real else (lambda a: np.where(a > 0, 1.0, -1.0).astype(np.float32))
    db = types.SimpleNamespace(output=f(rng.standard_normal((N, b), dtype=np.float32)), label=dl)
    q = types.SimpleNamespace(output=f(rng.standard_normal((Q, b), dtype=np.float32)), label=ql)
    probe = _native.Context(0)
    for _ in range(5):
        MAPs(R).get_maps_by_feature(db, q)
    gc_was = gc.isenabled()
    if "--no-gc" in sys.argv:
        gc.disable()
    rows = []
    for i in range(K):
        h0 = _native.host_phase_timers(probe)
        t0 = time.perf_counter()
        MAPs(R).get_maps_by_feature(db, q)
        dt = (time.perf_counter() - t0) * 1e3
        h1 = _native.host_phase_timers(probe)
        rows.append((dt, {k: round(h1[k][0] - h0[k][0], 3) for k in h0 if h1[k][0] - h0[k][0] > 0.0005}))
    if gc_was:
        gc.enable()
    ms = np.array([r[0] for r in rows])
    order = np.argsort(ms)
    print(name, "calls", K, "min %.3f  median %.3f  p90 %.3f  p99 %.3f  max %.3f ms" % (ms.min(), np.median(ms), np.percentile(ms, 90), np.percentile(ms, 99), ms.max()),
          "| calls beyond 1.5 x median:", int((ms > 1.5 * np.median(ms)).sum()))
    print("  median call:", round(rows[order[K // 2]][0], 3), rows[order[K // 2]][1])
    for i in order[::-1][:6]:
        print("  call %3d: %.3f ms" % (i, rows[i][0]), rows[i][1])
    probe.close()


if __name__ == "__main__":
    main()
