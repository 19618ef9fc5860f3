#!/usr/bin/env python3
"""Where a literal drop-in call (a new MAPs per evaluation, host arrays) spends its time, for the reference's own shapes:
load of the database (pack + upload), load of the queries, the ranking call, per kernel.  tools/gpu_r06_b.sh runs it.

    python tools/literal_breakdown.py [cifar|nus|c2] ...
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hashgan_amd import _native, metric  # noqa: E402

SHAPES = {"cifar": (1000, 54000, 64, 54000, 10, False), "nus": (5000, 168692, 64, 5000, 81, True), "c2": (10000, 1000000, 64, 5000, 10, False)}


def main():
    for name in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["cifar", "nus"]):
        Q, N, b, R, C, multi = SHAPES[name]
        rng = np.random.default_rng(7)
        if multi:
            dl = (rng.random((N, C)) < 0.03).astype(np.int64); dl[np.arange(N), rng.integers(0, C, N)] = 1
            ql = (rng.random((Q, C)) < 0.03).astype(np.int64); ql[np.arange(Q), rng.integers(0, C, Q)] = 1
        else:
            eye = np.eye(C, dtype=np.int64)
            dl, ql = eye[rng.integers(0, C, N)], eye[rng.integers(0, C, Q)]
        df = np.tanh(rng.standard_normal((N, b), dtype=np.float32))
        qf = np.tanh(rng.standard_normal((Q, b), dtype=np.float32))
        ctx = _native.Context(0)
        ctx.set_option("keep_floats", 2)
        rows = []
        for it in range(8):
            t0 = time.perf_counter()
            ctx.set_database_f32(df, dl)
            t1 = time.perf_counter()
            ctx.set_queries_f32(qf, ql)
            t2 = time.perf_counter()
            ap, rel = ctx.map_real(R)
            t3 = time.perf_counter()
            m = metric.mean_over_hits(ap, rel)
            t4 = time.perf_counter()
            rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
        print(name, "Q=%d N=%d b=%d R=%d C=%d" % (Q, N, b, R, C))
        for r in rows:
            print("   set_database_f32 %.3f  set_queries_f32 %.3f  map_real %.3f  mean %.3f   total %.3f ms" % (r + (sum(r),)))
        ctx.timing_enable(2)
        ctx.map_real(R)
        ctx.timing_reset()
        for _ in range(3):
            ctx.map_real(R)
        tm = ctx.timing_read()
        ctx.timing_enable(0)
        print("   kernels per call (ms):", {k: round(v[0] / 3, 4) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])}, "launches", {k: v[1] / 3 for k, v in tm.items()})
        print("   real_path", ctx.get_stat("real_path"), "attempts", ctx.get_stat("real_attempts"))
        # the same through the Python surface, a fresh MAPs per call
        import types
        from hashgan_amd import MAPs
        db, q = types.SimpleNamespace(output=df, label=dl), types.SimpleNamespace(output=qf, label=ql)
        each = []
        for _ in range(8):
            t0 = time.perf_counter()
            MAPs(R).get_maps_by_feature(db, q)
            each.append(round((time.perf_counter() - t0) * 1e3, 3))
        print("   MAPs(R).get_maps_by_feature, new object per call:", each)
        ctx.close()


if __name__ == "__main__" and "--after-c2" not in sys.argv:
    main()


def after_c2():
    """bench.py's drop_in_literal order: the pooled context serves the C2 +-1 shape first, then the CIFAR tanh shape."""
    import types
    from hashgan_amd import MAPs
    rng = np.random.default_rng(3)
    eye = np.eye(10, dtype=np.int64)

    def case(Q, N, real):
        x = rng.standard_normal((N, 64), dtype=np.float32)
        y = rng.standard_normal((Q, 64), dtype=np.float32)
        f = (lambda a: np.tanh(a)) if real else (lambda a: np.where(a > 0, 1.0, -1.0).astype(np.float32))
        return (types.SimpleNamespace(output=f(x), label=eye[rng.integers(0, 10, N)]), types.SimpleNamespace(output=f(y), label=eye[rng.integers(0, 10, Q)]))
    big = case(10000, 1000000, False)
    cif = case(1000, 54000, True)
    for rnd in range(2):
        for name, (db, q), R in (("c2 +-1", big, 5000), ("cifar tanh", cif, 54000)):
            m = MAPs(R)
            each = []
            for _ in range(5):
                t0 = time.perf_counter()
                m.get_maps_by_feature(db, q)
                each.append(round((time.perf_counter() - t0) * 1e3, 3))
            ctx = m._eng.ctx
            t0 = time.perf_counter(); metric._load_database(m._eng, db.output, db.label, "reference"); t1 = time.perf_counter()
            ctx.set_queries_f32(q.output, q.label); t2 = time.perf_counter()
            print("  ", name, "calls", each, "| load db %.3f  queries %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), "real_path", ctx.get_stat("real_path"),
                  "device_bytes", ctx.get_stat("device_bytes"), flush=True)
            m._resident = None
            m.close()


if "--after-c2" in sys.argv:
    after_c2()
