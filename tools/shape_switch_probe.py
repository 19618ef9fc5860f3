#!/usr/bin/env python3
"""What do the FIRST calls of a new shape cost on a pooled context that has served another shape?  (bench.py's drop_in_literal: CIFAR-tanh right
after C2 +-1 -- calls 1 and 2 of twelve ran 3.1 and 3.9 ms against a median of 1.9.)  Host phases per call."""
import os, sys, time, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hashgan_amd import MAPs, _native

rng = np.random.default_rng(3)
eye = np.eye(10, dtype=np.int64)
def case(Q, N, real):
    x = rng.standard_normal((N, 64), dtype=np.float32); y = rng.standard_normal((Q, 64), dtype=np.float32)
    f = np.tanh if real else (lambda a: np.where(a > 0, 1.0, -1.0).astype(np.float32))
    return types.SimpleNamespace(output=f(x), label=eye[rng.integers(0, 10, N)]), types.SimpleNamespace(output=f(y), label=eye[rng.integers(0, 10, Q)])
big, cif = case(10000, 1000000, False), case(1000, 54000, True)
probe = _native.Context(0)
for rnd in range(3):
    for name, (db, q), R in (("c2", big, 5000), ("cifar", cif, 54000)):
        out = []
        for i in range(6):
            h0 = _native.host_phase_timers(probe); t0 = time.perf_counter()
            MAPs(R).get_maps_by_feature(db, q)
            dt = (time.perf_counter() - t0) * 1e3; h1 = _native.host_phase_timers(probe)
            out.append("%.2f %s" % (dt, {k: round(h1[k][0] - h0[k][0], 2) for k in h0 if h1[k][0] - h0[k][0] > 0.05}))
        print(rnd, name, " | ".join(out), flush=True)
