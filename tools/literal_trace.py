#!/usr/bin/env python3
"""bench.py's drop_in_literal leg alone, with the call taken apart (load / rank / release), to find where the CIFAR shape's 5.4 ms go."""
import os, sys, time, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hashgan_amd import MAPs, metric, _native

spec = bench.WORKLOADS["c2"]
packed = bench.build_packed(spec, 0, spec["N"])
if "--bench-leg" in sys.argv:
    import json
    print(json.dumps({k: (v["ms_each_call"] if isinstance(v, dict) and "ms_each_call" in v else None) for k, v in bench.drop_in_literal(spec, packed).items()}))
rng = np.random.default_rng(0xD1)
eye = np.eye(10, dtype=np.int64)
N, Q = 54000, 1000
db = types.SimpleNamespace(output=np.tanh(rng.standard_normal((N, 64), dtype=np.float32)), label=eye[rng.integers(0, 10, N)])
q = types.SimpleNamespace(output=np.tanh(rng.standard_normal((Q, 64), dtype=np.float32)), label=eye[rng.integers(0, 10, Q)])
print("label dtype", db.label.dtype, db.label.flags["C_CONTIGUOUS"], db.output.flags["C_CONTIGUOUS"])
for rep in range(6):
    t0 = time.perf_counter()
    m = MAPs(54000)
    eng = m._engine()
    t1 = time.perf_counter()
    m._ensure_database(db)
    t2 = time.perf_counter()
    v = m.get_maps_by_feature(db, q)
    t3 = time.perf_counter()
    m.close()
    t4 = time.perf_counter()
    print("acquire %.3f  load db %.3f  rank (queries + map_real + mean) %.3f  release %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
