#!/usr/bin/env python3
"""One-off on the GPU box: the drop-in call from host arrays (C2) against the number of host packing threads."""
import sys, time, types
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from hashgan_amd import MAPs, metric
spec = bench.WORKLOADS["c2"]
c = bench.build_inputs(spec)
db = types.SimpleNamespace(output=c["dbbits"].astype(np.float32) * 2 - 1, label=c["dblab"].astype(np.int64))
q = types.SimpleNamespace(output=c["qbits"].astype(np.float32) * 2 - 1, label=c["qlab"].astype(np.int64))
m = MAPs(c["R"])
m.get_maps_by_feature(db, q)
for th in [0, 4, 8, 16, 24, 32, 48, 64, 96, 128]:
    m._eng.ctx.set_option("pack_threads", th)
    m.get_maps_by_feature(db, q)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); m.get_maps_by_feature(db, q); ts.append(time.perf_counter() - t0)
    print("pack_threads=%-4d  call %.2f ms (min %.2f)" % (th, 1e3 * float(np.median(ts)), 1e3 * min(ts)), flush=True)
m.close()
