#!/usr/bin/env python3
"""Phase timestamps of k_rank_cnt (build with -DHG_RANK_PROFILE=1, HG_LIBRARY=...): shader-clock cycles from a block's
start to the end of each phase, for the first 4096 queries of a C2 step."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from hashgan_amd import _native
spec = bench.WORKLOADS["c2"]
qw, ql, dw, dl = bench.build_packed(spec, 0, spec["N"])
ctx = _native.Context(0)
ctx.set_database(dw, dl, spec["b"], spec["C"]); ctx.set_queries(qw, ql)
for _ in range(3): ctx.map(spec["R"])
ptr = ctx.get_stat("dbg_hwq_ptr")
host = np.zeros(4096 * 16, np.uint32)
ctx.memcpy_dtoh(host, ptr, host.nbytes); ctx.synchronize()
t = host.reshape(4096, 16)[:, :8].astype(np.float64)
print("rank variant", ctx.get_stat("rank_variant"))
names = ["counts+prefix", "copy", "count", "totals", "plan", "offsets", "place", "bitmap out"]
if ctx.get_stat("rank_variant") == 6:      # k_rank_lean's phases
    names = ["loads + slice prefix", "compaction", "count", "totals", "plan + ranks", "place", "bitmap out", "AP"]
prev = np.zeros(4096)
print("phase            median cycles (cumulative)   median of the phase")
for k, n in enumerate(names):
    print("%-16s %10.0f %24.0f" % (n, np.median(t[:, k]), np.median(t[:, k] - prev)))
    prev = t[:, k]
ctx.close()
