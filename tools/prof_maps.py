import sys, time, types, cProfile, pstats
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from hashgan_amd import MAPs
spec = bench.WORKLOADS["c2"]
qw, ql, dw, dl = bench.build_packed(spec, 0, spec["N"])
b, C, R = spec["b"], spec["C"], spec["R"]
db = types.SimpleNamespace(output=bench.unpack_bits(dw, b).astype(np.float32) * 2 - 1, label=bench.unpack_bits(dl, C).astype(np.int64))
q = types.SimpleNamespace(output=bench.unpack_bits(qw, b).astype(np.float32) * 2 - 1, label=bench.unpack_bits(ql, C).astype(np.int64))
m = MAPs(R)
for _ in range(3): m.get_maps_by_feature(db, q)
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for _ in range(20): m.get_maps_by_feature(db, q)
dt = (time.perf_counter() - t) / 20
pr.disable()
print("ms per call (profiled) %.3f" % (dt * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
