import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from hashgan_amd import _native, metric
Q, N, b, R, C = 1000, 54000, 32, 54000, 10
rng = np.random.default_rng(Q + N)
proto = rng.standard_normal((C, b)).astype(np.float32)
cls, qcls = rng.integers(0, C, N), rng.integers(0, C, Q)
dl, ql = np.eye(C, dtype=np.int64)[cls], np.eye(C, dtype=np.int64)[qcls]
df = np.tanh(0.7 * proto[cls] + rng.standard_normal((N, b), dtype=np.float32)); qf = np.tanh(0.7 * proto[qcls] + rng.standard_normal((Q, b), dtype=np.float32))
ctx = _native.Context(0); ctx.set_option("keep_floats", 1)
ctx.set_database_f32(df, dl); ctx.set_queries_f32(qf, ql)
res = {}
for mode in (2, 1, 0):
    ctx.set_option("real_mfma", mode)
    a, r = ctx.map_real(R); ctx.map_real(R)
    t = time.perf_counter()
    for _ in range(5): ctx.map_real(R)
    dt = (time.perf_counter() - t) / 5
    ctx.timing_enable(2); ctx.timing_reset(); ctx.map_real(R); tm = ctx.timing_read(); ctx.timing_enable(0)
    print("real_mfma=%d  %.2f ms | %s" % (mode, dt * 1e3, " ".join("%s=%.2f" % (k.replace("k_", ""), v[0]) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.03 and k != "step_gpu_span")), flush=True)
    res[mode] = a
print("equal:", np.array_equal(res[2], res[1], equal_nan=True), np.array_equal(res[2], res[0], equal_nan=True))
