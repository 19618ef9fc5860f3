import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from hashgan_amd import _native, metric
Q, N, b, R, C = 4000, 1000000, 64, 5000, 10
rng = np.random.default_rng(9)
for gain in (1.0, 4.0, 12.0):
    df = np.tanh(gain * rng.standard_normal((N, b), dtype=np.float32)); qf = np.tanh(gain * rng.standard_normal((Q, b), dtype=np.float32))
    eye = np.eye(C, dtype=np.int64); dl = eye[rng.integers(0, C, N)]; ql = eye[rng.integers(0, C, Q)]
    ctx = _native.Context(0); ctx.set_option("keep_floats", 1)
    bad, _ = ctx.set_database_f32(df, dl); ctx.set_queries_f32(qf, ql)
    a, r = ctx.map_real(R); ctx.map_real(R)
    t = time.perf_counter()
    for _ in range(3): ctx.map_real(R)
    dt = (time.perf_counter() - t) / 3
    ctx.timing_enable(2); ctx.timing_reset(); ctx.map_real(R); tm = ctx.timing_read(); ctx.timing_enable(0)
    print("gain %.0f: exactly +-1 entries %.1f%%  %.2f ms attempts %d lds_ranked %d | %s" % (gain, 100 * np.mean(np.abs(df) == 1.0), dt * 1e3, ctx.get_stat("real_attempts"), ctx.get_stat("real_lds_ranked"),
          " ".join("%s=%.2f" % (k.replace("k_", ""), v[0]) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.05 and k != "step_gpu_span")), flush=True)
    ctx.set_option("real_mfma", 1); a2, _ = ctx.map_real(R)
    print("   equal to the exact pass:", np.array_equal(a, a2, equal_nan=True))
    ctx.close()
