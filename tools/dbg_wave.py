import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from hashgan_amd import _native, synth, metric
def run(Q, N, b, R, opts=()):
    dl, _ = synth.onehot_labels(1, N, 10); ql, _ = synth.onehot_labels(2, Q, 10)
    dw = synth.splitmix64(3, N * ((b + 63) // 64)).reshape(N, -1); qw = synth.splitmix64(4, Q * ((b + 63) // 64)).reshape(Q, -1)
    ctx = _native.Context(0)
    for k, v in opts: ctx.set_option(k, v)
    ctx.set_database(dw, metric.pack_labels(dl), b, 10); ctx.set_queries(qw, metric.pack_labels(ql))
    for i in range(8):
        t = time.perf_counter(); ctx.map(R); dt = time.perf_counter() - t
        st = {k: ctx.get_stat(k) for k in ("rank_variant", "optimistic_fallbacks", "optimistic_requeried", "optimistic_rebets", "rank_leftovers", "last_optimistic", "ap_fused")}
        print(R, opts, i, "%.3f ms" % (dt * 1e3), st, flush=True)
    ctx.close()
run(10000, 1000000, 64, 100)
run(10000, 1000000, 64, 100, [("rank_wave", 0)])
