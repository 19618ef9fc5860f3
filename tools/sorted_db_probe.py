#!/usr/bin/env python3
"""What a database stored class by class costs the bet: planted codes (a query's near rows share its class), the rows in
random order versus sorted by label -- the bet's slices are sized for hits spread evenly over the segments.
usage: sorted_db_probe.py [Q N b R flip [option=value ...]]"""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

OPTS = []
def run(tag, qw, ql, dw, dl, b, C, R, steps=10):
    ctx = _native.Context(0)
    for k, v in OPTS: ctx.set_option(k, v)
    ctx.set_database(dw, metric.pack_labels(dl), b, C)
    ctx.set_queries(qw, metric.pack_labels(ql))
    a0, r0 = ctx.map(R); ctx.map(R)
    f0 = ctx.get_stat("optimistic_fallbacks")
    t = time.perf_counter()
    for _ in range(steps): ctx.map(R)
    dt = (time.perf_counter() - t) / steps
    print("%-8s %8.3f ms/step  bet=%d fallbacks=%d requeried=%d rebets=%d cap_boost=%d S=%d  mAP=%.6f" % (tag, dt * 1e3, ctx.get_stat("last_optimistic"),
          ctx.get_stat("optimistic_fallbacks") - f0, ctx.get_stat("optimistic_requeried"), ctx.get_stat("optimistic_rebets"), ctx.get_stat("cap_boost"), ctx.get_stat("segments"),
          metric.mean_over_hits(a0, r0)), flush=True)
    ctx.timing_enable(2); ctx.timing_reset()
    for _ in range(3): ctx.map(R)
    tm = ctx.timing_read(); ctx.timing_enable(0)
    print("         " + " ".join("%s=%.3f" % (k.replace("k_", ""), v[0] / 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]) if k != "step_gpu_span" and v[0] / 3 >= 0.005), flush=True)
    ctx.close()
    return a0

if __name__ == "__main__":
    Q, N, b, R = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (10000, 1000000, 64, 5000)
    flip = float(sys.argv[5]) if len(sys.argv) > 5 else 0.30
    OPTS[:] = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[6:]]
    C = 10
    dl, _ = synth.onehot_labels(1, N, C)
    ql, _ = synth.onehot_labels(2, Q, C)
    db_bits = synth.planted_codes(3, dl, b, flip)
    q_bits = synth.planted_codes(3, ql, b, flip, noise_seed=5)       # same class prototypes, own noise
    assert b % 64 == 0
    pk = lambda bits: np.ascontiguousarray(np.packbits(bits.astype(np.uint8), axis=1, bitorder="little")).view(np.uint64)
    dw, qw = pk(db_bits), pk(q_bits)
    a = run("shuffled", qw, ql, dw, dl, b, C, R)
    order = np.argsort(dl.argmax(1), kind="stable")
    a2 = run("sorted", qw, ql, np.ascontiguousarray(dw[order]), np.ascontiguousarray(dl[order]), b, C, R)
    # ties break by index, so the two orders rank ties differently; on the mean the difference is small
    print("mean |dAP| between the two row orders: %.3e" % np.abs(a - a2).mean())
