#!/usr/bin/env python3
"""One-off stress run on the GPU: full-size shapes with odd sizes (ragged last segments, windows, tiles, query blocks):
the one-shot bet against the vector-ALU exact sequence."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

SHAPES = [(9999, 1000003, 64, 5000, 10), (10001, 999983, 64, 4999, 10), (777, 10000019, 64, 5000, 10), (10000, 1000000, 32, 5000, 10),
          (5003, 1500007, 128, 7001, 81), (20011, 500009, 48, 2503, 10), (311, 3000017, 255, 3001, 130), (10000, 1000000, 64, 100, 10),
          (10000, 999999, 64, 50000, 10), (50021, 400003, 64, 2000, 3), (63, 9999991, 64, 5000, 10), (257, 1000001, 100, 60000, 10)]

def one(i, Q, N, b, R, C):
    dl, _ = synth.onehot_labels(i * 3 + 1, N, C); ql, _ = synth.onehot_labels(i * 3 + 2, Q, C)
    W = (b + 63) // 64
    dw = synth.splitmix64(i * 7 + 3, N * W).reshape(N, W); qw = synth.splitmix64(i * 7 + 4, Q * W).reshape(Q, W)
    if b % 64:
        m = np.uint64((1 << (b % 64)) - 1); dw[:, -1] &= m; qw[:, -1] &= m
    ctx = _native.Context(0)
    try:
        ctx.set_database(dw, metric.pack_labels(dl), b, C); ctx.set_queries(qw, metric.pack_labels(ql))
        out = {}
        for name, opts in (("bet", {}), ("bet8", {"compact_records": 0}), ("exact_mx", {"optimistic": 0}), ("exact_valu", {"optimistic": 0, "hist_mfma": 0, "exact_mfma": 0, "select_mfma": 0})):
            for k, v in {"compact_records": 1, "optimistic": 1, "hist_mfma": 2, "exact_mfma": 1, "select_mfma": 1}.items(): ctx.set_option(k, v)
            for k, v in opts.items(): ctx.set_option(k, v)
            t = time.time(); out[name] = ctx.map(R); dt = time.time() - t
        ref = out["exact_valu"]
        bad = [n for n, (ap, rel) in out.items() if not (np.array_equal(ap, ref[0], equal_nan=True) and np.array_equal(rel, ref[1]))]
        return ("MISMATCH %s " % bad if bad else "ok ") + "Q=%d N=%d b=%d R=%d C=%d" % (Q, N, b, R, C)
    finally:
        ctx.close()

if __name__ == "__main__":
    for i, sh in enumerate(SHAPES):
        print(one(i, *sh), flush=True)
