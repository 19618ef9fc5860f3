#!/usr/bin/env python3
"""Per-kernel timings of the one-shot binary path under engine options (experiments on k_select_mx;
probe_select=2 no drain, =5 no record stores, =9 no emit -- k_select_mx's own time is what to read then).
usage: mx_probe.py [Q N b R] -- key=value[,key=value] ..."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hashgan_amd import _native, synth, metric

def main():
    args = sys.argv[1:]
    shape = (10000, 1000000, 64, 5000)
    if "--" in args:
        i = args.index("--")
        if i: shape = tuple(int(x) for x in args[:i])
        cfgs = args[i + 1:]
    else:
        cfgs = args
    Q, N, b, R = shape
    dl, _ = synth.onehot_labels(1, N, 10)
    ql, _ = synth.onehot_labels(2, Q, 10)
    dw = synth.splitmix64(3, N * ((b + 63) // 64)).reshape(N, -1)
    qw = synth.splitmix64(4, Q * ((b + 63) // 64)).reshape(Q, -1)
    if b % 64:
        m = np.uint64((1 << (b % 64)) - 1)
        dw[:, -1] &= m; qw[:, -1] &= m
    for cfg in cfgs or [""]:
        ctx = _native.Context(0)
        for kv in filter(None, cfg.split(",")):
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        ctx.set_database(dw, metric.pack_labels(dl), b, 10)
        ctx.set_queries(qw, metric.pack_labels(ql))
        ctx.map(R)
        ctx.timing_enable(True); ctx.timing_reset()
        t = time.perf_counter()
        for _ in range(5):
            ctx.set_option("optimistic", 1)      # clears the lost-bet latch (experiments that break the bet on purpose)
            ctx.map(R)
        dt = (time.perf_counter() - t) / 5
        tm = ctx.timing_read()
        print("%-40s %7.3f ms/step  " % (cfg, dt * 1e3) + "  ".join("%s=%.3f" % (k, v[0] / max(v[1], 1)) for k, v in tm.items()),
              "fallbacks=%d" % ctx.get_stat("optimistic_fallbacks"), flush=True)
        ctx.close()

if __name__ == "__main__":
    main()
