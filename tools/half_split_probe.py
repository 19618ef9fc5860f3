#!/usr/bin/env python3
"""Would ONE call split into two query halves on two streams beat the same call on one stream?  Emulated with what the C ABI has:
two contexts, each holding half of C2's queries and the whole database, a blind step enqueued on each (hg_map_begin), both
waited for (hg_map_end) -- against one context's synchronous hg_map on all queries.  `lag` delays the second half's enqueue by
that many microseconds (the halves otherwise run in lockstep: hist || hist, select || select, nothing overlaps another kind)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hashgan_amd import _native  # noqa: E402


def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


def main():
    spec = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
    qw, ql, dw, dl = bench.build_packed(spec, 0, spec["N"])
    R, Q = spec["R"], spec["Q"]
    h = (Q // 2 + 511) // 512 * 512
    full = _native.Context(0)
    full.set_database(dw, dl, spec["b"], spec["C"]); full.set_queries(qw, ql)
    halves = []
    for sl in (slice(0, h), slice(h, Q)):
        c = _native.Context(0)
        c.set_database(dw, dl, spec["b"], spec["C"]); c.set_queries(np.ascontiguousarray(qw[sl]), np.ascontiguousarray(ql[sl]))
        halves.append(c)
    for c in [full] + halves:
        for _ in range(30):
            c.map(R)
    a0, _ = full.map(R)
    K = 200
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            full.map(R)
        one = (time.perf_counter() - t0) / K
        out = ["one call, one stream %.4f ms" % (one * 1e3)]
        for lag in (0, 40, 80, 150, 300):
            t0 = time.perf_counter()
            for _ in range(K):
                halves[0].map_begin(R)
                if lag:
                    spin(lag)
                halves[1].map_begin(R)
                a, _ = halves[0].map_end()
                b, _ = halves[1].map_end()
            out.append("halves, lag %d us: %.4f" % (lag, (time.perf_counter() - t0) / K * 1e3))
        assert np.array_equal(np.concatenate([a, b]), a0, equal_nan=True)
        print(" | ".join(out), flush=True)
    print("segments: full", full.get_stat("segments"), "half", halves[0].get_stat("segments"))


if __name__ == "__main__":
    main()
