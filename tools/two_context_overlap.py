#!/usr/bin/env python3
"""How much of a step's tail (rank + AP, copy-out; the next step's sampled histogram and guess) would hide under another step's
select if consecutive steps ran on two streams?  Two contexts (a stream and work buffers each) on the same C2 tables, steps
enqueued alternately with hg_map_begin / hg_map_end, against one context with two steps in flight on ONE stream (bench.py's
headline mode).  An upper bound for what splitting a call across two streams can gain, measured without building it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hashgan_amd import _native  # noqa: E402


def main():
    spec = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
    qw, ql, dw, dl = bench.build_packed(spec, 0, spec["N"])
    R = spec["R"]
    ctxs = []
    for _ in range(2):
        c = _native.Context(0)
        c.set_database(dw, dl, spec["b"], spec["C"])
        c.set_queries(qw, ql)
        for _ in range(30):
            a0, r0 = c.map(R)
        ctxs.append(c)
    K = 200
    for rep in range(3):
        # one context, two steps in flight on one stream
        c = ctxs[0]
        c.synchronize()
        t0 = time.perf_counter()
        c.map_begin(R)
        for i in range(K):
            if i + 1 < K:
                c.map_begin(R)
            a, r = c.map_end()
        one = (time.perf_counter() - t0) / K
        # two contexts, one step in flight on each, alternating
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        ctxs[0].map_begin(R)
        for i in range(K):
            if i + 1 < K:
                ctxs[(i + 1) & 1].map_begin(R)
            a, r = ctxs[i & 1].map_end()
        two = (time.perf_counter() - t0) / K
        # two contexts, two steps in flight on each
        t0 = time.perf_counter()
        ctxs[0].map_begin(R); ctxs[1].map_begin(R); ctxs[0].map_begin(R)
        for i in range(K):
            if i + 3 < K:
                ctxs[(i + 3) & 1].map_begin(R)
            a, r = ctxs[i & 1].map_end()
        two2 = (time.perf_counter() - t0) / K
        assert np.array_equal(a, a0, equal_nan=True)
        print("one stream, two steps in flight: %.4f ms/step | two streams, one step each: %.4f | two streams, two steps each: %.4f" % (one * 1e3, two * 1e3, two2 * 1e3), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
