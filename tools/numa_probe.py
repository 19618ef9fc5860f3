#!/usr/bin/env python3
"""Is the box-to-box spread of the literal C2 call (2.5 ms on one box, 4.5 on the next) NUMA placement?  The caller's arrays are
first-touched under one node's processors; the calls (and the packing pool's threads, created after the switch) then run confined to
the same node, to the other node, or unconfined.  tools/literal_outliers.py's measurement, 150 calls each."""
import gc
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cpus(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def child(alloc_node, run_node):
    allowed = os.sched_getaffinity(0)
    os.sched_setaffinity(0, set(cpus(alloc_node)) & allowed)
    rng = np.random.default_rng(5)
    Q, N, b, R, C = 10000, 1000000, 64, 5000, 10
    eye = np.eye(C, dtype=np.int64)
    f = lambda a: np.where(a > 0, 1.0, -1.0).astype(np.float32)
    db = types.SimpleNamespace(output=f(rng.standard_normal((N, b), dtype=np.float32)), label=eye[rng.integers(0, C, N)])
    q = types.SimpleNamespace(output=f(rng.standard_normal((Q, b), dtype=np.float32)), label=eye[rng.integers(0, C, Q)])
    os.sched_setaffinity(0, allowed if run_node is None else set(cpus(run_node)) & allowed)
    from hashgan_amd import MAPs, _native
    probe = _native.Context(0)
    for _ in range(5):
        MAPs(R).get_maps_by_feature(db, q)
    ms, packs = [], []
    for _ in range(150):
        h0 = _native.host_phase_timers(probe)["pack"][0]
        t0 = time.perf_counter()
        MAPs(R).get_maps_by_feature(db, q)
        ms.append((time.perf_counter() - t0) * 1e3)
        packs.append(_native.host_phase_timers(probe)["pack"][0] - h0)
    print("arrays on node %d, threads %s: median call %.3f ms, median packing pass %.3f ms, max call %.3f" %
          (alloc_node, "unconfined" if run_node is None else "on node %d" % run_node, np.median(ms), np.median(packs), max(ms)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), None if sys.argv[2] == "x" else int(sys.argv[2]))
    else:
        import subprocess
        for a, r in ((0, "0"), (0, "1"), (0, "x"), (1, "1"), (1, "0"), (1, "x")):
            subprocess.run([sys.executable, os.path.abspath(__file__), str(a), r])
