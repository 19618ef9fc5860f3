#!/usr/bin/env python3
"""Where does the FIRST call on a NEW context spend its time?  (0.24 - 1.2 s first calls next to ~2 ms ones.)

For each of K new contexts on the C2 tables (database rows stably sorted by label with --sorted, like bench.py's class_sorted leg):
wall time of hg_init, hg_set_database, hg_set_queries, the first hg_map, the second, hg_destroy -- and, per stage, the
process-wide host-phase timers' deltas (hg_get_stat host_us_*: hipMalloc, hipFree, hipHostMalloc, hipHostFree, streams, events).

    python tools/new_context_probe.py [--contexts 8] [--sorted] [--keep-one]

--keep-one holds one other context open over the whole run (is the stall the device's first / last context?).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hashgan_amd import _native  # noqa: E402


def snap(ctx):
    return {k: v[0] for k, v in _native.host_phase_timers(ctx).items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, default=8)
    ap.add_argument("--sorted", action="store_true")
    ap.add_argument("--keep-one", action="store_true")
    ap.add_argument("--workload", default="c2")
    a = ap.parse_args()
    spec = bench.WORKLOADS[a.workload]
    qw, ql, dw, dl = bench.build_packed(spec, 0, spec["N"])
    if a.sorted:
        key = bench.unpack_bits(dl, spec["C"]).astype(np.int64) @ (1 << np.arange(spec["C"], dtype=np.int64))
        order = np.argsort(key, kind="stable")
        dw, dl = np.ascontiguousarray(dw[order]), np.ascontiguousarray(dl[order])
    keeper = _native.Context(0) if a.keep_one else None
    probe = keeper or _native.Context(0)
    rows = []
    for i in range(a.contexts):
        row = {"context": i}
        s0 = snap(probe)
        t0 = time.perf_counter()
        ctx = _native.Context(0)
        t1 = time.perf_counter()
        ctx.set_database(dw, dl, spec["b"], spec["C"])
        t2 = time.perf_counter()
        ctx.set_queries(qw, ql)
        ctx.synchronize()
        t3 = time.perf_counter()
        s3 = snap(probe)
        ctx.map(spec["R"])
        t4 = time.perf_counter()
        s4 = snap(probe)
        ctx.map(spec["R"])
        t5 = time.perf_counter()
        ctx.close()
        t6 = time.perf_counter()
        s6 = snap(probe)
        row["wall_ms"] = {"init": (t1 - t0) * 1e3, "set_database": (t2 - t1) * 1e3, "set_queries": (t3 - t2) * 1e3,
                          "first_map": (t4 - t3) * 1e3, "second_map": (t5 - t4) * 1e3, "destroy": (t6 - t5) * 1e3}
        row["host_phase_ms_load"] = {k: round(s3[k] - s0[k], 3) for k in s0 if s3[k] - s0[k] > 0.0005}
        row["host_phase_ms_first_map"] = {k: round(s4[k] - s3[k], 3) for k in s0 if s4[k] - s3[k] > 0.0005}
        row["host_phase_ms_rest"] = {k: round(s6[k] - s4[k], 3) for k in s0 if s6[k] - s4[k] > 0.0005}
        row["wall_ms"] = {k: round(v, 3) for k, v in row["wall_ms"].items()}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if not a.keep_one:
        probe.close()
    else:
        keeper.close()


if __name__ == "__main__":
    main()
