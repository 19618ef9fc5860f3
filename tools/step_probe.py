#!/usr/bin/env python3
"""Per-step wall time of the headline workload (C2) under engine options and kernel-timing levels:
   python tools/step_probe.py [key=value ...]   -> one line per timing level 0 / 1 / 2 with the steps' times and statistics."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from hashgan_amd import _native, metric

spec = bench.WORKLOADS[sys.argv[1]] if len(sys.argv) > 1 and sys.argv[1] in bench.WORKLOADS else bench.WORKLOADS["c2"]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith(("levels=", "every=", "n="))]
levels = [int(x) for x in next((a[7:] for a in sys.argv[1:] if a.startswith("levels=")), "0,1,1,2,0").split(",")]
every_opt = int(next((a[6:] for a in sys.argv[1:] if a.startswith("every=")), "4"))
nsteps = int(next((a[2:] for a in sys.argv[1:] if a.startswith("n=")), "24"))
qw, ql, dw, dl = bench.build_packed(spec, 0, spec["N"])
ctx = _native.Context(0)
for k, v in opts:
    ctx.set_option(k, int(v))
ctx.set_database(dw, dl, spec["b"], spec["C"])
ctx.set_queries(qw, ql)
R = spec["R"]
for _ in range(3):
    ctx.map(R)
for level in levels:
    ctx.set_option("timing_every", every_opt if level == 1 else 1)
    ctx.timing_enable(level)
    ctx.timing_reset()
    ts = []
    for _ in range(nsteps):
        ctx.synchronize()
        t0 = time.perf_counter()
        a, r = ctx.map(R)
        ts.append((time.perf_counter() - t0) * 1e3)
    tim = ctx.timing_read()
    print("timing=%d steps %s  fused=%d leftovers=%d fallbacks=%d requeried=%d rebets=%d cap_boost=%d bytes=%d" % (level, " ".join("%.3f" % t for t in ts), ctx.get_stat("ap_fused"),
          ctx.get_stat("rank_leftovers"), ctx.get_stat("optimistic_fallbacks"), ctx.get_stat("optimistic_requeried"), ctx.get_stat("optimistic_rebets"), ctx.get_stat("cap_boost"), ctx.get_stat("device_bytes")))
    print("   ", {k: round(v[0] / max(v[1], 1), 4) for k, v in tim.items()})
ctx.close()
