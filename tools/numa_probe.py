#!/usr/bin/env python3
"""Is the box-to-box spread of the literal C2 call (2.7 ms on one box, 4.5 on the next) NUMA placement?  The same 200 calls
(tools/literal_outliers.py c2) in child processes confined to node 0's processors, node 1's, and unconfined."""
import os
import subprocess
import sys


def cpus(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "literal_outliers.py")
nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
print("nodes", nodes, "gpu numa:", [open(p).read().strip() for p in
      [os.path.join(r, "numa_node") for r, ds, fs in os.walk("/sys/class/drm") if "numa_node" in fs][:2]] or "?")
for label, mask in [("unconfined", None)] + [("node %d" % n, cpus(n)) for n in nodes] + [("unconfined", None)]:
    def pre(mask=mask):
        if mask:
            os.sched_setaffinity(0, mask)
    r = subprocess.run([sys.executable, tool, "c2", "200"], capture_output=True, text=True, preexec_fn=pre)
    print("%-11s %s" % (label, (r.stdout.splitlines() or [r.stderr[-300:]])[0]), flush=True)
