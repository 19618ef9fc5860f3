#!/usr/bin/env python3
"""Kernel timeline of one steady-state step from a rocprofv3 --kernel-trace rocpd database: durations and the
idle gaps between consecutive GPU operations.  usage: step_timeline.py kt_results.db"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end from kernels order by start").fetchall()
try:
    rows += [("memcpy", s, e) for s, e in c.execute("select start,end from memory_copies")]
    rows.sort(key=lambda r: r[1])
except sqlite3.Error:
    pass
names = [re.sub(r"[<(].*", "", r[0]).replace("void ", "").replace("hg::", "") for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_hist"]
i0, i1 = idx[-3], idx[-2]
prev = None
busy = 0
for i in range(i0, i1 + 1):
    n, s, e = names[i], rows[i][1], rows[i][2]
    gap = (s - prev) / 1e3 if prev else 0
    print("%-30s dur %8.1f us  gap before %7.1f us" % (n, (e - s) / 1e3, gap))
    if i < i1: busy += e - s
    prev = e
span = (rows[i1][1] - rows[i0][1]) / 1e3
print("step span %.1f us, GPU busy %.1f us, idle %.1f us" % (span, busy / 1e3, span - busy / 1e3))
