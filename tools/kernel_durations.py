#!/usr/bin/env python3
"""Per-dispatch durations of the kernels whose name contains a pattern, from a rocprofv3 --kernel-trace database (rocpd sqlite).
usage: kernel_durations.py results.db pattern"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
try:
    rows = c.execute("select name, start, end from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall()
except sqlite3.Error as e:
    print("no `kernels` view:", e, [r[0] for r in c.execute("select name from sqlite_master")][:40])
    sys.exit(1)
d = [(e - s) / 1e3 for _, s, e in rows]
print(pat, "dispatches", len(d), "us each:", " ".join("%.1f" % x for x in d))
