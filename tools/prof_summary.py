#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: kernel-trace stats and PMC counters per kernel.
usage: prof_summary.py results.db [more.db ...]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("hg::", "")


def main():
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        print("== %s" % path)
        rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
        print("%-28s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for n, k, tot, avg, pct in rows:
            print("%-28s %6d %12.1f %12.2f %7.2f" % (short(n), k, tot, avg, pct))
        try:
            cur = c.execute("select * from counters_collection limit 1")
            cols = [d[0] for d in cur.description]
        except sqlite3.Error:
            cols = []
        if cols and c.execute("select count(*) from counters_collection").fetchone()[0]:
            kcol = "kernel_name" if "kernel_name" in cols else "name"
            q = "select %s, counter_name, avg(value), count(*) from counters_collection group by 1,2 order by 1,2" % kcol
            print("%-28s %-26s %16s %5s" % ("kernel", "counter", "avg/dispatch", "n"))
            for n, cn, v, k in c.execute(q):
                print("%-28s %-26s %16.1f %5d" % (short(n), cn, v, k))


if __name__ == "__main__":
    main()
