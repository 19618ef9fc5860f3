#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) into HBM bytes
per launch per kernel -> profiles/latest_traffic.json (read by bench.py's roofline.traffic).

usage: pmc_traffic.py fetch_results.db write_results.db out.json

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB... as
reported by rocprofv3 on gfx950 they are multiples of 1024 B; FETCH_SIZE under-reports wide
(16 B/lane) coalesced streaming reads by 2x.  The kernels here read through scalar loads and
8-byte gathers, not wide streams, so NO doubling is applied; both raw counters are kept in the
output so the reader can re-derive the figure.
"""
import json
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"[<(].*", "", name).replace("void ", "").replace("hg::", "")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for n, v, k in c.execute("select %s, avg(value), count(*) from counters_collection where counter_name=? group by 1" % kcol, (counter,)):
        out[short(n)] = (v, k)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0.0, 0))[0]
        w = write.get(k, (0.0, 0))[0]
        out[k] = {"FETCH_SIZE_avg": f, "WRITE_SIZE_avg": w, "hbm_bytes_per_launch": (f + w) * 1024.0,
                  "launches_sampled": fetch.get(k, (0, 0))[1]}
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        print("%-16s fetch %12.1f KiB  write %12.1f KiB  -> %10.1f MB/launch" % (k, v["FETCH_SIZE_avg"], v["WRITE_SIZE_avg"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
