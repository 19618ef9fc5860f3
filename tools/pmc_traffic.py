#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) into HBM bytes
per launch per kernel -> profiles/latest_traffic.json (read by bench.py's roofline.traffic).

usage: pmc_traffic.py fetch_results.db write_results.db out.json [workload [base.json]]
       with a workload name (c5): the passes were run with `--workload c5`; the result is merged under "_workloads"/<name> of
       base.json (default profiles/latest_traffic.json), whose own entries (the headline step's) stay as they are

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE come in KiB; on gfx950
FETCH_SIZE reports HALF the bytes of wide (16 B per lane) coalesced streaming reads.  The matrix-core
select kernels (k_select_mx*) stream the database image, codes and labels with 16-byte direct-to-LDS
loads, so their FETCH_SIZE is doubled; the other kernels read through scalar loads, 1-8 byte gathers
and LDS, and are left as reported.  WRITE_SIZE is uncalibrated (taken as reported).  Both raw counters
stay in the output so the figure can be re-derived.  The file is stamped with a fingerprint of
hashgan_amd/csrc (bench.py quotes it only for the sources it was measured on).
"""
import json
import os
import re
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WIDE_READERS = ("k_select_mx",)          # kernels whose reads are 16 B per lane streams


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
        fx = 2.0 if k.startswith(WIDE_READERS) else 1.0
        out[k] = {"FETCH_SIZE_avg_KiB": f, "WRITE_SIZE_avg_KiB": w, "fetch_correction": fx,
                  "hbm_bytes_per_launch": (fx * f + w) * 1024.0, "launches_sampled": fetch.get(k, (0, 0))[1]}
    from bench import kernel_sources_sha
    out["_kernel_sources_sha"] = kernel_sources_sha()
    wl = sys.argv[4] if len(sys.argv) > 4 else None
    out["_collected"] = time.strftime("%Y-%m-%d") + ", rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 10 --warmup 3%s` on MI355X" % (" --workload " + wl if wl else "")
    if wl:
        base = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "latest_traffic.json")
        try:
            with open(base) as fh:
                whole = json.load(fh)
        except (OSError, ValueError):
            whole = {}
        whole.setdefault("_workloads", {})[wl] = out
        with open(sys.argv[3], "w") as fh:
            json.dump(whole, fh, indent=1)
    else:
        with open(sys.argv[3], "w") as fh:
            json.dump(out, fh, indent=1)
    for k, v in out.items():
        if k.startswith("_"):
            continue
        print("%-16s fetch %12.1f KiB  write %12.1f KiB  -> %10.1f MB/launch" % (k, v["FETCH_SIZE_avg_KiB"], v["WRITE_SIZE_avg_KiB"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
