#!/usr/bin/env python3
"""Median FETCH_SIZE / WRITE_SIZE per kernel from a tools/pmc_traffic.sh run:  python tools/pmc_traffic_summary.py gpurun_out/<dir>

Units and the gfx950 correction as in tools/pmc_sq_summary.py: both counters are KiB; FETCH_SIZE reports half the bytes of wide
coalesced reads and is doubled."""
import collections
import csv
import glob
import os
import re
import statistics
import sys


def norm(name):
    name = name.split("(")[0].replace("void ", "").replace("nae::", "").strip()
    return re.sub(r"<.*", "", name)


def main():
    d = sys.argv[1]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, "p[1-9]*", "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = collections.defaultdict(float)
        names = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = norm(row["Kernel_Name"])
        for (disp, ctr), v in per_dispatch.items():
            vals[names[disp]][ctr].append(v)
    print("| kernel | launches seen | read bytes (FETCH_SIZE KiB x 1024 x 2) | write bytes (WRITE_SIZE KiB x 1024) | total |")
    print("|---|---|---|---|---|")
    for k, c in sorted(vals.items(), key=lambda kv: -statistics.median(kv[1].get("WRITE_SIZE", [0]))):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        rd = statistics.median(c["FETCH_SIZE"]) * 1024 * 2
        wr = statistics.median(c["WRITE_SIZE"]) * 1024
        if rd + wr < 1e8:
            continue
        print("| %s | %d | %.4g | %.4g | %.4g |" % (k, len(c["FETCH_SIZE"]), rd, wr, rd + wr))


if __name__ == "__main__":
    main()
