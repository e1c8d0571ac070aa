#!/usr/bin/env python3
"""kstats.py <dir> -- readable per-kernel table from the newest rocprofv3 *_kernel_stats.csv under <dir>."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    name = r["Name"]
    if "rocprim" in name:
        short = "rocprim:" + ("radix_sort" if "radix_sort" in name else "scan" if "scan" in name else "other")
    else:
        short = name.replace("void ", "").replace("secedo::(anonymous namespace)::", "").split("(")[0][:46]
    print("%-48s calls %5s avg %10.1f us total %10.1f us  %5.1f%%" % (
        short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, float(r["Percentage"])))
