#!/usr/bin/env python3
"""timeline.py <rocprofv3 out dir> -- the kernels of one bench step in start order (offset from the step's first
kernel, duration, queue), from the newest *kernel_trace.csv under the directory. Steps are delimited by
k_entry_locus, the first kernel of the device packing."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"))
               for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if "k_entry_locus" in r[2]]
a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0 = rows[a][0]
prev_end = t0
for s, e, name, q in rows[a:b]:
    if "rocprim" in name or "hipcub" in name:
        short = "rocprim:" + ("radix_sort" if "radix_sort" in name else "scan" if "scan" in name else "other")
    else:
        short = name.replace("void ", "").replace("secedo::(anonymous namespace)::", "").split("(")[0][:40]
    print("%9.1f  +%7.1f us  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q[-3:], short))
    prev_end = max(prev_end, e)
