#!/usr/bin/env python3
"""gaps.py <rocprofv3 out dir> -- busy time and idle gaps on the GPU per bench step, from the newest
*kernel_trace.csv under the directory (the steps are delimited by k_entry_locus, the first
kernel of the device packing)."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if "k_entry_locus" in r[2]]
steps = []
for a, b in zip(starts[2:-1], starts[3:]):  # skip warm-up
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0]
    busy = 0
    cur_end = seg[0][0]
    gaps = []
    for s, e, name in seg:
        if s > cur_end:
            gaps.append((s - cur_end, name))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    steps.append((span, busy, sorted(gaps, reverse=True)[:6], len(seg)))
span, busy, gaps, n = steps[len(steps) // 2]
print("kernels per step %d, span %.1f us, busy %.1f us, idle %.1f us" % (n, span / 1e3, busy / 1e3, (span - busy) / 1e3))
for g, name in gaps:
    print("   gap %.1f us before %s" % (g / 1e3, name.split("(")[0][-50:]))
