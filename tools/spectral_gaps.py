#!/usr/bin/env python3
"""spectral_gaps.py <rocprofv3 out dir> -- GPU busy time, idle time and the largest idle gaps (the host's
Rayleigh-Ritz steps) of the LAST solve in a kernel trace of tools/bench_spectral.py (solves are delimited by
k_init_block, the start vector of a solve)."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if "k_init_block" in r[2]]
seg = rows[starts[-1]:]
seg = [r for r in seg if "secedo" in r[2] or "rocclr" in r[2]]
# cut at the last spectral kernel (bench code follows)
last = max(i for i, r in enumerate(seg) if "spectral" in r[2])
seg = seg[:last + 1]
span = seg[-1][1] - seg[0][0]
busy, cur, gaps = 0, seg[0][0], []
for s, e, n in seg:
    if s > cur:
        gaps.append(s - cur)
    busy += max(0, e - max(s, cur))
    cur = max(cur, e)
gaps.sort(reverse=True)
big = [g for g in gaps if g > 200e3]
print("kernels %d, span %.2f ms, busy %.2f ms, idle %.2f ms" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
print("gaps > 0.2 ms: %d, total %.2f ms, mean %.2f ms; all other gaps: %d, total %.2f ms, mean %.1f us" % (
    len(big), sum(big) / 1e6, (sum(big) / max(1, len(big))) / 1e6, len(gaps) - len(big),
    (sum(gaps) - sum(big)) / 1e6, (sum(gaps) - sum(big)) / max(1, len(gaps) - len(big)) / 1e3))
