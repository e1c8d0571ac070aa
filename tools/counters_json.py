#!/usr/bin/env python3
"""counters_json.py <dir> -- profiles/r04_counters.json from the SQ counter tables tools/pmc.sh left under <dir>
(pmc_C3.txt, pmc_C3_clustered.txt): per workload the per-dispatch counters of the dominant accumulate kernel,
and kernel_cycles = its average duration in the kernel stats of the same run x the shader clock."""
import csv
import json
import os
import re
import sys

R = sys.argv[1]
CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md; the LDS-atomic microbenchmark reports the same clock_mhz


def table(path):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
            out.setdefault(name, {})
        else:
            m = re.match(r"\s+(\S+)\s+per-dispatch\s+(\S+)", line)
            if m and name:
                out[name][m.group(1)] = float(m.group(2))
    return out


def avg_ns(stats_csv, needle):
    for r in csv.DictReader(open(stats_csv)):
        if needle in r["Name"]:
            return float(r["AverageNs"])
    return None


res = {"_comment": "per-dispatch SQ counters of the dominant accumulate kernel (rocprofv3 --pmc passes over `python bench.py "
                   "--steps 3 --warmup 1`, tools/pmc.sh); kernel_cycles = average duration in the kernel stats x 2.4 GHz. "
                   "valu_issue_frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel_cycles)."}
for key, pmc, stats in (("C3", "pmc_C3.txt", "C3_kernel_stats.csv"),
                        ("C3_clustered", "pmc_C3_clustered.txt", "C3_clustered_kernel_stats.csv")):
    if not os.path.exists(os.path.join(R, pmc)):
        continue
    t = table(os.path.join(R, pmc))
    kern = next((k for k in t if "::accumulate_" in k), None)
    if not kern:
        continue
    c = dict(t[kern])
    c["kernel"] = kern
    ns = avg_ns(os.path.join(R, stats), kern.split("::")[-1].split("<")[0] + "<")
    if ns:
        c["kernel_avg_ns"] = ns
        c["kernel_cycles"] = ns * CLOCK_GHZ
    res[key] = c
print(json.dumps(res, indent=1))
