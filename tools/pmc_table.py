#!/usr/bin/env python3
"""pmc_table.py <dir> -- per-kernel per-dispatch averages of every counter in the rocprofv3 counter_collection.csv
files under <dir> (all kernels; tools/pmc.sh prints the accumulate kernels only)."""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s per-dispatch %.4g" % (c, v / n[(k, c)]))
