#!/bin/bash
# pmc.sh NAME "COUNTER LIST" -- bench args...   rocprofv3 PMC pass over bench.py; prints per-kernel sums
name=$1; ctrs=$2; shift 2
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rm -rf $out
rocprofv3 --pmc $ctrs --output-format csv -d $out -- python bench.py "$@" --no-cpu-baseline > /dev/null 2> $out.err
python - $(find $out -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in sys.argv[1:]:  # one file per process of the run (bench.py starts the LDS-atomic microbenchmark as a child)
  for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"][:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    if "accumulate" not in k and "correct_" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s per-dispatch %.4g" % (c, v / n[(k, c)]))
PY
