#!/bin/bash
# pmc_kernels.sh NAME "COUNTER LIST" "KERNEL REGEX" -- bench args...   one rocprofv3 PMC pass over bench.py;
# per-dispatch averages of every counter for the kernels whose name matches the regex
name=$1; ctrs=$2; pat=$3; shift 3
export TMPDIR=/tmp SECEDO_BENCH_NO_CHILD=1
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rm -rf $out
rocprofv3 --pmc $ctrs --output-format csv -d $out -- python bench.py "$@" --no-cpu-baseline > /dev/null 2> $out.err
python - "$pat" $(find $out -name "*counter_collection.csv") <<'PY'
import csv, sys, re, collections
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in sys.argv[2:]:
  for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if not pat.search(k): continue
    k = k.replace("void ", "").replace("secedo::(anonymous namespace)::", "").split("(")[0][:40]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s per-dispatch %.4g" % (c, v / n[(k, c)]))
PY
rm -rf $out
