#!/bin/bash
# pmc_any.sh TAG WORKLOAD COUNTER... -- one PMC pass (kernel trace only) over bench.py, packed-resident;
# prints the per-launch average of every counter for the accumulate kernel
tag=$1; wl=$2; shift 2
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --packed-resident > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'accumulate_tiles' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
