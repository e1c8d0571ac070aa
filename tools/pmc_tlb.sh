#!/bin/bash
# pmc_tlb.sh WORKLOAD -- TLB counters of the accumulate kernel (own PMC pass, kernel trace only)
wl=${1:-C2}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_tlb_$wl
rm -rf $out
rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $out -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --packed-resident > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'accumulate_tiles' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
