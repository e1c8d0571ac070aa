#!/bin/bash
# prof_pack.sh WORKLOAD STEPS -- rocprofv3 kernel stats of bench.py steps (device packing included), readable table
wl=$1; steps=${2:-10}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$wl
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline > $out.log 2>&1
python tools/kstats.py $out
