#!/bin/bash
# sweep_env.sh VAR "v1 v2 ..." WORKLOAD...  -- quick_bench under each value of an env knob
var=$1; vals=$2; shift 2
for v in $vals; do echo "== $var=$v"; env $var=$v tools/quick_bench.sh "$@"; done
