#!/bin/bash
# quick_bench.sh WORKLOAD[:clustered] ...  -- one line per workload (no CPU baseline)
for spec in "$@"; do
  wl=${spec%%:*}; extra=""; [[ "$spec" == *:clustered ]] && extra="--clustered"
  steps=20; [[ "$wl" == "C3" ]] && steps=5
  python bench.py --workload $wl $extra --steps $steps --warmup 2 --no-cpu-baseline 2>/dev/null > /tmp/qb.json
  python - <<'PY'
import json
d = json.load(open("/tmp/qb.json"))
print("%-40s ms/step %8.4f kernel_ms %8.4f frac %.3f value %.3e U %d prep %.2fs" % (
    d["config"]["workload"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
    d["value"], d["config"]["updates_per_step"], d["prepare_host_s"]))
PY
done
