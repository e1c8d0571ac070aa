#!/bin/bash
# quick_bench.sh WORKLOAD[:clustered] ...  -- one line per workload (no CPU baseline)
for spec in "$@"; do
  wl=${spec%%:*}; extra=""; [[ "$spec" == *:clustered ]] && extra="--clustered"
  steps=20; [[ "$wl" == "C3" ]] && steps=5
  python bench.py --workload $wl $extra $BENCH_EXTRA --steps $steps --warmup 2 --no-cpu-baseline 2>/dev/null > /tmp/qb.json
  python - <<'PY'
import json
d = json.load(open("/tmp/qb.json"))
print("%-40s ms/step %8.4f kernel_ms %8.4f frac %.3f value %.3e U %d pack %.3f fin %.3f ms (%s)" % (
    d["config"]["workload"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
    d["value"], d["config"]["updates_per_step"], d["phase_ms"]["pack"], d["phase_ms"]["finalize"], d["packing"]))
PY
done
