#!/bin/bash
# Sweep the accumulate kernel's workgroup-count target (SECEDO_TARGET_WGS) on one workload.
wl=${1:-C2}; shift
for w in "$@"; do
  SECEDO_TARGET_WGS=$w python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > /tmp/sweep.json
  python - "$w" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
print("target_wgs", sys.argv[1], "ms/step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
done
