#!/usr/bin/env python3
"""bench_split.py WORKLOAD MFL -- packing time when fragments outlive max_fragment_length (reads are cut by
flushes): device path against the host emulation; both must give the same counters."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import secedo_amd  # noqa: E402
from secedo_amd.synth import CONFIGS, synth_config  # noqa: E402

name, mfl = sys.argv[1], int(sys.argv[2])
n = CONFIGS[name][0]
p = synth_config(name)
out = {"workload": name, "mfl": mfl}
for mode in ("device", "host"):
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        plan.set_packing(mode)
        res = plan.upload(p, None, n)
        plan.prepare_resident(res, n, mfl, 8)  # warm-up (allocations)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.prepare_resident(res, n, mfl, 8)
        torch.cuda.synchronize()
        out[mode + "_pack_ms"] = (time.perf_counter() - t0) * 1e3
        acc = plan.new_acc()
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        torch.cuda.synchronize()
        out[mode + "_counts"] = list(plan.last_counts()) + [plan.num_reads]
out["equal"] = out["device_counts"] == out["host_counts"]
print(json.dumps(out))
