#!/usr/bin/env python3
"""traffic_json.py <dir> -- profiles/r04_traffic.json from the FETCH_SIZE / WRITE_SIZE passes that
tools/refresh_profiles_r04.sh left under <dir>: per accumulate_tiles launch, with the counters calibrated
on a known 1 GiB stream in the same run (tools/fetch_calib.hip) as MI355X_MICROARCH.md (HBM) prescribes."""
import collections
import csv
import glob
import json
import os
import sys

R = sys.argv[1]


def per_kernel(sub, counter):
    # (every process of the run leaves a file: bench.py also starts the LDS-atomic microbenchmark as a child)
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(R, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


GIB_KIB = float(1 << 20)
cal_f = per_kernel("calib_FETCH_SIZE", "FETCH_SIZE")
cal_w = per_kernel("calib_WRITE_SIZE", "WRITE_SIZE")
pick = lambda d, name: next(sum(v) / len(v) for k, v in d.items() if name in k)
f4, f16, w16 = pick(cal_f, "read4") / GIB_KIB, pick(cal_f, "read16") / GIB_KIB, pick(cal_w, "write16") / GIB_KIB
fetch = per_kernel("pmc_FETCH_SIZE", "FETCH_SIZE")
write = per_kernel("pmc_WRITE_SIZE", "WRITE_SIZE")
# the kernels of one accumulate: the pair kernel (dominant), the per-tile correction + slab reduction
names = [k for k in fetch if any(s in k for s in ("accumulate_counts", "accumulate_tiles", "reduce_slabs", "correct_tiles", "correct_flagged"))]
name = next(k for k in names if "accumulate_" in k)
avg = lambda d, k: sum(d[k]) / len(d[k])
fk, wk = avg(fetch, name), avg(write, name)
f_all, w_all = sum(avg(fetch, k) for k in names), sum(avg(write, k) for k in names)
# the loads are 4 bytes per lane (entry32 / offsets), the stores 16 bytes per lane (slab flush)
hbm = (fk / f4 + wk / w16) * 1024.0
hbm_all = (f_all / f4 + w_all / w16) * 1024.0
print(json.dumps({
    "_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes over `python bench.py --steps 3 "
                "--warmup 1 --no-cpu-baseline`), per accumulate_tiles launch, KiB as the counters report them; "
                "calibration = counter / true KiB on a 1 GiB stream (tools/fetch_calib.hip, same run). "
                "hbm_bytes_per_launch = (FETCH_SIZE / calib_read4 + WRITE_SIZE / calib_write16) * 1024.",
    "calibration": {"FETCH_SIZE_per_true_KiB_read_4B_per_lane": f4, "FETCH_SIZE_per_true_KiB_read_16B_per_lane": f16,
                    "WRITE_SIZE_per_true_KiB_write_16B_per_lane": w16},
    "C3": {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes_per_launch": hbm,
           "hbm_bytes_uncorrected": (fk + wk) * 1024.0, "launches": len(fetch[name]), "kernel": name[:80],
           "hbm_bytes_per_accumulate_all_kernels": hbm_all, "kernels": [k[:60] for k in names],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, calibrated"}}, indent=1))
