#!/usr/bin/env python3
"""bench_filter.py [C2|C3] -- throughput of the device locus filter on a resident pileup.

Algorithmic bytes (SURVEY.md 8d style): every input entry read once (4 B read id + 2 B id|base),
12 B per locus of offsets/positions, every kept entry written once (6 B)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import secedo_amd
from secedo_amd.synth import CONFIGS, synth_config

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_cells = CONFIGS[name][0]
p = synth_config(name)
rng = np.random.default_rng(1)
i2p = np.full(n_cells, secedo_amd.NO_POS, dtype=np.uint32)
inside = np.flatnonzero(rng.random(n_cells) < 0.5)
i2p[inside] = np.arange(len(inside), dtype=np.uint32)
with secedo_amd.SimilarityMatrixPlan(0) as plan:
    res = plan.upload(p, np.arange(n_cells, dtype=np.uint32), n_cells)
    times = []
    for it in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out, cov = secedo_amd.filter_resident(plan, res, i2p, 0.01, 4)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    ms = sorted(times[2:])[len(times[2:]) // 2]
    b_alg = 6 * p.n_entries + 12 * p.n_loci + 6 * out["n_entries"] + 12 * out["n_loci"]
    print(json.dumps({"workload": name, "entries_in": p.n_entries, "loci_in": p.n_loci,
                      "entries_out": out["n_entries"], "loci_out": out["n_loci"], "avg_coverage": cov,
                      "ms": ms, "algorithmic_bytes": b_alg, "GBps": b_alg / ms / 1e6,
                      "frac_of_8TBps": b_alg / ms / 1e6 / 8000}))
