#!/usr/bin/env python3
"""bench_em.py [WORKLOAD ...] -- EM refinement on a synthetic pileup resident in HBM: ms per call and
iterations. (The CPU figure quoted in DESIGN.md comes from the oracle under tests/, which tools may not import.)"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import secedo_amd  # noqa: E402
from secedo_amd import em  # noqa: E402
from secedo_amd.synth import CONFIGS, synth_config  # noqa: E402

for name in sys.argv[1:] or ["C2", "C3"]:
    n = CONFIGS[name][0]
    p = synth_config(name)
    rng = np.random.default_rng(1)
    prob0 = np.clip(np.where(np.arange(n) >= n // 2, 0.6, 0.4) + 0.2 * rng.standard_normal(n), 0.01, 0.99)
    idb = np.ascontiguousarray(p.id_base, dtype=np.uint16 if int(p.id_base.max()) <= 0xFFFF else np.uint32)
    d_off = torch.from_numpy(p.locus_entry_off.astype(np.int64)).cuda()
    d_idb = torch.from_numpy(idb.view(np.int16 if idb.dtype == np.uint16 else np.int32)).cuda()
    d_i2p = torch.from_numpy(np.arange(n, dtype=np.int32)).cuda()
    L, E = len(p.locus_pos), len(p.read_ids)
    times = []
    for rep in range(3):
        d_prob = torch.from_numpy(prob0).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = em.refine_resident(d_off, L, E, d_idb, d_i2p, 1e-3, d_prob)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    line = {"workload": name, "cells": n, "entries": E, "iterations": iters, "ms": min(times),
            "clones_separated": bool((d_prob[: n // 2] < 0.05).all() and (d_prob[n // 2:] > 0.95).all())}
    print(json.dumps(line))
