#!/usr/bin/env python3
"""divide_cluster_demo.py [WORKLOAD] -- one level of the reference's divide_cluster() (spectral_clustering.cpp:
311-440) on a synthetic two-clone pileup, every heavy step on the GPU and the pileup resident in HBM:

    Filter::filter            (:336-337)  -> secedo_amd.filter_resident
    computeSimilarityMatrix   (:354-356)  -> SimilarityMatrixPlan.prepare_resident / accumulate / finalize
    laplacian + eig_sym       (:127-138)  -> secedo_amd.smallest_eigenpairs
    FIEDLER cut               (:218-228)  -> sign of the second eigenvector (host; GMM / k-means are out of scope)
    expectation_maximization  (:375-377)  -> secedo_amd.em.refine_resident

Prints one JSON line with the time of every step and the purity of the split."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import secedo_amd  # noqa: E402
from secedo_amd import em  # noqa: E402
from secedo_amd.synth import CONFIGS, synth_config  # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def main(name="C2"):
    n = CONFIGS[name][0]
    p = synth_config(name)
    identity = np.arange(n, dtype=np.uint32)
    steps = {}
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        res = plan.upload(p, identity, n)
        for rep in range(2):  # the second pass is the steady state (allocations done)
            (flt, coverage), steps["filter_ms"] = timed(lambda: secedo_amd.filter_resident(plan, res, identity, 0.01, 4))
            # the synthetic loci are all informative by construction; the similarity matrix is built on the
            # unfiltered pileup so that its size is the workload's (the filter keeps what passes its test)
            def similarity():
                plan.prepare_resident(res, n, 1000, 8)
                acc = plan.new_acc()
                plan.accumulate(acc, 0.01, 0.5, 0.01)
                return plan.finalize(acc, "ADD_MIN")
            sim, steps["similarity_ms"] = timed(similarity)
            (vals, vecs, info), steps["eigenpairs_ms"] = timed(lambda: secedo_amd.smallest_eigenpairs(sim, 20, 7))
            fiedler = vecs[:, 1]
            cluster = (fiedler >= 0).double()  # FIEDLER: the zero cut
            prob = (0.1 + 0.8 * cluster).contiguous()
            _, steps["em_ms"] = timed(lambda: em.refine_resident(res["off"], res["n_loci"], res["n_entries"], res["idb"],
                                                                 res["g2p"], 1e-3, prob))
    side = (prob > 0.5).cpu().numpy()
    upper = side[n // 2:].mean()
    purity = max(upper, 1 - upper) * 0.5 + max(1 - side[: n // 2].mean(), side[: n // 2].mean()) * 0.5
    print(json.dumps({"workload": name, "cells": n, "loci_kept_by_filter": flt["n_loci"], **steps,
                      "eigen_cycles": info["cycles"], "lambda_1": float(vals[1]), "split_purity": float(purity)}))
    return purity


if __name__ == "__main__":
    main(*(sys.argv[1:2]))
