#!/usr/bin/env python3
"""fuzz_parity.py [N] [SEED] -- N random configurations (a wider net than tests/test_gpu_parity.py::
test_randomised_differential_sweep: more cells -- several 128-cell tiles, the sparse-loci kernels --, deeper
loci -- the count tile refused, ranges cut again --, group maps with shared rows, both tile sizes, fresh handles)
against the oracle: raw matrix, normalised matrix, both work counters; assign_finalize against accumulate +
finalize. Prints one line per failure and a summary; exit code 1 if any."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import secedo_amd  # noqa: E402
from oracle import bindings as ob  # noqa: E402
from tests import golden_util as gu  # noqa: E402
from tests.pileup_gen import random_pileup  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
START = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # skip the configurations before this one (same sequence)
VERBOSE = len(sys.argv) > 4
TOL = 1e-9
rng = np.random.default_rng(SEED)
fails = 0
t_start = time.time()
for it in range(N):
    shape = rng.choice(["small", "wide", "deep", "clustered"], p=[0.3, 0.3, 0.2, 0.2])
    if shape == "small":
        n, L, cov = int(rng.choice([3, 17, 64, 65, 130, 200])), int(rng.integers(40, 400)), int(rng.integers(3, 40))
    elif shape == "wide":   # several 128-cell blocks, few entries per block and locus
        n, L, cov = int(rng.choice([300, 700, 1500, 2600])), int(rng.integers(200, 1200)), int(rng.integers(4, 30))
    elif shape == "deep":   # few cells, many reads per locus
        n, L, cov = int(rng.choice([4, 12, 40, 100])), int(rng.integers(20, 120)), int(rng.integers(100, 900))
    else:                   # loci close together: multi-locus reads everywhere
        n, L, cov = int(rng.choice([50, 128, 129, 400])), int(rng.integers(100, 500)), int(rng.integers(10, 80))
    nchr = int(rng.integers(1, 5))
    gap = int(rng.choice([6, 40, 300, 3000])) if shape != "clustered" else int(rng.choice([3, 8, 20]))
    fmax = int(rng.choice([120, 400, 600]))
    mfl = int(rng.choice([90, 250, 1000]))
    # (the reference's tables have max_fragment_length rows, similarity_matrix.cpp:330: a read pair must not
    # share that many loci, or the reference -- and the oracle that restates it -- index out of bounds. Reads
    # longer than max_fragment_length linger until the next flush, so dense loci get the long limit.)
    if gap < 40 and mfl < 1000:
        mfl = 1000
    T = int(rng.choice([1, 2, 5, 8]))
    n_groups = n + int(rng.integers(0, 6))
    g2p = rng.integers(0, n, size=n_groups).astype(np.uint32) if rng.random() < 0.5 else None
    pseed, dup_frac, skip_frac = int(rng.integers(1, 1 << 30)), float(rng.choice([0.0, 0.05])), float(rng.choice([0.0, 0.2]))
    norm = secedo_amd.NORMALIZATIONS[it % 3]
    block = int(rng.choice([0, 64, 128]))
    if it < START:
        continue
    if VERBOSE:
        print("config", it, shape, n, nchr, L, cov, gap, fmax, mfl, T, block, norm, g2p is not None, flush=True)
    p = random_pileup(pseed, n, nchr, L, cov, gap, frag_min=30, frag_max=fmax, dup_frac=dup_frac, triple_frac=0.3,
                      skip_frac=skip_frac, n_groups=n_groups if g2p is not None else n)
    ctx = (it, shape, n, nchr, L, cov, gap, fmax, mfl, T, block, norm, g2p is not None, p.n_entries)
    ob.set_exact_binomials(0)  # the reference's own (wrapping) arithmetic everywhere, as the product returns it
    try:
        ref, raw = ob.oracle_compute(p, n, mfl, g2p, 0.01, 0.5, 0.02, T, norm, want_raw=True)
        counts_ref = (ob.oracle_last_updates(), ob.oracle_last_read_pairs())
        ob.set_direct_llr_sum(True)
        ref_direct, raw_direct = ob.oracle_compute(p, n, mfl, g2p, 0.01, 0.5, 0.02, T, norm, want_raw=True)
    finally:
        ob.set_exact_binomials(False)
        ob.set_direct_llr_sum(False)
    per_pair = counts_ref[1] / max(1.0, n * (n - 1) / 2)
    # One tolerance, 1e-9 norm-wise, against the oracle's EXACT per-pair sums (direct mode). The reference's own
    # arithmetic -- two large sums per cell pair, subtracted at the end, similarity_matrix.cpp:428 -- departs from
    # those sums by `cancel` on deep pileups (4e-7 at 2.5e6 pairs per cell pair); nobody can reproduce that
    # without its summation order, so against the reference-arithmetic matrix the allowance is TOL + cancel
    # (triangle inequality), and cancel is reported.
    cancel = gu.normwise_err(raw, raw_direct)
    tol_direct = TOL
    tol_ref = TOL + 2.0 * cancel
    # (EXPONENTIATE maps an absolute error of the raw matrix to at most a quarter of it in a matrix whose maximum is
    # about 1/2: relative to the raw maximum that is a factor max|raw| / 2)
    exp_factor = max(1.0, float(np.max(np.abs(raw)))) if norm == "EXPONENTIATE" else 1.0
    # (the normalised matrix: a division by the maximum, or an exponential, carries the reference's cancellation on in
    # its own way -- SCALE_MAX_1 divides by an entry that has it too --, so its allowance is measured on the normalised
    # matrices of the two oracle modes, not taken over from the raw ones)
    finite = np.all(np.isfinite(ref)) and np.all(np.isfinite(ref_direct))
    cancel_norm = gu.normwise_err(ref, ref_direct) if finite else cancel
    tol_norm = (TOL + 2.0 * max(cancel, cancel_norm)) * exp_factor
    problems = []
    fixed_point = ""
    try:
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.prepare(p, n, mfl, g2p, T, block_cells=block)
            acc = plan.new_acc()
            plan.accumulate(acc, 0.01, 0.5, 0.02)
            got = plan.finalize(acc, norm).cpu().numpy()
            got_raw = plan.finalize_raw(acc).cpu().numpy()
            counts = plan.last_counts()
            fixed_point = "scale 2^%d, pair bound %d, %.3g pairs per cell pair, the reference's own cancellation %.2g" % (
                plan.scale_log2, plan.pair_bound, per_pair, cancel)
            acc2 = torch.full_like(acc, 12345)
            got2 = plan.assign_finalize(acc2, 0.01, 0.5, 0.02, norm).cpu().numpy()
            if not torch.equal(acc[:plan.acc_elems], acc2[:plan.acc_elems]):
                problems.append("assign != accumulate into zeroes")
            if not np.array_equal(got, got2, equal_nan=True):
                problems.append("assign_finalize != finalize")
        if counts != counts_ref:
            problems.append("counters %r != %r" % (counts, counts_ref))
        if gu.normwise_err(got_raw, raw_direct) > tol_direct:
            problems.append("raw vs direct sums %.3g" % gu.normwise_err(got_raw, raw_direct))
        if gu.normwise_err(got_raw, raw) > tol_ref:
            problems.append("raw vs reference arithmetic %.3g" % gu.normwise_err(got_raw, raw))
        if np.all(np.isfinite(ref)) and not (norm == "EXPONENTIATE" and np.max(np.abs(raw)) > 30):
            if gu.normwise_err(got, ref) > tol_norm:
                problems.append("normalised %.3g" % gu.normwise_err(got, ref))
            if finite and gu.normwise_err(got, ref_direct) > TOL * exp_factor:
                problems.append("normalised vs direct sums %.3g" % gu.normwise_err(got, ref_direct))
        if not np.array_equal(got, got.T, equal_nan=True):
            problems.append("not symmetric")
    except Exception as exc:  # noqa: BLE001
        problems.append("exception %r" % (exc,))
    if problems:
        fails += 1
        print("FAIL", ctx, problems, fixed_point, flush=True)
    elif it % 20 == 0 or cancel > TOL:
        print("ok up to", it, "(%.0f s)" % (time.time() - t_start), ctx, fixed_point if cancel > TOL else "", flush=True)
print("fuzz: %d configurations, %d failures, %.0f s" % (N, fails, time.time() - t_start), flush=True)
sys.exit(1 if fails else 0)
