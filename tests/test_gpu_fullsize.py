"""BASELINE.json's full sizes. C2 (1000 cells x 50 K loci, the bench workload) is small enough for the
oracle (about a second), so it is compared outright. C3 (8000 cells x 100 K loci, 2.96e9 updates) would
take the oracle minutes: it is checked through what must hold at any size -- exact symmetry and zero
diagonal, the post-conditions of ADD_MIN, tile ranges composing bit for bit, and the GPU
packing agreeing with the sequential host emulation of the reference loop in every work counter and
in every bit of the matrix."""
import numpy as np
import pytest
import torch

import secedo_amd
from oracle import bindings as ob
from secedo_amd.synth import CONFIGS, synth_config
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def test_c2_full_size_against_the_oracle():
    n = CONFIGS["C2"][0]
    p = synth_config("C2")
    for norm in ("ADD_MIN", "EXPONENTIATE"):
        got = secedo_amd.compute_similarity_matrix(p, n, 1000, None, 0.01, 0.5, 0.01, 8, "", norm)
        ref = ob.oracle_compute(p, n, 1000, None, 0.01, 0.5, 0.01, 8, norm)
        assert gu.normwise_err(got, ref) <= 1e-9
        assert np.array_equal(got, got.T) and not np.any(np.diag(got))
        # the element-wise report over entries with |ref| > 1e-6 max|ref|
        big = np.abs(ref) > 1e-6 * np.max(np.abs(ref))
        rel = np.abs(got - ref)[big] / np.abs(ref)[big]
        print("\nC2 %s vs oracle, %d entries: norm-wise %.2e, element-wise relative max %.2e, 99.9-percentile %.2e"
              % (norm, rel.size, gu.normwise_err(got, ref), rel.max(), np.percentile(rel, 99.9)))
    u_ref, pairs_ref = ob.oracle_last_updates(), ob.oracle_last_read_pairs()
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        plan.prepare(p, n, 1000, None, 8)
        acc = plan.new_acc()
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        assert plan.last_counts() == (u_ref, pairs_ref) == (63956330, pairs_ref)


def test_c3_full_size_properties():
    n = CONFIGS["C3"][0]
    p = synth_config("C3")
    mats, counts = {}, {}
    for mode in ("device", "host"):
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.set_packing(mode)
            plan.prepare(p, n, 1000, None, 8)
            assert plan.used_device_packing == (mode == "device")
            acc = plan.new_acc()
            plan.accumulate(acc, 0.01, 0.5, 0.01)
            torch.cuda.synchronize()
            counts[mode] = plan.last_counts() + (plan.num_entries, plan.num_reads)
            mats[mode] = plan.finalize(acc, "ADD_MIN").clone()
            if mode == "device":
                # the same accumulator from three tile ranges, as three ranks would fill it
                parts = plan.new_acc()
                t = plan.num_tiles
                for lo, hi in ((0, t // 3), (t // 3, t // 2), (t // 2, t)):
                    plan.accumulate(parts, 0.01, 0.5, 0.01, lo, hi)
                assert torch.equal(parts, acc)
    assert counts["device"] == counts["host"]
    assert counts["device"][0] == 2955670215  # the updates the bench line reports for C3
    assert torch.equal(mats["device"], mats["host"])
    m = mats["device"]
    assert torch.equal(m, m.T) and not torch.any(torch.diagonal(m))
    off_diag_min = (m + torch.diag(torch.full((n,), float("inf"), device=m.device, dtype=m.dtype))).min()
    assert float(m.min()) == 0.0 and float(off_diag_min) == 0.0  # ADD_MIN: all >= 0, the best pair at 0


def digest_errors(z, sample_values, ref_max):
    """(norm-wise error, element-wise relative max, 99.9-percentile) of sampled entries against a digest."""
    err = np.abs(sample_values - z["sample_v"])
    big = np.abs(z["sample_v"]) > 1e-6 * ref_max
    rel = err[big] / np.abs(z["sample_v"][big])
    return float(np.max(err) / ref_max), float(np.max(rel)), float(np.percentile(rel, 99.9))


def test_c3_against_the_compiled_reference():
    """The headline configuration held against the REFERENCE itself: tests/golden/c3_reference_digest.npz is
    one run of the compiled, unmodified reference on C3 (8000 cells x 100 K loci, T = 8, ADD_MIN; 238 s in the
    build container, oracle/gen_golden.py c3_reference_run): 6000 sampled entries (uniform, inside diagonal
    tiles, in the last partial cell block, in the first cell block), the maximum, the sum and the sums of
    every 128-row block. Checked here: the sequence bench.py times (pileup resident in HBM -> prepare_resident
    -> assign_finalize) and the one-shot drop-in call, norm-wise 1e-9 (north_star), with the element-wise
    percentiles of SURVEY.md 8d printed."""
    z = np.load(gu.GOLDEN + "/c3_reference_digest.npz")
    n, mfl, eps, h, theta, T, _ = z["params"]
    n, mfl, T = int(n), int(mfl), int(T)
    p = synth_config("C3")
    assert p.n_entries == int(z["n_entries"]) and p.n_loci == int(z["n_loci"])
    ref_max, ii, jj = float(z["max_abs"]), z["sample_i"].astype(np.int64), z["sample_j"].astype(np.int64)
    ti, tj = torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda()
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        resident = plan.upload(p, None, n)
        plan.prepare_resident(resident, n, mfl, T)
        acc = plan.new_acc()
        out = plan.assign_finalize(acc, eps, h, theta, "ADD_MIN")
        torch.cuda.synchronize()
        assert plan.scale_log2 == 44
        normwise, rel_max, rel_999 = digest_errors(z, out[ti, tj].cpu().numpy(), ref_max)
        print("\nC3 assign_finalize vs compiled reference, %d sampled entries: norm-wise %.2e, element-wise "
              "relative max %.2e, 99.9-percentile %.2e" % (len(ii), normwise, rel_max, rel_999))
        assert normwise <= 1e-9
        assert abs(float(out.abs().max()) - ref_max) <= 1e-9 * ref_max
        assert abs(float(out.sum()) - float(z["total"])) <= 1e-9 * abs(float(z["total"]))
        blocks = torch.stack([out[lo:lo + 128].sum() for lo in range(0, n, 128)]).cpu().numpy()
        assert np.max(np.abs(blocks - z["row_block_sums"])) <= 1e-9 * np.max(np.abs(z["row_block_sums"]))
        assert rel_max <= 1e-7  # elements near 0 carry the reference's own cancellation error
        kept = out.clone()
    got = secedo_amd.compute_similarity_matrix(p, n, mfl, None, eps, h, theta, T, "", "ADD_MIN")
    assert np.array_equal(got, kept.cpu().numpy())  # the drop-in call returns the same bits
    assert digest_errors(z, got[ii, jj], ref_max)[0] <= 1e-9


@pytest.mark.parametrize("name", ["C2", "C3"])
def test_clustered_loci_against_the_compiled_reference(name):
    """SURVEY.md 8d "also run each with gap_max=300": the clustered-loci variants (about 2.8 loci per read, nine
    entries per cell block and locus: accumulate_masks + wide_pairs, the general packing path) held against the
    REFERENCE itself at full size: tests/golden/c{2,3}_clustered_reference_digest.npz are single runs of the
    compiled, unmodified reference (oracle/gen_golden.py c2_clustered_reference_run / c3_clustered_reference_run;
    5.4e8 and 2.5e10 updates) -- sampled entries (uniform, inside diagonal 64- and 128-cell tiles, last and first
    cell block), maximum, sum, 128-row block sums. Checked: the sequence `bench.py --clustered` times and the
    one-shot drop-in call, norm-wise 1e-9."""
    path = gu.GOLDEN + "/%s_clustered_reference_digest.npz" % name.lower()
    import os
    if not os.path.exists(path):
        pytest.skip("no digest of the reference for %s clustered" % name)
    z = np.load(path)
    n, mfl, eps, h, theta, T, _ = z["params"]
    n, mfl, T = int(n), int(mfl), int(T)
    p = synth_config(name, clustered=True)
    assert p.n_entries == int(z["n_entries"]) and p.n_loci == int(z["n_loci"])
    ref_max, ii, jj = float(z["max_abs"]), z["sample_i"].astype(np.int64), z["sample_j"].astype(np.int64)
    ti, tj = torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda()
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        resident = plan.upload(p, None, n)
        plan.prepare_resident(resident, n, mfl, T)
        acc = plan.new_acc()
        out = plan.assign_finalize(acc, eps, h, theta, "ADD_MIN")
        torch.cuda.synchronize()
        assert plan.pair_kernel == "accumulate_masks"
        normwise, rel_max, rel_999 = digest_errors(z, out[ti, tj].cpu().numpy(), ref_max)
        print("\n%s clustered assign_finalize vs compiled reference, %d sampled entries: norm-wise %.2e, "
              "element-wise relative max %.2e, 99.9-percentile %.2e" % (name, len(ii), normwise, rel_max, rel_999))
        assert normwise <= 1e-9
        assert abs(float(out.abs().max()) - ref_max) <= 1e-9 * ref_max
        assert abs(float(out.sum()) - float(z["total"])) <= 1e-9 * abs(float(z["total"]))
        blocks = torch.stack([out[lo:lo + 128].sum() for lo in range(0, n, 128)]).cpu().numpy()
        assert np.max(np.abs(blocks - z["row_block_sums"])) <= 1e-9 * np.max(np.abs(z["row_block_sums"]))
        assert torch.equal(out, out.T) and not torch.any(torch.diagonal(out))
        kept = out.clone()
    got = secedo_amd.compute_similarity_matrix(p, n, mfl, None, eps, h, theta, T, "", "ADD_MIN")
    assert np.array_equal(got, kept.cpu().numpy())  # the drop-in call returns the same bits
    assert digest_errors(z, got[ii, jj], ref_max)[0] <= 1e-9


def test_c5_full_size_properties():
    """C5 (32000 cells x 200 K loci, 1.05e10 updates; cell ids beyond the reference's 14 bits, so the
    32-bit id_base variant): GPU packing against the host emulation in counters and in every bit of the
    8.2 GB matrix, exact symmetry, zero diagonal."""
    n = CONFIGS["C5"][0]
    p = synth_config("C5")
    assert int(p.id_base.max()) > 0xFFFF
    counts, ref = {}, None
    for mode in ("host", "device"):
        with secedo_amd.SimilarityMatrixPlan(0) as plan:
            plan.set_packing(mode)
            plan.prepare(p, n, 1000, None, 8)
            acc = plan.new_acc()
            plan.accumulate(acc, 0.01, 0.5, 0.01)
            torch.cuda.synchronize()
            counts[mode] = plan.last_counts() + (plan.num_entries, plan.num_reads)
            m = plan.finalize(acc, "ADD_MIN")
            del acc
            if ref is None:
                ref = m.clone()
            else:
                assert torch.equal(m, ref)
                assert not torch.any(torch.diagonal(m))
                for lo in range(0, n, 4000):  # symmetry, a row block at a time (no 8 GB transpose copy)
                    assert torch.equal(m[lo:lo + 4000], m[:, lo:lo + 4000].T)
            del m
    assert counts["device"] == counts["host"]
    assert counts["device"][0] == 10517284027  # the updates the bench line reports for C5


def test_c2_matrix_through_the_spectral_step_and_em():
    """The consumers at C2 size: similarity matrix of the two-clone synthetic pileup (resident in HBM) ->
    smallest eigenpairs against LAPACK -> the second eigenvector splits the clones (the reference's
    FIEDLER rule) -> EM refinement started from that split keeps it and agrees with the oracle."""
    from oracle import spectral_oracle as so
    n = CONFIGS["C2"][0]
    p = synth_config("C2")
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        plan.prepare(p, n, 1000, None, 8)
        acc = plan.new_acc()
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        sim = plan.finalize(acc, "ADD_MIN").clone()
    vals, vecs, info = secedo_amd.smallest_eigenpairs(sim, 20, 7)
    assert info["converged"]
    w, v = so.eig_sym(so.laplacian_fast(sim.cpu().numpy()))
    assert np.max(np.abs(vals - w[:20])) <= 1e-8
    fiedler = vecs[:, 1].cpu().numpy()
    assert abs(abs(float(fiedler @ v[:, 1])) - 1.0) <= 1e-8  # separated eigenvalue: same vector up to sign
    side = fiedler >= 0
    assert np.all(side[: n // 2] == side[0]) and np.all(side[n // 2:] == side[-1]) and side[0] != side[-1]
    prob = np.where(side == side[-1], 0.9, 0.1)  # cluster b = the clone of the upper half
    ref, it_ref = ob.oracle_em(p, np.arange(n, dtype=np.uint32), 1e-3, prob)
    got, it = secedo_amd.expectation_maximization(p, np.arange(n, dtype=np.uint32), 8, 1e-3, prob)
    assert it == it_ref and np.max(np.abs(got - ref)) <= 1e-9
    assert np.all(got[: n // 2] < 0.05) and np.all(got[n // 2:] > 0.95)
