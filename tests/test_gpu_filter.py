"""Locus filter on the GPU (through the C-ABI) against the reference vectors and the CPU oracle:
exact equality of every output array (integer / index work) and of the average coverage."""
import numpy as np
import pytest

import secedo_amd
from oracle import bindings as ob
from tests import golden_util as gu
from tests.pileup_gen import from_rows, random_pileup

pytestmark = pytest.mark.gpu


def _same(got_pileup, got_cov, expect):
    o_chr, o_pos, o_off, o_rid, o_idb, cov = expect
    assert np.array_equal(got_pileup.chr_locus_off, o_chr)
    assert np.array_equal(got_pileup.locus_pos, o_pos)
    assert np.array_equal(got_pileup.locus_entry_off, o_off)
    assert np.array_equal(got_pileup.read_ids, o_rid)
    assert np.array_equal(got_pileup.id_base, o_idb)
    assert got_cov == cov


@pytest.mark.parametrize("name", gu.filter_fixture_names())
def test_hip_filter_matches_reference_vectors(name):
    p, i2p, theta, cp, expect = gu.load_filter(name)
    got, cov = secedo_amd.Filter(theta, cp).filter(p, i2p, "", 1)
    _same(got, cov, expect)


@pytest.mark.parametrize("seed,n,theta,cp", [(401, 80, 0.01, 4), (402, 300, 0.001, 1), (403, 20, 0.05, 3)])
def test_hip_filter_matches_oracle_random(seed, n, theta, cp):
    rng = np.random.default_rng(seed)
    p = random_pileup(seed, n, 3, 400, 40, 500, err=0.15)
    i2p = np.arange(n, dtype=np.uint32)
    i2p[rng.random(n) < 0.35] = secedo_amd.NO_POS
    got, cov = secedo_amd.Filter(theta, cp).filter(p, i2p)
    _same(got, cov, ob.oracle_filter(p, i2p, theta, cp))


def test_filter_edge_cases():
    f = secedo_amd.Filter(0.01)
    # empty pileup and empty chromosomes (reference tests/test_is_significant.cpp:108-112 Filter.Empty)
    got, cov = f.filter(from_rows([[], []]), np.arange(4, dtype=np.uint32))
    assert got.n_loci == 0 and got.n_entries == 0 and cov == 0 and got.chr_locus_off.tolist() == [0, 0, 0]
    # every cell outside the sub-cluster: nothing survives
    p = random_pileup(404, 30, 1, 100, 30, 300, err=0.2)
    got, cov = f.filter(p, np.full(30, secedo_amd.NO_POS, dtype=np.uint32))
    assert got.n_loci == 0 and cov == 0


def test_filter_then_similarity_matrix_stays_in_hbm():
    """divide_cluster's first two steps (spectral_clustering.cpp:336-337, :354-356): filter, then the
    similarity matrix of the filtered pileup, with the pileup resident in HBM in between."""
    n = 120
    rng = np.random.default_rng(405)
    p = random_pileup(405, n, 2, 500, 40, 400, err=0.15)
    i2p = np.full(n, secedo_amd.NO_POS, dtype=np.uint32)
    inside = np.flatnonzero(rng.random(n) < 0.6)
    i2p[inside] = np.arange(len(inside), dtype=np.uint32)
    o_chr, o_pos, o_off, o_rid, o_idb, cov_ref = ob.oracle_filter(p, i2p, 0.01, 4)
    fp = secedo_amd.FlatPileup(o_chr, o_pos, o_off, o_rid, o_idb)
    ref = ob.oracle_compute(fp, len(inside), 1000, i2p, 0.01, 0.5, 0.01, 4, "ADD_MIN")
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        res = plan.upload(p, np.arange(n, dtype=np.uint32), n)
        filtered, cov = secedo_amd.filter_resident(plan, res, i2p, 0.01, 4)
        assert cov == cov_ref and filtered["n_loci"] == len(o_pos) and filtered["n_entries"] == len(o_rid)
        plan.prepare_resident(filtered, len(inside), 1000, 4)
        acc = plan.new_acc()
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        got = plan.finalize(acc, "ADD_MIN").cpu().numpy()
    assert gu.normwise_err(got, ref) <= 1e-9
