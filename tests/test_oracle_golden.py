"""The CPU oracle against the vectors generated from the compiled reference (CPU only)."""
import numpy as np
import pytest

from oracle import bindings as ob
from tests import golden_util as gu

TOL = 1e-9  # norm-wise; the oracle actually lands at <= 1e-13


@pytest.mark.parametrize("name", gu.fixture_names())
def test_oracle_matches_reference_vectors(name):
    p, cases = gu.load(name)
    assert cases
    for c in cases:
        got = ob.oracle_compute(p, c["num_cells"], c["mfl"], c["g2p"], c["eps"], c["h"], c["theta"],
                                c["T"], c["norm"])
        ref = c["out"]
        assert got.shape == ref.shape
        assert gu.normwise_err(got, ref) <= 1e-12, (name, c["T"], c["norm"])
        assert np.array_equal(got, got.T, equal_nan=True)
        assert np.all(np.diag(got) == 0)


def test_oracle_llr_kat_table():
    """D(x_s,x_d) read off reference matrices vs the oracle's restated formulas."""
    z = np.load(gu.GOLDEN + "/kat_llr_table.npz")
    for pi, (eps, h, theta) in enumerate(z["params"]):
        for ci, (xs, xd) in enumerate(z["combos"]):
            d = ob.oracle_log_prob_diff(int(xs), int(xd), eps, h, theta) \
                - ob.oracle_log_prob_same(int(xs), int(xd), eps, h, theta)
            assert abs(d - z["table"][pi, ci]) <= 1e-12 * max(1.0, abs(d)), (eps, h, theta, xs, xd)


def test_probe_facts():
    """tail drop, thread-count dependence, joint multi-locus term."""
    p, cases = gu.load("probe_single_locus")
    assert np.all(cases[0]["out"] == 0)  # ADD_MIN of an all-zero D
    assert np.all(cases[1]["out"][~np.eye(4, dtype=bool)] == 0.5)  # EXPONENTIATE
    p, cases = gu.load("probe_flush_T1_vs_T2")
    t1, t2 = cases[0]["out"], cases[3]["out"]
    assert np.any(t1 != 0) and np.all(t2 == 0)
    p, cases = gu.load("probe_multi_locus_pair")
    expo = cases[1]["out"]
    d01 = np.log(1.0 / expo[0, 1] - 1.0)
    assert abs(d01 - 0.0171106405755368) < 1e-9  # D(1,1), not D(1,0)+D(0,1)


def test_invalid_normalization():
    p, cases = gu.load("probe_single_locus")
    with pytest.raises(ValueError):
        ob.oracle_compute(p, 4, 1000, None, 0.01, 0.5, 0.01, 1, "BOGUS")


def test_oracle_refuses_read_pairs_beyond_the_reference_tables():
    """A read pair that shares >= max_fragment_length loci makes the reference index past its
    max_fragment_length-sized tables (similarity_matrix.cpp:314-317, :330): the oracle returns an error for
    such a pileup instead of following it out of bounds (it used to crash the test process)."""
    import pytest
    from oracle import bindings as ob
    from tests.pileup_gen import from_rows
    # two reads of different cells over 12 adjacent loci, then far loci of a third cell so that they flush
    rows = [(100 + l, [(1, 0, l & 3), (2, 1, (l + l // 5) & 3)]) for l in range(12)]
    rows += [(50000 + 5000 * k, [(10 + k, 2, 0)]) for k in range(6)]
    p = from_rows([rows])
    ob.oracle_compute(p, 3, 13, None, 0.01, 0.5, 0.01, 1, "ADD_MIN")  # 12 shared loci < 13 table rows: fine
    with pytest.raises(ValueError, match="max_fragment_length"):
        ob.oracle_compute(p, 3, 12, None, 0.01, 0.5, 0.01, 1, "ADD_MIN")
