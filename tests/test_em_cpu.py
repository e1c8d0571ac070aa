"""CPU tests of the EM refinement's oracle (oracle/em_oracle.c) against the vectors of the compiled
reference (tests/golden/em_cases.npz: the five cases of the reference's own
tests/test_expectation_maximization.cpp and three random pileups) and, when oracle/_ref is present,
live against the reference."""
import os

import numpy as np
import pytest

from oracle import bindings as ob
from secedo_amd.pileup import FlatPileup

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def em_cases():
    z = np.load(os.path.join(GOLDEN, "em_cases.npz"), allow_pickle=False)
    names = sorted({k.split("__")[0] for k in z.files if "__" in k})
    out = []
    for n in names:
        g = lambda k: z[n + "__" + k]  # noqa: E731
        p = FlatPileup(g("chr_locus_off"), g("locus_pos"), g("locus_entry_off"), g("read_ids"), g("id_base"))
        out.append((n, p, g("id_to_pos"), g("prob_in"), g("prob_out"), int(g("iterations"))))
    return float(z["theta"]), out


THETA, CASES = em_cases()


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_oracle_matches_reference_vectors(name):
    _, p, i2p, prob_in, prob_out, iters = next(c for c in CASES if c[0] == name)
    got, it = ob.oracle_em(p, i2p, THETA, prob_in)
    assert np.array_equal(got, prob_out) and it == iters


def test_reference_test_expectations_hold_for_the_vectors():
    """The assertions of tests/test_expectation_maximization.cpp:15-85, on the stored reference output."""
    out = {c[0]: c[4] for c in CASES}
    assert out["one_cell"][0] == 1.0
    assert abs(out["two_cells_same"][1] - out["two_cells_same"][0]) <= 1e-3
    assert abs(abs(out["two_cells_different"][0] - out["two_cells_different"][1]) - 1.0) <= 1e-3
    assert np.max(np.abs(out["four_cells_22"] - np.array([0, 0, 1, 1]))) <= 1e-3
    assert np.max(np.abs(out["four_cells_31"] - np.array([0, 1, 0, 0]))) <= 1e-3


def test_oracle_rejects_what_the_reference_cannot_index():
    _, p, i2p, prob_in, _, _ = next(c for c in CASES if c[0] == "random_40")
    with pytest.raises(RuntimeError):
        ob.oracle_em(p, i2p, THETA, prob_in[:20])  # group ids up to 39 index a 20-vector
    with pytest.raises(RuntimeError):
        ob.oracle_em(p, i2p[:10], THETA, prob_in)  # groups outside id_to_pos


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_matches_reference_live():
    from tests.pileup_gen import random_pileup
    rng = np.random.default_rng(5)
    for seed, n in ((1, 30), (2, 90)):
        p = random_pileup(seed, n, 2, 250, 15, 300)
        prob = np.clip(rng.random(n), 0.02, 0.98)
        i2p = rng.permutation(n).astype(np.uint32)
        assert np.array_equal(ob.oracle_em(p, i2p, 1e-3, prob)[0], ob.ref_em(p, i2p, 1e-3, prob))
