"""Locus filter: the CPU oracle against the reference vectors and the compiled
reference, and the product's host-side significance test against both. CPU only."""
import numpy as np
import pytest

import secedo_amd
from oracle import bindings as ob
from tests import golden_util as gu
from tests.pileup_gen import random_pileup


def test_is_significant_reference_kats_and_vectors():
    """tests/golden/filter_kat.npz: the five known-answer strings of the reference's
    tests/test_is_significant.cpp:46-90 (as base counts) + 3000 decisions of the compiled reference."""
    z = np.load(gu.GOLDEN + "/filter_kat.npz")
    expect_first = [0, 1, 0, 0, 0]  # Cov52OneDifferent, Cov52TenDifferent, Cov59TwoDifferent, AtLimit, Paradox
    assert z["significant"][:5].tolist() == expect_first
    for c, th, cp, want in zip(z["counts"], z["theta"], z["cell_proportion"], z["significant"]):
        assert ob.oracle_is_significant(c, float(th), int(cp)) == bool(want)
        assert secedo_amd.Filter(float(th), int(cp)).is_significant(c) == bool(want)


@pytest.mark.parametrize("name", gu.filter_fixture_names())
def test_oracle_filter_matches_reference_vectors(name):
    p, i2p, theta, cp, expect = gu.load_filter(name)
    got = ob.oracle_filter(p, i2p, theta, cp)
    for a, b in zip(got[:5], expect[:5]):
        assert np.array_equal(a, b)
    assert got[5] == expect[5]


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built")
def test_oracle_filter_equals_reference_live():
    rng = np.random.default_rng(5)
    for seed in range(3):
        n = 50
        p = random_pileup(300 + seed, n, 2, 200, 30, 300, err=0.2)
        i2p = np.arange(n, dtype=np.uint32)
        drop = rng.random(n) < 0.4
        i2p[drop] = ob.NO_POS
        for cp in (0, 4):
            a = ob.oracle_filter(p, i2p, 0.01, cp)
            b = ob.ref_filter(p, i2p, 0.01, cp)
            assert all(np.array_equal(x, y) for x, y in zip(a[:5], b[:5])) and a[5] == b[5]


def test_filter_argument_errors():
    with pytest.raises(ValueError):
        secedo_amd.Filter(0.01, 7)
    with pytest.raises(ValueError):
        secedo_amd.Filter(0.01).is_significant([1, 2, 3])
