"""The CPU oracle against the compiled reference on fresh random pileups.

Runs only where oracle/_ref/libsecedo_ref.so exists (built in the container from
/root/reference by oracle/Makefile; the .so travels to the GPU box, the sources do not)."""
import numpy as np
import pytest

from oracle import bindings as ob
from tests import golden_util as gu
from tests.pileup_gen import random_pileup

pytestmark = pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built")

CASES = [
    # seed, cells, chr, loci, cov, gap_max, mfl, T
    (11, 24, 2, 300, 8, 250, 1000, 1),
    (12, 24, 2, 300, 8, 250, 1000, 3),
    (13, 24, 2, 300, 8, 250, 200, 1),
    (14, 24, 2, 300, 8, 250, 200, 2),
    (15, 50, 1, 800, 15, 3000, 1000, 8),
    (16, 16, 4, 100, 5, 40, 1000, 1),
]


@pytest.mark.parametrize("seed,n,nchr,L,cov,gap,mfl,T", CASES)
def test_oracle_equals_reference(seed, n, nchr, L, cov, gap, mfl, T):
    p = random_pileup(seed, n, nchr, L, cov, gap, dup_frac=0.05, triple_frac=0.3, skip_frac=0.15,
                      n_groups=n + 5)
    rng = np.random.default_rng(seed)
    g2p = rng.integers(0, n, size=n + 5).astype(np.uint32)  # several groups share a row
    for norm in ob.NORMALIZATIONS:
        got = ob.oracle_compute(p, n, mfl, g2p, 0.01, 0.5, 0.02, T, norm)
        ref = ob.ref_compute(p, n, mfl, g2p, 0.01, 0.5, 0.02, T, norm)
        assert gu.normwise_err(got, ref) <= 1e-12
        if T == 1:
            assert np.array_equal(got, ref)  # same summation order => bit-identical
