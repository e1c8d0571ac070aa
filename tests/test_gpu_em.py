"""GPU parity of the EM refinement (include/secedo_em.h) against oracle/em_oracle.c and the vectors of the
compiled reference. Floating point: the per-locus and per-cell sums are formed in a different order
than the reference's sequential loops, so probabilities agree to 1e-9 (observed ~1e-13), the number
of iterations exactly."""
import numpy as np
import pytest

import secedo_amd
from oracle import bindings as ob
from tests.test_em_cpu import CASES, THETA

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_hip_matches_reference_vectors(name):
    _, p, i2p, prob_in, prob_out, iters = next(c for c in CASES if c[0] == name)
    got, it = secedo_amd.expectation_maximization(p, i2p, 1, THETA, prob_in)
    assert np.max(np.abs(got - prob_out)) <= TOL if len(prob_out) else True
    assert it == iters


def test_hip_matches_oracle_on_a_larger_pileup_and_in_place_update():
    from secedo_amd.synth import synth_config
    p = synth_config("C1")  # 64 cells, two clones
    rng = np.random.default_rng(9)
    prob = np.clip(np.where(np.arange(64) >= 32, 0.65, 0.35) + 0.2 * rng.standard_normal(64), 0.01, 0.99)
    ref, it_ref = ob.oracle_em(p, np.arange(64, dtype=np.uint32), 1e-3, prob)
    mine = prob.copy()
    got, it = secedo_amd.expectation_maximization(p, np.arange(64, dtype=np.uint32), 8, 1e-3, mine)
    assert it == it_ref and np.max(np.abs(got - ref)) <= TOL
    assert np.array_equal(mine, got)  # the in/out vector of the reference signature
    assert np.all(got[:32] < 0.05) and np.all(got[32:] > 0.95)  # the two clones are told apart


def test_u32_ids_and_errors():
    from tests.pileup_gen import random_pileup
    n = 20000  # ids beyond the 14 bits of the reference's packing: the 32-bit id_base variant
    p = random_pileup(3, n, 1, 60, 400, 300)
    assert int(p.id_base.max()) > 0xFFFF
    rng = np.random.default_rng(4)
    prob = np.clip(rng.random(n), 0.05, 0.95)
    ref, it_ref = ob.oracle_em(p, np.arange(n, dtype=np.uint32), 1e-3, prob)
    got, it = secedo_amd.expectation_maximization(p, np.arange(n, dtype=np.uint32), 1, 1e-3, prob)
    assert it == it_ref and np.max(np.abs(got - ref)) <= TOL
    with pytest.raises(secedo_amd.SecedoError):  # group ids index past the probability vector
        secedo_amd.expectation_maximization(p, np.arange(n, dtype=np.uint32), 1, 1e-3, prob[:100])
    with pytest.raises(secedo_amd.SecedoError):  # groups outside id_to_pos
        secedo_amd.expectation_maximization(p, np.arange(50, dtype=np.uint32), 1, 1e-3, prob)
