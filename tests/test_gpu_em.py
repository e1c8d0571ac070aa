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


def test_many_iterations_and_the_iteration_limit():
    """Iterations are launched four at a time: a refinement that needs 16 of them, limits that fall inside
    and at the end of a batch, and the error when the limit is too small."""
    from tests.pileup_gen import random_pileup
    n = 60
    p = random_pileup(3, n, 1, 300, 6, 400, err=0.2)
    rng = np.random.default_rng(3)
    prob = np.clip(0.5 + 0.05 * rng.standard_normal(n), 0.05, 0.95)
    i2p = np.arange(n, dtype=np.uint32)
    ref, it_ref = ob.oracle_em(p, i2p, 0.2, prob)
    assert it_ref == 16
    for limit in (0, 16, 17, 18):
        got, it = secedo_amd.expectation_maximization(p, i2p, 1, 0.2, prob.copy(), max_iterations=limit)
        assert it == it_ref and np.max(np.abs(got - ref)) <= TOL
    for limit in (15, 14, 3):
        with pytest.raises(secedo_amd.SecedoError):
            secedo_amd.expectation_maximization(p, i2p, 1, 0.2, prob.copy(), max_iterations=limit)


def test_scratch_pool_release_and_concurrent_callers():
    """The scratch is kept per device between calls; releasing it and two threads refining at once (the
    second gets an allocation of its own) give the same results."""
    import threading
    from secedo_amd import _lib
    from tests.pileup_gen import random_pileup
    n = 80
    p = random_pileup(5, n, 2, 200, 10, 400)
    rng = np.random.default_rng(5)
    prob = np.clip(rng.random(n), 0.05, 0.95)
    i2p = np.arange(n, dtype=np.uint32)
    ref, it_ref = ob.oracle_em(p, i2p, 1e-3, prob)
    _lib.lib().secedo_simmat_release_cache()
    results = [None, None]

    def run(k):
        for _ in range(5):
            results[k] = secedo_amd.expectation_maximization(p, i2p, 1, 1e-3, prob.copy())

    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    _lib.lib().secedo_simmat_release_cache()
    for got, it in results + [secedo_amd.expectation_maximization(p, i2p, 1, 1e-3, prob.copy())]:
        assert it == it_ref and np.max(np.abs(got - ref)) <= TOL
