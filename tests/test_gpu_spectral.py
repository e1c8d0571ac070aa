"""GPU parity tests of the spectral step (include/secedo_spectral.h) against oracle/spectral_oracle.py:
the reference's laplacian() restated, and LAPACK's symmetric eigensolver where the reference calls
arma::eig_sym (spectral_clustering.cpp:33-52, :136-138). Eigenvalues to 1e-8, eigenvectors
up to sign where the eigenvalue is separated, as invariant subspaces where it is not."""
import os

import numpy as np
import pytest

import secedo_amd
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def planted(n, k, seed, noise=0.3, isolated=()):
    """Symmetric, zero diagonal, non-negative: k planted groups of cells over a noisy floor."""
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, k, size=n)
    a = 1.0 + noise * rng.random((n, n)) + 2.0 * (lab[:, None] == lab[None, :])
    a = 0.5 * (a + a.T)
    np.fill_diagonal(a, 0.0)
    for i in isolated:
        a[i, :] = 0.0
        a[:, i] = 0.0
    return a, lab


def check_against_lapack(a, n_values, n_vectors, val_tol=1e-8):
    lap = so.laplacian_fast(a)
    w, v = so.eig_sym(lap)
    vals, vecs, info = secedo_amd.smallest_eigenpairs(a, n_values, n_vectors)
    n_values, n_vectors = len(vals), vecs.shape[1]
    assert info["converged"], info
    assert np.max(np.abs(vals - w[:n_values])) <= val_tol
    assert np.all(np.diff(vals) >= -1e-12)
    # orthonormal, and eigenvectors of the oracle's Laplacian
    assert np.max(np.abs(vecs.T @ vecs - np.eye(n_vectors))) <= 1e-10
    assert np.max(np.abs(lap @ vecs - vecs * vals[:n_vectors])) <= 2e-8
    # against LAPACK's vectors: eigenvalues closer than 1e-6 form a group whose invariant subspace is
    # compared (projector); an eigenvector is determined to about residual / gap
    i = 0
    while i < n_vectors:
        j = i + 1
        while j < len(w) and w[j] - w[j - 1] <= 1e-6:
            j += 1
        if j <= n_vectors:  # the whole group was returned
            gaps = ([w[i] - w[i - 1]] if i > 0 else []) + ([w[j] - w[j - 1]] if j < len(w) else [])
            gap = min(gaps) if gaps else 1.0
            p_ref = v[:, i:j] @ v[:, i:j].T
            p_got = vecs[:, i:j] @ vecs[:, i:j].T
            assert np.max(np.abs(p_ref - p_got)) <= max(1e-8, 4e-9 / gap), (i, j, gap)
            if j == i + 1:  # sign convention: the component of largest magnitude is positive
                assert vecs[np.argmax(np.abs(vecs[:, i])), i] > 0
        i = j
    return vals, vecs, info


def test_laplacian_matches_reference_kat_and_oracle():
    kat = np.load(os.path.join(GOLDEN, "laplacian_kat.npz"))
    got = secedo_amd.laplacian(kat["a"])
    assert np.max(np.abs(got - kat["expected"])) <= float(kat["tolerance"])
    assert np.max(np.abs(got - so.laplacian(kat["a"]))) <= 1e-15
    a, _ = planted(300, 3, 5, isolated=(7, 120))
    got = secedo_amd.laplacian(a)
    assert np.max(np.abs(got - so.laplacian(a))) <= 1e-14
    assert np.array_equal(got, got.T) and got[7, 7] == 1.0 and np.all(got[7, :7] == 0.0)


def test_eigenpairs_of_the_reference_kat_matrix():
    kat = np.load(os.path.join(GOLDEN, "laplacian_kat.npz"))
    vals, vecs, info = check_against_lapack(kat["a"], 3, 3, val_tol=1e-12)
    assert abs(vals[0]) <= 1e-14


@pytest.mark.parametrize("n,k,seed", [(2, 1, 1), (5, 2, 2), (31, 2, 3), (33, 3, 4), (64, 2, 5), (100, 4, 6),
                                       (257, 2, 7), (1000, 3, 8)])
def test_eigenpairs_match_lapack(n, k, seed):
    a, _ = planted(n, k, seed)
    check_against_lapack(a, 20, 7)


@pytest.mark.parametrize("n,k,seed", [(230, 2, 41), (700, 3, 42), (1500, 5, 43)])
def test_thick_restart_matches_lapack(n, k, seed, monkeypatch):
    """The restart that keeps 64 Ritz vectors and the last Krylov block (the default from 12000 rows on),
    forced on small matrices; 230 rows: barely more than the 7 blocks of the basis."""
    monkeypatch.setenv("SECEDO_SPECTRAL_KEEP", "2")
    a, _ = planted(n, k, seed, isolated=(3,) if n > 500 else ())
    vals, vecs, info = check_against_lapack(a, 20, 7)
    assert info["max_residual_vectors"] <= 1e-9


def test_isolated_cells_disconnected_components_and_all_zero():
    # isolated cells: rows of zeros give eigenvalue exactly 1 (the Laplacian row is the unit vector)
    a, _ = planted(150, 2, 11, isolated=(0, 17, 149))
    check_against_lapack(a, 20, 7)
    # three components that do not touch: eigenvalue 0 three times
    blocks = [planted(40, 1, s)[0] for s in (21, 22, 23)]
    a = np.zeros((120, 120))
    for b, blk in enumerate(blocks):
        a[b * 40:(b + 1) * 40, b * 40:(b + 1) * 40] = blk
    vals, vecs, _ = check_against_lapack(a, 20, 7)
    assert np.max(np.abs(vals[:3])) <= 1e-12 and vals[3] > 0.5
    # all zero (the reference's SpectralClustering.AllZero input): L = I
    vals, vecs, info = secedo_amd.smallest_eigenpairs(np.zeros((50, 50)), 20, 7)
    assert np.max(np.abs(vals - 1.0)) <= 1e-14
    assert np.max(np.abs(vecs.T @ vecs - np.eye(7))) <= 1e-10


def test_fiedler_vector_separates_two_clones_end_to_end():
    """Pileup -> similarity matrix (resident in HBM) -> eigenvectors on the device: the sign of the
    second eigenvector splits the two planted clones (the reference's FIEDLER rule,
    spectral_clustering.cpp:218-228), and the device-resident path agrees with the host-array path."""
    import torch
    from secedo_amd.synth import synth_config
    p = synth_config("C1")  # 64 cells; cells >= 32 carry a different base at every third locus
    n = 64
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        plan.prepare(p, n, 1000, None, 8)
        acc = plan.new_acc()
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        sim = plan.finalize(acc, "ADD_MIN").clone()
    vals, vecs, info = secedo_amd.smallest_eigenpairs(sim, 20, 7)
    assert info["converged"] and isinstance(vecs, torch.Tensor) and vecs.is_cuda
    fiedler = vecs[:, 1].cpu().numpy()
    side = fiedler >= 0
    assert np.all(side[:32] == side[0]) and np.all(side[32:] == side[32]) and side[0] != side[32]
    vals_h, vecs_h, _ = secedo_amd.smallest_eigenpairs(sim.cpu().numpy(), 20, 7)
    assert np.array_equal(vals, vals_h) and np.array_equal(vecs.cpu().numpy(), vecs_h)
    w, _ = so.eig_sym(so.laplacian_fast(sim.cpu().numpy()))
    assert np.max(np.abs(vals - w[:20])) <= 1e-8


def test_larger_matrix_residuals_and_lapack_values():
    a, _ = planted(3000, 4, 31, noise=0.5)
    vals, vecs, info = check_against_lapack(a, 20, 7)
    assert info["max_residual_vectors"] <= 1e-9


def test_argument_errors():
    a, _ = planted(10, 2, 1)
    with pytest.raises(secedo_amd.SecedoError):
        secedo_amd._lib.check(secedo_amd._lib.lib().secedo_spectral_eigs(
            0, a.ctypes.data, 10, 11, 2, 0.0, 0, np.empty(11).ctypes.data, np.empty((2, 10)).ctypes.data, None))
    with pytest.raises(ValueError):
        secedo_amd.smallest_eigenpairs(np.zeros((3, 4)))


def test_divide_cluster_demo_runs_end_to_end():
    """tools/divide_cluster_demo.py: filter -> matrix -> eigenpairs -> Fiedler cut -> EM, resident in HBM, on the
    64-cell two-clone pileup: the split must be pure."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "divide_cluster_demo.py"), "C1"], check=True,
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    line = json.loads(out)
    assert line["cells"] == 64 and line["split_purity"] == 1.0


def test_reference_two_and_three_cluster_inputs():
    """The similarity matrices the reference's own spectral tests build (SpectralClustering.TwoClusters /
    ThreeClusters, tests/test_spectral_clustering.cpp:58-185; regenerated with the same libstdc++ generator,
    oracle/gen_golden.py: spectral_cases) through the GPU eigensolver: eigenpairs against LAPACK as everywhere,
    and the assignment those tests expect -- two clones of 50 by the sign of the Fiedler vector (the
    reference's FIEDLER rule, spectral_clustering.cpp:218-228; the test tolerates 3 misplaced cells, here none),
    three clones of 33 by the second and third eigenvector."""
    z = np.load(os.path.join(GOLDEN, "spectral_reference_inputs.npz"))
    a = z["two_clusters"]
    vals, vecs, _ = check_against_lapack(a, 20, 7)
    side = (vecs[:, 1] > 0).astype(np.int32)
    expect = z["two_expected"]
    assert np.array_equal(side, expect) or np.array_equal(1 - side, expect)
    assert vals[1] < 0.5 * vals[2]  # one clear gap after the two clone directions

    a = z["three_clusters"]
    vals, vecs, _ = check_against_lapack(a, 20, 7)
    emb = vecs[:, 1:3]
    expect = z["three_expected"]
    centres = np.stack([emb[expect == g].mean(axis=0) for g in range(3)])
    nearest = np.argmin(((emb[:, None, :] - centres[None, :, :]) ** 2).sum(axis=2), axis=1)
    assert np.array_equal(nearest, expect)  # 33 / 33 / 33, as the reference's test asks
    spread = max(np.linalg.norm(emb[expect == g] - centres[g], axis=1).max() for g in range(3))
    sep = min(np.linalg.norm(centres[g] - centres[h]) for g in range(3) for h in range(g))
    assert spread < 0.25 * sep
