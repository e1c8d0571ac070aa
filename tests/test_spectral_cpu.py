"""CPU tests of the spectral step's oracle (oracle/spectral_oracle.py) against the one known-answer
vector the reference holds for it (tests/test_spectral_clustering.cpp:15-26)."""
import os

import numpy as np

from oracle import spectral_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_laplacian_matches_reference_kat():
    kat = np.load(os.path.join(GOLDEN, "laplacian_kat.npz"))
    got = so.laplacian(kat["a"])
    assert np.max(np.abs(got - kat["expected"])) <= float(kat["tolerance"])  # the reference's own bound
    assert np.array_equal(got, got.T)
    # the printed digits of the expectation are good to 5e-8
    assert np.max(np.abs(got - kat["expected"])) <= 5e-8


def test_oracle_laplacian_zero_rows_and_fast_variant():
    rng = np.random.default_rng(3)
    a = rng.random((40, 40))
    a = a + a.T
    np.fill_diagonal(a, 0.0)
    a[5, :] = 0.0
    a[:, 5] = 0.0  # an isolated cell: 1/sqrt(0) := 0 (spectral_clustering.cpp:40-41)
    lap = so.laplacian(a)
    assert lap[5, 5] == 1.0 and np.all(lap[5, :5] == 0.0) and np.all(lap[5, 6:] == 0.0)
    assert np.max(np.abs(lap - so.laplacian_fast(a))) <= 1e-15
    w, v = so.eig_sym(lap)
    assert w[0] > -1e-14 and np.all(np.diff(w) >= 0)
    assert np.max(np.abs(lap @ v - v * w)) <= 1e-13


def test_host_eigensolver_of_the_rayleigh_ritz_step(tmp_path):
    """secedo_amd/csrc/sym_eig.cpp is plain C++ (the 192 x 192 projected eigenproblem of the block Lanczos
    iteration): compiled and run here on random, clustered and degenerate matrices."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sym_eig_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "secedo_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "sym_eig_test.cpp"),
                    os.path.join(root, "secedo_amd", "csrc", "sym_eig.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip() == "ok"
