"""spectral_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's spectral step for the parity tests of include/secedo_spectral.h:
  laplacian()          spectral_clustering.cpp:33-52
  eig_sym on it        spectral_clustering.cpp:136-138 (Armadillo -> LAPACK dsyevd; here numpy.linalg.eigh,
                       the same LAPACK driver family)
PARITY PIN: the reference's spectral code needs Armadillo/LAPACK headers and libraries that this image
does not have, so it cannot be compiled here. `laplacian` is pinned by the
reference's own known-answer test (tests/test_spectral_clustering.cpp:15-26, fixture
tests/golden/laplacian_kat.npz); the eigen-decomposition has no fixture in the reference (its tests
check cluster labels only) -- "parity unpinned" for the eigenpairs beyond agreement with LAPACK.
"""
import numpy as np


def laplacian(a):
    """spectral_clustering.cpp:33-52, loops restated with the reference's operation order."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    diag = np.zeros(n)
    for r in range(n):
        acc = 0.0
        for c in range(n):  # :38-39, sequential sum
            acc += a[r, c]
        diag[r] = acc
    diag = np.array([0.0 if v == 0 else 1.0 / np.sqrt(v) for v in diag])  # :40-41
    out = np.zeros((n, n))
    for r in range(n):
        for c in range(r + 1):  # :44-49
            out[r, c] = (1.0 if r == c else 0.0) - diag[r] * diag[c] * a[r, c]
            out[c, r] = out[r, c]
    return out


def laplacian_fast(a):
    """Vectorised variant for large matrices (row sums by numpy's pairwise summation: differs from the
    sequential sum in the last bits only)."""
    a = np.asarray(a, dtype=np.float64)
    d = a.sum(axis=1)
    s = np.where(d == 0, 0.0, 1.0 / np.sqrt(np.where(d == 0, 1.0, d)))
    return np.eye(a.shape[0]) - (s[:, None] * s[None, :]) * a


def eig_sym(lap):
    """Ascending eigenvalues and the eigenvectors in columns, as arma::eig_sym returns them."""
    w, v = np.linalg.eigh(np.asarray(lap, dtype=np.float64))
    return w, v
