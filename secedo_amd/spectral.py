"""Spectral step on the similarity matrix: host-side mirror of the reference's
``laplacian()`` (spectral_clustering.cpp:33-52) and of the eigen-decomposition inside
``spectral_clustering()`` (``arma::eig_sym``, spectral_clustering.cpp:136-138), of which the reference
uses the 20 smallest eigenvalues and the eigenvectors of the 7 smallest.

The matrix stays where `SimilarityMatrixPlan.finalize` left it (HBM); the C-ABI is
include/secedo_spectral.h. No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _as_device_matrix(similarity):
    import torch
    if not isinstance(similarity, torch.Tensor):
        raise TypeError("expected a torch tensor")
    if similarity.dim() != 2 or similarity.shape[0] != similarity.shape[1]:
        raise ValueError("the similarity matrix must be square")
    if similarity.dtype != torch.float64 or not similarity.is_cuda or not similarity.is_contiguous():
        raise ValueError("expected a contiguous float64 CUDA tensor")
    return similarity


def laplacian(similarity):
    """Normalised graph Laplacian ``I - D^-1/2 A D^-1/2`` (reference: laplacian(),
    spectral_clustering.cpp:33-52; rows that sum to zero keep 1/sqrt(0) := 0, :40-41).

    `similarity`: float64 CUDA tensor (n x n, symmetric, zero diagonal) -> CUDA tensor, or a numpy
    array -> numpy array (staged through the GPU)."""
    import torch
    host = isinstance(similarity, np.ndarray)
    a = torch.from_numpy(np.ascontiguousarray(similarity, dtype=np.float64)).cuda() if host else similarity
    a = _as_device_matrix(a)
    out = torch.empty_like(a)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().secedo_laplacian_device(a.data_ptr(), a.shape[0], out.data_ptr(), stream))
    return out.cpu().numpy() if host else out


def smallest_eigenpairs(similarity, n_values=20, n_vectors=7, tol=0.0, max_cycles=0, device=0):
    """The `n_values` smallest eigenvalues (ascending, as ``arma::eig_sym`` orders them) of the
    normalised Laplacian of `similarity`, and the eigenvectors of the `n_vectors` smallest as the
    columns of an (n x n_vectors) array -- what spectral_clustering() takes from eig_sym
    (spectral_clustering.cpp:141-143, :166, :171-172, :221, :235-237).

    `similarity`: numpy array (host in, host out) or float64 CUDA tensor (eigenvectors stay on the
    device, returned as a column-major (n x n_vectors) tensor view). Both counts are clipped to n.
    Returns (eigenvalues ndarray, eigenvectors, info dict)."""
    import torch
    L = _lib.lib()
    info = _lib.SpectralInfo()
    if isinstance(similarity, np.ndarray):
        a = np.ascontiguousarray(similarity, dtype=np.float64)
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("the similarity matrix must be square")
        n = a.shape[0]
        n_values = min(n_values, n, 32)
        n_vectors = min(n_vectors, n_values)
        vals = np.empty(n_values, dtype=np.float64)
        vecs = np.empty((max(n_vectors, 1), n), dtype=np.float64)  # column-major n x k == row-major k x n
        _lib.check(L.secedo_spectral_eigs(device, _lib.ptr(a), n, n_values, n_vectors, tol, max_cycles,
                                          _lib.ptr(vals), _lib.ptr(vecs), C.byref(info)))
        vec_out = vecs[:n_vectors].T
    else:
        a = _as_device_matrix(similarity)
        n = a.shape[0]
        n_values = min(n_values, n, 32)
        n_vectors = min(n_vectors, n_values)
        vals = np.empty(n_values, dtype=np.float64)
        vecs = torch.empty((max(n_vectors, 1), n), dtype=torch.float64, device=a.device)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _lib.check(L.secedo_spectral_eigs_device(a.device.index or 0, a.data_ptr(), n, n_values, n_vectors, tol,
                                                 max_cycles, _lib.ptr(vals), vecs.data_ptr(), C.byref(info),
                                                 stream))
        vec_out = vecs[:n_vectors].T
    return vals, vec_out, {"cycles": info.cycles, "block_products": info.block_products,
                           "converged": bool(info.converged),
                           "max_residual_vectors": info.max_residual_vectors,
                           "max_residual_values": info.max_residual_values}
