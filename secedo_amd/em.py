"""EM refinement of a two-way split: host-side mirror of the reference's
``expectation_maximization(pos_data, id_to_pos, num_threads, theta, &prob_cluster_b)``
(expectation_maximization.hpp:26-30; caller spectral_clustering.cpp:375-377). C-ABI: include/secedo_em.h.
No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .pileup import FlatPileup, flatten


def expectation_maximization(pos_data, id_to_pos, num_threads, theta, prob_cluster_b, max_iterations=0, device=0):
    """Refines `prob_cluster_b` (probability of every cell position to belong to the second cluster).

    `pos_data`: list of chromosomes of PosData (as the reference takes it) or a FlatPileup.
    `num_threads` is accepted for signature parity; the reference ignores it too
    (expectation_maximization.cpp:127). Returns (refined probabilities as a new ndarray, iterations);
    like the reference's in/out vector, a float64 ndarray passed in is also updated in place."""
    p = pos_data if isinstance(pos_data, FlatPileup) else flatten(pos_data)
    i2p = np.ascontiguousarray(id_to_pos, dtype=np.uint32)
    prob = np.array(prob_cluster_b, dtype=np.float64)
    if prob.ndim != 1 or len(prob) == 0:
        raise ValueError("prob_cluster_b must be a non-empty vector")
    idb = np.ascontiguousarray(p.id_base)
    b16 = b32 = None
    if idb.dtype == np.uint16 or (len(idb) == 0 or int(idb.max()) <= 0xFFFF):
        b16 = np.ascontiguousarray(idb, dtype=np.uint16)
    else:
        b32 = np.ascontiguousarray(idb, dtype=np.uint32)
    off = np.ascontiguousarray(p.locus_entry_off, dtype=np.uint64)
    iters = C.c_uint32(0)
    _lib.check(_lib.lib().secedo_em_refine(device, _lib.ptr(off), len(off) - 1, _lib.ptr(b16), _lib.ptr(b32),
                                           _lib.ptr(i2p), len(i2p), theta, _lib.ptr(prob), len(prob),
                                           max_iterations, C.byref(iters)))
    if isinstance(prob_cluster_b, np.ndarray) and prob_cluster_b.dtype == np.float64:
        prob_cluster_b[...] = prob
    return prob, iters.value


def refine_resident(d_locus_entry_off, n_loci, n_entries, d_id_base, d_id_to_pos, theta, d_prob_cluster_b,
                    max_iterations=0):
    """The same on tensors resident in HBM (uint64 offsets as int64 tensor, id_base as int16/int32-typed
    uint tensors, float64 probabilities in/out). Returns the number of iterations."""
    import torch
    iters = C.c_uint32(0)
    b16 = d_id_base.data_ptr() if d_id_base.element_size() == 2 else None
    b32 = d_id_base.data_ptr() if d_id_base.element_size() == 4 else None
    dev = d_prob_cluster_b.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().secedo_em_refine_device(dev.index or 0, d_locus_entry_off.data_ptr(), n_loci, n_entries,
                                                  b16, b32, d_id_to_pos.data_ptr(), d_id_to_pos.numel(), theta,
                                                  d_prob_cluster_b.data_ptr(), d_prob_cluster_b.numel(),
                                                  max_iterations, C.byref(iters), stream))
    return iters.value
