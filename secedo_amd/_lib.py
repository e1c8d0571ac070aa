"""ctypes loader of secedo_amd/libsecedo_simmat.so (the C-ABI of include/secedo_simmat.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C secedo_amd/csrc``.
There is no Python or CPU fallback: a missing library raises ImportError here, and a missing
GPU makes every compute entry point fail with SECEDO_E_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SECEDO_LIB_PATH") or os.path.join(HERE, "libsecedo_simmat.so")  # env: A/B builds

OK = 0
E_INVALID_ARG = -1
E_INVALID_NORMALIZATION = -2
E_NO_DEVICE = -3
E_HIP = -4
E_STATE = -5
E_LIMIT = -6

_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p


class PileupInfo(C.Structure):
    _fields_ = [("n_loci", C.c_uint64), ("n_entries", C.c_uint64), ("num_cells", C.c_uint32),
                ("max_read_length", C.c_uint32)]


class SpectralInfo(C.Structure):
    _fields_ = [("cycles", C.c_uint32), ("block_products", C.c_uint32), ("converged", C.c_uint32),
                ("reserved", C.c_uint32), ("max_residual_vectors", C.c_double),
                ("max_residual_values", C.c_double)]


# int (*secedo_allreduce_sum_fn)(void *ctx, double *d_buffer, uint64_t count, void *stream)
ALLREDUCE_SUM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class SynthSpec(C.Structure):
    _fields_ = [
        ("num_cells", C.c_uint32), ("num_loci", C.c_uint32), ("num_chromosomes", C.c_uint32),
        ("gap_max", C.c_uint32), ("new_frag_prob", C.c_double), ("frag_min", C.c_uint32),
        ("frag_max", C.c_uint32), ("base_error", C.c_double), ("mate_frac", C.c_double),
        ("seed", C.c_uint64),
    ]


# name -> (restype, argtypes): every symbol the headers under include/ declare
SIGNATURES = {
    "secedo_simmat_normalization_from_string": (C.c_int, [C.c_char_p]),
    "secedo_simmat_last_error": (C.c_char_p, []),
    "secedo_simmat_version": (C.c_char_p, []),
    "secedo_simmat_device_count": (C.c_int, []),
    "secedo_simmat_compute": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                        C.c_uint32, C.c_int, _vp]),
    "secedo_simmat_release_cache": (None, []),
    "secedo_simmat_staging_acquire": (C.c_int, [_vp, _vp]),
    "secedo_simmat_staging_release": (None, []),
    "secedo_simmat_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "secedo_simmat_destroy": (None, [_vp]),
    "secedo_simmat_set_pileup": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp,
                                           C.c_uint32]),
    "secedo_simmat_set_pileup_device": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp,
                                                  C.c_uint32, C.c_uint32, C.c_uint64]),
    "secedo_simmat_set_packing": (C.c_int, [_vp, C.c_int]),
    "secedo_simmat_used_device_packing": (C.c_int, [_vp]),
    "secedo_simmat_prepare": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "secedo_simmat_num_tiles": (C.c_uint32, [_vp]),
    "secedo_simmat_block_cells": (C.c_uint32, [_vp]),
    "secedo_simmat_acc_elems": (C.c_uint64, [_vp]),
    "secedo_simmat_num_entries": (C.c_uint64, [_vp]),
    "secedo_simmat_num_reads": (C.c_uint64, [_vp]),
    "secedo_simmat_num_loci": (C.c_uint64, [_vp]),
    "secedo_simmat_zero_acc": (C.c_int, [_vp, _vp, _vp]),
    "secedo_simmat_accumulate": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_uint32,
                                           C.c_uint32, _vp, _vp]),
    "secedo_simmat_finalize": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "secedo_simmat_tiles_of_rows": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, C.POINTER(C.c_uint32)]),
    "secedo_simmat_assign": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, _vp, _vp]),
    "secedo_simmat_assign_list": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, _vp, C.c_uint32, _vp, _vp]),
    "secedo_simmat_assign_finalize": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp]),
    "secedo_simmat_accumulate_list": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, _vp, C.c_uint32, _vp, _vp]),
    "secedo_simmat_max_of_tiles": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(C.c_double), _vp]),
    "secedo_simmat_finalize_rows_max": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_double, _vp, _vp]),
    "secedo_simmat_finalize_rows": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, C.c_uint32, _vp, _vp]),
    "secedo_simmat_finalize_raw": (C.c_int, [_vp, _vp, _vp, _vp]),
    "secedo_simmat_last_counts": (C.c_int, [_vp, _u64p, _u64p]),
    "secedo_simmat_last_accumulate_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "secedo_simmat_last_pair_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "secedo_simmat_pair_kernel": (C.c_char_p, [_vp]),
    "secedo_simmat_llr": (C.c_double, [C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double]),
    "secedo_simmat_llr_closed_form": (C.c_double, [C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double]),
    "secedo_simmat_pair_bound": (C.c_uint64, [_vp]),
    "secedo_simmat_set_pair_bound": (C.c_int, [_vp, C.c_uint64]),
    "secedo_simmat_max_read_entries": (C.c_uint32, [_vp]),
    "secedo_simmat_cell_squares": (C.c_int, [_vp, _vp, _vp]),
    "secedo_simmat_set_scale_bounds": (C.c_int, [_vp, C.c_uint64, C.c_uint32]),
    "secedo_simmat_scale_log2": (C.c_int, [_vp]),
    "secedo_simmat_scale_bounds_state": (C.c_int, [_vp]),
    "secedo_simmat_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_uint32]),
    "secedo_simmat_get_devices": (C.c_int, [C.POINTER(C.c_int), C.c_uint32]),
    "secedo_is_significant": (C.c_int, [_vp, C.c_double, C.c_uint32]),
    "secedo_filter": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.c_double,
                                C.c_uint32, _vp, _vp, _vp, _vp, _vp, _u64p, _u64p, _f64p]),
    "secedo_filter_device": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32,
                                       C.c_uint64, C.c_double, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _u64p,
                                       _u64p, _f64p, _vp]),
    "secedo_pileup_read": (C.c_int, [C.c_char_p, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint64, C.c_int, C.c_int,
                                     C.POINTER(PileupInfo), _vp, _vp, _vp, _vp]),
    "secedo_pileup_last_error": (C.c_char_p, []),
    "secedo_em_refine_device": (C.c_int, [C.c_int, _vp, C.c_uint32, C.c_uint64, _vp, _vp, _vp, C.c_uint32, C.c_double,
                                          _vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), _vp]),
    "secedo_em_refine": (C.c_int, [C.c_int, _vp, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_double, _vp, C.c_uint32,
                                   C.c_uint32, C.POINTER(C.c_uint32)]),
    "secedo_laplacian_device": (C.c_int, [_vp, C.c_uint32, _vp, _vp]),
    "secedo_spectral_eigs_device": (C.c_int, [C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                              C.c_uint32, _vp, _vp, C.POINTER(SpectralInfo), _vp]),
    "secedo_spectral_eigs_rows_device": (C.c_int, [C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                   C.c_uint32, C.c_double, C.c_uint32, _vp, _vp,
                                                   C.POINTER(SpectralInfo), _vp, _vp, _vp]),
    "secedo_spectral_eigs": (C.c_int, [C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32,
                                       _vp, _vp, C.POINTER(SpectralInfo)]),
}

_lib = None


class SecedoError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("secedo_simmat error %d: %s" % (code, message))
        self.code = code


class InvalidNormalization(ValueError):
    """The reference throws std::logic_error here (similarity_matrix.cpp:264)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C secedo_amd/csrc` (there is no fallback implementation)" % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 and the library links
        # /opt/rocm's. Whichever is loaded first serves both, so load torch's first when torch is
        # there (bench.py and the tests use torch for device memory, streams and RCCL); a process
        # that initialised /opt/rocm's runtime and imports torch afterwards sees no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc == OK:
        return
    msg = lib().secedo_simmat_last_error().decode(errors="replace")
    if rc == E_INVALID_NORMALIZATION:
        raise InvalidNormalization(msg)
    raise SecedoError(rc, msg)


def ptr(a):
    """Address of a numpy array's buffer (or None)."""
    return None if a is None else a.ctypes.data
