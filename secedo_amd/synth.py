"""SYNTH-v1 synthetic pileups: bench / test utility. The generator lives in its own
host-only library (secedo_amd/libsecedo_synth.so, csrc/synth.cpp), not in the product library."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .pileup import FlatPileup

_synth = None


def _synth_lib():
    global _synth
    if _synth is None:
        path = os.path.join(_lib.HERE, "libsecedo_synth.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: build it with `make -C secedo_amd/csrc`" % path)
        l = C.CDLL(path)
        l.secedo_synth_generate.restype = C.c_int
        l.secedo_synth_generate.argtypes = [C.POINTER(_lib.SynthSpec), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _synth = l
    return _synth

# name -> (cells, loci, chromosomes, gap_max, new-fragment probability)
CONFIGS = {
    "C1": (64, 2000, 1, 2000, 0.30),
    "C2": (1000, 50_000, 22, 30_000, 0.05),
    "C3": (8000, 100_000, 22, 30_000, 0.03),
    "C5": (32000, 200_000, 22, 30_000, 0.01),
}


def synth_pileup(num_cells, num_loci, num_chromosomes=1, gap_max=30000, new_frag_prob=0.05,
                 frag_min=50, frag_max=600, base_error=0.01, mate_frac=0.01, seed=42) -> FlatPileup:
    spec = _lib.SynthSpec(num_cells, num_loci, num_chromosomes, gap_max, new_frag_prob, frag_min,
                          frag_max, base_error, mate_frac, seed)
    nl, ne = C.c_uint64(), C.c_uint64()
    L = _synth_lib()
    if L.secedo_synth_generate(C.byref(spec), C.byref(nl), C.byref(ne), None, None, None, None, None) != 0:
        raise ValueError("invalid synthetic pileup spec")
    chr_off = np.zeros(max(num_chromosomes, 1) + 1, dtype=np.uint32)
    pos = np.zeros(nl.value, dtype=np.uint32)
    off = np.zeros(nl.value + 1, dtype=np.uint64)
    rid = np.zeros(ne.value, dtype=np.uint32)
    idb = np.zeros(ne.value, dtype=np.uint32)
    if L.secedo_synth_generate(C.byref(spec), C.byref(nl), C.byref(ne), _lib.ptr(chr_off), _lib.ptr(pos),
                               _lib.ptr(off), _lib.ptr(rid), _lib.ptr(idb)) != 0:
        raise ValueError("invalid synthetic pileup spec")
    return FlatPileup(chr_off, pos, off, rid, idb)


def synth_config(name: str, clustered: bool = False, seed: int = 42) -> FlatPileup:
    """One of the SURVEY.md 8d configurations; clustered=True uses gap_max=300 (~2.8 loci/read)."""
    cells, loci, chrs, gap, p = CONFIGS[name]
    return synth_pileup(cells, loci, chrs, 300 if clustered else gap, p, seed=seed)
