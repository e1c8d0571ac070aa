"""Host-side mirror of the reference's locus filter on top of the C-ABI.

``Filter(theta, cell_proportion).filter(pos_data, id_to_pos, marker, num_threads)`` keeps the names and
argument order of the reference (reference: util/is_significant.hpp:14-79); the counting, the
significance test and the compaction run on the GPU (secedo_amd/csrc/filter_device.hip).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib
from .pileup import FlatPileup
from .similarity_matrix import _as_flat, _id_arrays

NO_POS = 16383  # util/is_significant.hpp:11


class Filter:
    def __init__(self, theta: float, cell_proportion: int = 4):
        if not 0 <= cell_proportion <= 4:
            raise ValueError("cell_proportion must be in 0..4 (util/is_significant.hpp:30-40)")
        self.theta = float(theta)
        self.cell_proportion = int(cell_proportion)

    def is_significant(self, base_count) -> bool:
        """Filter::is_significant on counts of A, C, G, T (util/is_significant.cpp:78-138)."""
        c = np.ascontiguousarray(base_count, dtype=np.uint16)
        if c.shape != (4,):
            raise ValueError("base_count must hold four counts")
        rc = _lib.lib().secedo_is_significant(_lib.ptr(c), self.theta, self.cell_proportion)
        if rc < 0:
            _lib.check(rc)
        return bool(rc)

    def filter(self, pos_data, id_to_pos, marker: str = "", num_threads: int = 1) -> Tuple[FlatPileup, float]:
        """-> (filtered pileup, average coverage); Filter::filter (util/is_significant.cpp:149-193).
        ``marker`` only labels a log line and ``num_threads`` only sets the OpenMP width there."""
        del marker, num_threads
        p = _as_flat(pos_data)
        i2p = np.ascontiguousarray(id_to_pos, dtype=np.uint32)
        id16, id32 = _id_arrays(p)
        o_chr = np.zeros(p.n_chr + 1, dtype=np.uint32)
        o_pos = np.zeros(max(p.n_loci, 1), dtype=np.uint32)
        o_off = np.zeros(p.n_loci + 1, dtype=np.uint64)
        o_rid = np.zeros(max(p.n_entries, 1), dtype=np.uint32)
        o_idb = np.zeros(max(p.n_entries, 1), dtype=np.uint16 if id16 is not None else np.uint32)
        nl, ne, cov = C.c_uint64(), C.c_uint64(), C.c_double()
        _lib.check(_lib.lib().secedo_filter(
            _lib.ptr(p.chr_locus_off), p.n_chr, _lib.ptr(p.locus_pos), _lib.ptr(p.locus_entry_off),
            _lib.ptr(p.read_ids), _lib.ptr(id16), _lib.ptr(id32), _lib.ptr(i2p), len(i2p), self.theta,
            self.cell_proportion, _lib.ptr(o_chr), _lib.ptr(o_pos), _lib.ptr(o_off), _lib.ptr(o_rid),
            _lib.ptr(o_idb), C.byref(nl), C.byref(ne), C.byref(cov)))
        out = FlatPileup(o_chr, o_pos[:nl.value], o_off[:nl.value + 1], o_rid[:ne.value],
                         o_idb[:ne.value].astype(np.uint32))
        return out, float(cov.value)


def filter_resident(plan, res, id_to_pos, theta: float, cell_proportion: int = 4):
    """Device-resident variant: ``res`` from SimilarityMatrixPlan.upload -> (new resident pileup with
    ``g2p`` = the compacted rows of id_to_pos, average coverage). The filtered pileup stays in HBM and
    feeds plan.prepare_resident directly."""
    t = plan._torch
    dev = "cuda:%d" % plan.device
    i2p = np.ascontiguousarray(id_to_pos, dtype=np.uint32)
    d_i2p = t.from_numpy(i2p.view(np.int32)).to(dev)
    L, E = res["n_loci"], res["n_entries"]
    o_chr = t.empty(res["n_chr"] + 1, dtype=t.int32, device=dev)
    o_pos = t.empty(max(L, 1), dtype=t.int32, device=dev)
    o_off = t.empty(L + 1, dtype=t.int64, device=dev)
    o_rid = t.empty(max(E, 1), dtype=t.int32, device=dev)
    o_idb = t.empty(max(E, 1), dtype=res["idb"].dtype, device=dev)
    nl, ne, cov = C.c_uint64(), C.c_uint64(), C.c_double()
    idb = C.c_void_p(res["idb"].data_ptr())
    _lib.check(_lib.lib().secedo_filter_device(
        C.c_void_p(res["chr"].data_ptr()), res["n_chr"], C.c_void_p(res["pos"].data_ptr()),
        C.c_void_p(res["off"].data_ptr()), C.c_void_p(res["rid"].data_ptr()),
        idb if res["idb_is16"] else None, None if res["idb_is16"] else idb, C.c_void_p(d_i2p.data_ptr()),
        len(i2p), L, E, theta, cell_proportion, C.c_void_p(o_chr.data_ptr()), C.c_void_p(o_pos.data_ptr()),
        C.c_void_p(o_off.data_ptr()), C.c_void_p(o_rid.data_ptr()), C.c_void_p(o_idb.data_ptr()),
        C.byref(nl), C.byref(ne), C.byref(cov), plan._stream()))
    out = dict(chr=o_chr, pos=o_pos, off=o_off, rid=o_rid, idb=o_idb, idb_is16=res["idb_is16"], g2p=d_i2p,
               n_chr=res["n_chr"], n_loci=int(nl.value), n_entries=int(ne.value), n_groups=len(i2p))
    return out, float(cov.value)
