"""Host-side mirror of the reference's pileup reader on top of the C-ABI (secedo_pileup_read).

``read_pileup`` and ``get_grouping`` keep the names, argument order and return values of the reference
(reference: util/pileup_reader.hpp:33-54), but return the flat structure-of-arrays pileup directly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .pileup import FlatPileup


def get_grouping(merge_count: int = 1, merge_file: str = "", max_cell_count: int = 10_000) -> np.ndarray:
    """cell id -> group (reference: util/pileup_reader.cpp:273-291)."""
    if merge_file:
        if not os.path.exists(merge_file):
            raise FileNotFoundError("Cannot find merge file: " + merge_file)
        text = open(merge_file).read()
        return np.asarray([int(x) for x in text.split(",") if x.strip()], dtype=np.uint16)
    return (np.arange(max_cell_count, dtype=np.uint32) // merge_count).astype(np.uint16)


def read_pileup(fname: str, id_to_group: Sequence[int], progress=None, max_coverage: int = 100,
                positions: Optional[Sequence[int]] = None, compute_max_read_len: bool = True,
                write_bin: bool = False) -> Tuple[FlatPileup, int, int]:
    """-> (one-chromosome FlatPileup, num_cells, longest fragment); reference read_pileup
    (util/pileup_reader.cpp:259-270). ``progress`` is accepted for signature compatibility."""
    del progress
    i2g = np.ascontiguousarray(id_to_group, dtype=np.uint16)
    pos = np.ascontiguousarray(positions if positions is not None else [], dtype=np.uint32)
    info = _lib.PileupInfo()
    L = _lib.lib()
    args = (fname.encode(), _lib.ptr(i2g), len(i2g), max_coverage, _lib.ptr(pos) if len(pos) else None, len(pos),
            int(compute_max_read_len), int(write_bin), C.byref(info))
    rc = L.secedo_pileup_read(*args, None, None, None, None)
    if rc != 0:
        raise ValueError(L.secedo_pileup_last_error().decode(errors="replace"))
    lp = np.zeros(info.n_loci, dtype=np.uint32)
    off = np.zeros(info.n_loci + 1, dtype=np.uint64)
    rid = np.zeros(info.n_entries, dtype=np.uint32)
    idb = np.zeros(info.n_entries, dtype=np.uint16)
    rc = L.secedo_pileup_read(*args, _lib.ptr(lp), _lib.ptr(off), _lib.ptr(rid), _lib.ptr(idb))
    if rc != 0:
        raise ValueError(L.secedo_pileup_last_error().decode(errors="replace"))
    p = FlatPileup(np.asarray([0, info.n_loci], dtype=np.uint32), lp, off, rid, idb.astype(np.uint32))
    return p, int(info.num_cells), int(info.max_read_length)
