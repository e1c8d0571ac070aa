"""Host-side pileup containers for the similarity-matrix path.

``PosData`` mirrors the reference's per-locus record (reference: sequenced_data.hpp:11-47):
one genomic position with the read id and the packed ``group_id << 2 | base`` of every read
that covers it. ``FlatPileup`` is the structure-of-arrays form the C-ABI takes
(include/secedo_simmat.h): the reference's ``vector<vector<PosData>>`` (one vector per
chromosome, similarity_matrix.hpp:51) flattened into five arrays.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

import numpy as np


@dataclass
class PosData:
    """All reads of all cells at one position (reference: sequenced_data.hpp:11-47)."""

    position: int
    read_ids: np.ndarray  # u32[coverage]
    group_ids_bases: np.ndarray  # u32[coverage], group_id << 2 | base

    def __post_init__(self):
        self.read_ids = np.ascontiguousarray(self.read_ids, dtype=np.uint32)
        self.group_ids_bases = np.ascontiguousarray(self.group_ids_bases, dtype=np.uint32)
        if self.read_ids.shape != self.group_ids_bases.shape:
            raise ValueError("read_ids and group_ids_bases differ in length")

    def group_id(self, i: int) -> int:
        return int(self.group_ids_bases[i]) >> 2

    def base(self, i: int) -> int:
        return int(self.group_ids_bases[i]) & 3

    def size(self) -> int:
        return int(self.read_ids.shape[0])


@dataclass
class FlatPileup:
    """Structure-of-arrays pileup: the layout of include/secedo_simmat.h.

    chr_locus_off[c]..chr_locus_off[c+1] are the loci of chromosome c;
    locus_entry_off[l]..locus_entry_off[l+1] are the entries of locus l.
    """

    chr_locus_off: np.ndarray  # u32[n_chr + 1]
    locus_pos: np.ndarray  # u32[L]
    locus_entry_off: np.ndarray  # u64[L + 1]
    read_ids: np.ndarray  # u32[E]
    id_base: np.ndarray  # u32[E]  (group_id << 2 | base)

    def __post_init__(self):
        self.chr_locus_off = np.ascontiguousarray(self.chr_locus_off, dtype=np.uint32)
        self.locus_pos = np.ascontiguousarray(self.locus_pos, dtype=np.uint32)
        self.locus_entry_off = np.ascontiguousarray(self.locus_entry_off, dtype=np.uint64)
        self.read_ids = np.ascontiguousarray(self.read_ids, dtype=np.uint32)
        self.id_base = np.ascontiguousarray(self.id_base, dtype=np.uint32)
        self.validate()

    @property
    def n_chr(self) -> int:
        return int(self.chr_locus_off.shape[0]) - 1

    @property
    def n_loci(self) -> int:
        return int(self.locus_pos.shape[0])

    @property
    def n_entries(self) -> int:
        return int(self.read_ids.shape[0])

    def validate(self) -> None:
        if self.chr_locus_off.ndim != 1 or self.chr_locus_off.shape[0] < 1:
            raise ValueError("chr_locus_off must hold n_chr + 1 offsets")
        if int(self.chr_locus_off[0]) != 0 or int(self.chr_locus_off[-1]) != self.n_loci:
            raise ValueError("chr_locus_off must start at 0 and end at the number of loci")
        if np.any(np.diff(self.chr_locus_off.astype(np.int64)) < 0):
            raise ValueError("chr_locus_off must be non-decreasing")
        if self.locus_entry_off.shape[0] != self.n_loci + 1:
            raise ValueError("locus_entry_off must hold n_loci + 1 offsets")
        if int(self.locus_entry_off[0]) != 0 or int(self.locus_entry_off[-1]) != self.n_entries:
            raise ValueError("locus_entry_off must start at 0 and end at the number of entries")
        if np.any(np.diff(self.locus_entry_off.astype(np.int64)) < 0):
            raise ValueError("locus_entry_off must be non-decreasing")
        if self.id_base.shape[0] != self.n_entries:
            raise ValueError("id_base and read_ids differ in length")

    def pair_locus_upper_bound(self) -> int:
        """Sum over loci of C(coverage, 2): an upper bound on the updates."""
        cov = np.diff(self.locus_entry_off.astype(np.int64))
        return int(np.sum(cov * (cov - 1) // 2))

    def to_pos_data(self) -> List[List[PosData]]:
        out: List[List[PosData]] = []
        for c in range(self.n_chr):
            chrom = []
            for l in range(int(self.chr_locus_off[c]), int(self.chr_locus_off[c + 1])):
                b, e = int(self.locus_entry_off[l]), int(self.locus_entry_off[l + 1])
                chrom.append(PosData(int(self.locus_pos[l]), self.read_ids[b:e], self.id_base[b:e]))
            out.append(chrom)
        return out


def flatten(pos_data: Sequence[Sequence[PosData]]) -> FlatPileup:
    """``vector<vector<PosData>>`` (one list per chromosome) -> FlatPileup."""
    chr_off = [0]
    pos: List[int] = []
    off = [0]
    rid: List[np.ndarray] = []
    idb: List[np.ndarray] = []
    for chrom in pos_data:
        for pd in chrom:
            pos.append(pd.position)
            rid.append(pd.read_ids)
            idb.append(pd.group_ids_bases)
            off.append(off[-1] + pd.size())
        chr_off.append(len(pos))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.uint32)
    return FlatPileup(
        np.asarray(chr_off, dtype=np.uint32),
        np.asarray(pos, dtype=np.uint32),
        np.asarray(off, dtype=np.uint64),
        cat(rid),
        cat(idb),
    )
