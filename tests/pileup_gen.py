"""Small random fragment pileups for the parity tests (numpy, pure-Python loops: small sizes).

Independent of the product's SYNTH-v1 generator (secedo_amd/csrc/synth.cpp): the tests also use
this one so that a bug in the product's generator cannot hide a bug in the product's kernels.
"""
from __future__ import annotations

import numpy as np

from secedo_amd.pileup import FlatPileup


def random_pileup(seed, n_cells, n_chr, loci_per_chr, cov, gap_max, frag_min=50, frag_max=600,
                  dup_frac=0.02, conflict_frac=0.5, skip_frac=0.0, err=0.02, n_groups=None,
                  shuffle_ids=True, triple_frac=0.0):
    """Fragments start at loci, cover the following loci inside their length.

    dup_frac      fraction of (read, locus) entries that get a second mate entry
    conflict_frac of those, fraction whose second base differs (similarity_matrix.cpp:387-395)
    triple_frac   fraction of duplicated entries that get a THIRD entry at the same locus
    skip_frac     probability that a fragment skips a locus inside its span (paired-end insert)
    n_groups      number of group ids appearing in the pileup (default n_cells)
    """
    rng = np.random.default_rng(seed)
    n_groups = n_groups or n_cells
    chr_off = [0]
    locus_pos, locus_off = [], [0]
    read_ids, id_base = [], []
    next_read = 0
    for c in range(n_chr):
        L = loci_per_chr if np.isscalar(loci_per_chr) else loci_per_chr[c]
        gaps = 1 + rng.integers(0, gap_max, size=L)
        pos = np.cumsum(gaps) + 1000
        clone_diff = rng.random(L) < 0.35
        ref_base = rng.integers(0, 4, size=L)
        # fragments: (start locus, end position, group, id)
        per_locus = [[] for _ in range(L)]
        n_new = rng.poisson(cov * 0.6, size=L) if gap_max > frag_max else rng.poisson(
            max(cov * gap_max / (2.0 * (frag_min + frag_max) / 2.0), 0.3), size=L)
        for l in range(L):
            for _ in range(int(n_new[l])):
                g = int(rng.integers(0, n_groups))
                length = int(rng.integers(frag_min, frag_max + 1))
                rid = next_read
                next_read += 1
                end = pos[l] + length
                k = l
                while k < L and pos[k] < end:
                    if k == l or rng.random() >= skip_frac:
                        per_locus[k].append((rid, g))
                    k += 1
        for l in range(L):
            entries = []
            for (rid, g) in per_locus[l]:
                b = int(ref_base[l])
                if clone_diff[l] and g >= n_groups // 2:
                    b = (b + 1) & 3
                if rng.random() < err:
                    b = int(rng.integers(0, 4))
                entries.append((rid, g, b))
                if rng.random() < dup_frac:
                    b2 = (b + int(rng.integers(1, 4))) & 3 if rng.random() < conflict_frac else b
                    entries.append((rid, g, b2))
                    if rng.random() < triple_frac:
                        entries.append((rid, g, int(rng.integers(0, 4))))
            for (rid, g, b) in entries:
                read_ids.append(rid)
                id_base.append((g << 2) | b)
            locus_pos.append(int(pos[l]))
            locus_off.append(len(read_ids))
        chr_off.append(len(locus_pos))
    read_ids = np.asarray(read_ids, dtype=np.uint32)
    if shuffle_ids and next_read:
        # arbitrary (non-consecutive) ids, still unique per fragment
        perm = rng.permutation(next_read).astype(np.uint32) * np.uint32(7) + np.uint32(13)
        read_ids = perm[read_ids]
    return FlatPileup(np.asarray(chr_off, dtype=np.uint32), np.asarray(locus_pos, dtype=np.uint32),
                      np.asarray(locus_off, dtype=np.uint64), read_ids,
                      np.asarray(id_base, dtype=np.uint32))


def from_rows(rows, n_chr_split=None):
    """rows: list of chromosomes, each a list of (position, [(read_id, group, base), ...])."""
    chr_off = [0]
    pos, off, rid, idb = [], [0], [], []
    for chrom in rows:
        for (p, ents) in chrom:
            pos.append(p)
            for (r, g, b) in ents:
                rid.append(r)
                idb.append((g << 2) | b)
            off.append(len(rid))
        chr_off.append(len(pos))
    return FlatPileup(np.asarray(chr_off, dtype=np.uint32), np.asarray(pos, dtype=np.uint32),
                      np.asarray(off, dtype=np.uint64), np.asarray(rid, dtype=np.uint32),
                      np.asarray(idb, dtype=np.uint32))
