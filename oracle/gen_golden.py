#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED, UNMODIFIED reference -- TEST INFRASTRUCTURE.

Run in the build container only (needs oracle/_ref/libsecedo_ref.so, i.e. /root/reference):

    make -C oracle ref && python oracle/gen_golden.py

Every fixture stores the flat input pileup, the call parameters and the reference's output
matrix, so the tests need neither the reference nor this script's random generators.
The pileup text files under tests/golden/data/ are data files of the reference's own test-suite
(reference: tests/data/{ten_rows,six_cells,three_rows,one_row}.pileup, six_cells.pileup.group);
they are parsed here by the reference's reader (util/pileup_reader.cpp) in a scratch directory.
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import bindings as ob  # noqa: E402
from secedo_amd.pileup import FlatPileup  # noqa: E402
from tests.pileup_gen import from_rows, random_pileup  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
NORMS = ob.NORMALIZATIONS


def save(name, p: FlatPileup, cases, **extra):
    """cases: list of dicts(num_cells, mfl, g2p, eps, h, theta, T, norm) -> adds 'out'."""
    arrays = dict(chr_locus_off=p.chr_locus_off, locus_pos=p.locus_pos,
                  locus_entry_off=p.locus_entry_off, read_ids=p.read_ids, id_base=p.id_base)
    params = []
    for i, c in enumerate(cases):
        out = ob.ref_compute(p, c["num_cells"], c["mfl"], c["g2p"], c["eps"], c["h"], c["theta"],
                             c["T"], c["norm"])
        arrays["out_%d" % i] = out
        arrays["g2p_%d" % i] = np.asarray(c["g2p"], dtype=np.uint32)
        params.append([c["num_cells"], c["mfl"], c["eps"], c["h"], c["theta"], c["T"],
                       NORMS.index(c["norm"])])
    arrays["params"] = np.asarray(params, dtype=np.float64)
    for k, v in extra.items():
        arrays[k] = np.asarray(v)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s loci=%6d entries=%7d cases=%2d  %7.1f KB" % (
        name, p.n_loci, p.n_entries, len(cases), os.path.getsize(path) / 1024))


def case(n, mfl, T, norm, g2p=None, eps=0.01, h=0.5, theta=0.01):
    return dict(num_cells=n, mfl=mfl, g2p=np.arange(n, dtype=np.uint32) if g2p is None else g2p,
                eps=eps, h=h, theta=theta, T=T, norm=norm)


def far_dummies(first_id, pos0, count, group=0, step=5000):
    """`count` loci far downstream, each with one read of cell `group` (same cell => no pair):
    they push the position past start + mfl so that earlier reads complete and get flushed."""
    return [(pos0 + step * (k + 1), [(first_id + k, group, 0)]) for k in range(count)]


def semantic_probes():
    # SURVEY.md section 0 facts, 4 cells. Reads 1..4 at one locus; cells 0,1 'A', cells 2,3 'C'.
    locus = (1000, [(1, 0, 0), (2, 1, 0), (3, 2, 1), (4, 3, 1)])
    # (1) single-locus chromosome: nothing is ever flushed -> zero comparisons (fact 2)
    save("probe_single_locus", from_rows([[locus]]),
         [case(4, 1000, 1, n) for n in NORMS])
    # (2) + far dummy loci of cell 0: reads complete; with T=1 four completed reads trigger a
    # flush at the next locus; with T=2 eight are needed and are not reached (fact 3)
    rows = [locus] + far_dummies(100, 1000, 3)
    save("probe_flush_T1_vs_T2", from_rows([rows]),
         [case(4, 1000, 1, n) for n in NORMS] + [case(4, 1000, 2, n) for n in NORMS])
    # (3) two chromosomes: live reads are dropped at the chromosome end; `completed` carries over
    rows2 = [(500, [(1, 0, 0), (2, 2, 1)]), (700, [(3, 1, 0), (4, 3, 1)])] + far_dummies(200, 700, 2)
    save("probe_two_chromosomes", from_rows([rows, rows2]),
         [case(4, 1000, 1, "ADD_MIN"), case(4, 1000, 2, "ADD_MIN"), case(4, 1000, 1, "EXPONENTIATE")])
    # (4) multi-locus pair: reads 1 (cell 0) and 2 (cell 1) share one matching and one
    # mismatching locus -> ONE joint (1,1) term, not (1,0)+(0,1) (fact 1)
    rows = [(1000, [(1, 0, 0), (2, 1, 0)]), (1100, [(1, 0, 2), (2, 1, 3)])] + far_dummies(100, 1100, 4, group=2)
    save("probe_multi_locus_pair", from_rows([rows]),
         [case(4, 1000, 1, n) for n in NORMS])
    # (5) paired-end duplicates (similarity_matrix.cpp:387-395): equal second base is ignored;
    # a conflicting second base removes the position; a third entry is then appended again
    rows = [(1000, [(1, 0, 0), (2, 1, 0), (1, 0, 0), (2, 1, 1), (3, 2, 0), (3, 2, 1), (3, 2, 2)]),
            (1050, [(1, 0, 1), (2, 1, 1), (3, 2, 1)])] + far_dummies(100, 1050, 4, group=3)
    save("probe_paired_end_rule", from_rows([rows]),
         [case(4, 1000, 1, n) for n in NORMS])
    # (6) read longer than max_fragment_length: flushed, then re-opened as a new read
    rows = [(1000, [(1, 0, 0), (2, 1, 0)]), (1200, [(1, 0, 1), (5, 2, 1)]),
            (1250, [(6, 2, 0)]), (1290, [(7, 2, 0)]), (1300, [(8, 2, 0)]),
            (1400, [(1, 0, 2), (2, 1, 3), (9, 3, 2)]), (1500, [(1, 0, 0), (9, 3, 1)])] \
        + far_dummies(100, 1500, 5, group=2, step=400)
    save("probe_split_long_read", from_rows([rows]),
         [case(4, 300, 1, "ADD_MIN"), case(4, 300, 2, "ADD_MIN"), case(4, 1000, 1, "ADD_MIN")])
    # (7) group_id_to_pos remap: 6 group ids onto 3 matrix rows (same row => skipped pair)
    rows = [(1000, [(1, 0, 0), (2, 1, 0), (3, 2, 1), (4, 3, 1), (5, 4, 2), (6, 5, 0)])] \
        + far_dummies(100, 1000, 4, group=0)
    g2p = np.asarray([0, 0, 1, 1, 2, 2], dtype=np.uint32)
    save("probe_group_remap", from_rows([rows]), [case(3, 1000, 1, n, g2p=g2p) for n in NORMS])


def kat_llr_table():
    """D(x_s, x_d) = logP_diff - logP_same for small (x_s, x_d): 3 cells, reads of cells 0 and 1
    share x_s matching and x_d mismatching loci, cell 2 stays empty, ADD_MIN:
    D = M[0][2] - M[0][1] (both entries carry the same additive constant)."""
    params = [(0.01, 0.5, 0.01), (0.01, 0.5, 0.001), (0.01, 0.15, 0.001), (0.01, 0.5, 0.05)]
    combos = [(1, 0), (0, 1), (2, 0), (1, 1), (0, 2), (3, 1), (10, 0), (5, 5), (0, 8), (12, 3),
              (20, 20), (30, 2)]
    table = np.zeros((len(params), len(combos)), dtype=np.float64)
    for pi, (eps, h, theta) in enumerate(params):
        for ci, (xs, xd) in enumerate(combos):
            rows = []
            for k in range(xs + xd):
                b1 = 0
                b2 = 0 if k < xs else 1
                rows.append((1000 + 3 * k, [(1, 0, b1), (2, 1, b2)]))
            rows += far_dummies(100, 1000 + 3 * (xs + xd), 4, group=0)
            p = from_rows([rows])
            m = ob.ref_compute(p, 3, 1000, None, eps, h, theta, 1, "ADD_MIN")
            table[pi, ci] = m[0, 2] - m[0, 1]
    path = os.path.join(GOLDEN, "kat_llr_table.npz")
    np.savez_compressed(path, params=np.asarray(params), combos=np.asarray(combos, dtype=np.uint32),
                        table=table)
    print("kat_llr_table: D(1,0)=%.15g D(0,1)=%.15g D(1,1)=%.15g (eps,h,theta)=(0.01,0.5,0.01)"
          % (table[0, 0], table[0, 1], table[0, 3]))


def reference_pileup_files():
    data = os.path.join(GOLDEN, "data")
    with tempfile.TemporaryDirectory() as tmp:
        for f in os.listdir(data):
            shutil.copy(os.path.join(data, f), tmp)

        def read(name, merge_count=1, merge_file=""):
            pos, off, rid, idb, ncell, mlen = ob.ref_read_pileup(
                os.path.join(tmp, name), merge_count, merge_file)
            return FlatPileup(np.asarray([0, len(pos)], dtype=np.uint32), pos, off, rid, idb), ncell, mlen

        # ten_rows: real chr22 data, reads span several loci, 2208 cell ids -> identity grouping
        p, _, mlen = read("ten_rows.pileup")
        n = int((p.id_base >> 2).max()) + 1
        save("ref_ten_rows", p,
             [case(n, 1000, T, norm) for T in (1, 2) for norm in NORMS]
             + [case(n, max(mlen, 2), 1, "ADD_MIN")], max_len=mlen)
        # six_cells with merge_count=2 (3 groups) and with the .group file (2 groups)
        p, _, _ = read("six_cells.pileup", merge_count=2)
        save("ref_six_cells_merge2", p, [case(3, 1000, 1, norm) for norm in NORMS]
             + [case(3, 50, 1, norm) for norm in NORMS])
        p, _, _ = read("six_cells.pileup", merge_file=os.path.join(tmp, "six_cells.pileup.group"))
        save("ref_six_cells_groupfile", p, [case(2, 1000, 1, norm) for norm in NORMS]
             + [case(2, 50, 1, norm) for norm in NORMS])
        p, _, _ = read("three_rows.pileup")
        n = int((p.id_base >> 2).max()) + 1
        save("ref_three_rows", p, [case(n, 1000, 1, norm) for norm in NORMS])


def files_pipeline():
    """SURVEY.md 8f-3 end to end, from the compiled reference: the reference's own tests/data pileup
    FILES through its reader (util/pileup_reader.cpp), its locus filter (Filter::filter, all cells inside
    the cluster) and computeSimilarityMatrix -- what divide_cluster does with a file
    (spectral_clustering.cpp:336-356). ten_rows keeps 4 of its 10 loci; six_cells keeps none (the empty
    pileup is a case of its own: every matrix entry is the constant of its normalisation)."""
    data = os.path.join(GOLDEN, "data")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in os.listdir(data):
            if not f.endswith(".bin"):
                shutil.copy(os.path.join(data, f), tmp)
        for name in ("ten_rows", "six_cells"):
            pos, off, rid, idb, ncell, mlen = ob.ref_read_pileup(os.path.join(tmp, name + ".pileup"), 1, "")
            p = FlatPileup(np.asarray([0, len(pos)], dtype=np.uint32), pos, off, rid, idb)
            n = int((p.id_base >> 2).max()) + 1
            i2p = np.arange(n, dtype=np.uint32)
            o_chr, o_pos, o_off, o_rid, o_idb, cov = ob.ref_filter(p, i2p, 0.01, 4)
            fp = FlatPileup(o_chr, o_pos, o_off, o_rid, o_idb)
            mfl = max(int(mlen), 2)
            for k, v in dict(n_cells=np.uint32(n), mfl=np.uint32(mfl), kept_pos=o_pos, kept_off=o_off,
                             kept_rid=o_rid, kept_idb=o_idb, avg_coverage=np.float64(cov)).items():
                out[name + "__" + k] = v
            for norm in NORMS:
                out[name + "__" + norm] = ob.ref_compute(fp, n, mfl, i2p, 0.01, 0.5, 0.01, 1, norm)
            print("files_pipeline %-10s cells %5d loci %3d -> %3d  mfl %d" % (name, n, p.n_loci, len(o_pos), mfl))
    np.savez_compressed(os.path.join(GOLDEN, "ref_files_pipeline.npz"), theta=np.float64(0.01),
                        cell_proportion=np.uint32(4), **out)


def divide_clusters_shaped():
    """Input shaped like the reference's only test that reaches the path
    (tests/test_spectral_clustering.cpp:210-259): 100 cells, 5000 consecutive positions, all
    read ids distinct, coverage per position uniform in 1..40, half of the positions split the
    two 50-cell clones, 5% base errors; parameters of :201-207 (mfl 500, eps .01, h .5,
    theta .05, T 4, ADD_MIN). The test's own generator uses libstdc++-specific distributions
    and asserts labels only, so the input here is drawn with numpy and stored."""
    rng = np.random.default_rng(20240607)
    n_cells, n_pos = 100, 5000
    pos, off, rid, idb = [], [0], [], []
    next_id = 0
    for x in range(n_pos):
        cov = rng.integers(1, 41)
        significant = rng.random() < 0.5
        for cell in range(n_cells):
            if rng.integers(1, 41) <= cov:
                base = (0 if cell < n_cells // 2 else 1) if significant else 2
                if rng.random() < 0.05:
                    base = int(rng.integers(0, 4))
                rid.append(next_id)
                next_id += 1
                idb.append((cell << 2) | base)
        pos.append(x)
        off.append(len(rid))
    p = FlatPileup(np.asarray([0, n_pos], dtype=np.uint32), np.asarray(pos, dtype=np.uint32),
                   np.asarray(off, dtype=np.uint64), np.asarray(rid, dtype=np.uint32),
                   np.asarray(idb, dtype=np.uint32))
    save("divide_clusters_shaped", p,
         [case(n_cells, 500, 4, "ADD_MIN", theta=0.05), case(n_cells, 500, 1, "ADD_MIN", theta=0.05)])


def random_cases():
    # config 1 (BASELINE.json configs[0]): 64 cells x 2000 loci, 1 chromosome, sparse loci
    p = random_pileup(101, 64, 1, 2000, 19, 2000)
    save("config1_64cells", p, [case(64, 1000, T, "ADD_MIN") for T in (1, 4, 8)]
         + [case(64, 1000, 8, "EXPONENTIATE"), case(64, 1000, 8, "SCALE_MAX_1")])
    # clustered loci (most read pairs share >= 2 loci), duplicates/conflicts/triples, inserts,
    # 2 chromosomes, reversed group map, flag-file rates
    p = random_pileup(102, 40, 2, 500, 10, 250, dup_frac=0.05, triple_frac=0.3, skip_frac=0.2)
    rev = np.arange(40, dtype=np.uint32)[::-1].copy()
    save("clustered_40cells", p,
         [case(40, 1000, T, norm, g2p=rev) for T in (1, 4) for norm in NORMS]
         + [case(40, 300, T, "ADD_MIN", g2p=rev) for T in (1, 4)]
         + [case(40, 1000, 8, "ADD_MIN", g2p=rev, h=0.15, theta=0.001)])
    # very dense overlaps: long reads over tightly packed loci, up to 46 loci per read (beyond one
    # 32-locus window) but x_s + x_d < 48, where the reference's u64 binomial products still hold
    # (oracle/simmat_oracle.c, oracle_set_exact_binomials)
    p = random_pileup(103, 12, 1, 400, 6, 12, frag_min=120, frag_max=260, dup_frac=0.01)
    save("dense_12cells", p, [case(12, 1000, 1, norm) for norm in NORMS]
         + [case(12, 1000, 3, "ADD_MIN", theta=0.001)])


def wrap_cases():
    """Read pairs sharing 48-64 loci: there the reference's uint64_t binomial PRODUCTS wrap
    (similarity_matrix.cpp:125, :159; single binomials up to row 64 still fit), and what it returns is
    the wrapped sum -- D(60,4) = 0.464 where its own formula gives 0.578. The product reproduces exactly
    that (llr_table.cpp: reference_llr), so these vectors come straight from the compiled reference.
    `wrap_beyond64` goes further (reads of 70-100 loci): the product follows the reference there too (up to 128
    shared loci since round 3). `wrap_beyond128` (reads of 140-170 loci): from 129 shared loci on the product
    returns the closed form instead; the fixture records what the reference does there so that the test can
    state the difference."""
    # (a) known answers D(x_s, x_d) up to x_s + x_d = 64, as in kat_llr_table
    params = [(0.01, 0.5, 0.01), (0.01, 0.15, 0.001)]
    combos = [(48, 0), (50, 3), (60, 4), (32, 32), (64, 0), (0, 64), (40, 24), (24, 40), (63, 1), (47, 1),
              (45, 10), (30, 30)]
    table = np.zeros((len(params), len(combos)), dtype=np.float64)
    for pi, (eps, h, theta) in enumerate(params):
        for ci, (xs, xd) in enumerate(combos):
            rows = [(1000 + 3 * k, [(1, 0, 0), (2, 1, 0 if k < xs else 1)]) for k in range(xs + xd)]
            rows += far_dummies(100, 1000 + 3 * (xs + xd), 4, group=0)
            m = ob.ref_compute(from_rows([rows]), 3, 1000, None, eps, h, theta, 1, "ADD_MIN")
            table[pi, ci] = m[0, 2] - m[0, 1]
    np.savez_compressed(os.path.join(GOLDEN, "kat_llr_wrap.npz"), params=np.asarray(params),
                        combos=np.asarray(combos, dtype=np.uint32), table=table)
    print("kat_llr_wrap: D(60,4)=%.15g D(64,0)=%.15g (eps,h,theta)=(0.01,0.5,0.01)" % (table[0, 2], table[0, 4]))
    # (b) random dense pileups: fragments of 200-250 bp over loci 4 bp apart on average
    p = random_pileup(104, 10, 1, 420, 5, 7, frag_min=200, frag_max=250, dup_frac=0.01)
    save("wrap_dense_10cells", p, [case(10, 1000, 1, norm) for norm in NORMS]
         + [case(10, 1000, 2, "ADD_MIN", h=0.15, theta=0.001)])
    p = random_pileup(105, 8, 1, 500, 4, 7, frag_min=300, frag_max=400, dup_frac=0.0)
    save("wrap_beyond64", p, [case(8, 1000, 1, "ADD_MIN")])
    # reads of 140-170 loci: beyond the 128 shared loci up to which the product restates the reference's sums
    p = random_pileup(106, 8, 1, 700, 4, 7, frag_min=560, frag_max=680, dup_frac=0.0)
    save("wrap_beyond128", p, [case(8, 1000, 1, "ADD_MIN")])


def c2_reference_run():
    """one 1000-cell run of the compiled reference (BASELINE config 2,
    SYNTH-v1 seed 42: 1000 cells x 50000 loci, T = 8, ADD_MIN), kept as a digest: sha256 of the 8 MB
    output, its maximum, its sum, and 1000 sampled entries (i, j, value). The input is regenerated by the
    product's deterministic generator (secedo_amd.synth), whose entry/locus counts are stored too."""
    import hashlib
    from secedo_amd.synth import CONFIGS, synth_config
    n_cells = CONFIGS["C2"][0]
    p = synth_config("C2")
    out = ob.ref_compute(p, n_cells, 1000, None, 0.01, 0.5, 0.01, 8, "ADD_MIN")
    rng = np.random.default_rng(2024)
    ii = rng.integers(0, n_cells, size=1000)
    jj = rng.integers(0, n_cells, size=1000)
    np.savez_compressed(os.path.join(GOLDEN, "c2_reference_digest.npz"),
                        sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest(), dtype=np.uint8),
                        max_abs=np.float64(np.max(np.abs(out))), total=np.float64(np.sum(out)),
                        sample_i=ii.astype(np.uint32), sample_j=jj.astype(np.uint32), sample_v=out[ii, jj],
                        n_entries=np.uint64(p.n_entries), n_loci=np.uint64(p.n_loci),
                        params=np.asarray([n_cells, 1000, 0.01, 0.5, 0.01, 8, 0], dtype=np.float64))
    print("c2_reference_digest: max %.6f sum %.6f" % (np.max(np.abs(out)), np.sum(out)))


def c3_reference_run():
    """The headline configuration (BASELINE configs[2], SYNTH-v1 seed 42: 8000 cells x 100000 loci, 22
    chromosomes, T = 8, ADD_MIN) through the compiled reference ONCE (about five minutes and 1.5 GB here;
    only run when named: `python oracle/gen_golden.py c3_reference_run`), kept as a digest: sha256 of the 512 MB
    output, its maximum, its sum, per-row-block sums, and 6000 sampled entries (i, j, value): 3000 uniform, 1000
    inside diagonal 128-cell tiles, 1000 with a cell of the last (partial: 8000 = 62.5 x 128) cell block, 1000
    from the rows/columns of the first cell block. The input is regenerated by the product's deterministic
    generator (secedo_amd.synth), whose entry/locus counts are stored too."""
    import hashlib
    import time
    from secedo_amd.synth import CONFIGS, synth_config
    n = CONFIGS["C3"][0]
    p = synth_config("C3")
    t0 = time.time()
    out = ob.ref_compute(p, n, 1000, None, 0.01, 0.5, 0.01, 8, "ADD_MIN")
    dt = time.time() - t0
    rng = np.random.default_rng(2025)
    ii = [rng.integers(0, n, size=3000)]
    jj = [rng.integers(0, n, size=3000)]
    blk = rng.integers(0, (n + 127) // 128, size=1000)          # same 128-cell block on both sides
    ii.append(np.minimum(blk * 128 + rng.integers(0, 128, size=1000), n - 1))
    jj.append(np.minimum(blk * 128 + rng.integers(0, 128, size=1000), n - 1))
    last0 = (n - 1) // 128 * 128                                  # a cell of the last block
    ii.append(rng.integers(last0, n, size=1000)); jj.append(rng.integers(0, n, size=1000))
    ii.append(rng.integers(0, n, size=1000)); jj.append(rng.integers(0, 128, size=1000))
    ii = np.concatenate(ii); jj = np.concatenate(jj)
    swap = rng.random(len(ii)) < 0.5
    ii, jj = np.where(swap, jj, ii), np.where(swap, ii, jj)
    row_block_sums = np.add.reduceat(out.sum(axis=1), np.arange(0, n, 128))
    np.savez_compressed(os.path.join(GOLDEN, "c3_reference_digest.npz"),
                        sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest(), dtype=np.uint8),
                        max_abs=np.float64(np.max(np.abs(out))), total=np.float64(np.sum(out)),
                        row_block_sums=row_block_sums,
                        sample_i=ii.astype(np.uint32), sample_j=jj.astype(np.uint32), sample_v=out[ii, jj],
                        n_entries=np.uint64(p.n_entries), n_loci=np.uint64(p.n_loci),
                        reference_seconds=np.float64(dt),
                        params=np.asarray([n, 1000, 0.01, 0.5, 0.01, 8, 0], dtype=np.float64))
    print("c3_reference_digest: max %.6f sum %.6f  (reference took %.0f s)" % (np.max(np.abs(out)), np.sum(out), dt))


def clustered_reference_runs(which=("C2", "C3")):
    """SURVEY.md section 8d "also run each with gap_max=300": the clustered-loci variants of configs 2 and 3
    (SYNTH-v1 seed 42, gap_max 300: about 2.8 loci per read, T = 8, ADD_MIN) through the compiled reference
    ONCE each (C2 clustered: 5.4e8 updates, about a minute; C3 clustered: 2.5e10 updates, tens of minutes -- only
    when named: `python oracle/gen_golden.py c2_clustered_reference_run` / `c3_clustered_reference_run`), kept
    as digests like c3_reference_run(): sha256, maximum, sum, 128-row block sums and sampled entries (uniform,
    inside diagonal 64- and 128-cell tiles, in the last cell block, in the first cell block)."""
    import hashlib
    import time
    from secedo_amd.synth import CONFIGS, synth_config
    for name in which:
        n = CONFIGS[name][0]
        p = synth_config(name, clustered=True)
        t0 = time.time()
        out = ob.ref_compute(p, n, 1000, None, 0.01, 0.5, 0.01, 8, "ADD_MIN")
        dt = time.time() - t0
        rng = np.random.default_rng(2026 + n)
        k = 1000 if n <= 1000 else 3000
        ii = [rng.integers(0, n, size=k)]
        jj = [rng.integers(0, n, size=k)]
        for edge in (64, 128):                                        # same cell block on both sides
            blk = rng.integers(0, (n + edge - 1) // edge, size=500)
            ii.append(np.minimum(blk * edge + rng.integers(0, edge, size=500), n - 1))
            jj.append(np.minimum(blk * edge + rng.integers(0, edge, size=500), n - 1))
        last0 = (n - 1) // 128 * 128                                  # a cell of the last block
        ii.append(rng.integers(last0, n, size=500)); jj.append(rng.integers(0, n, size=500))
        ii.append(rng.integers(0, n, size=500)); jj.append(rng.integers(0, 64, size=500))
        ii = np.concatenate(ii); jj = np.concatenate(jj)
        swap = rng.random(len(ii)) < 0.5
        ii, jj = np.where(swap, jj, ii), np.where(swap, ii, jj)
        row_block_sums = np.add.reduceat(out.sum(axis=1), np.arange(0, n, 128))
        np.savez_compressed(os.path.join(GOLDEN, "%s_clustered_reference_digest.npz" % name.lower()),
                            sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest(), dtype=np.uint8),
                            max_abs=np.float64(np.max(np.abs(out))), total=np.float64(np.sum(out)),
                            row_block_sums=row_block_sums,
                            sample_i=ii.astype(np.uint32), sample_j=jj.astype(np.uint32), sample_v=out[ii, jj],
                            n_entries=np.uint64(p.n_entries), n_loci=np.uint64(p.n_loci),
                            reference_seconds=np.float64(dt),
                            params=np.asarray([n, 1000, 0.01, 0.5, 0.01, 8, 0], dtype=np.float64))
        print("%s_clustered_reference_digest: max %.6f sum %.6f  (reference took %.0f s)" % (
            name.lower(), np.max(np.abs(out)), np.sum(out), dt), flush=True)


def filter_cases():
    """Locus filter (Filter::filter / is_significant) vectors from the compiled reference."""
    rng = np.random.default_rng(7)
    # (a) the known-answer strings of the reference's tests/test_is_significant.cpp:46-90 as base counts,
    #     plus random counts, all decided by the reference
    kat_counts = [[0, 51, 0, 1], [3, 44, 2, 3], [2, 0, 0, 57], [2, 39, 0, 0], [1, 0, 3, 0]]
    kat_theta = [0.01, 0.01, 0.01, 0.001, 0.001]
    counts, thetas, props, verdicts = [], [], [], []
    for c, th in zip(kat_counts, kat_theta):
        counts.append(c); thetas.append(th); props.append(4)
        verdicts.append(ob.ref_is_significant(c, th, 4))
    for t in range(3000):
        cov = int(rng.integers(2, 260))
        major = int(cov * rng.uniform(0.5, 1.0)); rest = cov - major
        a = int(rng.integers(0, rest + 1)); b = int(rng.integers(0, rest - a + 1))
        c = np.array([major, a, b, rest - a - b]); rng.shuffle(c)
        th = [0.01, 0.001, 0.05][t % 3]; cp = t % 5
        counts.append(c.tolist()); thetas.append(th); props.append(cp)
        verdicts.append(ob.ref_is_significant(c, th, cp))
    np.savez_compressed(os.path.join(GOLDEN, "filter_kat.npz"), counts=np.asarray(counts, dtype=np.uint16),
                        theta=np.asarray(thetas), cell_proportion=np.asarray(props, dtype=np.uint32),
                        significant=np.asarray(verdicts, dtype=np.uint8), n_reference_kats=len(kat_counts))
    print("filter_kat: %d decisions, %d significant" % (len(verdicts), int(np.sum(verdicts))))
    # (b) whole-pileup filtering with sub-cluster restriction
    for name, seed, n, nchr, L, cov, err, theta, cp, frac_out in [
            ("filter_60cells", 201, 60, 3, 300, 25, 0.15, 0.01, 4, 0.3),
            ("filter_200cells_all_in", 202, 200, 2, 250, 60, 0.10, 0.001, 2, 0.0),
            ("filter_40cells_deep", 203, 40, 1, 120, 150, 0.2, 0.05, 0, 0.5)]:
        p = random_pileup(seed, n, nchr, L, cov, 400, err=err)
        i2p = np.arange(n, dtype=np.uint32)
        drop = rng.random(n) < frac_out
        i2p[drop] = ob.NO_POS
        i2p[~drop] = np.arange(int((~drop).sum()), dtype=np.uint32)
        o_chr, o_pos, o_off, o_rid, o_idb, cov_avg = ob.ref_filter(p, i2p, theta, cp)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), chr_locus_off=p.chr_locus_off,
                            locus_pos=p.locus_pos, locus_entry_off=p.locus_entry_off, read_ids=p.read_ids,
                            id_base=p.id_base, id_to_pos=i2p, theta=theta, cell_proportion=cp,
                            out_chr_locus_off=o_chr, out_locus_pos=o_pos, out_locus_entry_off=o_off,
                            out_read_ids=o_rid, out_id_base=o_idb, avg_coverage=cov_avg)
        print("%-26s loci %5d -> %5d entries %6d -> %6d avg cov %.3f" % (
            name, p.n_loci, len(o_pos), p.n_entries, len(o_rid), cov_avg))


def reader_cases():
    """Pileup reader vectors: the reference reader (util/pileup_reader.cpp) on the reference's own
    tests/data/*.pileup files (text), and on the .bin files it writes next to them (binary; stored
    too, they are the binary-format fixtures)."""
    data = os.path.join(GOLDEN, "data")
    with tempfile.TemporaryDirectory() as tmp:
        for f in os.listdir(data):
            if not f.endswith(".bin"):
                shutil.copy(os.path.join(data, f), tmp)
        out = {}
        # (the text reader rewrites name.pileup.bin with the loci it kept: the max_coverage 60 case goes
        # first so that the stored .bin fixtures are the complete ones)
        combos = [("ten_rows", 1, "", 60), ("ten_rows", 1, "", 100), ("six_cells", 2, "", 100),
                  ("six_cells", 1, "six_cells.pileup.group", 100), ("three_rows", 1, "", 100),
                  ("one_row", 1, "", 100)]
        for name, mc, mf, maxcov in combos:
            mfp = os.path.join(tmp, mf) if mf else ""
            # the .bin the text reader just wrote holds only the loci it kept (renumbered read ids), so
            # the binary variant is recorded for the complete files only
            for ext in (("", ".bin") if maxcov == 100 else ("",)):
                path = os.path.join(tmp, name + ".pileup" + ext)
                pos, off, rid, idb, ncell, mlen = ob.ref_read_pileup(path, mc, mfp, maxcov)
                key = "%s|%d|%s|%d|%s" % (name, mc, mf, maxcov, "bin" if ext else "text")
                out[key + "|pos"] = pos
                out[key + "|off"] = off
                out[key + "|rid"] = rid
                out[key + "|idb"] = idb
                out[key + "|meta"] = np.asarray([ncell, mlen], dtype=np.uint32)
        for name in ("ten_rows", "six_cells", "three_rows", "one_row"):
            shutil.copy(os.path.join(tmp, name + ".pileup.bin"), os.path.join(data, name + ".pileup.bin"))
    np.savez_compressed(os.path.join(GOLDEN, "reader_vectors.npz"), **out)
    print("reader_vectors: %d arrays" % len(out))


def laplacian_kat():
    """The one known-answer vector the reference holds for the spectral step: the input matrix and the
    expected Laplacian of TEST(Laplacian, SomeMatrix) (tests/test_spectral_clustering.cpp:15-26; the
    test compares with 1e-3). Data of the reference's test, not an output of a run."""
    a = np.array([[0, .5, .2], [.5, 0, .5], [.2, .5, 0]], dtype=np.float64)
    expected = np.array([[1., -0.5976143, -0.28571429], [-0.5976143, 1., -0.5976143],
                         [-0.28571429, -0.5976143, 1.]], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, "laplacian_kat.npz"), a=a, expected=expected,
                        tolerance=np.float64(1e-3))
    print("laplacian_kat: 3 x 3")


SPECTRAL_GEN_CPP = r"""
// Input matrices of the reference's SpectralClustering.TwoClusters / ThreeClusters tests
// (tests/test_spectral_clustering.cpp:58-129 and :131-185), regenerated with the same standard-library
// generator the reference's tests run with (std::default_random_engine, default seed, and
// std::uniform_int_distribution of this libstdc++): background similarity 0..5, plus 20 four times out
// of five; inside a clone 100..200. Writes n, then n*n doubles, per matrix.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
static void emit(uint32_t n, uint32_t groups, FILE *f) {
    std::default_random_engine generator;
    std::uniform_int_distribution<uint32_t> dissimilar(0, 5);
    std::uniform_int_distribution<uint32_t> similar(100, 200);
    std::vector<double> m((size_t)n * n, 0.0);
    auto set = [&](uint32_t i, uint32_t j, double v) { m[(size_t)i * n + j] = v; m[(size_t)j * n + i] = v; };
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < i; ++j) {
            if (similar(generator) % 5) set(i, j, dissimilar(generator) + 20);
            else set(i, j, dissimilar(generator));
        }
    const uint32_t part = n / groups;
    for (uint32_t i = 0; i < part; ++i)
        for (uint32_t j = 0; j < i; ++j) {
            if (similar(generator) % 2) {
                set(i, j, similar(generator));
            } else {
                for (uint32_t g = 1; g < groups; ++g) set(i + g * part, j + g * part, similar(generator));
            }
        }
    double dn = n;
    fwrite(&dn, 8, 1, f);
    fwrite(m.data(), 8, m.size(), f);
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "wb");
    emit(100, 2, f);
    emit(99, 3, f);
    fclose(f);
    return 0;
}
"""


def spectral_cases():
    """The similarity matrices the reference's own spectral tests build (TwoClusters: 100 cells, two clones of
    50; ThreeClusters: 99 cells, three clones of 33) and the assignment those tests expect
    (tests/test_spectral_clustering.cpp:58-185). The matrices come out of libstdc++'s random distributions,
    so they are generated here, in the build container, with the same library, and stored."""
    import subprocess
    with tempfile.TemporaryDirectory() as tmp:
        src, exe, out = os.path.join(tmp, "g.cpp"), os.path.join(tmp, "g"), os.path.join(tmp, "m.bin")
        open(src, "w").write(SPECTRAL_GEN_CPP)
        subprocess.run(["g++", "-std=c++17", "-O1", src, "-o", exe], check=True)
        subprocess.run([exe, out], check=True)
        blob = np.fromfile(out, dtype=np.float64)
    n2 = int(blob[0])
    two = blob[1:1 + n2 * n2].reshape(n2, n2)
    rest = blob[1 + n2 * n2:]
    n3 = int(rest[0])
    three = rest[1:1 + n3 * n3].reshape(n3, n3)
    assert np.array_equal(two, two.T) and np.array_equal(three, three.T) and not np.any(np.diag(two))
    np.savez_compressed(os.path.join(GOLDEN, "spectral_reference_inputs.npz"), two_clusters=two,
                        two_expected=np.repeat([0, 1], 50).astype(np.int32), three_clusters=three,
                        three_expected=np.repeat([0, 1, 2], 33).astype(np.int32))
    print("spectral_reference_inputs: two clusters mean in/out %.1f / %.1f, three clusters %d x %d" % (
        two[:50, :50].mean(), two[:50, 50:].mean(), n3, n3))


def em_cases():
    """EM refinement (expectation_maximization.cpp): the five inputs of the reference's own
    tests/test_expectation_maximization.cpp:15-85 and three random pileups (one with a permuted
    id_to_pos: the centres are weighted by prob[group id], the sums land at id_to_pos[group id]),
    each with the compiled reference's output."""
    def three_loci(cells_bases):
        ents = [(1000 + i, c, b) for i, (c, b) in enumerate(cells_bases)]
        return from_rows([[(1234, ents), (1235, ents), (1236, ents)]])  # positions are not used by EM

    cases = {
        "one_cell": (from_rows([[]]), np.zeros(0, dtype=np.uint32), [1.0]),
        "two_cells_same": (three_loci([(0, 1), (1, 1)]), np.arange(2, dtype=np.uint32), [0.3, 0.4]),
        "two_cells_different": (three_loci([(0, 1), (1, 2)]), np.arange(2, dtype=np.uint32), [0.01, 0.02]),
        "four_cells_22": (three_loci([(0, 1), (1, 1), (2, 2), (3, 2)]), np.arange(4, dtype=np.uint32),
                          [0.9, 0.02, 0.03, 0.9]),
        "four_cells_31": (three_loci([(0, 2), (1, 1), (2, 2), (3, 2)]), np.arange(4, dtype=np.uint32),
                          [0.9, 0.9, 0.03, 0.1]),
    }
    rng = np.random.default_rng(77)
    for name, (n, nchr, L, cov, permute) in {"random_40": (40, 2, 300, 12, False),
                                             "random_200": (200, 3, 400, 30, False),
                                             "random_60_permuted": (60, 1, 500, 20, True)}.items():
        p = random_pileup(int(rng.integers(1 << 30)), n, nchr, L, cov, 400)
        # two clones: flip the base of the upper half of the cells at every third locus
        idb = p.id_base.copy()
        for l in range(0, len(p.locus_pos), 3):
            b, e = int(p.locus_entry_off[l]), int(p.locus_entry_off[l + 1])
            upper = (idb[b:e] >> 2) >= n // 2
            idb[b:e] = np.where(upper, (idb[b:e] & ~np.uint32(3)) | ((idb[b:e] + 1) & 3), idb[b:e])
        p = FlatPileup(p.chr_locus_off, p.locus_pos, p.locus_entry_off, p.read_ids, idb)
        i2p = rng.permutation(n).astype(np.uint32) if permute else np.arange(n, dtype=np.uint32)
        prob = np.clip(np.where(np.arange(n) >= n // 2, 0.7, 0.3) + 0.25 * rng.standard_normal(n), 0.01, 0.99)
        cases[name] = (p, i2p, prob)
    out = {}
    for name, (p, i2p, prob) in cases.items():
        ref = ob.ref_em(p, i2p, 1e-3, prob)
        mine, iters = ob.oracle_em(p, i2p, 1e-3, prob)
        assert np.max(np.abs(ref - mine)) <= 1e-12, (name, np.max(np.abs(ref - mine)))
        for k, v in dict(chr_locus_off=p.chr_locus_off, locus_pos=p.locus_pos, locus_entry_off=p.locus_entry_off,
                         read_ids=p.read_ids, id_base=p.id_base, id_to_pos=i2p,
                         prob_in=np.asarray(prob, dtype=np.float64), prob_out=ref,
                         iterations=np.int64(iters)).items():
            out[name + "__" + k] = v
        print("em %-22s cells %4d entries %6d iterations %d  max|ref - oracle| %.1e" % (
            name, len(prob), len(p.read_ids), iters, np.max(np.abs(ref - mine)) if len(prob) else 0))
    np.savez_compressed(os.path.join(GOLDEN, "em_cases.npz"), theta=np.float64(1e-3), **out)


def main():
    only = set(sys.argv[1:])
    if not ob.have_ref():
        sys.exit("oracle/_ref/libsecedo_ref.so missing: run `make -C oracle ref` in the container")
    os.makedirs(GOLDEN, exist_ok=True)
    for fn in (semantic_probes, kat_llr_table, reference_pileup_files, divide_clusters_shaped,
               random_cases, wrap_cases, c2_reference_run, files_pipeline, spectral_cases, filter_cases, reader_cases, laplacian_kat, em_cases):
        if not only or fn.__name__ in only:
            fn()
    if "c3_reference_run" in only:   # five minutes of the reference: only when asked for by name
        c3_reference_run()
    if "c2_clustered_reference_run" in only:
        clustered_reference_runs(("C2",))
    if "c3_clustered_reference_run" in only:   # tens of minutes of the reference
        clustered_reference_runs(("C3",))


if __name__ == "__main__":
    main()
