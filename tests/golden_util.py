"""Loader for tests/golden/*.npz (written by oracle/gen_golden.py from the compiled reference)."""
import glob
import os

import numpy as np

from secedo_amd.pileup import FlatPileup

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NORMS = ("ADD_MIN", "EXPONENTIATE", "SCALE_MAX_1")


def fixture_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not os.path.basename(f).startswith(("kat_", "filter_", "reader_", "laplacian_", "em_", "c2_reference", "c3_reference", "wrap_beyond128", "ref_files_pipeline", "spectral_"))
                  and not os.path.basename(f).endswith("_reference_digest.npz"))


def filter_fixture_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "filter_*.npz"))
                  if not os.path.basename(f).startswith("filter_kat"))


def load_filter(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    p = FlatPileup(z["chr_locus_off"], z["locus_pos"], z["locus_entry_off"], z["read_ids"], z["id_base"])
    expect = (z["out_chr_locus_off"], z["out_locus_pos"], z["out_locus_entry_off"], z["out_read_ids"],
              z["out_id_base"], float(z["avg_coverage"]))
    return p, z["id_to_pos"], float(z["theta"]), int(z["cell_proportion"]), expect


def load(name):
    """-> (FlatPileup, [case dict with num_cells,mfl,eps,h,theta,T,norm,g2p,out])"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    p = FlatPileup(z["chr_locus_off"], z["locus_pos"], z["locus_entry_off"], z["read_ids"],
                   z["id_base"])
    cases = []
    for i, row in enumerate(z["params"]):
        cases.append(dict(num_cells=int(row[0]), mfl=int(row[1]), eps=float(row[2]), h=float(row[3]),
                          theta=float(row[4]), T=int(row[5]), norm=NORMS[int(row[6])],
                          g2p=z["g2p_%d" % i], out=z["out_%d" % i]))
    return p, cases


def normwise_err(got, ref):
    """max|got - ref| / max|ref| over the finite entries (SURVEY.md section 8d parity norm).

    Non-finite entries (the reference yields NaN for SCALE_MAX_1 of an all-zero matrix: 0 * 1/0)
    must sit at the same places with the same value; otherwise the error is infinite."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    fin = np.isfinite(ref)
    if not np.array_equal(fin, np.isfinite(got)):
        return float("inf")
    if not np.array_equal(np.isnan(ref), np.isnan(got)):
        return float("inf")
    if np.any(~fin & ~np.isnan(ref)) and not np.array_equal(got[~fin & ~np.isnan(ref)],
                                                            ref[~fin & ~np.isnan(ref)]):
        return float("inf")
    if not np.any(fin):
        return 0.0
    denom = float(np.max(np.abs(ref[fin])))
    diff = float(np.max(np.abs(got[fin] - ref[fin])))
    return diff if denom == 0.0 else diff / denom
