"""ctypes bindings of the CPU oracle and of the compiled reference -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under secedo_amd/ does. ``oracle_compute`` runs oracle/simmat_oracle.c (the plain-C
restatement); ``ref_compute`` runs oracle/_ref/libsecedo_ref.so (the unmodified reference,
compiled from /root/reference by oracle/Makefile) when that file is present.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libsimmat_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsecedo_ref.so")

NORMALIZATIONS = ("ADD_MIN", "EXPONENTIATE", "SCALE_MAX_1")

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

_oracle = None
_ref = None


def build(target: str = "oracle") -> None:
    subprocess.run(["make", "-C", HERE, target], check=True, capture_output=True)


def _load_oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build("oracle")
        lib = C.CDLL(ORACLE_SO)
        lib.oracle_simmat_compute.restype = C.c_int
        lib.oracle_simmat_compute.argtypes = [
            _u32p, C.c_uint32, _u32p, _u64p, _u32p, _u32p, _u32p, C.c_uint32, C.c_uint32,
            C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_int, _f64p,
            C.c_void_p,
        ]
        for f in (lib.oracle_log_prob_same, lib.oracle_log_prob_diff):
            f.restype = C.c_double
            f.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32]
        lib.oracle_normalize.restype = C.c_int
        lib.oracle_normalize.argtypes = [C.c_int, _f64p, C.c_uint32]
        lib.oracle_set_exact_binomials.restype = None
        lib.oracle_set_exact_binomials.argtypes = [C.c_int]
        lib.oracle_set_direct_llr_sum.restype = None
        lib.oracle_set_direct_llr_sum.argtypes = [C.c_int]
        lib.oracle_is_significant.restype = C.c_int
        lib.oracle_is_significant.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS"),
                                              C.c_double, C.c_uint32]
        lib.oracle_significance_terms.restype = None
        lib.oracle_significance_terms.argtypes = [
            np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS"), C.c_double, C.c_uint32,
            C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.oracle_filter.restype = C.c_int
        lib.oracle_filter.argtypes = [_u32p, C.c_uint32, _u32p, _u64p, _u32p, _u32p, _u32p, C.c_uint32,
                                      C.c_double, C.c_uint32, _u32p, _u32p, _u64p, _u32p, _u32p,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        lib.oracle_em.restype = C.c_int
        lib.oracle_em.argtypes = [_u32p, C.c_uint32, _u64p, _u32p, _u32p, C.c_uint32, C.c_double, _f64p,
                                  C.c_uint32, C.c_uint32]
        lib.oracle_last_updates.restype = C.c_uint64
        lib.oracle_last_read_pairs.restype = C.c_uint64
        _oracle = lib
    return _oracle


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def _load_ref():
    global _ref
    if _ref is None:
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        lib = C.CDLL(REF_SO)
        lib.ref_simmat_compute.restype = C.c_int
        lib.ref_simmat_compute.argtypes = [
            _u32p, C.c_uint32, _u32p, _u64p, _u32p, _u32p, _u32p, C.c_uint32, C.c_uint32,
            C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_char_p, _f64p,
        ]
        lib.ref_read_pileup.restype = C.c_int
        lib.ref_read_pileup.argtypes = [
            C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64),
            C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
        ]
        lib.ref_is_significant.restype = C.c_int
        lib.ref_is_significant.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS"),
                                           C.c_double, C.c_uint32]
        lib.ref_filter.restype = C.c_int
        lib.ref_filter.argtypes = [_u32p, C.c_uint32, _u32p, _u64p, _u32p, _u32p, _u32p, C.c_uint32,
                                   C.c_double, C.c_uint32, C.c_uint32, _u32p, _u32p, _u64p, _u32p, _u32p,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        lib.ref_em.restype = C.c_int
        lib.ref_em.argtypes = [_u32p, C.c_uint32, _u32p, _u64p, _u32p, _u32p, _u32p, C.c_uint32, C.c_double,
                               _f64p, C.c_uint32]
        lib.ref_read_pileup_fetch.restype = None
        lib.ref_read_pileup_fetch.argtypes = [_u32p, _u64p, _u32p, _u32p]
        _ref = lib
    return _ref


def _g2p(group_id_to_pos, num_cells):
    if group_id_to_pos is None:
        group_id_to_pos = np.arange(num_cells, dtype=np.uint32)
    return np.ascontiguousarray(group_id_to_pos, dtype=np.uint32)


def oracle_compute(p, num_cells, max_fragment_length, group_id_to_pos=None, mutation_rate=0.01,
                   homozygous_rate=0.5, seq_error_rate=0.01, num_threads=8,
                   normalization="ADD_MIN", want_raw=False):
    """Normalised matrix (and optionally the pre-normalisation D) from the C restatement.

    ``p`` is any object with the five FlatPileup arrays. Raises ValueError for an unknown
    normalisation (the reference throws std::logic_error, similarity_matrix.cpp:264).
    """
    lib = _load_oracle()
    if normalization not in NORMALIZATIONS:
        raise ValueError("Invalid normalization: " + str(normalization))
    g2p = _g2p(group_id_to_pos, num_cells)
    out = np.zeros((num_cells, num_cells), dtype=np.float64)
    raw = np.zeros((num_cells, num_cells), dtype=np.float64) if want_raw else None
    rc = lib.oracle_simmat_compute(
        p.chr_locus_off, len(p.chr_locus_off) - 1, p.locus_pos, p.locus_entry_off, p.read_ids,
        p.id_base, g2p, len(g2p), num_cells, max_fragment_length, mutation_rate,
        homozygous_rate, seq_error_rate, num_threads, NORMALIZATIONS.index(normalization), out,
        raw.ctypes.data if want_raw else None)
    if rc == -3:
        raise ValueError("a read pair shares >= max_fragment_length loci: the reference indexes past its tables "
                         "there (similarity_matrix.cpp:314-317, :330), so there is no reference answer to restate")
    if rc != 0:
        raise RuntimeError("oracle_simmat_compute failed: %d" % rc)
    return (out, raw) if want_raw else out


def set_exact_binomials(on) -> None:
    """See simmat_oracle.c: False/0 the reference bit for bit (default); True/1 the reference's formula
    without the u64 binomial wrap (x_s + x_d > ~48); 2 exact only beyond x_s + x_d = 64 (what the MI355X
    path documents for read pairs sharing more than 64 loci)."""
    _load_oracle().oracle_set_exact_binomials(int(on))


def set_direct_llr_sum(on: bool) -> None:
    """See simmat_oracle.c: per-pair (logP_diff - logP_same) sums, i.e. without the reference's
    final-subtraction cancellation (for inputs with ~1e5+ pairs per cell pair)."""
    _load_oracle().oracle_set_direct_llr_sum(1 if on else 0)


def oracle_last_updates() -> int:
    return int(_load_oracle().oracle_last_updates())


def oracle_last_read_pairs() -> int:
    return int(_load_oracle().oracle_last_read_pairs())


def oracle_log_prob_same(x_s, x_d, eps, h, theta, table_size=1000) -> float:
    return float(_load_oracle().oracle_log_prob_same(x_s, x_d, eps, h, theta, table_size))


def oracle_log_prob_diff(x_s, x_d, eps, h, theta, table_size=1000) -> float:
    return float(_load_oracle().oracle_log_prob_diff(x_s, x_d, eps, h, theta, table_size))


def oracle_normalize(normalization, mat):
    m = np.ascontiguousarray(mat, dtype=np.float64).copy()
    if normalization not in NORMALIZATIONS:
        raise ValueError("Invalid normalization: " + str(normalization))
    rc = _load_oracle().oracle_normalize(NORMALIZATIONS.index(normalization), m, m.shape[0])
    if rc != 0:
        raise RuntimeError("oracle_normalize failed: %d" % rc)
    return m


def ref_compute(p, num_cells, max_fragment_length, group_id_to_pos=None, mutation_rate=0.01,
                homozygous_rate=0.5, seq_error_rate=0.01, num_threads=8,
                normalization="ADD_MIN"):
    """The unmodified reference's computeSimilarityMatrix on the same flat pileup."""
    lib = _load_ref()
    g2p = _g2p(group_id_to_pos, num_cells)
    out = np.zeros((num_cells, num_cells), dtype=np.float64)
    rc = lib.ref_simmat_compute(
        p.chr_locus_off, len(p.chr_locus_off) - 1, p.locus_pos, p.locus_entry_off, p.read_ids,
        p.id_base, g2p, len(g2p), num_cells, max_fragment_length, mutation_rate,
        homozygous_rate, seq_error_rate, num_threads, normalization.encode(), out)
    if rc == -2:
        raise ValueError("Invalid normalization: " + str(normalization))
    if rc != 0:
        raise RuntimeError("ref_simmat_compute failed: %d" % rc)
    return out


def ref_read_pileup(fname, merge_count=1, merge_file="", max_coverage=100):
    """Reference reader (util/pileup_reader.cpp) -> (locus_pos, locus_entry_off, read_ids,
    id_base, num_cells, max_len). ``fname`` must live in a writable scratch directory."""
    lib = _load_ref()
    nl, ne = C.c_uint64(), C.c_uint64()
    nc, ml = C.c_uint32(), C.c_uint32()
    rc = lib.ref_read_pileup(fname.encode(), merge_count, merge_file.encode(), max_coverage,
                             C.byref(nl), C.byref(ne), C.byref(nc), C.byref(ml))
    if rc != 0:
        raise RuntimeError("ref_read_pileup failed: %d" % rc)
    pos = np.zeros(nl.value, dtype=np.uint32)
    off = np.zeros(nl.value + 1, dtype=np.uint64)
    rid = np.zeros(ne.value, dtype=np.uint32)
    idb = np.zeros(ne.value, dtype=np.uint32)
    lib.ref_read_pileup_fetch(pos, off, rid, idb)
    return pos, off, rid, idb, nc.value, ml.value


NO_POS = 16383  # util/is_significant.hpp:11


def _counts(base_count):
    return np.ascontiguousarray(base_count, dtype=np.uint16)


def oracle_is_significant(base_count, theta, cell_proportion=4) -> bool:
    return bool(_load_oracle().oracle_is_significant(_counts(base_count), theta, cell_proportion))


def oracle_significance_terms(base_count, theta, cell_proportion=4):
    s, k = C.c_double(), C.c_double()
    _load_oracle().oracle_significance_terms(_counts(base_count), theta, cell_proportion, C.byref(s), C.byref(k))
    return s.value, k.value


def ref_is_significant(base_count, theta, cell_proportion=4) -> bool:
    return bool(_load_ref().ref_is_significant(_counts(base_count), theta, cell_proportion))


def _run_filter(fn, p, id_to_pos, theta, cell_proportion, extra=()):
    i2p = np.ascontiguousarray(id_to_pos, dtype=np.uint32)
    L, E = len(p.locus_pos), len(p.read_ids)
    o_chr = np.zeros(len(p.chr_locus_off), dtype=np.uint32)
    o_pos = np.zeros(max(L, 1), dtype=np.uint32)
    o_off = np.zeros(L + 1, dtype=np.uint64)
    o_rid = np.zeros(max(E, 1), dtype=np.uint32)
    o_idb = np.zeros(max(E, 1), dtype=np.uint32)
    nl, ne, cov = C.c_uint64(), C.c_uint64(), C.c_double()
    rc = fn(p.chr_locus_off, len(p.chr_locus_off) - 1, p.locus_pos, p.locus_entry_off, p.read_ids, p.id_base,
            i2p, len(i2p), theta, cell_proportion, *extra, o_chr, o_pos, o_off, o_rid, o_idb, C.byref(nl),
            C.byref(ne), C.byref(cov))
    if rc != 0:
        raise RuntimeError("filter failed: %d" % rc)
    return (o_chr, o_pos[:nl.value].copy(), o_off[:nl.value + 1].copy(), o_rid[:ne.value].copy(),
            o_idb[:ne.value].copy(), cov.value)


def oracle_filter(p, id_to_pos, theta, cell_proportion=4):
    """-> (chr_locus_off, locus_pos, locus_entry_off, read_ids, id_base, avg_coverage)"""
    return _run_filter(_load_oracle().oracle_filter, p, id_to_pos, theta, cell_proportion)


def ref_filter(p, id_to_pos, theta, cell_proportion=4, num_threads=2):
    return _run_filter(_load_ref().ref_filter, p, id_to_pos, theta, cell_proportion, extra=(num_threads,))


def oracle_em(p, id_to_pos, theta, prob_cluster_b, max_iterations=1000):
    """oracle/em_oracle.c on a FlatPileup -> (refined probabilities, iterations)."""
    i2p = np.ascontiguousarray(id_to_pos, dtype=np.uint32)
    prob = np.array(prob_cluster_b, dtype=np.float64)
    rc = _load_oracle().oracle_em(p.chr_locus_off, len(p.chr_locus_off) - 1, p.locus_entry_off, p.id_base, i2p,
                                  len(i2p), theta, prob, len(prob), max_iterations)
    if rc < 0:
        raise RuntimeError("oracle_em failed: %d" % rc)
    return prob, rc


def ref_em(p, id_to_pos, theta, prob_cluster_b):
    """The compiled reference's expectation_maximization (expectation_maximization.cpp:125-161)."""
    i2p = np.ascontiguousarray(id_to_pos, dtype=np.uint32)
    prob = np.array(prob_cluster_b, dtype=np.float64)
    rc = _load_ref().ref_em(p.chr_locus_off, len(p.chr_locus_off) - 1, p.locus_pos, p.locus_entry_off, p.read_ids,
                            p.id_base, i2p, len(i2p), theta, prob, len(prob))
    if rc != 0:
        raise RuntimeError("ref_em failed: %d" % rc)
    return prob
