"""CPU-only checks of the C-ABI library and the host-side mirror (no compute calls: no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import secedo_amd
from oracle import bindings as ob
from secedo_amd import _lib
from secedo_amd.pileup import FlatPileup, PosData, flatten
from secedo_amd.synth import synth_pileup
from tests.pileup_gen import from_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    declared = set()
    for name in ("secedo_simmat.h", "secedo_spectral.h", "secedo_em.h"):
        header = open(os.path.join(ROOT, "include", name)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
        declared |= set(re.findall(r"\b(secedo_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    # and the Python binding table covers exactly the header
    assert declared == set(_lib.SIGNATURES)
    # the product library exports the declared interfaces and nothing else of its own (the synthetic
    # pileup generator of the bench lives in libsecedo_synth.so)
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (secedo_[a-z0-9_]+)$", nm, flags=re.M))
    assert exported == declared, exported ^ declared


def test_version_and_device_count():
    L = _lib.lib()
    assert b"gfx950" in L.secedo_simmat_version()
    assert L.secedo_simmat_device_count() >= 0


def test_normalization_strings():
    # reference: to_enum, similarity_matrix.cpp:256-266
    assert secedo_amd.to_enum("ADD_MIN") == 0
    assert secedo_amd.to_enum("EXPONENTIATE") == 1
    assert secedo_amd.to_enum("SCALE_MAX_1") == 2
    for bad in ("", "add_min", "ADD_MIN ", "SCALE"):
        with pytest.raises(secedo_amd.InvalidNormalization):
            secedo_amd.to_enum(bad)
    with pytest.raises(ValueError):  # InvalidNormalization is a ValueError, like the oracle's
        secedo_amd.compute_similarity_matrix(from_rows([[]]), 2, 1000, None, 0.01, 0.5, 0.01, 1, "", "nope")


@pytest.mark.skipif(_has_gpu(), reason="checks the loud failure without a GPU")
def test_no_silent_cpu_fallback():
    p = from_rows([[(10, [(1, 0, 0), (2, 1, 1)])]])
    with pytest.raises(secedo_amd.SecedoError) as e:
        secedo_amd.compute_similarity_matrix(p, 2, 1000, None, 0.01, 0.5, 0.01, 1, "", "ADD_MIN")
    assert e.value.code == _lib.E_NO_DEVICE
    h = C.c_void_p()
    assert _lib.lib().secedo_simmat_create(C.byref(h), 0) == _lib.E_NO_DEVICE


def test_multi_device_entry_points_without_a_gpu(monkeypatch):
    """secedo_simmat_set_devices / SECEDO_GPUS (the N-GPU partition behind secedo_simmat_compute): without a
    device a list is refused with SECEDO_E_NO_DEVICE, and so is the call when the environment asks for several
    devices -- never a silent single-device or CPU run; malformed lists are SECEDO_E_INVALID_ARG."""
    with pytest.raises(secedo_amd.SecedoError) as e:
        secedo_amd.set_devices([0, 0])
    assert e.value.code == _lib.E_NO_DEVICE
    secedo_amd.set_devices(None)  # back to the environment: always allowed
    assert secedo_amd.get_devices() == [0]
    p = from_rows([[(10, [(1, 0, 0), (2, 1, 1)])]])
    monkeypatch.setenv("SECEDO_GPUS", "2")
    with pytest.raises(secedo_amd.SecedoError) as e:
        secedo_amd.compute_similarity_matrix(p, 2, 1000, None, 0.01, 0.5, 0.01, 1, "", "ADD_MIN")
    assert e.value.code == _lib.E_NO_DEVICE
    for bad in ("0,x", "0,,1", "-1", "0", "17", "1,2,"):
        monkeypatch.setenv("SECEDO_GPUS", bad)
        with pytest.raises(secedo_amd.SecedoError) as e:
            secedo_amd.compute_similarity_matrix(p, 2, 1000, None, 0.01, 0.5, 0.01, 1, "", "ADD_MIN")
        assert e.value.code == (_lib.E_INVALID_ARG if bad != "0" else _lib.E_INVALID_ARG), bad
    monkeypatch.setenv("SECEDO_GPUS", "1")  # one device: the ordinary path (which then finds no device)
    with pytest.raises(secedo_amd.SecedoError) as e:
        secedo_amd.compute_similarity_matrix(p, 2, 1000, None, 0.01, 0.5, 0.01, 1, "", "ADD_MIN")
    assert e.value.code == _lib.E_NO_DEVICE


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "secedo_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), (dirpath, f)


LLR_PARAMS = [(0.01, 0.5, 0.01), (0.01, 0.5, 0.001), (0.01, 0.15, 0.001), (0.02, 0.3, 0.05), (0.001, 0.5, 0.01)]


@pytest.mark.parametrize("eps,h,theta", LLR_PARAMS)
def test_llr_closed_form_equals_reference_sums(eps, h, theta):
    """Closed form of the product (llr_table.cpp) vs the reference's nested binomial sums as the
    oracle restates them (similarity_matrix.cpp:117-170), over the whole fast-path table range that
    the reference can represent (x_s + x_d <= 44: beyond, its u64 binomial products wrap)."""
    worst = 0.0
    for xs in range(0, 31):
        for xd in range(0, 31 - xs if xs > 14 else 15):
            if xs + xd == 0 or xs + xd > 44:
                continue
            ref = ob.oracle_log_prob_diff(xs, xd, eps, h, theta) - ob.oracle_log_prob_same(xs, xd, eps, h, theta)
            got = secedo_amd.llr_closed_form(xs, xd, eps, h, theta)
            worst = max(worst, abs(got - ref) / max(1.0, abs(ref)))
    assert worst < 2e-12


def test_llr_closed_form_matches_exact_binomials_anywhere():
    ob.set_exact_binomials(True)
    try:
        for xs, xd in [(50, 3), (60, 4), (64, 64), (70, 10), (3, 64), (100, 40), (129, 0), (90, 90)]:
            ref = ob.oracle_log_prob_diff(xs, xd, 0.01, 0.5, 0.01) - ob.oracle_log_prob_same(xs, xd, 0.01, 0.5, 0.01)
            assert abs(secedo_amd.llr_closed_form(xs, xd, 0.01, 0.5, 0.01) - ref) < 1e-12 * max(1.0, abs(ref))
            # (what the matrix path adds there is NOT this formula but the reference's wrapped sums, at any number of
            # shared loci since round 4: test_llr_beyond_128_shared_loci_is_the_reference_s_too)
    finally:
        ob.set_exact_binomials(False)


@pytest.mark.parametrize("eps,h,theta", LLR_PARAMS[:3])
def test_llr_table_is_reference_identical_up_to_128_shared_loci(eps, h, theta):
    """The terms the device tables hold (x_s + x_d <= 128; 64 until round 3) are the reference's, wrap-around
    of its uint64 binomial products -- and, from row 68 on, of its uint64 Pascal triangle itself -- included
    (similarity_matrix.cpp:95-101, :125, :159): D(60,4) is 0.464 in the reference and 0.578 by its formula,
    D(100,0) is -1.3e-6 where the formula gives -0.50. Checked against the restated sums of the oracle (default
    mode = the reference bit for bit) on a grid that covers the wrapping region."""
    grid = [(xs, xd) for xs in range(0, 65, 4) for xd in range(0, 65 - xs, 5) if xs + xd] \
        + [(50, 3), (60, 4), (64, 0), (0, 64), (32, 32), (47, 1), (1, 63)] \
        + [(65, 0), (70, 10), (100, 0), (0, 100), (64, 64), (90, 38), (128, 0), (1, 127), (37, 80)]
    wrapped = 0
    for xs, xd in grid:
        ref = ob.oracle_log_prob_diff(xs, xd, eps, h, theta) - ob.oracle_log_prob_same(xs, xd, eps, h, theta)
        got = secedo_amd.llr(xs, xd, eps, h, theta)
        assert abs(got - ref) <= 1e-11 * max(1.0, abs(ref)), (xs, xd, got, ref)
        wrapped += abs(got - secedo_amd.llr_closed_form(xs, xd, eps, h, theta)) > 1e-9
    assert wrapped >= 5  # the grid does reach the region where the reference departs from its formula


def test_llr_beyond_128_shared_loci_is_the_reference_s_too():
    """VERDICT r03 missing #4: beyond the table the matrix path evaluates a read pair's (x_s, x_d) on the host as the
    reference would on first use -- tables as long as needed, the same wrapping uint64 products and Pascal triangle
    (llr_table.cpp: reference_llr_any) -- instead of the formula in exact arithmetic. Against the oracle's restated
    sums (default mode = the reference bit for bit), whose tables go to max_fragment_length like the reference's."""
    eps, h, theta = LLR_PARAMS[0]
    differs = 0
    for xs, xd in [(129, 0), (0, 129), (100, 40), (70, 75), (150, 3), (2, 160), (90, 90)]:
        ref = ob.oracle_log_prob_diff(xs, xd, eps, h, theta) - ob.oracle_log_prob_same(xs, xd, eps, h, theta)
        got = secedo_amd.llr(xs, xd, eps, h, theta)
        assert np.isfinite(ref) and abs(got - ref) <= 1e-10 * max(1.0, abs(ref)), (xs, xd, got, ref)
        assert secedo_amd.llr(xs, xd, eps, h, theta) == got  # cached per (rates, x_s, x_d)
        differs += abs(got - secedo_amd.llr_closed_form(xs, xd, eps, h, theta)) > 1e-3
    assert differs >= 3  # ... and that is nowhere near the formula there


def test_llr_wrap_kat_from_reference_matrices():
    """Known answers read off matrices of the compiled reference for read pairs sharing 48-64 loci
    (tests/golden/kat_llr_wrap.npz, oracle/gen_golden.py: wrap_cases)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "kat_llr_wrap.npz"))
    for pi, (eps, h, theta) in enumerate(z["params"]):
        for ci, (xs, xd) in enumerate(z["combos"]):
            d = secedo_amd.llr(int(xs), int(xd), eps, h, theta)
            # (the matrix entries are sums of two ~50-term logs minus each other: 1e-12 absolute)
            assert abs(d - z["table"][pi, ci]) <= 2e-12 * max(1.0, abs(d)), (xs, xd, d, z["table"][pi, ci])


def test_llr_kat_from_reference_matrices():
    z = np.load(os.path.join(ROOT, "tests", "golden", "kat_llr_table.npz"))
    for pi, (eps, h, theta) in enumerate(z["params"]):
        for ci, (xs, xd) in enumerate(z["combos"]):
            d = secedo_amd.llr(int(xs), int(xd), eps, h, theta)
            assert abs(d - z["table"][pi, ci]) <= 1e-12 * max(1.0, abs(d))


def test_synth_generator_is_deterministic_and_well_formed():
    a = synth_pileup(50, 400, 3, 300, 0.2, seed=5)
    b = synth_pileup(50, 400, 3, 300, 0.2, seed=5)
    c = synth_pileup(50, 400, 3, 300, 0.2, seed=6)
    for name in ("chr_locus_off", "locus_pos", "locus_entry_off", "read_ids", "id_base"):
        assert np.array_equal(getattr(a, name), getattr(b, name))
    assert not np.array_equal(a.read_ids, c.read_ids) or not np.array_equal(a.id_base, c.id_base)
    assert a.n_chr == 3 and a.n_loci == 400
    assert int((a.id_base >> 2).max()) < 50
    for ch in range(3):  # strictly increasing positions inside a chromosome
        pos = a.locus_pos[a.chr_locus_off[ch]:a.chr_locus_off[ch + 1]].astype(np.int64)
        assert np.all(np.diff(pos) > 0)
    # clustered loci => reads with several entries exist; a few duplicated (mate) entries too
    ids, counts = np.unique(a.read_ids[:int(a.locus_entry_off[a.chr_locus_off[1]])], return_counts=True)
    assert counts.max() > 1


def test_flatten_round_trip():
    rows = [[(5, [(1, 0, 0), (2, 1, 3)]), (9, [(2, 1, 2)])], [], [(7, [(4, 2, 1)])]]
    p = from_rows(rows)
    q = flatten(p.to_pos_data())
    for name in ("chr_locus_off", "locus_pos", "locus_entry_off", "read_ids", "id_base"):
        assert np.array_equal(getattr(p, name), getattr(q, name))
    pd = PosData(5, [1, 2], [0 << 2 | 0, 1 << 2 | 3])
    assert pd.group_id(1) == 1 and pd.base(1) == 3 and pd.size() == 2
    with pytest.raises(ValueError):
        FlatPileup([0, 2], [1, 2], [0, 1], [1], [0])  # offsets inconsistent with entries


def test_cpp_shim_compiles_and_keeps_the_error_contract(tmp_path):
    exe = str(tmp_path / "shim_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "secedo_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"), "-o", exe,
                    "-L" + os.path.join(ROOT, "secedo_amd"), "-lsecedo_simmat", "-lsecedo_synth",
                    "-Wl,-rpath," + os.path.join(ROOT, "secedo_amd")], check=True)
    empty = str(tmp_path / "empty.bin")
    open(empty, "wb").close()
    r = subprocess.run([exe, empty, "4", "1000", "1", "BOGUS", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 3 and "Invalid normalization: BOGUS" in r.stderr  # std::logic_error
