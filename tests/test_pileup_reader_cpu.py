"""Pileup reader: text and binary pileup files straight into the flat layout,
against vectors produced by the reference reader on the reference's own tests/data files (the .bin
fixtures are what the reference's text reader wrote). Host code: CPU only."""
import os
import shutil

import numpy as np
import pytest

import secedo_amd
from oracle import bindings as ob
from tests import golden_util as gu

DATA = os.path.join(gu.GOLDEN, "data")
VEC = np.load(os.path.join(gu.GOLDEN, "reader_vectors.npz"))
COMBOS = sorted({k.rsplit("|", 1)[0] for k in VEC.files})


@pytest.mark.parametrize("key", COMBOS)
def test_reader_matches_reference_vectors(key, tmp_path):
    name, mc, mf, maxcov, kind = key.split("|")
    for f in os.listdir(DATA):
        shutil.copy(os.path.join(DATA, f), tmp_path)
    path = str(tmp_path / (name + ".pileup" + (".bin" if kind == "bin" else "")))
    i2g = secedo_amd.get_grouping(int(mc), str(tmp_path / mf) if mf else "")
    p, num_cells, max_len = secedo_amd.read_pileup(path, i2g, None, int(maxcov))
    assert np.array_equal(p.locus_pos, VEC[key + "|pos"])
    assert np.array_equal(p.locus_entry_off, VEC[key + "|off"])
    assert np.array_equal(p.read_ids, VEC[key + "|rid"])
    assert np.array_equal(p.id_base, VEC[key + "|idb"])
    assert [num_cells, max_len] == VEC[key + "|meta"].tolist()


def test_text_reader_writes_the_reference_binary(tmp_path):
    """write_bin reproduces, byte for byte, the .bin the reference's text reader writes
    (util/pileup_reader.cpp:107-116)."""
    for name in ("ten_rows", "six_cells", "three_rows", "one_row"):
        src = str(tmp_path / (name + ".pileup"))
        shutil.copy(os.path.join(DATA, name + ".pileup"), src)
        secedo_amd.read_pileup(src, secedo_amd.get_grouping(), write_bin=True)
        assert open(src + ".bin", "rb").read() == open(os.path.join(DATA, name + ".pileup.bin"), "rb").read()


def test_positions_and_defaults(tmp_path):
    src = str(tmp_path / "ten_rows.pileup.bin")
    shutil.copy(os.path.join(DATA, "ten_rows.pileup.bin"), src)
    i2g = secedo_amd.get_grouping()
    full, _, _ = secedo_amd.read_pileup(src, i2g)
    want = full.locus_pos[[1, 4, 7]]
    some, _, max_len = secedo_amd.read_pileup(src, i2g, positions=want, compute_max_read_len=False)
    assert np.array_equal(some.locus_pos, want) and max_len == 1000  # util/pileup_reader.cpp:256
    # positions absent from the file select nothing; the scan stops after the last listed position
    none, _, _ = secedo_amd.read_pileup(src, i2g, positions=[5, 7])
    assert none.n_loci == 0


def test_reader_errors(tmp_path):
    with pytest.raises(ValueError):
        secedo_amd.read_pileup(str(tmp_path / "missing.pileup"), secedo_amd.get_grouping())
    src = str(tmp_path / "ten_rows.pileup.bin")
    shutil.copy(os.path.join(DATA, "ten_rows.pileup.bin"), src)
    with pytest.raises(ValueError):  # cell ids go up to 2211: a 100-entry mapping is too small
        secedo_amd.read_pileup(src, secedo_amd.get_grouping(max_cell_count=100))
    assert secedo_amd.get_grouping(2, "", 6).tolist() == [0, 0, 1, 1, 2, 2]


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built")
def test_reader_equals_reference_reader_live(tmp_path):
    """A synthetic binary pileup written in the reference's record format, read by both readers."""
    from tests.pileup_gen import random_pileup
    p = random_pileup(501, 300, 1, 400, 30, 500)
    path = str(tmp_path / "synthetic.bin")
    with open(path, "wb") as f:
        for l in range(p.n_loci):
            b, e = int(p.locus_entry_off[l]), int(p.locus_entry_off[l + 1])
            f.write(np.uint32(p.locus_pos[l]).tobytes() + np.uint16(e - b).tobytes())
            f.write(p.read_ids[b:e].astype(np.uint32).tobytes() + p.id_base[b:e].astype(np.uint16).tobytes())
    for merge in (1, 3):
        got, nc, ml = secedo_amd.read_pileup(path, secedo_amd.get_grouping(merge), max_coverage=35)
        pos, off, rid, idb, rnc, rml = ob.ref_read_pileup(path, merge, "", 35)
        assert np.array_equal(got.locus_pos, pos) and np.array_equal(got.locus_entry_off, off)
        assert np.array_equal(got.read_ids, rid) and np.array_equal(got.id_base, idb) and (nc, ml) == (rnc, rml)
