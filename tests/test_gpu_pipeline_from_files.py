"""SURVEY.md 8f-3 end to end on the GPU: a pileup FILE of the reference's own test-suite -> secedo_pileup_read
-> HBM -> secedo_filter_device -> secedo_simmat_* -> matrix, against vectors the compiled reference produced
for the same files with its reader, its Filter::filter and computeSimilarityMatrix
(tests/golden/ref_files_pipeline.npz, oracle/gen_golden.py: files_pipeline)."""
import os

import numpy as np
import pytest

import secedo_amd
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

DATA = os.path.join(gu.GOLDEN, "data")


@pytest.mark.parametrize("name", ["ten_rows", "six_cells"])
@pytest.mark.parametrize("ext", ["", ".bin"])
def test_file_to_matrix_matches_the_reference_pipeline(name, ext, tmp_path):
    z = np.load(os.path.join(gu.GOLDEN, "ref_files_pipeline.npz"))
    n, mfl = int(z[name + "__n_cells"]), int(z[name + "__mfl"])
    theta, cp = float(z["theta"]), int(z["cell_proportion"])
    # text format, and the reference's binary format (the .bin the reference reader wrote next to the file)
    src = os.path.join(DATA, name + ".pileup" + ext)
    path = str(tmp_path / (name + ".pileup" + ext))
    with open(src, "rb") as fi, open(path, "wb") as fo:
        fo.write(fi.read())
    p, num_cells, max_len = secedo_amd.read_pileup(path, secedo_amd.get_grouping(1), None, 100, None, True)
    assert max(max_len, 2) == mfl
    i2p = np.arange(n, dtype=np.uint32)
    with secedo_amd.SimilarityMatrixPlan(0) as plan:
        res = plan.upload(p, i2p, n)  # the file's pileup, now in HBM
        filtered, cov = secedo_amd.filter_resident(plan, res, i2p, theta, cp)
        assert filtered["n_loci"] == len(z[name + "__kept_pos"])
        assert filtered["n_entries"] == len(z[name + "__kept_rid"])
        assert cov == float(z[name + "__avg_coverage"])
        L, E = filtered["n_loci"], filtered["n_entries"]
        assert np.array_equal(filtered["pos"][:L].cpu().numpy().view(np.uint32), z[name + "__kept_pos"])
        assert np.array_equal(filtered["rid"][:E].cpu().numpy().view(np.uint32), z[name + "__kept_rid"])
        assert np.array_equal(filtered["idb"][:E].cpu().numpy().view(np.uint16).astype(np.uint32),
                              z[name + "__kept_idb"])
        plan.prepare_resident(filtered, n, mfl, 1)
        acc = plan.new_acc()
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        for norm in secedo_amd.NORMALIZATIONS:
            got = plan.finalize(acc, norm).cpu().numpy()
            assert gu.normwise_err(got, z[name + "__" + norm]) <= 1e-9, (name, norm)
            assert np.array_equal(got, got.T, equal_nan=True)
            assert np.all(np.diag(got) == 0)
