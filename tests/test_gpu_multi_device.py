"""The multi-GPU partition behind the C-ABI / the reference's C++ signature (VERDICT r03 row 8e'; north_star: "the
N x N output is block-partitioned across the 8 GPUs"; SURVEY.md 8b secedo_simmat_set_devices, section 5 SECEDO_GPUS).
secedo_simmat_compute deals tile ranges to N devices from ONE process (a host thread and two streams per device),
exchanges the int64 tiles with hipMemcpyPeerAsync and lets every device normalise and download its block of rows.
The test box has one GPU: the N "devices" are all device 0 -- every lane, stream, event, peer copy and row-block
download of the N-device path runs, only the links are missing. The matrix must equal the single-device one bit for
bit (integer accumulators), for every normalisation, ragged tile and row counts, more lanes than tiles, clustered
and sparse loci."""
import json
import os
import subprocess

import numpy as np
import pytest

import secedo_amd
from oracle import bindings as ob
from secedo_amd.synth import synth_pileup
from tests import golden_util as gu
from tests.pileup_gen import random_pileup

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _single_device_afterwards():
    yield
    secedo_amd.set_devices(None)


def test_n_lanes_reproduce_the_single_device_matrix_bitwise():
    cases = [
        (random_pileup(701, 400, 3, 500, 60, 1500, dup_frac=0.02), 400, 1000, 8),    # 7 blocks of 64 / 4 of 128
        (synth_pileup(300, 4000, 3, 300, 0.05, seed=5), 300, 1000, 4),                # clustered loci: accumulate_masks
        (random_pileup(702, 70, 2, 300, 25, 700, dup_frac=0.05, triple_frac=0.2), 70, 400, 2),  # reads longer than mfl
    ]
    for p, n, mfl, threads in cases:
        for norm in secedo_amd.NORMALIZATIONS:
            secedo_amd.set_devices(None)
            assert secedo_amd.get_devices() == [0]
            want = secedo_amd.compute_similarity_matrix(p, n, mfl, None, 0.01, 0.5, 0.01, threads, "", norm)
            for lanes in (2, 3, 8):
                secedo_amd.set_devices([0] * lanes)
                assert secedo_amd.get_devices() == [0] * lanes
                got = secedo_amd.compute_similarity_matrix(p, n, mfl, None, 0.01, 0.5, 0.01, threads, "", norm)
                assert np.array_equal(got, want, equal_nan=True), (n, norm, lanes)
    # ... and what the lanes computed is what the reference computes
    p, n, mfl, threads = cases[1]
    secedo_amd.set_devices([0, 0, 0])
    got = secedo_amd.compute_similarity_matrix(p, n, mfl, None, 0.01, 0.5, 0.01, threads, "", "ADD_MIN")
    ref = ob.oracle_compute(p, n, mfl, None, 0.01, 0.5, 0.01, threads, "ADD_MIN")
    assert gu.normwise_err(got, ref) <= 1e-9


def test_more_lanes_than_tiles_and_an_empty_pileup():
    p = random_pileup(703, 40, 1, 200, 12, 500)  # one 64-cell block: ONE tile, 16 lanes
    want = secedo_amd.compute_similarity_matrix(p, 40, 1000, None, 0.01, 0.5, 0.01, 4, "", "ADD_MIN")
    secedo_amd.set_devices([0] * 16)
    got = secedo_amd.compute_similarity_matrix(p, 40, 1000, None, 0.01, 0.5, 0.01, 4, "", "ADD_MIN")
    assert np.array_equal(got, want)
    from secedo_amd.pileup import FlatPileup
    empty = FlatPileup(np.zeros(2, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint32),
                       np.zeros(0, np.uint32))
    secedo_amd.set_devices(None)
    want = secedo_amd.compute_similarity_matrix(empty, 10, 1000, None, 0.01, 0.5, 0.01, 4, "", "EXPONENTIATE")
    secedo_amd.set_devices([0, 0])
    got = secedo_amd.compute_similarity_matrix(empty, 10, 1000, None, 0.01, 0.5, 0.01, 4, "", "EXPONENTIATE")
    assert np.array_equal(got, want)


def test_device_lists_are_checked():
    with pytest.raises(secedo_amd.SecedoError):
        secedo_amd.set_devices([0, 99])
    with pytest.raises(secedo_amd.SecedoError):
        secedo_amd.set_devices([0] * 17)
    assert secedo_amd.get_devices() == [0]  # a refused list changes nothing
    # an invalid normalisation is refused before any lane starts, as on one device
    secedo_amd.set_devices([0, 0])
    p = random_pileup(704, 20, 1, 50, 8, 300)
    with pytest.raises(secedo_amd.InvalidNormalization):
        secedo_amd.compute_similarity_matrix(p, 20, 1000, None, 0.01, 0.5, 0.01, 4, "", "NOPE")
    # an error inside the lanes (a group id outside group_id_to_pos) comes back as the single-device error does
    with pytest.raises(secedo_amd.SecedoError):
        secedo_amd.compute_similarity_matrix(p, 20, 1000, [0, 1], 0.01, 0.5, 0.01, 4, "", "ADD_MIN")
    # ... and leaves the library usable
    got = secedo_amd.compute_similarity_matrix(p, 20, 1000, None, 0.01, 0.5, 0.01, 4, "", "ADD_MIN")
    secedo_amd.set_devices(None)
    assert np.array_equal(got, secedo_amd.compute_similarity_matrix(p, 20, 1000, None, 0.01, 0.5, 0.01, 4, "", "ADD_MIN"))


def test_cpp_shim_with_the_reference_signature_on_two_and_four_lanes():
    """tests/cpp/shim_test.cpp --gpus ...: the C++ host with the reference's signature, unchanged, the devices from
    the environment's SECEDO_GPUS (what a deployment of the reference's binary sets): every bit of the matrix equal
    to the single-device call's (FNV-1a over the matrix, printed by the program)."""
    exe = os.path.join(ROOT, "secedo_amd", "csrc", "build", "shim_test")
    if not os.path.exists(exe):
        pytest.skip("shim_test not built")
    spec = ["--synth", "1000", "20000", "3", "3000", "0.05", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("SECEDO_GPUS", "SECEDO_DEVICE")}

    def run(extra):
        r = subprocess.run([exe] + extra + spec, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    single = run([])
    assert single["devices"] == 1
    for gpus, lanes in (("0,0", 2), ("0,0,0,0", 4)):
        multi = run(["--gpus", gpus])
        assert multi["devices"] == lanes
        assert multi["matrix_hash"] == single["matrix_hash"] and multi["checksum"] == single["checksum"]
    # a count beyond the visible devices is an error of the call, not a silent single-device run
    r = subprocess.run([exe, "--gpus", "9"] + spec, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "device" in r.stderr
