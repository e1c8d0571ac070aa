"""One pass with poisoned device memory (SECEDO_POISON=2: every new device allocation and, before each packing,
the packing's scratch and output arenas are filled with 0xA5 bytes), in a child process: results must not
depend on what an allocation happened to hold. Round 2's out-of-bounds read in k_fix_locus_rel (the word behind
blk_off, deep pileups, fresh handles only) is the kind of defect this catches; its test is part of the pass,
together with a 20-configuration differential fuzz against the oracle (tools/fuzz_parity.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_poisoned(args, timeout):
    env = dict(os.environ, SECEDO_POISON="2")
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    return r.stdout


def test_deep_loci_and_fuzz_with_poisoned_device_memory():
    out = run_poisoned(["-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        "tests/test_gpu_parity.py::test_very_deep_loci_unstaged_ranges",
                        "tests/test_gpu_parity.py::test_device_packing_equals_host_packing"], 900)
    assert "passed" in out
    out = run_poisoned(["tools/fuzz_parity.py", "20", "7"], 900)
    assert "20 configurations, 0 failures" in out
