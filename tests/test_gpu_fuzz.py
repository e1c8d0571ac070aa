"""A short run of tools/fuzz_parity.py (random shapes the fixed parity cases do not cover: several 128-cell
tiles of sparse loci, deep loci that refuse the count tile, clustered loci, shared matrix rows; fresh handles;
assign_finalize against accumulate + finalize) against the oracle. The long runs are the tool's own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_forty_random_configurations_match_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "40", "7"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "40 configurations, 0 failures" in r.stdout
