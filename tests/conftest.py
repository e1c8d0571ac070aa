import os
import sys

import pytest

# the HIP runtime aborts the process on a queue error without a word unless its error log is on
os.environ.setdefault("AMD_LOG_LEVEL", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The product library is a build artefact (git-ignored): if a checkout has not been built yet, build it
    once (hipcc cross-compiles without a GPU; __graft_entry__.build() does the same and more)."""
    lib = os.path.join(ROOT, "secedo_amd", "libsecedo_simmat.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "secedo_amd", "csrc"), "-j4"], check=True,
                       capture_output=True)
    yield
