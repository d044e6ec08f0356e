import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session", autouse=True)
def _native_library_built():
    """The shared library is a build artefact (git-ignored): build it in-tree when it is missing, exactly like
    `__graft_entry__.build()` (hipcc cross-compiles gfx950 without a GPU).  The oracle's C helper builds itself."""
    so = os.path.join(ROOT, "crisperwhisper_amd", "libcrisperwhisper.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "crisperwhisper_amd", "csrc")])
    yield
