import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build
    the product library once, as __graft_entry__.build() does.  Nothing
    happens when it is there already (e.g. on the GPU box, where it arrives
    with the snapshot)."""
    import subprocess
    lib = os.path.join(ROOT, "vstree_amd", "libvstree_amd.so")
    if not os.path.exists(lib) and shutil.which("hipcc"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vstree_amd",
                                                          "csrc"), "-j8"],
                              stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def V():
    import vstree_amd
    return vstree_amd
