import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The CPU oracle is compiled on demand (gcc, seconds); the HIP library and oracle/_ref are
    built by __graft_entry__.build() and travel to the GPU box as prebuilt .so files."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libfaiss_oracle.so")
    src = os.path.join(ROOT, "oracle", "faiss_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    # torch wheels bundle their own HIP runtime: when torch is to share the process with libfaiss_amd.so (the tests use
    # it for device pointers and streams), it has to initialise first so that both end up on one runtime -- initialising
    # it after the library has created and destroyed streams / pinned buffers fails with hipErrorNoDevice.  bench.py
    # imports in the same order.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield


@pytest.fixture(scope="session")
def res():
    import faiss_amd
    return faiss_amd.StandardGpuResources(0)
