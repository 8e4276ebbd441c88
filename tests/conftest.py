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


# ---------------------------------------------------------------- gfx950 builds for the CPU-side kernel checks
KERNEL_FILES = ["flat_filter.hip", "flat_kernels.hip", "ivf_fused.hip", "ivf_kernels.hip", "ivf_listmajor.hip",
                "ivf_lm_filter.hip", "select_kernels.hip", "selector_kernels.hip"]


@pytest.fixture(scope="session")
def kernel_builds(tmp_path_factory):
    """{file: (path of the gfx950 assembly, hipcc's kernel-resource-usage remarks)} for every .hip file of the library: ONE
    `hipcc -S` per file (cross-compiles without a GPU), shared by the register-budget test and the ISA lint.  Results are
    cached under tests/.isa_cache/ by a hash of the sources so that a second pytest run here costs nothing."""
    import concurrent.futures
    import hashlib
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    if not os.path.exists(isa_lint.HIPCC):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "faiss_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    cache = os.path.join(ROOT, "tests", ".isa_cache")
    tag = os.path.join(cache, "tag")
    digest = h.hexdigest()
    if not (os.path.exists(tag) and open(tag).read() == digest):
        shutil.rmtree(cache, ignore_errors=True)
        os.makedirs(cache)

        def one(name):
            err = isa_lint.compile_asm(os.path.join(csrc, name), os.path.join(cache, name + ".s"),
                                       extra=["-Rpass-analysis=kernel-resource-usage"])
            with open(os.path.join(cache, name + ".remarks"), "w") as f:
                f.write(err)

        with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(one, KERNEL_FILES))
        with open(tag, "w") as f:
            f.write(digest)
    return {name: (os.path.join(cache, name + ".s"), open(os.path.join(cache, name + ".remarks")).read()) for name in KERNEL_FILES}
