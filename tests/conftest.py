import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # PyTorch ships its own copy of the HIP runtime.  If libvp8hip.so (linked against /opt/rocm) is the
    # first to initialise HIP in a process, torch later reports "No HIP GPUs are available"; the other
    # order works (bench.py also initialises torch first).  Only one test hands torch tensors to the ABI.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle_stages():
    from oracle_lib import Oracle
    return Oracle.stages()


@pytest.fixture(scope="session")
def reference_stages():
    """The reference's own kernels compiled for x86 (oracle/_ref), or skip where it was not built."""
    from oracle_lib import ref_stages
    st = ref_stages()
    if st is None:
        pytest.skip("oracle/_ref/libvp8ref.so not built (needs /root/reference)")
    return st
