import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # ONE GPU runtime per test process: the tests hold their frames in memory the library allocates (api.to_device) and never
    # import PyTorch in-process on the GPU box; the few tests that need torch.distributed over RCCL run it in a child process.


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


def pytest_collection_modifyitems(config, items):
    """The tests that run bench.py as a CHILD process go first: a process keeps every hardware queue it ever used, and behind two
    hundred in-process GPU tests the parent holds enough of them that parent + child pass the 24 queues at which the part's
    scheduler starts context-switching running waves (seen in whole-suite runs only: the child's loop filter counted switched
    waves once, and once the child died in the runtime with `double free or corruption`; never in 100 stand-alone runs)."""
    first = [i for i in items if "test_bench_" in i.nodeid or "gather_frames_over_rccl" in i.nodeid]
    if first:
        rest = [i for i in items if i not in first]
        items[:] = first + rest
