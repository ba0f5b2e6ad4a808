"""Runs last (file order): the GPU test process has ONE GPU runtime -- the one libvp8hip.so was built for -- and no PyTorch.
(pytest imports every test module when it collects: a module-level `import torch` anywhere under tests/ would load PyTorch's bundled HIP
runtime and RCCL into this process before the library, which then runs on them.)"""
import sys

import pytest

from vp8oclenc_amd import api

pytestmark = pytest.mark.gpu


def test_the_gpu_test_process_holds_one_runtime_and_no_pytorch():
    assert "torch" not in sys.modules, "a test module imports torch at module level"
    maps = open("/proc/self/maps").read()
    hip = sorted({line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line})
    rccl = sorted({line.split()[-1] for line in maps.splitlines() if "librccl" in line})
    assert len(hip) == 1 and "/torch/" not in hip[0], hip
    assert all("/torch/" not in p for p in rccl), rccl          # (loaded by the by-reference split's in-process tests: the ROCm one)
    assert api.load_library().vp8hip_runtime_version() // 10_000_000 == 7
