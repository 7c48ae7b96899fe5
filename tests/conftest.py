import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build once per session if the tree is fresh (a few seconds; hipcc cross-compiles on CPU).
    lib = os.path.join(ROOT, "ehyb_spmv_gpu_amd", "libehyb.so")
    ora = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "ehyb_spmv_gpu_amd", "csrc"), "-j8", "-s"], check=True)
    if not os.path.exists(ora):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)


@pytest.fixture(scope="session")
def E():
    import ehyb_spmv_gpu_amd as E

    return E


@pytest.fixture(scope="session")
def O():
    from oracle import oracle as O

    return O


@pytest.fixture(scope="session")
def gpu(E):
    if E.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the GPU box (there is no CPU fallback)")
    return True
