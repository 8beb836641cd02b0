import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_executor_factory():
    """Executor factory; fails loudly (no fallback) when the HIP library or a device is missing."""
    from hdk_amd.executor import Executor
    from hdk_amd.hip_mgr import HipMgr
    mgr = HipMgr()

    def make(storage, device_id=0):
        return Executor(storage, device_id, mgr)

    return make
