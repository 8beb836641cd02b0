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


@pytest.fixture(autouse=True)
def _library_sees_switch_changes(monkeypatch):
    """libhdk_hip.so reads its HDK_HIP_* switches once per process (hdk_amd/csrc/switches.h); tests change them between
    launches.  The executor re-syncs when it prepares a step; tests that call the C ABI directly after
    monkeypatch.setenv / delenv get the re-read here (and every test starts from the environment as it is)."""
    from hdk_amd import _lib

    def sync():
        if _lib._lib is not None:
            _lib.sync_switches()

    sync()
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        orig_set(name, value, prepend)
        if name.startswith("HDK_HIP_"):
            sync()

    def delenv(name, raising=True):
        orig_del(name, raising)
        if name.startswith("HDK_HIP_"):
            sync()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
