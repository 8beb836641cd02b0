"""The HDK_HIP_* switches are read once per process (hdk_amd/csrc/switches.h) and re-read on request."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd._lib import lib, sync_switches


def test_reload_is_callable_without_a_device(monkeypatch):
    L = lib()
    monkeypatch.setenv("HDK_HIP_NO_BH_LDS", "1")
    L.hdk_hip_reload_switches()  # (getenv only: no HIP call)
    monkeypatch.delenv("HDK_HIP_NO_BH_LDS")
    L.hdk_hip_reload_switches()
    sync_switches()


@pytest.mark.gpu
def test_a_switch_takes_effect_only_after_a_reload(gpu_executor_factory):
    """Changing the environment alone changes nothing (no getenv on the launch path); hdk_hip_reload_switches does."""
    import ctypes as C
    import os
    from hdk_amd._lib import check
    from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
    from hdk_amd.storage import ArrowStorage
    rng = np.random.default_rng(3)
    st = ArrowStorage()
    st.import_numpy("t", {"k": rng.integers(0, 50, 10_000).astype(np.int64) * 1_000_003, "v": rng.integers(0, 9, 10_000).astype(np.int64)})
    q = QueryUnit("t", groupby=[ColRef("k")], force_baseline=True, baseline_entry_count=131, targets=[KeyRef(0), Agg("sum", ColRef("v"))])
    ex = gpu_executor_factory(st)
    step = ex.prepare(q)
    L = lib()

    def names():  # (straight through the ABI: PreparedStep.kernel_names() would sync the switches itself)
        out = C.create_string_buffer(256)
        check(L.hdk_hip_describe_launch(C.byref(step.plan), C.byref(step.ko), step.dev, out, 256))
        return out.value.decode()

    assert names().startswith("hdk_scan_agg_bh_")
    os.environ["HDK_HIP_NO_BH_LDS"] = "1"
    try:
        assert names().startswith("hdk_scan_agg_bh_")  # not re-read
        L.hdk_hip_reload_switches()
        assert not names().startswith("hdk_scan_agg_bh_")
    finally:
        del os.environ["HDK_HIP_NO_BH_LDS"]
        L.hdk_hip_reload_switches()
    assert names().startswith("hdk_scan_agg_bh_")
    step.free()
