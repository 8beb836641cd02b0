"""The C-ABI library loads and exports every symbol include/hdk_hip.h declares; the ctypes mirror
agrees with the compiled struct sizes.  No compute calls (CPU only)."""
import ctypes as C
import os
import re

from hdk_amd import _abi as A
from hdk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "hdk_hip.h")).read()
    declared = set(re.findall(r"\b(hdk_hip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hdk_hip_kern_param"}
    assert len(declared) >= 30
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, f"not exported: {missing}"
    # and the ctypes binding table covers the header
    assert declared <= set(_lib.SIGNATURES), sorted(declared - set(_lib.SIGNATURES))


def test_struct_sizes_match():
    L = _lib.lib()
    assert C.sizeof(A.Plan) == L.hdk_hip_sizeof_plan()
    assert C.sizeof(A.Expr) == L.hdk_hip_sizeof_expr()
    assert C.sizeof(A.Target) == L.hdk_hip_sizeof_target()
    assert C.sizeof(A.Qual) == L.hdk_hip_sizeof_qual()
    assert C.sizeof(A.Join) == L.hdk_hip_sizeof_join()
    assert C.sizeof(A.DeviceProperties) == L.hdk_hip_sizeof_device_properties()
    assert C.sizeof(A.KernelOptions) == L.hdk_hip_sizeof_kernel_options()


def test_no_gpu_is_reported_not_hidden():
    """Without a device the manager says so (status + message); nothing falls back to the CPU."""
    L = _lib.lib()
    n = C.c_int32(-1)
    st = L.hdk_hip_mgr_get_device_count(C.byref(n))
    if n.value == 0:
        assert st != 0 and L.hdk_hip_last_error()
        p = C.c_void_p()
        assert L.hdk_hip_mgr_allocate_device_mem(16, 0, C.byref(p)) != 0


def test_version():
    assert _lib.lib().hdk_hip_version() >= 1000
