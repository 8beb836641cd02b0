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


def test_plan_validation_errors_without_a_gpu():
    """validate_plan runs on the host before anything touches a device: malformed plans come back as
    HDK_HIP_ERR_INVALID_ARG / UNSUPPORTED with a message, never as a crash (no exception crosses the ABI)."""
    import copy
    import numpy as np
    from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
    from hdk_amd.plan import compile_query
    from hdk_amd.storage import ArrowStorage
    L = _lib.lib()
    L.hdk_hip_last_error.restype = C.c_char_p
    st = ArrowStorage()
    st.import_numpy("t", {"k": np.arange(100, dtype=np.int64) * 1_000_003, "v": np.arange(100, dtype=np.int64)})
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("k")], force_baseline=True, baseline_entry_count=1000,
                                     targets=[KeyRef(0), Agg("sum", ColRef("v"))]))
    q = C.c_int64(0)
    assert L.hdk_hip_baseline_table_quads(C.byref(cp.plan), 10, C.byref(q)) == A.OK and q.value == 10 * cp.plan.row_size_quad

    def broken(mutate):
        p = copy.deepcopy(cp.plan) if False else type(cp.plan).from_buffer_copy(cp.plan)
        mutate(p)
        rc = L.hdk_hip_baseline_table_quads(C.byref(p), 10, C.byref(q))
        return rc, (L.hdk_hip_last_error() or b"").decode()

    rc, msg = broken(lambda p: setattr(p, "abi_version", 1))
    assert rc == A.ERR_INVALID_ARG and "ABI" in msg
    rc, msg = broken(lambda p: setattr(p, "num_targets", 99))
    assert rc == A.ERR_INVALID_ARG and "num_targets" in msg
    rc, msg = broken(lambda p: setattr(p, "query_kind", 17))
    assert rc == A.ERR_INVALID_ARG
    rc, msg = broken(lambda p: setattr(p.targets[1], "slot_width", 3))
    assert rc == A.ERR_INVALID_ARG and "slot width" in msg
    rc, msg = broken(lambda p: setattr(p, "query_kind", A.Q_PERFECT_HASH))  # not a baseline plan any more
    assert rc == A.ERR_INVALID_ARG
    assert L.hdk_hip_baseline_table_quads(None, 10, C.byref(q)) == A.ERR_INVALID_ARG


def test_join_build_scratch_geometry_is_host_only():
    """hdk_hip_join_build_scratch_bytes (the partitioned one-to-one build, join_build_part.h) is plain host arithmetic: 0 for
    tables the build does not partition, 8 / 16 / 24 / 32 bytes per row and scatter level otherwise, a second level from
    256 slices on (32 768 slots a slice for the table alone, 4 096 with payload columns)."""
    from hdk_amd._lib import lib
    L = lib()
    f = L.hdk_hip_join_build_scratch_bytes
    assert f(0, 100, 0) == 0 and f(100, 0, 0) == 0 and f(10**6, 10**6, 4) == 0 and f(2**31, 10**6, 0) == 0
    one = f(4_000_000, 4_000_000, 0)         # 123 slices: one level, 8-byte tuples
    assert 4_000_000 * 8 <= one <= 4_000_000 * 8 * 1.25 + (123 * 8 * 4096 + 4096) * 8 + 2**20
    two = f(100_000_000, 100_000_000, 0)     # 3 052 slices: level-1 sub-slabs + 32 768-tuple slabs per slice
    assert two > 100_000_000 * 8 * 2 and two < 100_000_000 * 8 * 2.5
    fused = f(100_000_000, 100_000_000, 1)   # 16-byte tuples
    assert 1.9 * two < fused < 2.2 * two
    assert f(5_000_000, 50_000_000, 0) > f(5_000_000, 5_000_000, 0)  # a sparse range: more slices, hence more slabs
