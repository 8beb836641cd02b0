"""Loader for the product library hdk_amd/libhdk_hip.so (C ABI of include/hdk_hip.h).

The library is the ONLY compute path: there is no CPU or PyTorch fallback.  If it is missing (or
a GPU call fails) the caller gets a loud error, never a silent eager path.
"""
import ctypes as C
import os
import subprocess

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# HDK_HIP_LIB: load another build of the same ABI instead (A/B measurements of two builds; never a fallback)
LIB_PATH = os.environ.get("HDK_HIP_LIB") or os.path.join(_HERE, "libhdk_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class HdkHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hdk_hip status {code}: {msg}")
        self.code = code


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); every symbol include/hdk_hip.h declares
v, i32, u32, i64, sz, i8 = C.c_void_p, C.c_int32, C.c_uint32, C.c_int64, C.c_size_t, C.c_int8
SIGNATURES = {
    "hdk_hip_last_error": (C.c_char_p, []),
    "hdk_hip_version": (i32, []),
    "hdk_hip_reload_switches": (None, []),
    "hdk_hip_mgr_get_device_count": (i32, [C.POINTER(i32)]),
    "hdk_hip_mgr_set_context": (i32, [i32]),
    "hdk_hip_set_interrupt": (i32, [i32, i32]),
    "hdk_hip_mgr_measure_hbm": (i32, [i32, sz, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hdk_hip_mgr_allocate_device_mem": (i32, [sz, i32, C.POINTER(v)]),
    "hdk_hip_mgr_free_device_mem": (i32, [v]),
    "hdk_hip_mgr_allocate_pinned_host_mem": (i32, [sz, C.POINTER(v)]),
    "hdk_hip_mgr_free_pinned_host_mem": (i32, [v]),
    "hdk_hip_mgr_copy_host_to_device": (i32, [v, v, sz, i32]),
    "hdk_hip_mgr_copy_host_to_device_async": (i32, [v, v, sz, i32]),
    "hdk_hip_mgr_synchronize_stream": (i32, [i32]),
    "hdk_hip_mgr_copy_device_to_host": (i32, [v, v, sz, i32]),
    "hdk_hip_mgr_copy_device_to_device": (i32, [v, v, sz, i32, i32]),
    "hdk_hip_mgr_zero_device_mem": (i32, [v, sz, i32]),
    "hdk_hip_mgr_set_device_mem": (i32, [v, C.c_ubyte, sz, i32]),
    "hdk_hip_mgr_synchronize_devices": (i32, []),
    "hdk_hip_mgr_get_stream": (i32, [i32, C.POINTER(v)]),
    "hdk_hip_mgr_get_device_properties": (i32, [i32, C.POINTER(A.DeviceProperties)]),
    "hdk_hip_init_group_by_buffer": (i32, [v, v, u32, u32, u32, u32, i32, i8, sz, sz, i32, v]),
    "hdk_hip_init_columnar_group_by_buffer": (i32, [v, v, u32, u32, u32, v, i32, i32, i8, sz, sz, i32, v]),
    "hdk_hip_validate_plan": (i32, [C.POINTER(A.Plan), i32]),
    "hdk_hip_workspace_size": (i32, [C.POINTER(A.Plan), C.POINTER(A.KernelOptions), i32, C.POINTER(sz)]),
    "hdk_hip_launch": (i32, [C.POINTER(A.Plan), C.POINTER(v), C.POINTER(A.KernelOptions), i32, v, v, sz]),
    "hdk_hip_collect_scan_times": (i32, [i32, C.POINTER(C.c_float), i32, C.POINTER(i32)]),
    "hdk_hip_describe_launch": (i32, [C.POINTER(A.Plan), C.POINTER(A.KernelOptions), i32, C.c_char_p, sz]),
    "hdk_hip_reduce_buffers": (i32, [C.POINTER(A.Plan), v, u32, C.POINTER(v), C.POINTER(u32), i32, v, v,
                                     i32, v]),
    "hdk_hip_graph_begin_capture": (i32, [i32, v]),
    "hdk_hip_graph_end_capture": (i32, [i32, v, C.POINTER(v)]),
    "hdk_hip_graph_launch": (i32, [v, i32, v]),
    "hdk_hip_graph_destroy": (i32, [v]),
    "hdk_hip_baseline_table_quads": (i32, [C.POINTER(A.Plan), u32, C.POINTER(i64)]),
    "hdk_hip_partition_baseline_count": (i32, [C.POINTER(A.Plan), v, u32, v, i32, C.POINTER(u32), i32, v]),
    "hdk_hip_partition_baseline": (i32, [C.POINTER(A.Plan), v, u32, v, i32, C.POINTER(u32), C.POINTER(v), i32, v]),
    "hdk_hip_exchange_shape_for": (i32, [C.POINTER(A.Plan), C.POINTER(A.KernelOptions), i32, u32, i32,
                                         C.POINTER(A.ExchangeShape)]),
    "hdk_hip_scatter_to_owners": (i32, [C.POINTER(A.Plan), C.POINTER(v), C.POINTER(A.KernelOptions),
                                        C.POINTER(A.ExchangeShape), v, i32, v, v, sz]),
    "hdk_hip_aggregate_from_ranks": (i32, [C.POINTER(A.Plan), C.POINTER(v), C.POINTER(A.KernelOptions),
                                           C.POINTER(A.ExchangeShape), v, i32, v, v, sz]),
    "hdk_hip_init_baseline_hash_join_buff": (i32, [v, i64, sz, i32, i32, i32, i32, v]),
    "hdk_hip_fill_baseline_hash_join_buff": (i32, [v, i64, i32, i32, sz, i32, i32, v, C.POINTER(A.JoinColumn),
                                                   C.POINTER(A.JoinColumnTypeInfo), i32, v]),
    "hdk_hip_fill_one_to_many_baseline_hash_table": (i32, [v, v, i64, i32, sz, i32, C.POINTER(A.JoinColumn),
                                                           C.POINTER(A.JoinColumnTypeInfo), i32, v]),
    "hdk_hip_build_fused_join_table": (i32, [v, i64, C.POINTER(v), C.POINTER(i32), C.POINTER(i32), i32, v, i32, v]),
    "hdk_hip_join_build_scratch_bytes": (sz, [i64, i64, i32]),
    "hdk_hip_fill_hash_join_buff_fused": (i32, [v, i32, i32, v, A.JoinColumn, A.JoinColumnTypeInfo, i64, C.POINTER(v),
                                                C.POINTER(i32), C.POINTER(i32), i32, v, v, sz, i32, v]),
    "hdk_hip_init_hash_join_buff": (i32, [v, i64, i32, i32, v]),
    "hdk_hip_fill_hash_join_buff": (i32, [v, i32, i32, v, A.JoinColumn, A.JoinColumnTypeInfo, i32, v]),
    "hdk_hip_fill_hash_join_buff_bucketized": (i32, [v, i32, i32, v, A.JoinColumn, A.JoinColumnTypeInfo,
                                                     i64, i32, v]),
    "hdk_hip_fill_one_to_many_hash_table": (i32, [v, A.HashEntryInfo, i32, A.JoinColumn,
                                                  A.JoinColumnTypeInfo, i32, v]),
    "hdk_hip_fill_one_to_many_hash_table_bucketized": (i32, [v, A.HashEntryInfo, i32, A.JoinColumn,
                                                             A.JoinColumnTypeInfo, i32, v]),
}
# introspection helpers (not part of the drop-in surface)
EXTRA_SIGNATURES = {
    "hdk_hip_sizeof_plan": (sz, []),
    "hdk_hip_sizeof_expr": (sz, []),
    "hdk_hip_sizeof_target": (sz, []),
    "hdk_hip_sizeof_qual": (sz, []),
    "hdk_hip_sizeof_join": (sz, []),
    "hdk_hip_sizeof_device_properties": (sz, []),
    "hdk_hip_sizeof_kernel_options": (sz, []),
}


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    return _load()


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so (soname
    libamdhip64.so.7) that libtorch_hip.so asks for by the unversioned name: if libhdk_hip.so pulled in
    /opt/rocm's copy first, a later `import torch` would load a SECOND runtime that finds no devices.
    Loading the wheel's copy first makes both sides resolve to it (the dynamic loader matches our
    NEEDED libamdhip64.so.7 by soname).  Only relevant to Python processes that use torch.distributed
    next to this library; set HDK_HIP_SYSTEM_RUNTIME=1 to skip."""
    import importlib.util
    import sys
    if os.environ.get("HDK_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HdkHipError(A.ERR_RUNTIME,
                          f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C hdk_amd/csrc).  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in {**SIGNATURES, **EXTRA_SIGNATURES}.items():
        f = getattr(L, name)  # AttributeError here == the library does not export its header
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


_switch_env = None


def sync_switches():
    """The library reads its HDK_HIP_* switches once per process (csrc/switches.h).  Tests and A/B scripts change them
    between launches: when this process's HDK_HIP_* environment differs from what the library last read, ask for a re-read.
    Called where a launch is prepared, not per launch."""
    global _switch_env
    now = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("HDK_HIP_")))
    if now != _switch_env:
        lib().hdk_hip_reload_switches()
        _switch_env = now


def check(status: int):
    if status != A.OK:
        msg = lib().hdk_hip_last_error()
        raise HdkHipError(status, msg.decode() if msg else "")
    return status
