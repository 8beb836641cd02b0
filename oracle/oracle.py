"""ctypes bindings of the CPU oracle (oracle/libhdk_oracle.so) and, when present, of the
reference's own compiled runtime (oracle/_ref/libhdk_ref_runtime.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by hdk_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from hdk_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# HDK_ORACLE_LIB: another build of the same checker (the sanitizer build, scripts/run_sanitizers.sh)
_LIB = os.environ.get("HDK_ORACLE_LIB") or os.path.join(_HERE, "libhdk_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libhdk_ref_runtime.so")

i8p = C.POINTER(C.c_int8)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def build(force=False):
    """(Re)build the oracle library (and _ref when /root/reference exists)."""
    # always go through make: it is a no-op when up to date and rebuilds when include/hdk_hip.h (the
    # plan POD the oracle's row function reads) changed
    subprocess.check_call(["make", "-C", _HERE, "--no-print-directory", os.path.basename(_LIB)],
                          stdout=subprocess.DEVNULL)


def _sig(lib, name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    v = C.c_void_p
    _sig(L, "orc_murmur_hash3", C.c_uint32, v, C.c_int, C.c_uint32)
    _sig(L, "orc_murmur_hash1", C.c_uint32, v, C.c_int, C.c_uint32)
    _sig(L, "orc_murmur_hash64a", C.c_uint64, v, C.c_int, C.c_uint64)
    _sig(L, "orc_key_hash", C.c_uint32, v, C.c_uint32, C.c_uint32)
    _sig(L, "orc_baseline_hash_join_idx_32", C.c_int64, v, v, C.c_size_t, C.c_size_t)
    _sig(L, "orc_baseline_hash_join_idx_64", C.c_int64, v, v, C.c_size_t, C.c_size_t)
    _sig(L, "orc_get_composite_key_index_32", C.c_int64, v, C.c_size_t, v, C.c_size_t)
    _sig(L, "orc_get_composite_key_index_64", C.c_int64, v, C.c_size_t, v, C.c_size_t)
    _sig(L, "orc_init_baseline_hash_join_buff", None, v, C.c_int64, C.c_size_t, C.c_int32, C.c_int32, C.c_int32)
    _sig(L, "orc_fill_baseline_hash_join_buff", C.c_int, v, C.c_int64, C.c_int32, C.c_size_t, C.c_int32, v, v)
    _sig(L, "orc_fill_baseline_hash_join_buff_semi", C.c_int, v, C.c_int64, C.c_int32, C.c_int32, C.c_size_t, C.c_int32, v, v)
    _sig(L, "orc_fill_one_to_many_baseline_hash_table", C.c_int, v, C.c_int64, C.c_int32, C.c_size_t, C.c_int32, v, v)
    _sig(L, "orc_fixed_width_int_decode", C.c_int64, v, C.c_int32, C.c_int64)
    _sig(L, "orc_fixed_width_unsigned_decode", C.c_int64, v, C.c_int32, C.c_int64)
    _sig(L, "orc_get_group_value", v, v, C.c_uint32, v, C.c_uint32, C.c_uint32, C.c_uint32)
    _sig(L, "orc_get_group_value_columnar_slot", C.c_int32, v, C.c_uint32, v, C.c_uint32, C.c_uint32)
    _sig(L, "orc_get_group_value_columnar", v, v, C.c_uint32, v, C.c_uint32)
    _sig(L, "orc_get_group_value_fast", v, v, C.c_int64, C.c_int64, C.c_int64, C.c_uint32)
    _sig(L, "orc_get_group_value_fast_keyless", v, v, C.c_int64, C.c_int64, C.c_int64, C.c_uint32)
    _sig(L, "orc_get_columnar_group_bin_offset", C.c_uint32, v, C.c_int64, C.c_int64, C.c_int64)
    for n in ("sum", "max", "min"):
        _sig(L, f"orc_agg_{n}", C.c_int64 if n == "sum" else None, v, C.c_int64)
        _sig(L, f"orc_agg_{n}_skip_val", C.c_int64 if n == "sum" else None, v, C.c_int64, C.c_int64)
        _sig(L, f"orc_agg_{n}_int32", C.c_int32 if n == "sum" else None, v, C.c_int32)
        _sig(L, f"orc_agg_{n}_int32_skip_val", C.c_int32 if n == "sum" else None, v, C.c_int32, C.c_int32)
        _sig(L, f"orc_agg_{n}_double", None, v, C.c_double)
        _sig(L, f"orc_agg_{n}_double_skip_val", None, v, C.c_double, C.c_double)
    _sig(L, "orc_agg_count", C.c_uint64, v, C.c_int64)
    _sig(L, "orc_agg_count_skip_val", C.c_uint64, v, C.c_int64, C.c_int64)
    _sig(L, "orc_checked_single_agg_id", C.c_int32, v, C.c_int64, C.c_int64)
    _sig(L, "orc_checked_single_agg_id_int32", C.c_int32, v, C.c_int32, C.c_int32)
    _sig(L, "orc_checked_single_agg_id_double", C.c_int32, v, C.c_double, C.c_double)
    _sig(L, "orc_checked_single_agg_id_float", C.c_int32, v, C.c_float, C.c_float)
    _sig(L, "orc_agg_count_int32", C.c_uint32, v, C.c_int32)
    _sig(L, "orc_agg_count_int32_skip_val", C.c_uint32, v, C.c_int32, C.c_int32)
    _sig(L, "orc_agg_count_double_skip_val", C.c_uint64, v, C.c_double, C.c_double)
    _sig(L, "orc_agg_sum_float", None, v, C.c_float)
    _sig(L, "orc_agg_sum_float_skip_val", None, v, C.c_float, C.c_float)
    _sig(L, "orc_scale_decimal_down_nullable", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_scale_decimal_down_not_nullable", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_floor_div_lhs", C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_floor_div_nullable_lhs", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_extract_year", C.c_int64, C.c_int64)
    _sig(L, "orc_logical_and", C.c_int8, C.c_int8, C.c_int8, C.c_int8)
    _sig(L, "orc_logical_or", C.c_int8, C.c_int8, C.c_int8, C.c_int8)
    _sig(L, "orc_logical_not", C.c_int8, C.c_int8, C.c_int8)
    _sig(L, "orc_hash_join_idx", C.c_int64, v, C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_bucketized_hash_join_idx", C.c_int64, v, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_hash_join_idx_nullable", C.c_int64, v, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_hash_join_idx_bitwise", C.c_int64, v, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
         C.c_int64)
    _sig(L, "orc_bucketized_hash_join_idx_nullable", C.c_int64, v, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(L, "orc_bucketized_hash_join_idx_bitwise", C.c_int64, v, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
         C.c_int64, C.c_int64)
    _sig(L, "orc_fixed_width_small_date_decode", C.c_int64, v, C.c_int32, C.c_int32, C.c_int64, C.c_int64)
    _sig(L, "orc_allowed_cpu_count", C.c_int32)
    _sig(L, "orc_host_stream_read_gbps", C.c_double, C.c_int32, C.c_size_t, C.c_int32)
    _sig(L, "orc_host_stream_read_gbps_placed", C.c_double, C.c_int32, C.c_size_t, C.c_int32, C.POINTER(C.c_double), C.c_int32)
    _sig(L, "orc_physical_core_count", C.c_int32)
    _sig(L, "orc_init_hash_join_buff", None, v, C.c_int64, C.c_int32)
    _sig(L, "orc_fill_hash_join_buff", C.c_int, v, C.c_int32, C.c_int32, v, C.c_size_t,
         C.POINTER(A.JoinColumnTypeInfo), C.c_int64)
    _sig(L, "orc_fill_one_to_many_hash_table", None, v, C.c_int64, C.c_int32, v, C.c_size_t,
         C.POINTER(A.JoinColumnTypeInfo), C.c_int64)
    _sig(L, "orc_init_group_by_buffer", None, v, v, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
         C.c_int32, C.c_int8)
    _sig(L, "orc_init_columnar_group_by_buffer", None, v, v, C.c_uint32, C.c_uint32, C.c_uint32, v,
         C.c_int32, C.c_int32, C.c_int8)
    _sig(L, "orc_run_plan", C.c_int32, C.POINTER(A.Plan), v, C.c_uint64, v, C.c_uint32, v, v)
    _sig(L, "orc_run_plan_range", C.c_int32, C.POINTER(A.Plan), v, C.c_uint64, C.c_uint64, v,
         C.c_uint32, v, v)
    _sig(L, "orc_run_projection", C.c_int32, C.POINTER(A.Plan), v, C.c_uint64, v, C.c_uint32, v, v, C.c_int32, v)
    _sig(L, "orc_reduce", C.c_int32, C.POINTER(A.Plan), v, C.c_uint32, v, C.c_uint32, v)
    _sig(L, "orc_is_empty_entry", C.c_int32, C.POINTER(A.Plan), v, C.c_uint32, C.c_uint32, v)
    _sig(L, "orc_run_plan_parallel", C.c_int32, C.POINTER(A.Plan), v, C.c_uint64, v, C.c_uint32, v,
         v, C.c_size_t, v, C.c_int32, v)
    _sig(L, "orc_c2_jit_shaped", C.c_double, v, v, v, C.c_uint64, C.c_int64, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
         C.c_int32, C.c_int32, C.c_int64, v, C.c_int32, C.c_int32, C.c_int32, v)
    _sig(L, "orc_max_threads", C.c_int32)
    _sig(L, "orc_sizeof_plan", C.c_size_t)
    if L.orc_sizeof_plan() != C.sizeof(A.Plan):
        raise RuntimeError("oracle/libhdk_oracle.so was built against a different include/hdk_hip.h "
                           f"(plan is {L.orc_sizeof_plan()} B there, {C.sizeof(A.Plan)} B here): make -C oracle")
    _lib = L
    return L


_ref = None


def ref():
    """The reference's own runtime (only where oracle/_ref was built); None otherwise."""
    global _ref
    if _ref is not None:
        return _ref
    if not os.path.exists(_REF):
        return None
    R = C.CDLL(_REF)
    v = C.c_void_p
    _sig(R, "MurmurHash3", C.c_uint32, v, C.c_int, C.c_uint32)
    _sig(R, "MurmurHash1", C.c_uint32, v, C.c_int, C.c_uint32)
    _sig(R, "MurmurHash64A", C.c_uint64, v, C.c_int, C.c_uint64)
    _sig(R, "key_hash", C.c_uint32, v, C.c_uint32, C.c_uint32)
    _sig(R, "fixed_width_int_decode", C.c_int64, v, C.c_int32, C.c_int64)
    _sig(R, "fixed_width_unsigned_decode", C.c_int64, v, C.c_int32, C.c_int64)
    _sig(R, "get_group_value", v, v, C.c_uint32, v, C.c_uint32, C.c_uint32, C.c_uint32)
    _sig(R, "get_group_value_columnar_slot", C.c_int32, v, C.c_uint32, v, C.c_uint32, C.c_uint32)
    _sig(R, "get_group_value_columnar", v, v, C.c_uint32, v, C.c_uint32)
    _sig(R, "get_group_value_fast", v, v, C.c_int64, C.c_int64, C.c_int64, C.c_uint32)
    _sig(R, "get_group_value_fast_keyless", v, v, C.c_int64, C.c_int64, C.c_int64, C.c_uint32)
    _sig(R, "get_columnar_group_bin_offset", C.c_uint32, v, C.c_int64, C.c_int64, C.c_int64)
    for n in ("sum", "max", "min"):
        _sig(R, f"agg_{n}", C.c_int64 if n == "sum" else None, v, C.c_int64)
        _sig(R, f"agg_{n}_skip_val", C.c_int64 if n == "sum" else None, v, C.c_int64, C.c_int64)
        _sig(R, f"agg_{n}_int32", C.c_int32 if n == "sum" else None, v, C.c_int32)
        _sig(R, f"agg_{n}_int32_skip_val", C.c_int32 if n == "sum" else None, v, C.c_int32, C.c_int32)
        _sig(R, f"agg_{n}_double", None, v, C.c_double)
        _sig(R, f"agg_{n}_double_skip_val", None, v, C.c_double, C.c_double)
    _sig(R, "agg_count", C.c_uint64, v, C.c_int64)
    _sig(R, "agg_count_skip_val", C.c_uint64, v, C.c_int64, C.c_int64)
    _sig(R, "checked_single_agg_id", C.c_int32, v, C.c_int64, C.c_int64)
    _sig(R, "checked_single_agg_id_int32", C.c_int32, v, C.c_int32, C.c_int32)
    _sig(R, "checked_single_agg_id_double", C.c_int32, v, C.c_double, C.c_double)
    _sig(R, "checked_single_agg_id_float", C.c_int32, v, C.c_float, C.c_float)
    _sig(R, "agg_count_int32", C.c_uint32, v, C.c_int32)
    _sig(R, "agg_count_int32_skip_val", C.c_uint32, v, C.c_int32, C.c_int32)
    _sig(R, "agg_count_double_skip_val", C.c_uint64, v, C.c_double, C.c_double)
    _sig(R, "agg_sum_float", None, v, C.c_float)
    _sig(R, "agg_sum_float_skip_val", None, v, C.c_float, C.c_float)
    _sig(R, "scale_decimal_down_nullable", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "scale_decimal_down_not_nullable", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "floor_div_lhs", C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "floor_div_nullable_lhs", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "extract_year", C.c_int64, C.c_int64)
    _sig(R, "logical_and", C.c_int8, C.c_int8, C.c_int8, C.c_int8)
    _sig(R, "logical_or", C.c_int8, C.c_int8, C.c_int8, C.c_int8)
    _sig(R, "logical_not", C.c_int8, C.c_int8, C.c_int8)
    for op in ("add", "sub", "mul", "div", "mod"):
        _sig(R, f"{op}_int64_t_nullable", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "add_double_nullable", C.c_double, C.c_double, C.c_double, C.c_double)
    _sig(R, "mul_double_nullable", C.c_double, C.c_double, C.c_double, C.c_double)
    for op in ("eq", "lt", "ge"):
        _sig(R, f"{op}_int64_t_nullable", C.c_int8, C.c_int64, C.c_int64, C.c_int64, C.c_int8)
    _sig(R, "gt_double_nullable", C.c_int8, C.c_double, C.c_double, C.c_double, C.c_int8)
    _sig(R, "cast_int64_t_to_double_nullable", C.c_double, C.c_int64, C.c_int64, C.c_double)
    _sig(R, "cast_double_to_int64_t_nullable", C.c_int64, C.c_double, C.c_double, C.c_int64)
    # hash_buff is passed as an int64 in the reference (GroupByRuntime.cpp:298-308)
    _sig(R, "hash_join_idx", C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "bucketized_hash_join_idx", C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "hash_join_idx_nullable", C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "hash_join_idx_bitwise", C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
         C.c_int64)
    _sig(R, "bucketized_hash_join_idx_nullable", C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
         C.c_int64)
    _sig(R, "bucketized_hash_join_idx_bitwise", C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
         C.c_int64, C.c_int64)
    _sig(R, "fixed_width_small_date_decode", C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64)
    _sig(R, "translate_null_key_int64_t", C.c_int64, C.c_int64, C.c_int64, C.c_int64)
    _sig(R, "baseline_hash_join_idx_32", C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t)
    _sig(R, "baseline_hash_join_idx_64", C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t)
    _ref = R
    return R


# ---------------------------------------------------------------------------------------------
# numpy-level helpers
# ---------------------------------------------------------------------------------------------

def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class HostFragments:
    """col_buffers[frag][buf_idx] / num_rows[frag*num_tables+t] marshalled for orc_run_plan.

    `frags` is a list (per fragment) of lists of numpy arrays (per input column buffer)."""

    def __init__(self, frags, num_rows, num_tables=1):
        self.keep = frags
        nf = len(frags)
        self.num_fragments = nf
        self.num_tables = num_tables
        self.num_rows = np.ascontiguousarray(num_rows, dtype=np.int64)
        assert self.num_rows.size == nf * num_tables
        self.inner = []
        outer = (C.c_void_p * max(nf, 1))()
        for f, cols in enumerate(frags):
            arr = (C.c_void_p * max(len(cols), 1))()
            for i, c in enumerate(cols):
                assert c.flags["C_CONTIGUOUS"]
                arr[i] = c.ctypes.data
            self.inner.append(arr)
            outer[f] = C.cast(arr, C.c_void_p).value
        self.outer = outer


def run_plan(plan, frags: HostFragments, out: np.ndarray, join_tables=None):
    """Run the oracle's row function for `plan`; `out` is the initialised output buffer."""
    jt = None
    keep = None
    if join_tables:
        if len(join_tables) == 1:
            jt = C.c_void_p(join_tables[0].ctypes.data)
        else:
            keep = np.array([t.ctypes.data for t in join_tables], dtype=np.int64)
            jt = _ptr(keep)
    err = lib().orc_run_plan(C.byref(plan), C.cast(frags.outer, C.c_void_p), frags.num_fragments,
                             _ptr(frags.num_rows), frags.num_tables, jt, _ptr(out))
    return err


def run_projection(plan, frags: HostFragments, out: np.ndarray, max_matched: int, join_tables=None):
    """Filter/project plan; returns (err, number of rows claimed)."""
    jt = None
    keep = None
    if join_tables:
        if len(join_tables) == 1:
            jt = C.c_void_p(join_tables[0].ctypes.data)
        else:
            keep = np.array([t.ctypes.data for t in join_tables], dtype=np.int64)
            jt = _ptr(keep)
    total = np.zeros(1, dtype=np.int32)
    err = lib().orc_run_projection(C.byref(plan), C.cast(frags.outer, C.c_void_p), frags.num_fragments,
                                   _ptr(frags.num_rows), frags.num_tables, jt, _ptr(out), max_matched, _ptr(total))
    return err, int(total[0])


def run_plan_parallel(plan, frags: HostFragments, init_buffer: np.ndarray, init_vals: np.ndarray,
                      num_threads: int, join_tables=None):
    out = np.empty_like(init_buffer)
    jt = None
    keep = None
    if join_tables:
        if len(join_tables) == 1:
            jt = C.c_void_p(join_tables[0].ctypes.data)
        else:
            keep = np.array([t.ctypes.data for t in join_tables], dtype=np.int64)
            jt = _ptr(keep)
    iv = np.ascontiguousarray(init_vals, dtype=np.int64)
    err = lib().orc_run_plan_parallel(C.byref(plan), C.cast(frags.outer, C.c_void_p),
                                      frags.num_fragments, _ptr(frags.num_rows), frags.num_tables, jt,
                                      _ptr(init_buffer), init_buffer.size, _ptr(iv), num_threads,
                                      _ptr(out))
    return err, out


def reduce(plan, this_buf, this_entry_count, that_buf, that_entry_count, init_vals):
    iv = np.ascontiguousarray(init_vals, dtype=np.int64)
    return lib().orc_reduce(C.byref(plan), _ptr(this_buf), this_entry_count, _ptr(that_buf),
                            that_entry_count, _ptr(iv))


def make_join_chunks(arrays):
    """JoinChunk[] over host numpy arrays (row ids are running offsets)."""
    chunks = (A.JoinChunk * len(arrays))()
    rid = 0
    for i, a in enumerate(arrays):
        chunks[i].col_buff = a.ctypes.data
        chunks[i].num_elems = a.size
        chunks[i].row_id = rid
        rid += a.size
    return chunks


def c2_jit_shaped(keys, vals, plan, init_buffer, threads, first_touch=True, reps=3):
    """CPU baseline leg of bench.py: GROUP BY key SUM(val) over 8-byte columns with the row loop HDK's JIT would emit
    (hdk_oracle.c: orc_c2_jit_shaped).  keys / vals: lists of int64 numpy arrays, one per fragment.
    Returns (seconds of the best of `reps` timed regions, output buffer)."""
    n = len(keys)
    kp = (C.c_void_p * n)(*[k.ctypes.data for k in keys])
    vp = (C.c_void_p * n)(*[x.ctypes.data for x in vals])
    nr = np.array([len(k) for k in keys], dtype=np.int64)
    init = np.ascontiguousarray(init_buffer, dtype=np.int64)
    out = np.empty_like(init)
    # the C2 shape: perfect hash on one 8-byte key, row-wise, targets = [projected key,] SUM(8-byte integer column)
    p = plan
    if not (p.query_kind == A.Q_PERFECT_HASH and p.key_count == 1 and not p.output_columnar and
            p.key_bucket[0] in (0, 1) and not p.key_has_nulls[0]):
        raise ValueError("not the C2 shape")
    key_slot, sum_slot, skip, nullv = -1, -1, 0, 0
    for t in range(p.num_targets):
        tg = p.targets[t]
        if tg.agg == A.AGG_ID and tg.slot_width == 8:
            key_slot = tg.slot_off // 8
        elif tg.agg == A.AGG_SUM and tg.slot_width == 8 and not tg.arg_is_fp and sum_slot < 0:
            sum_slot, skip, nullv = tg.slot_off // 8, int(tg.skip_null), int(tg.null_val)
        else:
            raise ValueError("not the C2 shape")
    sec = lib().orc_c2_jit_shaped(kp, vp, nr.ctypes.data, n, int(p.key_min[0]), int(p.entry_count), int(p.row_size_quad),
                                  (int(p.idx_target_as_key) + 1) if p.keyless else 0, key_slot, sum_slot, skip, nullv, init.ctypes.data, int(threads), int(bool(first_touch)),
                                  int(reps), out.ctypes.data)
    if sec < 0:
        raise MemoryError("orc_c2_jit_shaped could not allocate its fragment copies")
    return sec, out
