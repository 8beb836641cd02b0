"""ctypes mirror of include/hdk_hip.h (the C ABI's PODs and constants).

Pure interface definition: no compute, no library loading (that is `_lib.py`).
Field order and types must match include/hdk_hip.h exactly; tests/test_abi.py checks sizes
against the compiled library (`hdk_hip_sizeof_*`).
"""
import ctypes as C

# --- limits (hdk_hip.h) ---------------------------------------------------------------------
MAX_COLS = 24
MAX_KEYS = 4
MAX_TARGETS = 8
MAX_QUALS = 6
MAX_JOINS = 2
MAX_JOIN_KEYS = 3
MAX_EXPR_STEPS = 3
MAX_FILTER_OPS = 16
F_AND, F_OR, F_NOT = 64, 65, 66
PLAN_ABI = 4

# --- sentinels: reference omniscidb/Shared/InlineNullValues.h:33-39, QueryEngine/GpuRtConstants.h:29-32
EMPTY_KEY_64 = 2**63 - 1
EMPTY_KEY_32 = 2**31 - 1
EMPTY_KEY_16 = 2**15 - 1
EMPTY_KEY_8 = 2**7 - 1
NULL_BIGINT = -(2**63)
NULL_INT = -(2**31)
NULL_SMALLINT = -(2**15)
NULL_TINYINT = -(2**7)
NULL_DOUBLE_BITS = 0x0010000000000000  # DBL_MIN
NULL_FLOAT_BITS = 0x00800000  # FLT_MIN
FP_SLOT_NONE, FP_SLOT_DOUBLE, FP_SLOT_FLOAT = 0, 1, 2  # hdk_hip_fp_slot (target.arg_is_fp)
JOIN_INVALID_SLOT = -1

# --- status codes (reference QueryEngine/Execute.h:1019-1031 + library-level) ----------------
OK = 0
ERR_DIV_BY_ZERO = 1
ERR_OUT_OF_GPU_MEM = 2
ERR_OUT_OF_SLOTS = 3
ERR_OVERFLOW_OR_UNDERFLOW = 7
ERR_OUT_OF_TIME = 9
ERR_INTERRUPTED = 10
ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES = 15
ERR_UNSUPPORTED = 100
ERR_INVALID_ARG = 101
ERR_RUNTIME = 102
ERR_EXCHANGE_INCOMPLETE = 103

# --- enums -----------------------------------------------------------------------------------
VC_INT, VC_FP = 0, 1
COL_INT, COL_UNSIGNED, COL_FLOAT, COL_DOUBLE, COL_SMALL_DATE = 0, 1, 2, 3, 4
LEAF_NONE, LEAF_COL, LEAF_INT, LEAF_FP = 0, 1, 2, 3
(OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD, OP_EXTRACT_YEAR, OP_SCALE_DOWN, OP_FLOOR_DIV,
 OP_CAST_INT_TO_FP, OP_CAST_FP_TO_INT) = range(1, 11)
CMP_EQ, CMP_NE, CMP_LT, CMP_GT, CMP_LE, CMP_GE = range(1, 7)
JOIN_ONE_TO_ONE, JOIN_ONE_TO_MANY, JOIN_ONE_TO_ONE_FUSED, JOIN_KEYED_ONE_TO_ONE, JOIN_KEYED_ONE_TO_MANY = 0, 1, 2, 3, 4
JOIN_INNER, JOIN_LEFT, JOIN_SEMI, JOIN_ANTI = 0, 1, 2, 3
JOIN_NULL_NONE, JOIN_NULL_NULLABLE, JOIN_NULL_BITWISE = 0, 1, 2
Q_NON_GROUPED, Q_PERFECT_HASH, Q_BASELINE_HASH, Q_PROJECTION = 0, 1, 2, 3
AGG_COUNT, AGG_SUM, AGG_MIN, AGG_MAX, AGG_AVG, AGG_ID, AGG_SINGLE_VALUE = 0, 1, 2, 3, 4, 5, 6
JC_SMALL_DATE, JC_SIGNED, JC_UNSIGNED, JC_DOUBLE = 0, 1, 2, 3
(KP_COL_BUFFERS, KP_NUM_FRAGMENTS, KP_LITERALS, KP_NUM_ROWS, KP_FRAG_ROW_OFFSETS, KP_MAX_MATCHED,
 KP_TOTAL_MATCHED, KP_INIT_AGG_VALS, KP_GROUPBY_BUF, KP_ERROR_CODE, KP_NUM_TABLES,
 KP_JOIN_HASH_TABLES, KP_COUNT) = range(13)
LAUNCH_FORCE_GLOBAL_ATOMICS = 1
LAUNCH_RECORD_EVENTS = 2
LAUNCH_FORCE_GENERIC = 4
LAUNCH_FORCE_SCALAR = 8
LAUNCH_FORCE_PARTITIONED = 16
LAUNCH_WIDE_TUPLES = 1024
LAUNCH_ACCUMULATE = 2048
LAUNCH_PLAN_RESIDENT = 32
LAUNCH_CHECK_INTERRUPT = 64
LAUNCH_INIT_OUTPUT = 128
LAUNCH_CLUSTER_PROBES = 256
LAUNCH_NO_CLUSTER_PROBES = 512


class Col(C.Structure):
    _fields_ = [("buf_idx", C.c_int32), ("table", C.c_int32), ("width", C.c_int32),
                ("kind", C.c_int32),
                # ChunkStats of the column over the launch's fragments (DataMgr/ChunkMetadata.h); has_stats = 0: unknown
                ("has_stats", C.c_int32), ("has_nulls", C.c_int32), ("min_val", C.c_int64), ("max_val", C.c_int64)]


class Leaf(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col", C.c_int32), ("ival", C.c_int64),
                ("null_val", C.c_int64), ("nullable", C.c_int32), ("pad_", C.c_int32)]


class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("out_class", C.c_int32), ("rhs", Leaf),
                ("null_out", C.c_int64), ("check_width", C.c_int32), ("pad_", C.c_int32)]


class Expr(C.Structure):
    _fields_ = [("vclass", C.c_int32), ("nsteps", C.c_int32), ("leaf0", Leaf),
                ("steps", Step * MAX_EXPR_STEPS), ("null_val", C.c_int64),
                ("nullable", C.c_int32), ("pad_", C.c_int32)]


class Qual(C.Structure):
    _fields_ = [("lhs", Expr), ("rhs", Leaf), ("cmp", C.c_int32), ("after_joins", C.c_int32)]


class Join(C.Structure):
    _fields_ = [("outer_key", Expr), ("min_key", C.c_int64), ("max_key", C.c_int64),
                ("null_val", C.c_int64), ("translated_null", C.c_int64), ("bucket", C.c_int64),
                ("kind", C.c_int32), ("type", C.c_int32), ("null_mode", C.c_int32),
                ("table_idx", C.c_int32), ("fused_stride", C.c_int32), ("key_component_count", C.c_int32),
                ("entry_count", C.c_int64), ("key_component_width", C.c_int32), ("pad_", C.c_int32),
                ("extra_keys", Expr * (MAX_JOIN_KEYS - 1))]


class Target(C.Structure):
    _fields_ = [("agg", C.c_int32), ("has_arg", C.c_int32), ("arg", Expr),
                ("skip_null", C.c_int32), ("slot_width", C.c_int32), ("slot_off", C.c_int32),
                ("slot2_width", C.c_int32), ("slot2_off", C.c_int32), ("arg_is_fp", C.c_int32),
                ("key_idx", C.c_int32), ("pad_", C.c_int32), ("null_val", C.c_int64)]


class Plan(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("query_kind", C.c_int32),
                ("num_cols", C.c_int32), ("cols", Col * MAX_COLS),
                ("num_quals", C.c_int32), ("quals", Qual * MAX_QUALS),
                ("num_filter_ops", C.c_int32), ("filter_after_joins", C.c_int32),
                ("filter_ops", C.c_uint8 * MAX_FILTER_OPS),
                ("num_joins", C.c_int32), ("joins", Join * MAX_JOINS),
                ("key_count", C.c_int32), ("keys", Expr * MAX_KEYS),
                ("key_min", C.c_int64 * MAX_KEYS), ("key_bucket", C.c_int64 * MAX_KEYS),
                ("key_card", C.c_int64 * MAX_KEYS), ("key_null_translated", C.c_int64 * MAX_KEYS),
                ("key_has_nulls", C.c_int32 * MAX_KEYS),
                ("entry_count", C.c_uint32), ("key_width", C.c_int32), ("keyless", C.c_int32),
                ("idx_target_as_key", C.c_int32),
                ("output_columnar", C.c_int32), ("row_size_quad", C.c_uint32),
                ("num_targets", C.c_int32), ("targets", Target * MAX_TARGETS)]


class ExchangeShape(C.Structure):
    """hdk_hip_exchange_shape: geometry of the multi-GPU tuple exchange (include/hdk_hip.h)."""
    _fields_ = [("num_owners", C.c_uint32), ("owner_entry_count", C.c_uint32), ("tuple_bytes", C.c_uint32),
                ("coarse_per_owner", C.c_uint32), ("regions_log2", C.c_uint32), ("reserved_", C.c_uint32),
                ("sub_slab_tuples", C.c_uint64), ("segment_header_bytes", C.c_uint64), ("segment_bytes", C.c_uint64),
                ("rows_bound", C.c_uint64), ("scatter_workspace_bytes", C.c_uint64),
                ("aggregate_workspace_bytes", C.c_uint64)]


class KernelOptions(C.Structure):
    _fields_ = [("grid_dim_x", C.c_uint32), ("block_dim_x", C.c_uint32),
                ("shared_mem_bytes", C.c_uint32), ("flags", C.c_uint32), ("total_rows", C.c_uint64),
                ("watchdog_ms", C.c_uint32), ("reserved_", C.c_uint32)]


class DeviceProperties(C.Structure):
    _fields_ = [("global_mem", C.c_size_t), ("num_cu", C.c_int32),
                ("max_threads_per_block", C.c_int32), ("wavefront_size", C.c_int32),
                ("grid_size", C.c_int32), ("shared_mem_per_block", C.c_size_t),
                ("has_shared_memory_atomics", C.c_int32), ("can_load_async", C.c_int32),
                ("has_fp64", C.c_int32), ("clock_khz", C.c_int32),
                ("memory_clock_khz", C.c_int32), ("memory_bus_width", C.c_int32),
                ("arch_name", C.c_char * 64)]


class JoinChunk(C.Structure):
    _fields_ = [("col_buff", C.c_void_p), ("num_elems", C.c_size_t), ("row_id", C.c_size_t)]


class JoinColumn(C.Structure):
    _fields_ = [("col_chunks_buff", C.c_void_p), ("col_chunks_buff_sz", C.c_size_t),
                ("num_chunks", C.c_size_t), ("num_elems", C.c_size_t), ("elem_sz", C.c_size_t)]


class JoinColumnTypeInfo(C.Structure):
    _fields_ = [("elem_sz", C.c_size_t), ("min_val", C.c_int64), ("max_val", C.c_int64),
                ("null_val", C.c_int64), ("uses_bw_eq", C.c_int32), ("column_type", C.c_int32),
                ("translated_null_val", C.c_int64)]


class HashEntryInfo(C.Structure):
    _fields_ = [("hash_entry_count", C.c_size_t), ("bucket_normalization", C.c_int64)]


def to_i64(v: int) -> int:
    """Wrap a Python int into the signed 64-bit range (two's complement)."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v
