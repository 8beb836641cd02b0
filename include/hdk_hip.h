/*
 * hdk_hip.h -- C ABI of the MI355X (gfx950) operator library for HDK's per-fragment hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes / PODs, returns an int32 status
 * (0 = ok; 1..16 = HDK's own Executor error codes, reference omniscidb/QueryEngine/Execute.h:1019-1031;
 * <0 = out of slots, as in RuntimeFunctions.cpp:1123-1135) and never lets an exception cross the
 * boundary.  `hdk_hip_last_error()` returns a thread-local message for the last non-zero status.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference
 * root, `QE/` = omniscidb/QueryEngine/).  INTEGRATION.md shows the HDK-side bindings
 * (HipMgr : GpuMgr, HipKernel : DeviceKernel, the *_on_device free functions).
 *
 * The JIT'ed row function that HDK generates per query (QE/RowFuncBuilder.cpp,
 * QE/QueryTemplateGenerator.cpp) is replaced by a POD *plan* (`hdk_hip_plan`) interpreted by a
 * fixed library of hand-written kernels; shapes the library does not cover are rejected with
 * HDK_HIP_ERR_UNSUPPORTED so the caller can take HDK's existing QueryMustRunOnCpu retry
 * (QE/RelAlgExecutor.cpp:183-192).
 */
#ifndef HDK_HIP_H
#define HDK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Constants shared with the reference
 * ---------------------------------------------------------------------------------------- */
#define HDK_EMPTY_KEY_64 INT64_MAX /* QE/GpuRtConstants.h:29 */
#define HDK_EMPTY_KEY_32 INT32_MAX /* :30 */
#define HDK_EMPTY_KEY_16 INT16_MAX /* :31 */
#define HDK_EMPTY_KEY_8 INT8_MAX   /* :32 */
#define HDK_NULL_BIGINT INT64_MIN  /* Shared/InlineNullValues.h:37 */
#define HDK_NULL_INT INT32_MIN
#define HDK_NULL_SMALLINT INT16_MIN
#define HDK_NULL_TINYINT INT8_MIN
/* NULL_DOUBLE = DBL_MIN, NULL_FLOAT = FLT_MIN (smallest positive normal), compared bit-wise. */
#define HDK_NULL_DOUBLE_BITS INT64_C(0x0010000000000000)
#define HDK_NULL_FLOAT_BITS INT32_C(0x00800000)
#define HDK_JOIN_INVALID_SLOT (-1) /* QE/JoinHashTable/Runtime/JoinHashTableQueryRuntime.cpp:38 */

/* status / error codes: QE/Execute.h:1019-1031 */
#define HDK_HIP_OK 0
#define HDK_HIP_ERR_DIV_BY_ZERO 1
#define HDK_HIP_ERR_OUT_OF_GPU_MEM 2
#define HDK_HIP_ERR_OUT_OF_SLOTS 3
#define HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW 7
#define HDK_HIP_ERR_OUT_OF_TIME 9
#define HDK_HIP_ERR_INTERRUPTED 10
#define HDK_HIP_ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES 15
/* library-level failures (not produced by HDK's runtime): */
#define HDK_HIP_ERR_UNSUPPORTED 100 /* plan shape outside the fixed library -> QueryMustRunOnCpu */
#define HDK_HIP_ERR_INVALID_ARG 101
#define HDK_HIP_ERR_RUNTIME 102 /* a HIP runtime call failed; see hdk_hip_last_error() */
#define HDK_HIP_ERR_EXCHANGE_INCOMPLETE 103 /* device code of hdk_hip_aggregate_from_ranks: a rank's segment was flagged
                                               (a sub-slab overflowed: skewed keys; stale column statistics; or the
                                               scatter was interrupted), or the owner's spill list overflowed (its
                                               table is all but full) -- redo the step with the table exchange.  A
                                               heavy hitter that only overflows the OWNER's slabs is not an error: the
                                               owner then applies its inbox with atomics (hdk_part_owner_fallback) */
#define HDK_HIP_ERR_JOIN_SLOT_TAKEN (-1) /* one-to-one build hit a duplicate key (JoinHashImpl.h:55-66) */

const char* hdk_hip_last_error(void);
/* Library/ABI version (major*1000+minor). */
int32_t hdk_hip_version(void);

/* ------------------------------------------------------------------------------------------
 * HipMgr -- device manager under BufferProvider.
 * Replaces CudaMgr/L0Manager behind `struct GpuMgr` (omniscidb/DataMgr/GpuMgr.h:29-79,
 * omniscidb/CudaMgr/CudaMgr.h:83-260); one function per virtual.  There is no handle: state is a
 * per-device table (stream, properties) initialised lazily and thread-safe.
 * ---------------------------------------------------------------------------------------- */
int32_t hdk_hip_mgr_get_device_count(int32_t* count);               /* GpuMgr::getDeviceCount */
int32_t hdk_hip_mgr_set_context(int32_t device_num);                /* GpuMgr::setContext */
int32_t hdk_hip_mgr_allocate_device_mem(size_t num_bytes, int32_t device_num,
                                        int8_t** device_ptr);       /* GpuMgr::allocateDeviceMem;
                                                                       OOM -> ERR_OUT_OF_GPU_MEM */
int32_t hdk_hip_mgr_free_device_mem(int8_t* device_ptr);            /* GpuMgr::freeDeviceMem */
int32_t hdk_hip_mgr_allocate_pinned_host_mem(size_t num_bytes, int8_t** host_ptr); /* CudaMgr.h:120 */
int32_t hdk_hip_mgr_free_pinned_host_mem(int8_t* host_ptr);
int32_t hdk_hip_mgr_copy_host_to_device(int8_t* device_ptr, const int8_t* host_ptr, size_t num_bytes,
                                        int32_t device_num);        /* GpuMgr::copyHostToDevice */
int32_t hdk_hip_mgr_copy_host_to_device_async(int8_t* device_ptr, const int8_t* host_ptr,
                                              size_t num_bytes, int32_t device_num);
int32_t hdk_hip_mgr_synchronize_stream(int32_t device_num);         /* GpuMgr::synchronizeStream */
int32_t hdk_hip_mgr_copy_device_to_host(int8_t* host_ptr, const int8_t* device_ptr, size_t num_bytes,
                                        int32_t device_num);        /* GpuMgr::copyDeviceToHost */
int32_t hdk_hip_mgr_copy_device_to_device(int8_t* dest_ptr, int8_t* src_ptr, size_t num_bytes,
                                          int32_t dest_device_num, int32_t src_device_num);
int32_t hdk_hip_mgr_zero_device_mem(int8_t* device_ptr, size_t num_bytes, int32_t device_num);
int32_t hdk_hip_mgr_set_device_mem(int8_t* device_ptr, unsigned char uc, size_t num_bytes,
                                   int32_t device_num);
int32_t hdk_hip_mgr_synchronize_devices(void);                      /* GpuMgr::synchronizeDevices */
/* The per-device stream the manager's async copies and (when `stream`==NULL) the kernels use. */
int32_t hdk_hip_mgr_get_stream(int32_t device_num, void** stream);

typedef struct hdk_hip_device_properties { /* CudaMgr.h:43-66 `DeviceProperties` */
  size_t global_mem;                /* GpuMgr::getTotalMem */
  int32_t num_cu;                   /* GpuMgr::getMinEUNumForAllDevices (256 on MI355X) */
  int32_t max_threads_per_block;    /* GpuMgr::getMaxBlockSize */
  int32_t wavefront_size;           /* GpuMgr::getSubGroupSize -> 64 on CDNA4 */
  int32_t grid_size;                /* GpuMgr::getGridSize: blocks a persistent launch uses */
  size_t shared_mem_per_block;      /* GpuMgr::getMinSharedMemoryPerBlockForAllDevices */
  int32_t has_shared_memory_atomics;/* GpuMgr::hasSharedMemoryAtomicsSupport -> 1 */
  int32_t can_load_async;           /* GpuMgr::canLoadAsync -> 1 */
  int32_t has_fp64;                 /* GpuMgr::hasFP64Support -> 1 */
  int32_t clock_khz;
  int32_t memory_clock_khz;
  int32_t memory_bus_width;
  char arch_name[64];               /* "gfx950..." */
} hdk_hip_device_properties;
int32_t hdk_hip_mgr_get_device_properties(int32_t device_num, hdk_hip_device_properties* out);
/* Measurement helper (no reference counterpart): what plain streaming kernels reach on this device -- a 16 B/lane
 * copy (bytes read + written per second) and a 16 B/lane read, best of `reps` over `bytes` of scratch.  bench.py
 * reports them as roofline.peak_measured next to the nominal 8 TB/s. */
int32_t hdk_hip_mgr_measure_hbm(int32_t device_num, size_t bytes, int32_t reps, double* copy_gbps, double* read_gbps);

/* ------------------------------------------------------------------------------------------
 * Output-buffer initialisation (kernel #1 of every group-by launch).
 * Replaces init_group_by_buffer_on_device / init_columnar_group_by_buffer_on_device
 * (QE/GpuInitGroups.h:23-48, kernels QE/GpuInitGroups.cu:17-232).  Same arguments minus the
 * platform enum, plus device + stream.  block/grid sizes are accepted for signature parity and
 * only used as hints.
 * ---------------------------------------------------------------------------------------- */
int32_t hdk_hip_init_group_by_buffer(int64_t* groups_buffer, const int64_t* init_vals,
                                     uint32_t groups_buffer_entry_count, uint32_t key_count,
                                     uint32_t key_width, uint32_t row_size_quad, int32_t keyless,
                                     int8_t warp_size, size_t block_size_x, size_t grid_size_x,
                                     int32_t device_id, void* stream);
int32_t hdk_hip_init_columnar_group_by_buffer(int64_t* groups_buffer, const int64_t* init_vals,
                                              uint32_t groups_buffer_entry_count, uint32_t key_count,
                                              uint32_t agg_col_count, const int8_t* col_sizes,
                                              int32_t need_padding, int32_t keyless, int8_t key_size,
                                              size_t block_size_x, size_t grid_size_x,
                                              int32_t device_id, void* stream);

/* ------------------------------------------------------------------------------------------
 * The plan: POD replacement for the JIT'ed row function.
 * ---------------------------------------------------------------------------------------- */
#define HDK_HIP_MAX_COLS 24
#define HDK_HIP_MAX_KEYS 4
#define HDK_HIP_MAX_TARGETS 8
#define HDK_HIP_MAX_QUALS 6
#define HDK_HIP_MAX_JOINS 2
#define HDK_HIP_MAX_JOIN_KEYS 3 /* key components of a keyed (composite-key) join */
#define HDK_HIP_MAX_EXPR_STEPS 3
#define HDK_HIP_MAX_FILTER_OPS 16
enum hdk_hip_filter_op { HDK_F_AND = 64, HDK_F_OR = 65, HDK_F_NOT = 66 }; /* < 64: push quals[op] */

enum hdk_hip_value_class { HDK_VC_INT = 0, HDK_VC_FP = 1 };

/* How a fixed-width column element is decoded (QE/DecodersImpl.h:30-150). */
enum hdk_hip_col_kind {
  HDK_COL_INT = 0,      /* fixed_width_int_decode: sign-extend 1/2/4/8 bytes */
  HDK_COL_UNSIGNED = 1, /* fixed_width_unsigned_decode (dictionary ids of 1/2 bytes) */
  HDK_COL_FLOAT = 2,    /* fixed_width_float_decode, widened to double */
  HDK_COL_DOUBLE = 3,   /* fixed_width_double_decode */
  HDK_COL_SMALL_DATE = 4 /* fixed_width_small_date_decode (QE/DecodersImpl.h:151-159; chosen by get_col_decoder for a DATE
                           stored in days, QE/ColumnIR.cpp:46-49): a 2- or 4-byte day count, read as epoch SECONDS
                           (x 86400); the column's narrow NULL (INT16_MIN / INT32_MIN, QE/Codec.cpp:86-87) becomes
                           NULL_BIGINT.  Read by the plan interpreters only (no specialised kernel takes such a column) */
};

typedef struct hdk_hip_col {
  int32_t buf_idx;   /* index into col_buffers[frag][...] (the COL_BUFFERS kernel param) */
  int32_t table;     /* 0 = outer table (row = pos); j>0 = inner table of join j-1 (row = matched id);
                        -j (j>0) = PAYLOAD word `buf_idx` of join j-1's fused table (see
                        HDK_JOIN_ONE_TO_ONE_FUSED): no column buffer is read */
  int32_t width;     /* bytes per element */
  int32_t kind;      /* hdk_hip_col_kind */
  /* Statistics of the column over the fragments of the launch: the ChunkStats {min, max, has_nulls} every chunk's
   * metadata carries (omniscidb/DataMgr/ChunkMetadata.h) and the reference's planner reads through
   * getExpressionRange (QE/ExpressionRange.cpp) to choose hash layouts.  Here they also let a multi-pass strategy
   * move a column in fewer bytes than its physical width (8-byte tuples in the radix-partitioned group-by when key
   * and argument both fit 32 bits).  Optional: has_stats = 0 means unknown.  They must hold for every element read;
   * the kernels that rely on them notice a value outside and redo the launch at full width (no wrong result from
   * stale metadata, only a slower one). */
  int32_t has_stats; /* 1: min_val / max_val / has_nulls below are valid (integer columns only) */
  int32_t has_nulls; /* ChunkStats::has_nulls: some element is the in-band NULL */
  int64_t min_val;   /* bounds of the non-NULL elements */
  int64_t max_val;
} hdk_hip_col;

/* Expression: a short left-to-right chain  acc = leaf0; acc = op_i(acc, leaf_i)  i < nsteps.
 * Integer values are carried as int64, fp as double.  NULLs stay in-band exactly like the
 * reference's *_nullable runtime functions (QE/RuntimeFunctions.cpp:49-230): each step names the
 * sentinel of its inputs and of its result. */
enum hdk_hip_leaf_kind { HDK_LEAF_NONE = 0, HDK_LEAF_COL = 1, HDK_LEAF_INT = 2, HDK_LEAF_FP = 3 };
typedef struct hdk_hip_leaf {
  int32_t kind;     /* hdk_hip_leaf_kind */
  int32_t col;      /* index into plan->cols when kind == HDK_LEAF_COL */
  int64_t ival;     /* literal (int), or bit pattern of the double literal */
  int64_t null_val; /* in-band NULL of this leaf, widened (int) or double bits (fp); */
  int32_t nullable; /* 0 => the leaf can never be NULL (null_val ignored) */
  int32_t pad_;
} hdk_hip_leaf;

enum hdk_hip_op {
  HDK_OP_ADD = 1, HDK_OP_SUB, HDK_OP_MUL, HDK_OP_DIV, HDK_OP_MOD, /* DEF_ARITH_NULLABLE */
  HDK_OP_EXTRACT_YEAR,     /* omniscidb/Utils/ExtractFromTime.cpp: extract_year on epoch seconds */
  HDK_OP_SCALE_DOWN,       /* scale_decimal_down_[not_]nullable (RuntimeFunctions.cpp:245-262) */
  HDK_OP_FLOOR_DIV,        /* floor_div_[nullable_]lhs (:266-279) */
  HDK_OP_CAST_INT_TO_FP,   /* cast_int64_t_to_double_nullable (:311-345) */
  HDK_OP_CAST_FP_TO_INT    /* DEF_ROUND_NULLABLE(double,int64_t) */
};
typedef struct hdk_hip_step {
  int32_t op;        /* hdk_hip_op */
  int32_t out_class; /* hdk_hip_value_class of the result */
  hdk_hip_leaf rhs;  /* second operand (unused by unary ops) */
  int64_t null_out;  /* in-band NULL of the result */
  int32_t check_width; /* + - * on integers: byte width (1/2/4/8) of the operation's SQL type; a result outside that
                          type's range ends the query with ERR_OVERFLOW_OR_UNDERFLOW, like the checked arithmetic the
                          reference generates (QE/ArithmeticIR.cpp:277-520; NULL operands are not checked).  0: unchecked */
  int32_t pad_;
} hdk_hip_step;
typedef struct hdk_hip_expr {
  int32_t vclass;    /* value class of the final result */
  int32_t nsteps;
  hdk_hip_leaf leaf0;
  hdk_hip_step steps[HDK_HIP_MAX_EXPR_STEPS];
  int64_t null_val;  /* in-band NULL of the final result */
  int32_t nullable;  /* whether the result may be NULL */
  int32_t pad_;
} hdk_hip_expr;

/* Filter conjunct `lhs cmp rhs`, three-valued (DEF_CMP_NULLABLE, RuntimeFunctions.cpp:83-117);
 * a row passes when every conjunct is TRUE (logical_and, :357-372; NULL does not pass). */
enum hdk_hip_cmp { HDK_CMP_EQ = 1, HDK_CMP_NE, HDK_CMP_LT, HDK_CMP_GT, HDK_CMP_LE, HDK_CMP_GE };
typedef struct hdk_hip_qual {
  hdk_hip_expr lhs;
  hdk_hip_leaf rhs;
  int32_t cmp;
  int32_t after_joins; /* 0: reads outer-table columns only, evaluated before any probe (the reference
                          hoists such filters in front of the join loops, IRCodegen.cpp:569-570);
                          1: reads joined columns, evaluated once per matching row combination */
} hdk_hip_qual;

/* Equi-join probe.  Perfect-hash tables: hash_join_idx family (QE/GroupByRuntime.cpp:274-366);
 * keyed ("baseline") tables: baseline_hash_join_idx_{32,64}
 * (QE/JoinHashTable/Runtime/JoinHashTableQueryRuntime.cpp:42-98).  The join loops follow
 * Executor::buildJoinLoops (QE/IRCodegen.cpp:497-667): a one-to-one table yields at most one inner
 * row (JoinLoopKind::Singleton), a one-to-many table a set of them (JoinLoopKind::Set, the matching
 * set of HashJoin::codegenMatchingSet, QE/JoinHashTable/HashJoin.cpp:149-197); a LEFT join with no
 * match runs the rest of the row once with every column of the inner table NULL. */
enum hdk_hip_join_kind {
  HDK_JOIN_ONE_TO_ONE = 0,  /* the reference's table: int32 row id per slot (PerfectHashTableBuilder.h:35-38) */
  HDK_JOIN_ONE_TO_MANY = 1, /* int32 [offsets(entry_count) | counts(entry_count) | row ids]
                               (fill_one_to_many_hash_table, HashJoinRuntime.cpp:770-853) */
  /* MI355X addition (no reference counterpart): int64 entries [row id | payload 1 | ... ] per slot,
   * `fused_stride` quads apart, built by hdk_hip_build_fused_join_table from the reference table and
   * the referenced inner columns.  One 16..32-B gather per probing row replaces the dependent
   * slot -> row id -> inner column chain (two cache lines) of the reference layout. */
  HDK_JOIN_ONE_TO_ONE_FUSED = 2,
  /* keyed tables (wide or composite keys), key components `key_component_width` bytes wide:
   *   one-to-one : entry_count x [key components | row id]        (width-sized row id)
   *   one-to-many: entry_count x [key components], then int32 [offsets | counts | row ids]
   * (BaselineJoinHashTable layout, QE/JoinHashTable/Runtime/HashJoinRuntime.cpp:357-507,855-1000) */
  HDK_JOIN_KEYED_ONE_TO_ONE = 3,
  HDK_JOIN_KEYED_ONE_TO_MANY = 4
};
enum hdk_hip_join_type {
  HDK_JOIN_INNER = 0,
  HDK_JOIN_LEFT = 1,
  /* JoinType::SEMI / ANTI (Shared/sqldefs.h:33; JoinLoop::codegen, QE/LoopControlFlow/JoinLoop.cpp:254-262): the table
   * is a one-to-one table filled with for_semi_join = 1 (first row of a key wins, duplicates are not an error:
   * fill_hashtable_for_semi_join, JoinHashImpl.h:68-77).  SEMI runs the rest of the row when the probe finds a slot
   * (exactly like INNER over that table: every outer row at most once), ANTI when it finds none (`slot_lookup_result
   * < 0`); an ANTI join's inner columns are never read. */
  HDK_JOIN_SEMI = 2,
  HDK_JOIN_ANTI = 3
};
enum hdk_hip_join_null { HDK_JOIN_NULL_NONE = 0, HDK_JOIN_NULL_NULLABLE = 1, HDK_JOIN_NULL_BITWISE = 2 };
typedef struct hdk_hip_join {
  hdk_hip_expr outer_key;
  int64_t min_key;
  int64_t max_key;
  int64_t null_val;
  int64_t translated_null; /* [bucketized_]hash_join_idx_bitwise: the key a NULL probes with -- what getHashJoinArgs passes
                              (QE/JoinHashTable/PerfectJoinHashTable.cpp:803-810): max_key + 1, or for a bucketized (DATE)
                              key max_key / bucket + 1 */
  int64_t bucket;          /* bucket_normalization (DATE keys: 86400, PerfectJoinHashTable.cpp:81); 0/1 => plain.  The
                              table then has ceil((max - min + 1 [+ 1 when NULLs match]) / bucket) slots
                              (HashEntryInfo::getNormalizedHashEntryCount, HashJoinRuntime.h:46-55) and `entry_count` of a
                              one-to-many table is that number */
  int32_t kind;            /* hdk_hip_join_kind */
  int32_t type;            /* hdk_hip_join_type */
  int32_t null_mode;       /* hdk_hip_join_null */
  int32_t table_idx;       /* which entry of JOIN_HASH_TABLES */
  int32_t fused_stride;    /* HDK_JOIN_ONE_TO_ONE_FUSED: int64 words per slot (1 + payload columns) */
  int32_t key_component_count; /* keyed tables: 1 + number of `extra_keys` in use */
  int64_t entry_count;     /* slots of a one-to-many or keyed table */
  int32_t key_component_width; /* keyed tables: 4 or 8 */
  int32_t pad_;
  hdk_hip_expr extra_keys[HDK_HIP_MAX_JOIN_KEYS - 1]; /* keyed tables: outer-side components 2.. */
} hdk_hip_join;

/* Query shape: RS/QueryMemoryDescriptor.h `QueryDescriptionType`. */
enum hdk_hip_query_kind {
  HDK_Q_NON_GROUPED = 0,        /* NonGroupedAggregate */
  HDK_Q_PERFECT_HASH = 1,       /* GroupByPerfectHash */
  HDK_Q_BASELINE_HASH = 2,      /* GroupByBaselineHash */
  HDK_Q_PROJECTION = 3          /* Projection: filter/project, one output row per passing input row.
                                   Layout (RS/QueryMemoryDescriptor.cpp:314-342): row-wise
                                   [int64 row position | slots], columnar [int64 positions | slot
                                   columns at their logical widths]; rows land at atomically claimed
                                   positions (TOTAL_MATCHED), `entry_count` = MAX_MATCHED rows. */
};

enum hdk_hip_agg {
  HDK_AGG_COUNT = 0, HDK_AGG_SUM = 1, HDK_AGG_MIN = 2, HDK_AGG_MAX = 3, HDK_AGG_AVG = 4,
  HDK_AGG_ID = 5, /* non-aggregate target, written with agg_id (QE/RuntimeFunctions.cpp:473-476).
                    Group-by plans: a projected group-by key, `key_idx` names it and `arg` repeats its
                    expression.  Projection plans: any expression `arg`, key_idx = -1. */
  HDK_AGG_SINGLE_VALUE = 6 /* SINGLE_VALUE(x) (hdk::ir::AggType::kSingleValue): the slot starts at the argument's NULL,
                    the first non-NULL value is stored, a DIFFERENT non-NULL value ends the launch with
                    HDK_HIP_ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES (checked_single_agg_id[_int32|_double|_float],
                    QE/RuntimeFunctions.cpp:489-506,567-583,743-760; *_shared, QE/cuda_mapd_rt.cu:670-782; partial
                    results: reduceOneSlotSingleValue, QE/ResultSetReduction.cpp:1186-1230).  `skip_null` = 1 and
                    `null_val` = the slot's initial value; 4- or 8-byte slots.  Plans with such a target run on the
                    global-atomics kernel (hdk_scan_agg_global). */
};
enum hdk_hip_fp_slot { HDK_FP_SLOT_NONE = 0, HDK_FP_SLOT_DOUBLE = 1, HDK_FP_SLOT_FLOAT = 2 };
typedef struct hdk_hip_target {
  int32_t agg;        /* hdk_hip_agg */
  int32_t has_arg;    /* 0 => COUNT(*) */
  hdk_hip_expr arg;
  int32_t skip_null;  /* TargetInfo::skip_null_val -> *_skip_val runtime (RuntimeFunctions.cpp:612-875) */
  int32_t slot_width; /* 4 or 8: padded slot width of the first slot (RS/ColSlotContext); 0 for a projected
                         group key of a GroupByBaselineHash plan, which has no slot and is read back
                         from the key columns (target_groupby_indices, QE/MemoryLayoutBuilder.cpp:921-927;
                         ColSlotContext::addSlotForColumn(0, 0), RS/ColSlotContext.cpp:43-48) */
  int32_t slot_off;   /* row-wise: byte offset of the first slot inside the row (after keys);
                         columnar: byte offset of the slot column from the buffer start */
  int32_t slot2_width;/* AVG: width of the count slot */
  int32_t slot2_off;  /* AVG: offset of the count slot */
  int32_t arg_is_fp;  /* hdk_hip_fp_slot: 0 integer slot; 1 the slot holds a double bit pattern; 2 float accumulator
                         (takes_float_argument, Shared/TargetInfo.h:170-179: SUM / MIN / MAX / AVG over a FLOAT
                         argument): the slot's LOW 4 BYTES hold a float whatever its padded width, updated like
                         agg_{sum,min,max}_float[_skip_val] (QE/RuntimeFunctions.cpp:770-875); the other bytes of an
                         8-byte slot keep the init pattern.  Values travel through the scan as doubles (HDK_COL_FLOAT
                         widens), so `null_val` is the float sentinel WIDENED to double; the slot's own sentinel is its
                         float bits, and INIT_AGG_VALS carries that sign-extended from 32 bits, as the reference's
                         init_agg_val_vec does (QE/OutputBufferInitialization.cpp:52-65) */
  int32_t key_idx;    /* HDK_AGG_ID: index of the projected group-by key */
  int32_t pad_;
  int64_t null_val;   /* skip value == init value of the slot for nullable args (slot-typed bits) */
} hdk_hip_target;

typedef struct hdk_hip_plan {
  uint32_t abi_version;     /* must be HDK_HIP_PLAN_ABI */
  int32_t query_kind;       /* hdk_hip_query_kind */
  /* inputs */
  int32_t num_cols;
  hdk_hip_col cols[HDK_HIP_MAX_COLS];
  int32_t num_quals;
  hdk_hip_qual quals[HDK_HIP_MAX_QUALS];
  /* Filter as a boolean expression over the conjuncts above, in postfix order (AND / OR / NOT with the reference's
   * three-valued logical_and / logical_or / logical_not, QE/RuntimeFunctions.cpp:357-384): a byte < HDK_F_AND pushes
   * the value of quals[byte] (TRUE / FALSE / NULL), HDK_F_AND and HDK_F_OR combine the two topmost values, HDK_F_NOT
   * negates the topmost; a row passes when the result is TRUE.  num_filter_ops == 0: the plain conjunction of all
   * quals (each staged by its own after_joins).  With a program the whole filter is evaluated at one stage:
   * filter_after_joins. */
  int32_t num_filter_ops;
  int32_t filter_after_joins;
  uint8_t filter_ops[HDK_HIP_MAX_FILTER_OPS];
  int32_t num_joins;
  hdk_hip_join joins[HDK_HIP_MAX_JOINS];
  /* group-by keys */
  int32_t key_count;
  hdk_hip_expr keys[HDK_HIP_MAX_KEYS];
  int64_t key_min[HDK_HIP_MAX_KEYS];       /* perfect hash: ColRangeInfo.min */
  int64_t key_bucket[HDK_HIP_MAX_KEYS];    /* ColRangeInfo.bucket (0 => none) */
  int64_t key_card[HDK_HIP_MAX_KEYS];      /* getBucketedCardinality (multi-col stride factors) */
  int64_t key_null_translated[HDK_HIP_MAX_KEYS]; /* max + (bucket?bucket:1) (RowFuncBuilder.cpp:456-461) */
  int32_t key_has_nulls[HDK_HIP_MAX_KEYS]; /* translate NULL keys (perfect hash only) */
  /* output layout (subset of RS/QueryMemoryDescriptor) */
  uint32_t entry_count;
  int32_t key_width;        /* effective key width in the buffer: 4 or 8 (perfect hash: 8) */
  int32_t keyless;          /* hasKeylessHash */
  int32_t idx_target_as_key;/* keyless: SLOT index whose init value marks an empty entry
                               (QueryMemoryDescriptor::getTargetIdxForKey; AVG counts as 2 slots) */
  int32_t output_columnar;  /* didOutputColumnar */
  uint32_t row_size_quad;   /* row-wise: row bytes / 8 (keys + slots) */
  int32_t num_targets;
  hdk_hip_target targets[HDK_HIP_MAX_TARGETS];
} hdk_hip_plan;
#define HDK_HIP_PLAN_ABI 4u

/* ------------------------------------------------------------------------------------------
 * Kernel launch.
 * Replaces DeviceKernel::launch(const KernelOptions&, std::vector<int8_t*>& kernelParams)
 * (QE/DeviceKernel.h:33-61) for the JIT'ed `multifrag_query[_hoisted_literals]`
 * (QE/RuntimeFunctions.cpp:1692-1768).  `params` holds exactly the reference's 12 *device*
 * pointers in the order of QE/QueryExecutionContext.h:111-125, laid out as
 * QueryExecutionContext::prepareKernelParams builds them (QE/QueryExecutionContext.cpp:788-964).
 * ---------------------------------------------------------------------------------------- */
enum hdk_hip_kern_param {
  HDK_KP_COL_BUFFERS = 0,   /* const int8_t***  [num_fragments][num_cols] */
  HDK_KP_NUM_FRAGMENTS,     /* const uint64_t*  */
  HDK_KP_LITERALS,          /* const int8_t*    (unused: literals live in the plan) */
  HDK_KP_NUM_ROWS,          /* const int64_t*   [num_fragments * num_tables] */
  HDK_KP_FRAG_ROW_OFFSETS,  /* const uint64_t*  [num_fragments * num_tables] */
  HDK_KP_MAX_MATCHED,       /* const int32_t*   */
  HDK_KP_TOTAL_MATCHED,     /* int32_t*         */
  HDK_KP_INIT_AGG_VALS,     /* const int64_t*   */
  HDK_KP_GROUPBY_BUF,       /* int64_t**        group-by: [0] = the shared output buffer;
                                                non-grouped: out_vec, [i] = slot of target i */
  HDK_KP_ERROR_CODE,        /* int32_t*         [grid*block]; entry 0 is written */
  HDK_KP_NUM_TABLES,        /* const uint32_t*  */
  HDK_KP_JOIN_HASH_TABLES,  /* 1 table: the table itself; >1: const int64_t* array of tables */
  HDK_KP_COUNT
};

typedef struct hdk_hip_kernel_options { /* KernelOptions, QE/DeviceKernel.h:33-43 */
  uint32_t grid_dim_x;  /* 0 => library default (persistent grid sized from the CU count) */
  uint32_t block_dim_x; /* 0 => library default */
  uint32_t shared_mem_bytes; /* ignored: LDS use is decided by the kernel choice */
  uint32_t flags;       /* HDK_HIP_LAUNCH_* */
  uint64_t total_rows;  /* upper bound on the outer rows of this launch (sum of NUM_ROWS), 0 = unknown.  NUM_ROWS is
                           device memory; multi-pass strategies size their stream-ordered scratch from this instead
                           of reading it back: the radix-partitioned group-by (skipped when 0) and the selection
                           bitmask the filter/project counting pass hands to its writing pass (when 0, or for rows
                           past the bound, the writing pass evaluates the filter again) */
  uint32_t watchdog_ms; /* > 0: the launch's kernels give up with ERR_OUT_OF_TIME once this much time has passed
                           since the launch reached the device (dynamic watchdog: QE/DynamicWatchdog.cpp:36-84,
                           QE/cuda_mapd_rt.cu:105-135; the reference counts cycles, this counts the 100 MHz
                           s_memrealtime clock) */
  uint32_t reserved_;
} hdk_hip_kernel_options;
#define HDK_HIP_LAUNCH_CHECK_INTERRUPT 64u     /* poll the device's interrupt flag (hdk_hip_set_interrupt) once per
                                                  tile and stop with ERR_INTERRUPTED when it is set
                                                  (check_interrupt, QE/cuda_mapd_rt.cu:137-148) */
#define HDK_HIP_LAUNCH_INIT_OUTPUT 128u        /* row-wise group-by plans: the output buffer is NOT initialised yet -- the
                                                  launch writes the image hdk_hip_init_group_by_buffer would have (EMPTY
                                                  keys, INIT_AGG_VALS) itself.  The radix-partitioned group-by builds each
                                                  region's image in LDS instead of loading it and stores every region, which
                                                  saves one write and one read of the whole table (a 200 M-entry table is
                                                  3.2 GB); every other strategy simply runs the init kernel first.  Never
                                                  set it when launching into a buffer that holds an earlier launch's groups */
#define HDK_HIP_LAUNCH_CLUSTER_PROBES 256u      /* aggregate plans with ONE inner one-to-one perfect-hash join over 8-byte outer
                                                  columns: permute the outer rows by join-key range first (scan_cluster.h), so
                                                  that the probes of consecutive rows share a slice of the table that stays in L2
                                                  instead of costing one 128-byte memory line each.  Off unless asked for: with
                                                  the batched interpreter as the consumer the pre-pass (32 B/row of traffic)
                                                  costs more than the locality returns (C3: 2.2 + 4.2 ms against 5.0 ms) */
#define HDK_HIP_LAUNCH_NO_CLUSTER_PROBES 512u   /* overrides the flag above */
#define HDK_HIP_LAUNCH_ACCUMULATE 2048u         /* hdk_hip_aggregate_from_ranks: GROUPBY_BUF[0] already holds an owner table (an
                                                  earlier CHUNK of the same exchange put it there): merge this call's tuples
                                                  into it instead of writing the table completely -- what lets a rank's rows be
                                                  exchanged in chunks, scatter of chunk k + 1 beside the all-to-all of chunk k
                                                  beside the aggregation of chunk k - 1 */
#define HDK_HIP_LAUNCH_WIDE_TUPLES 1024u       /* multi-pass strategies: do NOT narrow tuples from the column statistics (8-byte
                                                  tuples of the radix-partitioned group-by / the tuple exchange).  The
                                                  statistics are per rank: ranks of one exchange must agree on the tuple
                                                  width, so a rank whose columns would allow 8 bytes sets this when another
                                                  rank's do not (hdk_amd/distributed.py: TupleExchange) */
#define HDK_HIP_LAUNCH_FORCE_PARTITIONED 16u   /* take the radix-partitioned group-by whenever the plan shape
                                                  allows it, whatever the table size (testing) */
#define HDK_HIP_LAUNCH_PLAN_RESIDENT 32u       /* the head of `workspace` already holds this plan (an earlier
                                                  launch with the same workspace put it there): skip the
                                                  host-to-device copy -- what makes a launch capturable as
                                                  pure kernel nodes (hdk_hip_graph_*) */
#define HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS 1u /* skip the LDS-privatised strategy (testing) */
#define HDK_HIP_LAUNCH_FORCE_GENERIC 4u        /* use the (batched) plan-interpreter kernel even when a
                                                  specialised kernel matches (testing) */
#define HDK_HIP_LAUNCH_FORCE_SCALAR 8u         /* use the row-at-a-time interpreter kernel (testing) */
#define HDK_HIP_LAUNCH_RECORD_EVENTS 2u        /* bracket the scan kernel with HIP events on the launch
                                                  stream (DeviceClock, QE/DeviceKernel.cpp:25-43) */

/* Runtime interrupt (Executor::interrupt -> the `runtime_interrupt_flag` of the GPU module, QE/GpuInterrupt.cpp,
 * QE/cuda_mapd_rt.cu:137-148): value != 0 makes every running and future launch on `device_id` that was started with
 * HDK_HIP_LAUNCH_CHECK_INTERRUPT record ERR_INTERRUPTED and stop at its next tile; 0 re-arms (what
 * DeviceKernel::initializeRuntimeInterrupter does before a launch).  Written on a stream of its own, so it overtakes
 * the kernels it is meant to stop. */
int32_t hdk_hip_set_interrupt(int32_t device_id, int32_t value);

/* Host-only check of a plan: ABI version, every count, index, width and enumerator a kernel or a matcher reads
 * (column widths and tables, leaf column indices, expression steps, comparison / join / aggregate kinds, slot
 * offsets inside the row, the filter program).  Every entry point that takes a plan runs it first; a malformed plan
 * gets HDK_HIP_ERR_INVALID_ARG (or _UNSUPPORTED) and a message, never an out-of-bounds read.  No device is touched.
 * layout_only != 0: only what describes the OUTPUT buffer (query kind, keys' count and width, targets' aggregates and
 * slots, entry count) -- what hdk_hip_reduce_buffers, hdk_hip_partition_baseline* and hdk_hip_baseline_table_quads
 * need and check; a caller that reduces buffers may leave columns, filters, joins and expressions unset, as the
 * reference's ResultSetReduction works from the QueryMemoryDescriptor alone. */
int32_t hdk_hip_validate_plan(const hdk_hip_plan* plan, int32_t layout_only);
/* Bytes of device scratch `hdk_hip_launch` needs for this plan (per-block partial tables). */
int32_t hdk_hip_workspace_size(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko,
                               int32_t device_id, size_t* bytes);
/* Enqueue the scan -> filter -> join-probe -> group/aggregate pass over all fragments described by
 * `params` on `stream` (NULL = the manager's stream for `device_id`).  Asynchronous: results and
 * error codes are complete when the stream is.  The output buffer must already be initialised
 * (hdk_hip_init_*_group_by_buffer, or INIT_AGG_VALS copies for non-grouped out slots). */
int32_t hdk_hip_launch(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT],
                       const hdk_hip_kernel_options* ko, int32_t device_id, void* stream,
                       void* workspace, size_t workspace_bytes);
/* hipGraph capture of a launch sequence (no reference counterpart: the reference launches kernel by kernel
 * through cuLaunchKernel, CudaMgr/DeviceKernel.cpp).  For callers that issue one small step at a time and
 * wait for it, the step is launch latency; recording the sequence removes the per-kernel submission.
 * (Measured here with back-to-back asynchronous steps of 1 M rows the replay is NOT faster -- 25.8 vs
 * 20.7 us, scripts/small_query_latency.py -- so the Python executor does not use it by default.)  Record the
 * sequence once and replay it:
 *     hdk_hip_graph_begin_capture(dev, stream);
 *       hdk_hip_init_*_group_by_buffer(..., stream);
 *       hdk_hip_launch(plan, params, ko with HDK_HIP_LAUNCH_PLAN_RESIDENT, dev, stream, ws, ws_bytes);
 *     hdk_hip_graph_end_capture(dev, stream, &graph);       then per execution:  hdk_hip_graph_launch(graph, dev, stream)
 * Only sequences made of kernels are capturable: the LDS-strategy launches (perfect hash / non-grouped) with
 * PLAN_RESIDENT and without RECORD_EVENTS.  `stream` NULL = the manager's stream. */
int32_t hdk_hip_graph_begin_capture(int32_t device_id, void* stream);
int32_t hdk_hip_graph_end_capture(int32_t device_id, void* stream, void** graph_exec);
int32_t hdk_hip_graph_launch(void* graph_exec, int32_t device_id, void* stream);
int32_t hdk_hip_graph_destroy(void* graph_exec);

/* Elapsed milliseconds of every scan kernel launched with HDK_HIP_LAUNCH_RECORD_EVENTS on
 * `device_id` since the last call (waits for them); at most `capacity` values are written, `*count`
 * receives how many there were. */
int32_t hdk_hip_collect_scan_times(int32_t device_id, float* ms_out, int32_t capacity, int32_t* count);
/* Names of the device kernels a launch of `plan` dispatches (for profiling), comma separated.
 * device_id HDK_HIP_DEVICE_ASSUMED_MI355X: answered for an MI355X (256 CUs, 160 KB of LDS per block) without touching
 * a device -- routing is host arithmetic, and a CPU-only test-suite can pin it. */
#define HDK_HIP_DEVICE_ASSUMED_MI355X (-355)
int32_t hdk_hip_describe_launch(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko,
                                int32_t device_id, char* out, size_t out_len);

/* ------------------------------------------------------------------------------------------
 * Partial-result reduction on device.
 * Replaces the host-side ResultSetReduction::reduce (QE/ResultSetReduction.cpp:174-330:
 * perfect hash = slot-wise reduceOneSlot :1234-1330; baseline = reduceOneEntryBaseline :694-731)
 * for per-GPU / per-launch partial buffers that already sit in HBM (e.g. after an RCCL
 * all-gather).  `that_bufs[i]` are merged into `this_buf`; layouts come from `plan`.
 * For HDK_Q_BASELINE_HASH `this_entry_count` may exceed `plan->entry_count` (Execute.cpp:1241-1253), and `this_buf`
 * may be a fresh (initialised) table or the output of ANY launch of this library: every kernel, the
 * radix-partitioned group-by included, leaves a group on the reference's probe sequence
 * (key_hash % entry_count, then linearly on), which is where the re-insert looks for it
 * (tests/test_gpu_baseline.py::test_reduce_and_relaunch_into_a_partitioned_table).
 * ---------------------------------------------------------------------------------------- */
int32_t hdk_hip_reduce_buffers(const hdk_hip_plan* plan, int64_t* this_buf, uint32_t this_entry_count,
                               const int64_t* const* that_bufs, const uint32_t* that_entry_counts,
                               int32_t num_that, const int64_t* init_vals, int32_t* dev_error,
                               int32_t device_id, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU exchange step of a baseline-hash group-by.  No reference counterpart:
 * the reference copies every device's partial ResultSet to the host and re-inserts the entries
 * there (Executor::reduceMultiDeviceResultSets, QE/Execute.cpp:1224-1336, with
 * reduceOneEntryBaseline, QE/ResultSetReduction.cpp:694-731).  Here the non-empty entries of a
 * rank's table are split by owner = mulhi32(key_hash(key), num_owners) into `num_owners` compact
 * tables that keep the plan's layout (row-wise or columnar) with entry_count = counts[o]; after an
 * all-to-all each owner folds what it received with hdk_hip_reduce_buffers, and the result is the
 * concatenation of the owners' disjoint tables.
 *   hdk_hip_baseline_table_quads     size (int64 words) of a table of `entry_count` entries
 *   hdk_hip_partition_baseline_count entries per owner -> HOST array counts[num_owners] (synchronises)
 *   hdk_hip_partition_baseline       scatter into seg_bufs[o] (HOST array of device pointers, each
 *                                    hdk_hip_baseline_table_quads(counts[o]) words); `counts` is the
 *                                    HOST array the count step returned
 * `init_vals` is the HOST array of hdk_hip_reduce_buffers (keyless tables cannot be baseline, but the
 * emptiness rule is shared).
 * ---------------------------------------------------------------------------------------- */
int32_t hdk_hip_baseline_table_quads(const hdk_hip_plan* plan, uint32_t entry_count, int64_t* quads);
int32_t hdk_hip_partition_baseline_count(const hdk_hip_plan* plan, const int64_t* buf, uint32_t entry_count,
                                         const int64_t* init_vals, int32_t num_owners, uint32_t* counts,
                                         int32_t device_id, void* stream);
int32_t hdk_hip_partition_baseline(const hdk_hip_plan* plan, const int64_t* buf, uint32_t entry_count,
                                   const int64_t* init_vals, int32_t num_owners, const uint32_t* counts,
                                   int64_t* const* seg_bufs, int32_t device_id, void* stream);

/* ------------------------------------------------------------------------------------------
 * Environment switches (MI355X addition; no reference counterpart: the reference's knobs are Config fields,
 * Shared/Config.h).  libhdk_hip.so reads its HDK_HIP_* variables (tests and A/B measurements, none needed
 * in production) ONCE per process, at the first launch that asks for one -- never per launch, so a host that calls
 * setenv() while queries run races with nothing.  A caller that changes them on purpose (the test harness) asks for a
 * re-read; not to be called while another thread is inside a launch.
 * ---------------------------------------------------------------------------------------- */
void hdk_hip_reload_switches(void);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU open-addressing group-by by TUPLE exchange (MI355X addition; replaces, for plans of the
 * radix-partitioned shape, per-device tables + host merge: Executor::reduceMultiDeviceResultSets,
 * QE/Execute.cpp:1224-1336, reduceOneEntryBaseline, QE/ResultSetReduction.cpp:694-731).  Keys are split by owner =
 * mulhi32(key_hash(key), num_owners); owner o ends with an open-addressing table of `owner_entry_count` entries in
 * the plan's layout holding exactly its keys, each group where the reference's probe sequence
 * (key_hash % owner_entry_count, then linearly on) finds it; the query result is the concatenation of the owners'
 * tables.  One step of rank r:
 *     hdk_hip_exchange_shape_for(plan, ko, G, owner_entry_count, &shape)      the same on every rank (host only)
 *     hdk_hip_scatter_to_owners(plan, params, ko, &shape, send, ...)          pass 1 over the rank's fragments:
 *                                                                              (filtered) rows -> tuples -> G segments
 *     all-to-all with EQUAL splits of shape.segment_bytes (RCCL over xGMI; segment o of `send` goes to rank o and
 *                                                          lands as segment r of its `recv`) -- sizes are static,
 *                                                          no counts are exchanged and the host never waits
 *     hdk_hip_aggregate_from_ranks(plan, params, ko, &shape, recv, ...)       passes 2-4 over the G received segments
 *                                                                              into GROUPBY_BUF[0] (the owner's table,
 *                                                                              initialised by the call itself)
 * Tuples are as narrow as the plan's column statistics allow (8 bytes when key and argument fit 32 bits each, else
 * 16 / 24).  `ko->total_rows` must be the SAME upper bound on a rank's outer rows on every rank (it sizes the
 * segments).  Skewed keys (one owner sub-slab overflowing), stale statistics or an interrupt flag the segments; the
 * owner's call then leaves ERROR_CODE = HDK_HIP_ERR_EXCHANGE_INCOMPLETE and the caller redoes the step with partial
 * tables (hdk_hip_launch + hdk_hip_partition_baseline + hdk_hip_reduce_buffers), which has no such limit.
 * Plans outside the radix-partitioned shape (hdk_scan_agg_baseline_direct's: row-wise, 1-2 plain integer key columns,
 * plain-column arguments, `column cmp literal` filters) get HDK_HIP_ERR_UNSUPPORTED from shape_for.
 * `stream` of the two calls must be the stream the all-to-all runs on, or be ordered with it by the caller: NULL means the
 * manager's own stream, which is ordered neither with the legacy default stream nor with a communicator's.
 * ---------------------------------------------------------------------------------------- */
typedef struct hdk_hip_exchange_shape {
  uint32_t num_owners;         /* G, 1 ... 32 (one owner: a rank exchanging with itself -- every kernel and the collective run,
                                  which is how a one-GPU box exercises the RCCL all-to-all) */
  uint32_t owner_entry_count;  /* entries of every owner's table */
  uint32_t tuple_bytes;        /* 8, 16 or 24 */
  uint32_t coarse_per_owner;   /* level-1 bins per owner (G x this <= 256) */
  uint32_t regions_log2;       /* regions of the owner's table per coarse slab = 1 << this */
  uint32_t reserved_;
  uint64_t sub_slab_tuples;    /* capacity of one (coarse slab, XCD) sub-slab */
  uint64_t segment_header_bytes; /* header: uint32 tuples per (coarse slab, XCD) sub-slab, a flag word (0 = complete) and a
                                    tag word (tuple bytes | coarse_per_owner << 8) that the owner checks against its own
                                    shape: a sender that chose another tuple width makes the exchange INCOMPLETE */
  uint64_t segment_bytes;      /* one rank -> owner segment: header + coarse_per_owner x 8 sub-slabs; `send` and `recv`
                                  are num_owners segments each */
  uint64_t rows_bound;         /* ko->total_rows the shape was made for */
  uint64_t scatter_workspace_bytes;   /* `workspace` of hdk_hip_scatter_to_owners */
  uint64_t aggregate_workspace_bytes; /* `workspace` of hdk_hip_aggregate_from_ranks */
} hdk_hip_exchange_shape;
int32_t hdk_hip_exchange_shape_for(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko, int32_t num_owners,
                                   uint32_t owner_entry_count, int32_t device_id, hdk_hip_exchange_shape* shape);
/* `params`: the 12 launch pointers of hdk_hip_launch (GROUPBY_BUF unused); `send`: num_owners x segment_bytes of device
 * memory, 256-byte aligned.  Honours HDK_HIP_LAUNCH_CHECK_INTERRUPT / watchdog_ms / RECORD_EVENTS of `ko`. */
int32_t hdk_hip_scatter_to_owners(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT],
                                  const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape, int8_t* send,
                                  int32_t device_id, void* stream, void* workspace, size_t workspace_bytes);
/* `params`: GROUPBY_BUF[0] = the owner's table (hdk_hip_baseline_table_quads(plan, owner_entry_count) words, written
 * completely: no initialisation needed), INIT_AGG_VALS and ERROR_CODE as for hdk_hip_launch; the other entries are
 * not read.  `recv`: the num_owners segments this owner received, in rank order. */
int32_t hdk_hip_aggregate_from_ranks(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT],
                                     const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape,
                                     const int8_t* recv, int32_t device_id, void* stream, void* workspace,
                                     size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * Hash-join table build (perfect hash).
 * Replaces the *_on_device free functions of QE/JoinHashTable/Runtime/HashJoinRuntime.h:66-68,
 * 158-200 (GPU bodies QE/JoinHashTable/Runtime/HashJoinRuntimeGpu.cu:32-190).  The structs carry the same fields
 * as the reference's PODs (HashJoinRuntime.h:43-57,100-124) but are NOT layout-compatible with them: fixed-width
 * members, `bool` widened to int32, and hdk_hip_join_column_type_info keeps column_type in front of
 * translated_null_val (the reference: `translated_null_val, uses_bw_eq, column_type`).  Convert field by field
 * (hdk_amd/glue/HipRuntimeOnDevice.h: to_abi / to_abi_column), never by memcpy.
 * ---------------------------------------------------------------------------------------- */
typedef struct hdk_hip_join_chunk {  /* JoinChunk */
  const int8_t* col_buff;
  size_t num_elems;
  size_t row_id;
} hdk_hip_join_chunk;
typedef struct hdk_hip_join_column { /* JoinColumn */
  const int8_t* col_chunks_buff;     /* device array of hdk_hip_join_chunk */
  size_t col_chunks_buff_sz;
  size_t num_chunks;
  size_t num_elems;
  size_t elem_sz;
} hdk_hip_join_column;
enum hdk_hip_column_type { HDK_JC_SMALL_DATE = 0, HDK_JC_SIGNED = 1, HDK_JC_UNSIGNED = 2, HDK_JC_DOUBLE = 3 };
typedef struct hdk_hip_join_column_type_info { /* JoinColumnTypeInfo */
  size_t elem_sz;
  int64_t min_val;
  int64_t max_val;
  int64_t null_val;
  int32_t uses_bw_eq;
  int32_t column_type; /* hdk_hip_column_type */
  int64_t translated_null_val;
} hdk_hip_join_column_type_info;
typedef struct hdk_hip_hash_entry_info { /* HashEntryInfo */
  size_t hash_entry_count;
  int64_t bucket_normalization;
} hdk_hip_hash_entry_info;

/* init_hash_join_buff_on_device (HashJoinRuntime.h:66-68) */
int32_t hdk_hip_init_hash_join_buff(int32_t* buff, int64_t entry_count, int32_t invalid_slot_val,
                                    int32_t device_id, void* stream);
/* fill_hash_join_buff_on_device[_bucketized] (HashJoinRuntime.h:158-171): one-to-one; on a duplicate
 * key `*dev_err_buff` becomes -1 (the caller then rebuilds one-to-many, PerfectHashTableBuilder.h:134-141).
 * ONE call fills ONE table: from 2 M rows on (and a table of at most 16 slots per row) the fill goes through slot-range
 * partitions whose build pass writes EVERY slot of `buff` (the claimed row id, or `invalid_slot_val`), so entries of an
 * earlier call into the same buffer do not survive -- the reference's callers (PerfectHashTableBuilder::initOneToOneHashTable
 * OnGpu) hand the whole JoinColumn to one call as well. */
int32_t hdk_hip_fill_hash_join_buff(int32_t* buff, int32_t invalid_slot_val, int32_t for_semi_join,
                                    int32_t* dev_err_buff, hdk_hip_join_column join_column,
                                    hdk_hip_join_column_type_info type_info, int32_t device_id,
                                    void* stream);
int32_t hdk_hip_fill_hash_join_buff_bucketized(int32_t* buff, int32_t invalid_slot_val,
                                               int32_t for_semi_join, int32_t* dev_err_buff,
                                               hdk_hip_join_column join_column,
                                               hdk_hip_join_column_type_info type_info,
                                               int64_t bucket_normalization, int32_t device_id,
                                               void* stream);
/* fill_one_to_many_hash_table_on_device[_bucketized] (HashJoinRuntime.h:188-200): layout
 * [pos | count | row ids] of int32 (Builders/PerfectHashTableBuilder.h:35-38). `buff` must have been
 * initialised with hdk_hip_init_hash_join_buff over 2*entries + num_elems slots' first 2*entries. */
int32_t hdk_hip_fill_one_to_many_hash_table(int32_t* buff, hdk_hip_hash_entry_info hash_entry_info,
                                            int32_t invalid_slot_val, hdk_hip_join_column join_column,
                                            hdk_hip_join_column_type_info type_info, int32_t device_id,
                                            void* stream);
int32_t hdk_hip_fill_one_to_many_hash_table_bucketized(int32_t* buff,
                                                       hdk_hip_hash_entry_info hash_entry_info,
                                                       int32_t invalid_slot_val,
                                                       hdk_hip_join_column join_column,
                                                       hdk_hip_join_column_type_info type_info,
                                                       int32_t device_id, void* stream);

/* Keyed ("baseline") join tables for composite or wide keys -- the *_on_device_{32,64} functions of
 * HashJoinRuntime.h:181-204,225-281 (GPU bodies HashJoinRuntimeGpu.cu:192-330), with the
 * GenericKeyHandler (HashJoinKeyHandlers.h:36-100) passed as its two arrays: `join_column_per_key`
 * and `type_info_per_key` are HOST arrays of `key_component_count` entries; `key_component_width`
 * (4 or 8) selects the _32 / _64 form.  A row with a NULL component is skipped.
 *   init : keys = EMPTY_KEY, payload (with_val_slot) = invalid_slot_val
 *   fill : one-to-one table (with_val_slot=1; *dev_err_buff becomes -1 on a duplicate key, -2 when
 *          the table is full) or just the composite-key dictionary (with_val_slot=0)
 *   fill_one_to_many : `buff` = int32 [offsets | counts | row ids] right after the dictionary
 *          `composite_key_dict` (which fill with_val_slot=0 built) */
int32_t hdk_hip_init_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count, size_t key_component_count,
                                             int32_t key_component_width, int32_t with_val_slot,
                                             int32_t invalid_slot_val, int32_t device_id, void* stream);
int32_t hdk_hip_fill_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                             int32_t for_semi_join, size_t key_component_count,
                                             int32_t key_component_width, int32_t with_val_slot,
                                             int32_t* dev_err_buff, const hdk_hip_join_column* join_column_per_key,
                                             const hdk_hip_join_column_type_info* type_info_per_key,
                                             int32_t device_id, void* stream);
int32_t hdk_hip_fill_one_to_many_baseline_hash_table(int32_t* buff, const int8_t* composite_key_dict,
                                                     int64_t hash_entry_count, int32_t invalid_slot_val,
                                                     size_t key_component_count, int32_t key_component_width,
                                                     const hdk_hip_join_column* join_column_per_key,
                                                     const hdk_hip_join_column_type_info* type_info_per_key,
                                                     int32_t device_id, void* stream);

/* Build a fused join table (HDK_JOIN_ONE_TO_ONE_FUSED) from a one-to-one table:
 *   out[slot * (1 + ncols) + 0]     = table[slot]                      (row id or invalid)
 *   out[slot * (1 + ncols) + 1 + c] = decode(inner_cols[c][row id])     (0 when the slot is empty)
 * `inner_cols` / `widths` / `kinds` are HOST arrays of ncols device pointers / element widths /
 * hdk_hip_col_kind; inner columns must be linearised (one buffer per column). */
int32_t hdk_hip_build_fused_join_table(const int32_t* table, int64_t entry_count, const int8_t* const* inner_cols,
                                       const int32_t* widths, const int32_t* kinds, int32_t ncols, int64_t* out,
                                       int32_t device_id, void* stream);

/* A one-to-one table AND its fused form in one sweep over the inner rows (MI355X addition; the table is what
 * fill_hash_join_buff_on_device_bucketized leaves, QE/JoinHashTable/Runtime/HashJoinRuntimeGpu.cu:57-75, the fused form
 * what hdk_hip_build_fused_join_table derives from it).  From a few million rows on, and unless the table is built for a
 * semi join, the rows are PARTITIONED by slot range (one or two radix levels) and each 32 768-slot slice is built in LDS
 * and written front to back, payloads travelling with the rows: no random atomics (1e8 of them cost 4.4 ms), no gather
 * through the row id (2.9 ms per 1e8 slots).  hdk_hip_fill_hash_join_buff[_bucketized] take the same route for the table
 * alone.  A taken slot is -1 and a key outside [min, max] -2 in *dev_err_buff, as there; `buff` must have been initialised
 * (hdk_hip_init_hash_join_buff).  `scratch`: hdk_hip_join_build_scratch_bytes(rows, slots, ncols) bytes of device memory
 * (0: this table is not partitioned, none needed), or NULL -- the library then takes it from the stream's memory pool
 * (hipMallocAsync).  HDK_HIP_BUILD_PARTITION_MIN_ROWS overrides the threshold (0 = never partition). */
size_t hdk_hip_join_build_scratch_bytes(int64_t num_rows, int64_t entry_count, int32_t ncols);
int32_t hdk_hip_fill_hash_join_buff_fused(int32_t* buff, int32_t invalid_slot_val, int32_t for_semi_join, int32_t* dev_err_buff,
                                          hdk_hip_join_column join_column, hdk_hip_join_column_type_info type_info,
                                          int64_t bucket_normalization, const int8_t* const* inner_cols, const int32_t* widths,
                                          const int32_t* kinds, int32_t ncols, int64_t* fused_out, void* scratch,
                                          size_t scratch_bytes, int32_t device_id, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HDK_HIP_H */
