/*
 * hdk_oracle.h -- CPU restatement of HDK's per-row runtime for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker.
 *
 * Parity status: PINNED.  The primitives below are checked (tests/test_oracle_vs_ref.py) against
 * the reference's own runtime compiled from /root/reference (oracle/_ref/libhdk_ref_runtime.so,
 * recipe oracle/Makefile) when that tree is present, and (tests/test_oracle_golden.py) against
 * golden vectors generated from it (tests/golden/, generator tests/golden/gen_golden.py) plus the
 * literal known-answer tests of the reference's own test-suite (GroupByHashTest.cpp,
 * NoCatalogRelAlgTest.cpp, JoinHashTableTest.cpp, taxi Q1-Q4).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root;
 * QE/ = omniscidb/QueryEngine/).
 */
#ifndef HDK_ORACLE_H
#define HDK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/hdk_hip.h" /* the plan POD + constants (interface definition only) */

#ifdef __cplusplus
extern "C" {
#endif

/* hashes: QE/MurmurHash3Inl.h:11-76, QE/MurmurHash1Inl.h:6-52 */
uint32_t orc_murmur_hash3(const void* key, int len, uint32_t seed);
uint32_t orc_murmur_hash1(const void* key, int len, uint32_t seed);
uint64_t orc_murmur_hash64a(const void* key, int len, uint64_t seed);
/* QE/GroupByRuntime.cpp:24-29 */
uint32_t orc_key_hash(const int64_t* key, uint32_t key_count, uint32_t key_byte_width);

/* decoders: QE/DecodersImpl.h:30-150 */
int64_t orc_fixed_width_int_decode(const int8_t* byte_stream, int32_t byte_width, int64_t pos);
int64_t orc_fixed_width_unsigned_decode(const int8_t* byte_stream, int32_t byte_width, int64_t pos);
int64_t orc_fixed_width_small_date_decode(const int8_t* byte_stream, int32_t byte_width, int32_t null_val,
                                          int64_t ret_null_val, int64_t pos); /* QE/DecodersImpl.h:151-159 */
float orc_fixed_width_float_decode(const int8_t* byte_stream, int64_t pos);
double orc_fixed_width_double_decode(const int8_t* byte_stream, int64_t pos);

/* group lookup: QE/GroupByRuntime.cpp:31-246, QE/RuntimeFunctions.cpp:1209-1406 */
int64_t* orc_get_matching_group_value(int64_t* groups_buffer, uint32_t h, const int64_t* key,
                                      uint32_t key_count, uint32_t key_width, uint32_t row_size_quad);
int64_t* orc_get_group_value(int64_t* groups_buffer, uint32_t groups_buffer_entry_count,
                             const int64_t* key, uint32_t key_count, uint32_t key_width,
                             uint32_t row_size_quad);
int32_t orc_get_matching_group_value_columnar_slot(int64_t* groups_buffer, uint32_t entry_count,
                                                   uint32_t h, const int64_t* key, uint32_t key_count,
                                                   uint32_t key_width);
int32_t orc_get_group_value_columnar_slot(int64_t* groups_buffer, uint32_t groups_buffer_entry_count,
                                          const int64_t* key, uint32_t key_count, uint32_t key_width);
int64_t* orc_get_group_value_columnar(int64_t* groups_buffer, uint32_t groups_buffer_entry_count,
                                      const int64_t* key, uint32_t key_qw_count);
int64_t* orc_get_group_value_fast(int64_t* groups_buffer, int64_t key, int64_t min_key, int64_t bucket,
                                  uint32_t row_size_quad);
int64_t* orc_get_group_value_fast_keyless(int64_t* groups_buffer, int64_t key, int64_t min_key,
                                          int64_t bucket, uint32_t row_size_quad);
uint32_t orc_get_columnar_group_bin_offset(int64_t* key_base_ptr, int64_t key, int64_t min_key,
                                           int64_t bucket);
int64_t* orc_get_matching_group_value_perfect_hash(int64_t* groups_buffer, uint32_t hashed_index,
                                                   const int64_t* key, uint32_t key_count,
                                                   uint32_t row_size_quad);
int64_t* orc_get_matching_group_value_perfect_hash_keyless(int64_t* groups_buffer,
                                                           uint32_t hashed_index,
                                                           uint32_t row_size_quad);
void orc_set_matching_group_value_perfect_hash_columnar(int64_t* groups_buffer, uint32_t hashed_index,
                                                        const int64_t* key, uint32_t key_count,
                                                        uint32_t entry_count);

/* aggregates: QE/RuntimeFunctions.cpp:387-391,456-476,528-538,612-875 */
uint64_t orc_agg_count(uint64_t* agg, int64_t val);
uint32_t orc_agg_count_int32(uint32_t* agg, int32_t val);
int64_t orc_agg_sum(int64_t* agg, int64_t val);
int32_t orc_agg_sum_int32(int32_t* agg, int32_t val);
void orc_agg_max(int64_t* agg, int64_t val);
void orc_agg_min(int64_t* agg, int64_t val);
void orc_agg_max_int32(int32_t* agg, int32_t val);
void orc_agg_min_int32(int32_t* agg, int32_t val);
int64_t orc_agg_sum_skip_val(int64_t* agg, int64_t val, int64_t skip_val);
int32_t orc_agg_sum_int32_skip_val(int32_t* agg, int32_t val, int32_t skip_val);
uint64_t orc_agg_count_skip_val(uint64_t* agg, int64_t val, int64_t skip_val);
uint32_t orc_agg_count_int32_skip_val(uint32_t* agg, int32_t val, int32_t skip_val);
void orc_agg_max_skip_val(int64_t* agg, int64_t val, int64_t skip_val);
void orc_agg_min_skip_val(int64_t* agg, int64_t val, int64_t skip_val);
/* checked_single_agg_id[_int32|_double|_float] (QE/RuntimeFunctions.cpp:489-506,567-583,743-760,799-816): 0 or 15 */
int32_t orc_checked_single_agg_id(int64_t* agg, int64_t val, int64_t null_val);
int32_t orc_checked_single_agg_id_int32(int32_t* agg, int32_t val, int32_t null_val);
int32_t orc_checked_single_agg_id_double(int64_t* agg, double val, double null_val);
int32_t orc_checked_single_agg_id_float(int32_t* agg, float val, float null_val);
void orc_agg_max_int32_skip_val(int32_t* agg, int32_t val, int32_t skip_val);
void orc_agg_min_int32_skip_val(int32_t* agg, int32_t val, int32_t skip_val);
uint64_t orc_agg_count_double(uint64_t* agg, double val);
void orc_agg_sum_double(int64_t* agg, double val);
void orc_agg_max_double(int64_t* agg, double val);
void orc_agg_min_double(int64_t* agg, double val);
uint64_t orc_agg_count_double_skip_val(uint64_t* agg, double val, double skip_val);
void orc_agg_sum_double_skip_val(int64_t* agg, double val, double skip_val);
void orc_agg_max_double_skip_val(int64_t* agg, double val, double skip_val);
void orc_agg_min_double_skip_val(int64_t* agg, double val, double skip_val);
void orc_agg_sum_float(int32_t* agg, float val);
void orc_agg_sum_float_skip_val(int32_t* agg, float val, float skip_val);

/* scalar helpers: QE/RuntimeFunctions.cpp:49-384; omniscidb/Utils/ExtractFromTime.cpp:156-272 */
int64_t orc_scale_decimal_down_nullable(int64_t operand, int64_t scale, int64_t null_val);
int64_t orc_scale_decimal_down_not_nullable(int64_t operand, int64_t scale, int64_t null_val);
int64_t orc_floor_div_lhs(int64_t dividend, int64_t divisor);
int64_t orc_floor_div_nullable_lhs(int64_t dividend, int64_t divisor, int64_t null_val);
int64_t orc_extract_year(int64_t timeval);
int8_t orc_logical_and(int8_t lhs, int8_t rhs, int8_t null_val);
int8_t orc_logical_or(int8_t lhs, int8_t rhs, int8_t null_val);
int8_t orc_logical_not(int8_t operand, int8_t null_val);

/* join probe: QE/GroupByRuntime.cpp:274-366 */
int64_t orc_hash_join_idx(const int32_t* hash_buff, int64_t key, int64_t min_key, int64_t max_key);
int64_t orc_bucketized_hash_join_idx(const int32_t* hash_buff, int64_t key, int64_t min_key,
                                     int64_t max_key, int64_t bucket_normalization);
int64_t orc_hash_join_idx_nullable(const int32_t* hash_buff, int64_t key, int64_t min_key,
                                   int64_t max_key, int64_t null_val);
int64_t orc_bucketized_hash_join_idx_nullable(const int32_t* hash_buff, int64_t key, int64_t min_key, int64_t max_key,
                                              int64_t null_val, int64_t bucket_normalization);
int64_t orc_bucketized_hash_join_idx_bitwise(const int32_t* hash_buff, int64_t key, int64_t min_key, int64_t max_key,
                                             int64_t null_val, int64_t translated_val, int64_t bucket_normalization);
int64_t orc_hash_join_idx_bitwise(const int32_t* hash_buff, int64_t key, int64_t min_key,
                                  int64_t max_key, int64_t null_val, int64_t translated_val);

/* join build: QE/JoinHashTable/Runtime/HashJoinRuntime.cpp:127-147,197-293,589-853,1140-1190;
 * JoinHashImpl.h:55-97.  Host pointers; chunks are walked in order (cpu_thread_count = 1). */
void orc_init_hash_join_buff(int32_t* buff, int64_t entry_count, int32_t invalid_slot_val);
int orc_fill_hash_join_buff(int32_t* buff, int32_t invalid_slot_val, int32_t for_semi_join,
                            const hdk_hip_join_chunk* chunks, size_t num_chunks,
                            const hdk_hip_join_column_type_info* type_info,
                            int64_t bucket_normalization /* 0 or 1 => not bucketized */);
void orc_fill_one_to_many_hash_table(int32_t* buff, int64_t hash_entry_count,
                                     int32_t invalid_slot_val, const hdk_hip_join_chunk* chunks,
                                     size_t num_chunks, const hdk_hip_join_column_type_info* type_info,
                                     int64_t bucket_normalization);

/* keyed ("baseline") join tables: JoinHashTableQueryRuntime.cpp:25-172, HashJoinRuntime.cpp:296-573,723-950 */
int64_t orc_baseline_hash_join_idx_32(const int8_t* hash_buff, const int8_t* key, size_t key_bytes,
                                      size_t entry_count);
int64_t orc_baseline_hash_join_idx_64(const int8_t* hash_buff, const int8_t* key, size_t key_bytes,
                                      size_t entry_count);
int64_t orc_get_composite_key_index_32(const int32_t* key, size_t key_component_count,
                                       const int32_t* composite_key_dict, size_t entry_count);
int64_t orc_get_composite_key_index_64(const int64_t* key, size_t key_component_count,
                                       const int64_t* composite_key_dict, size_t entry_count);
void orc_init_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count, size_t key_component_count,
                                      int32_t key_component_width, int32_t with_val_slot,
                                      int32_t invalid_slot_val);
int orc_fill_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                     size_t key_component_count, int32_t key_component_width,
                                     const hdk_hip_join_column* cols, const hdk_hip_join_column_type_info* ti);
int orc_fill_baseline_hash_join_buff_semi(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                          int32_t for_semi_join, size_t key_component_count,
                                          int32_t key_component_width, const hdk_hip_join_column* cols,
                                          const hdk_hip_join_column_type_info* ti);
int orc_fill_one_to_many_baseline_hash_table(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                             size_t key_component_count, int32_t key_component_width,
                                             const hdk_hip_join_column* cols,
                                             const hdk_hip_join_column_type_info* ti);

/* output buffer init: QE/GpuInitGroups.cu:17-166 (the CPU twin is
 * QE/QueryMemoryInitializer.cpp initGroupByBuffer/initColumnarGroups). */
void orc_init_group_by_buffer(int64_t* groups_buffer, const int64_t* init_vals,
                              uint32_t groups_buffer_entry_count, uint32_t key_count,
                              uint32_t key_width, uint32_t row_size_quad, int32_t keyless,
                              int8_t warp_size);
void orc_init_columnar_group_by_buffer(int64_t* groups_buffer, const int64_t* init_vals,
                                       uint32_t groups_buffer_entry_count, uint32_t key_count,
                                       uint32_t agg_col_count, const int8_t* col_sizes,
                                       int32_t need_padding, int32_t keyless, int8_t key_size);

/* The row function HDK would JIT for `plan`, run the way the CPU path runs it
 * (QE/RuntimeFunctions.cpp:1741-1768 multifrag_query -> query_group_by_template -> row_func):
 * fragments in order, rows in order, one output buffer.
 *   col_buffers[frag][buf_idx], num_rows[frag*num_tables + t]
 * group-by: `out` is the (already initialised) group-by buffer of plan->entry_count entries;
 * non-grouped: `out[i]` is the slot of target i (AVG uses two consecutive slots), pre-set to init vals.
 * Returns 0 or an HDK error code (3 = out of slots). */
int32_t orc_run_plan(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                     uint64_t num_fragments, const int64_t* num_rows, uint32_t num_tables,
                     const int64_t* join_hash_tables, int64_t* out);
/* As orc_run_plan, but for fragments [frag_begin, frag_end) only (one ExecutionKernel). */
int32_t orc_run_plan_range(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                           uint64_t frag_begin, uint64_t frag_end, const int64_t* num_rows,
                           uint32_t num_tables, const int64_t* join_hash_tables, int64_t* out);

/* Projection (filter/project) plans: rows are claimed in scan order through *total_matched
 * (QE/RowFuncBuilder.cpp:162-215, QE/GroupByRuntime.cpp:248-272). */
int32_t orc_run_projection(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                           uint64_t num_fragments, const int64_t* num_rows, uint32_t num_tables,
                           const int64_t* join_hash_tables, int64_t* out, int32_t max_matched,
                           int32_t* total_matched);

/* Partial-result reduction: QE/ResultSetReduction.cpp:174-330 (perfect hash / non-grouped:
 * entry-wise reduceOneSlot :1234-1330; baseline: reduceOneEntryBaseline :694-731 re-insert).
 * `that` is merged into `this_`.  Non-grouped buffers are the per-target slot vectors. */
int32_t orc_reduce(const hdk_hip_plan* plan, int64_t* this_buf, uint32_t this_entry_count,
                   const int64_t* that_buf, uint32_t that_entry_count, const int64_t* init_vals);

/* Whether entry `idx` of a result buffer is empty: RS/ResultSetStorage.cpp:439-521. */
int32_t orc_is_empty_entry(const hdk_hip_plan* plan, const int64_t* buf, uint32_t entry_count,
                           uint32_t idx, const int64_t* init_vals);

/* HDK-semantics CPU path for the timed baseline (bench.py cpu_baseline): one kernel per fragment
 * on a thread pool, private output buffer per kernel, then reduction of the partials
 * (QE/Execute.cpp:2776-2788 launchKernels, :1290-1317 reduceMultiDeviceResultSets).
 * `out` receives the reduced buffer; `buffer_quads` = its size in int64 words. */
int32_t orc_run_plan_parallel(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                              uint64_t num_fragments, const int64_t* num_rows, uint32_t num_tables,
                              const int64_t* join_hash_tables, const int64_t* init_buffer,
                              size_t buffer_quads, const int64_t* init_vals, int32_t num_threads,
                              int64_t* out);
/* CPU baseline only (bench.py): the row loop HDK's JIT would emit for C2, hand-inlined; see hdk_oracle.c */
double orc_c2_jit_shaped(const int64_t* const* keys, const int64_t* const* vals, const int64_t* num_rows,
                         uint64_t num_fragments, int64_t min_key, uint32_t entry_count, uint32_t row_quads,
                         int32_t keyless, int32_t key_slot, int32_t sum_slot, int32_t skip_null, int64_t null_val,
                         const int64_t* init_buffer, int32_t num_threads, int32_t first_touch, int32_t reps,
                         int64_t* out);
int32_t orc_max_threads(void);

/* placement-aware helpers of the CPU baseline (bench.py) */
int32_t orc_allowed_cpu_count(void);
double orc_host_stream_read_gbps(int32_t num_threads, size_t bytes_per_thread, int32_t reps);
/* the same figure with the threads placed one per physical core first, round robin over the NUMA nodes (hdk_oracle.c) */
double orc_host_stream_read_gbps_placed(int32_t num_threads, size_t bytes_per_thread, int32_t reps, double* per_node, int32_t max_nodes);
int32_t orc_physical_core_count(void);

#ifdef __cplusplus
}
#endif
#endif
