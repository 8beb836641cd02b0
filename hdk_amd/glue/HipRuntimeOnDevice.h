// HipRuntimeOnDevice.h -- HDK-side forwards for the free functions the executor and the join-table
// builders call with GpuMgrPlatform-style dispatch (reference QE/GpuInitGroupsImpl.cpp:86-159 switches
// on the platform; QE/JoinHashTable/Runtime/HashJoinRuntime.h:66-68,158-200 are CUDA-only today).
// Each forward keeps the reference's name and argument list and adds (device_id, stream = nullptr).
// PODs: the reference's JoinChunk/JoinColumn/JoinColumnTypeInfo/HashEntryInfo are layout-compatible
// with hdk_hip_join_* except JoinColumnTypeInfo (bool/enum members): `to_abi` converts it.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "hdk_hip.h"

namespace hip_rt {

inline void check(int32_t status) {
  if (status != HDK_HIP_OK) {
    throw std::runtime_error(std::string("hdk_hip: ") + hdk_hip_last_error());
  }
}

// init_group_by_buffer_on_device(..., GpuMgrPlatform::HIP) -> here (QE/GpuInitGroups.h:23-34)
inline void init_group_by_buffer_on_device_hip(int64_t* groups_buffer, const int64_t* init_vals,
                                               const uint32_t groups_buffer_entry_count, const uint32_t key_count,
                                               const uint32_t key_width, const uint32_t agg_col_count /*row_size_quad*/,
                                               const bool keyless, const int8_t warp_size, const size_t block_size_x,
                                               const size_t grid_size_x, const int device_id) {
  check(hdk_hip_init_group_by_buffer(groups_buffer, init_vals, groups_buffer_entry_count, key_count, key_width,
                                     agg_col_count, keyless, warp_size, block_size_x, grid_size_x, device_id, nullptr));
}

// init_columnar_group_by_buffer_on_device(..., GpuMgrPlatform::HIP) (QE/GpuInitGroups.h:36-48)
inline void init_columnar_group_by_buffer_on_device_hip(int64_t* groups_buffer, const int64_t* init_vals,
                                                        const uint32_t groups_buffer_entry_count,
                                                        const uint32_t key_count, const uint32_t agg_col_count,
                                                        const int8_t* col_sizes, const bool need_padding,
                                                        const bool keyless, const int8_t key_size,
                                                        const size_t block_size_x, const size_t grid_size_x,
                                                        const int device_id) {
  check(hdk_hip_init_columnar_group_by_buffer(groups_buffer, init_vals, groups_buffer_entry_count, key_count,
                                              agg_col_count, col_sizes, need_padding, keyless, key_size, block_size_x,
                                              grid_size_x, device_id, nullptr));
}

// JoinColumnTypeInfo (HashJoinRuntime.h:114-122) -> ABI POD
template <class RefTypeInfo>
inline hdk_hip_join_column_type_info to_abi(const RefTypeInfo& t) {
  hdk_hip_join_column_type_info o;
  o.elem_sz = t.elem_sz;
  o.min_val = t.min_val;
  o.max_val = t.max_val;
  o.null_val = t.null_val;
  o.uses_bw_eq = t.uses_bw_eq ? 1 : 0;
  o.column_type = static_cast<int32_t>(t.column_type);  // SmallDate=0, Signed=1, Unsigned=2, Double=3
  o.translated_null_val = t.translated_null_val;
  return o;
}
template <class RefJoinColumn>
inline hdk_hip_join_column to_abi_column(const RefJoinColumn& c) {
  hdk_hip_join_column o;
  o.col_chunks_buff = c.col_chunks_buff;
  o.col_chunks_buff_sz = c.col_chunks_buff_sz;
  o.num_chunks = c.num_chunks;
  o.num_elems = c.num_elems;
  o.elem_sz = c.elem_sz;
  return o;
}

inline void init_hash_join_buff_on_device(int32_t* buff, const int64_t entry_count, const int32_t invalid_slot_val,
                                          const int device_id) {
  check(hdk_hip_init_hash_join_buff(buff, entry_count, invalid_slot_val, device_id, nullptr));
}

template <class RefJoinColumn, class RefTypeInfo>
inline void fill_hash_join_buff_on_device(int32_t* buff, const int32_t invalid_slot_val, const bool for_semi_join,
                                          int* dev_err_buff, const RefJoinColumn& join_column,
                                          const RefTypeInfo& type_info, const int device_id) {
  check(hdk_hip_fill_hash_join_buff(buff, invalid_slot_val, for_semi_join, dev_err_buff, to_abi_column(join_column),
                                    to_abi(type_info), device_id, nullptr));
}

template <class RefJoinColumn, class RefTypeInfo>
inline void fill_hash_join_buff_on_device_bucketized(int32_t* buff, const int32_t invalid_slot_val,
                                                     const bool for_semi_join, int* dev_err_buff,
                                                     const RefJoinColumn& join_column, const RefTypeInfo& type_info,
                                                     const int64_t bucket_normalization, const int device_id) {
  check(hdk_hip_fill_hash_join_buff_bucketized(buff, invalid_slot_val, for_semi_join, dev_err_buff,
                                               to_abi_column(join_column), to_abi(type_info), bucket_normalization,
                                               device_id, nullptr));
}

// The table AND its fused form ([row id | payload words], HDK_JOIN_ONE_TO_ONE_FUSED) in one sweep over the inner rows: what
// PerfectJoinHashTableBuilder::initOneToOneHashTableOnGpu (Builders/PerfectHashTableBuilder.h:82-141) calls in place of
// fill_hash_join_buff_on_device_bucketized when the plan's kernels read the fused table.  `scratch` comes from the
// BufferProvider (hdk_hip_join_build_scratch_bytes(rows, slots, ncols) bytes; 0 = none needed) or is NULL (stream pool).
template <class RefJoinColumn, class RefTypeInfo>
inline void fill_hash_join_buff_fused_on_device(int32_t* buff, const int32_t invalid_slot_val, const bool for_semi_join,
                                                int* dev_err_buff, const RefJoinColumn& join_column,
                                                const RefTypeInfo& type_info, const int64_t bucket_normalization,
                                                const int8_t* const* inner_cols, const int32_t* widths, const int32_t* kinds,
                                                const int32_t ncols, int64_t* fused_out, int8_t* scratch,
                                                const size_t scratch_bytes, const int device_id) {
  check(hdk_hip_fill_hash_join_buff_fused(buff, invalid_slot_val, for_semi_join, dev_err_buff, to_abi_column(join_column),
                                          to_abi(type_info), bucket_normalization, inner_cols, widths, kinds, ncols, fused_out,
                                          scratch, scratch_bytes, device_id, nullptr));
}

template <class RefHashEntryInfo, class RefJoinColumn, class RefTypeInfo>
inline void fill_one_to_many_hash_table_on_device(int32_t* buff, const RefHashEntryInfo& hash_entry_info,
                                                  const int32_t invalid_slot_val, const RefJoinColumn& join_column,
                                                  const RefTypeInfo& type_info, const int device_id) {
  hdk_hip_hash_entry_info h{hash_entry_info.hash_entry_count, hash_entry_info.bucket_normalization};
  check(hdk_hip_fill_one_to_many_hash_table(buff, h, invalid_slot_val, to_abi_column(join_column), to_abi(type_info),
                                            device_id, nullptr));
}

// ---- keyed ("baseline") join tables: *_on_device_{32,64} of HashJoinRuntime.h:181-204,225-281 ----------
// The reference passes a device-resident GenericKeyHandler; the ABI takes its two arrays on the host
// (join_column_per_key / type_info_per_key, HashJoinKeyHandlers.h:36-50), so the HDK-side builder
// (BaselineJoinHashTableBuilder::initHashTableOnGpu) forwards the vectors it already owns.
template <int W>  // key component width in bytes: 4 -> *_32, 8 -> *_64
inline void init_baseline_hash_join_buff_on_device(int8_t* hash_join_buff, const int64_t entry_count,
                                                   const size_t key_component_count, const bool with_val_slot,
                                                   const int32_t invalid_slot_val, const int device_id) {
  check(hdk_hip_init_baseline_hash_join_buff(hash_join_buff, entry_count, key_component_count, W, with_val_slot,
                                             invalid_slot_val, device_id, nullptr));
}

template <int W, class RefJoinColumn, class RefTypeInfo>
inline void fill_baseline_hash_join_buff_on_device(int8_t* hash_buff, const int64_t entry_count,
                                                   const int32_t invalid_slot_val, const bool for_semi_join,
                                                   const size_t key_component_count, const bool with_val_slot,
                                                   int* dev_err_buff, const RefJoinColumn* join_column_per_key,
                                                   const RefTypeInfo* type_info_per_key, const int device_id) {
  hdk_hip_join_column cols[HDK_HIP_MAX_JOIN_KEYS];
  hdk_hip_join_column_type_info tis[HDK_HIP_MAX_JOIN_KEYS];
  if (key_component_count > HDK_HIP_MAX_JOIN_KEYS) {
    throw std::runtime_error("hdk_hip: more key components than the fixed kernel library takes");
  }
  for (size_t k = 0; k < key_component_count; ++k) {
    cols[k] = to_abi_column(join_column_per_key[k]);
    tis[k] = to_abi(type_info_per_key[k]);
  }
  check(hdk_hip_fill_baseline_hash_join_buff(hash_buff, entry_count, invalid_slot_val, for_semi_join,
                                             key_component_count, W, with_val_slot, dev_err_buff, cols, tis, device_id,
                                             nullptr));
}

template <int W, class RefJoinColumn, class RefTypeInfo>
inline void fill_one_to_many_baseline_hash_table_on_device(int32_t* buff, const int8_t* composite_key_dict,
                                                           const int64_t hash_entry_count,
                                                           const int32_t invalid_slot_val,
                                                           const size_t key_component_count,
                                                           const RefJoinColumn* join_column_per_key,
                                                           const RefTypeInfo* type_info_per_key, const int device_id) {
  hdk_hip_join_column cols[HDK_HIP_MAX_JOIN_KEYS];
  hdk_hip_join_column_type_info tis[HDK_HIP_MAX_JOIN_KEYS];
  if (key_component_count > HDK_HIP_MAX_JOIN_KEYS) {
    throw std::runtime_error("hdk_hip: more key components than the fixed kernel library takes");
  }
  for (size_t k = 0; k < key_component_count; ++k) {
    cols[k] = to_abi_column(join_column_per_key[k]);
    tis[k] = to_abi(type_info_per_key[k]);
  }
  check(hdk_hip_fill_one_to_many_baseline_hash_table(buff, composite_key_dict, hash_entry_count, invalid_slot_val,
                                                     key_component_count, W, cols, tis, device_id, nullptr));
}

}  // namespace hip_rt
