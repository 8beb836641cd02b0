// HipPlanBuilder.h -- HDK-side construction of the POD plan: the output-layout half of `hdk_hip_plan` filled from a
// QueryMemoryDescriptor, the expression half handed over by the caller.
//
// In HDK this runs where Executor::compileWorkUnit has the descriptor in hand (QE/NativeCodegen.cpp:1451-1461:
// MemoryLayoutBuilder::build -> QueryMemoryDescriptor) and would otherwise start emitting IR.  `QMD` is any type with
// the accessors of ResultSet/QueryMemoryDescriptor.h used below -- the real class, or the stand-in of the C++
// harness (tests/cpp/qmd_standin.h) -- so that the mapping is compiled and EXECUTED without an HDK build:
//     getQueryDescriptionType()  getEntryCount()  hasKeylessHash()  getTargetIdxForKey()  didOutputColumnar()
//     getRowSize()  getGroupbyColCount()  getEffectiveKeyWidth()  getColOffInBytes(slot)
//     getPaddedSlotWidthBytes(slot)  getMinVal()  getMaxVal()  getBucket()  hasNulls()
// The enumerators of QueryDescriptionType are compared by value (ResultSet/ResultType.h:28-34: GroupByPerfectHash = 0,
// GroupByBaselineHash = 1, Projection = 2, NonGroupedAggregate = 3).
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "hdk_hip.h"

namespace hip_rt {

// one input column of the step: index into col_buffers[frag] = its position in this vector
struct HipInputCol {
  int32_t table;  // 0 outer, j > 0 inner table of join j - 1
  int32_t width;
  int32_t kind;   // hdk_hip_col_kind
  // ChunkStats over the fragments the step reads (ChunkMetadata::chunkStats, DataMgr/ChunkMetadata.h); 0 = unknown
  int32_t has_stats = 0;
  int32_t has_nulls = 0;
  int64_t min_val = 0;
  int64_t max_val = 0;
};

// one target as get_target_info sees it (Shared/TargetInfo.h:89-160), argument already in ABI form
struct HipTargetDesc {
  int32_t agg;        // hdk_hip_agg
  bool has_arg;       // false: COUNT(*)
  hdk_hip_expr arg;
  bool skip_null;     // TargetInfo::skip_null_val
  int32_t arg_is_fp;  // hdk_hip_fp_slot: NONE, DOUBLE (agg_*_double family), FLOAT (takes_float_argument: agg_*_float)
  int32_t key_idx;    // HDK_AGG_ID: projected group-by key
  int64_t null_val;   // the skip value TargetExprBuilder passes (QE/TargetExprBuilder.cpp:437-452)
};

struct HipWorkUnit {
  std::vector<HipInputCol> cols;
  std::vector<hdk_hip_qual> quals;     // simple_quals + quals of the RelAlgExecutionUnit
  std::vector<uint8_t> filter_ops;     // postfix program over `quals` when the filter is not a plain conjunction
  int32_t filter_after_joins{0};       // ... and the stage the program runs at
  std::vector<hdk_hip_join> joins;
  std::vector<hdk_hip_expr> keys;      // groupby_exprs
  std::vector<int64_t> key_card;       // ColRangeInfo::getBucketedCardinality per key (multi-column perfect hash)
  // get_expr_range_info per group-by expression (QE/ColRangeInfo.cpp:28-66), for perfect hash over several keys: each
  // key's own range feeds the key index (RowFuncBuilder.cpp:748-801).  Empty: the descriptor's single range is used.
  struct KeyRange {
    int64_t min, max, bucket;
    bool has_nulls;
  };
  std::vector<KeyRange> key_ranges;
  std::vector<HipTargetDesc> targets;  // target_exprs
};

inline hdk_hip_expr column_expr(int32_t col, int64_t null_val, bool nullable) {
  hdk_hip_expr e;
  std::memset(&e, 0, sizeof(e));
  e.vclass = HDK_VC_INT;
  e.leaf0.kind = HDK_LEAF_COL;
  e.leaf0.col = col;
  e.leaf0.null_val = null_val;
  e.leaf0.nullable = nullable;
  e.null_val = null_val;
  e.nullable = nullable;
  return e;
}

template <class QMD>
hdk_hip_plan make_plan(const HipWorkUnit& wu, const QMD& qmd) {
  if (wu.cols.size() > HDK_HIP_MAX_COLS || wu.quals.size() > HDK_HIP_MAX_QUALS || wu.joins.size() > HDK_HIP_MAX_JOINS ||
      wu.keys.size() > HDK_HIP_MAX_KEYS || wu.targets.empty() || wu.targets.size() > HDK_HIP_MAX_TARGETS) {
    throw std::runtime_error("QueryMustRunOnCpu: step outside the limits of the fixed kernel library");
  }
  hdk_hip_plan p;
  std::memset(&p, 0, sizeof(p));
  p.abi_version = HDK_HIP_PLAN_ABI;
  switch (static_cast<int>(qmd.getQueryDescriptionType())) {
    case 0: p.query_kind = HDK_Q_PERFECT_HASH; break;
    case 1: p.query_kind = HDK_Q_BASELINE_HASH; break;
    case 2: p.query_kind = HDK_Q_PROJECTION; break;
    case 3: p.query_kind = HDK_Q_NON_GROUPED; break;
    default: throw std::runtime_error("QueryMustRunOnCpu: query description type outside the fixed kernel library");
  }
  p.num_cols = static_cast<int32_t>(wu.cols.size());
  for (size_t i = 0; i < wu.cols.size(); ++i) {
    p.cols[i].buf_idx = static_cast<int32_t>(i);
    p.cols[i].table = wu.cols[i].table;
    p.cols[i].width = wu.cols[i].width;
    p.cols[i].kind = wu.cols[i].kind;
    // ChunkStats of the column over the fragments of the work unit (DataMgr/ChunkMetadata.h; 0 = unknown)
    p.cols[i].has_stats = wu.cols[i].has_stats;
    p.cols[i].has_nulls = wu.cols[i].has_nulls;
    p.cols[i].min_val = wu.cols[i].min_val;
    p.cols[i].max_val = wu.cols[i].max_val;
  }
  p.num_quals = static_cast<int32_t>(wu.quals.size());
  for (size_t i = 0; i < wu.quals.size(); ++i) p.quals[i] = wu.quals[i];
  if (wu.filter_ops.size() > HDK_HIP_MAX_FILTER_OPS) {
    throw std::runtime_error("QueryMustRunOnCpu: filter program too long for the fixed kernel library");
  }
  p.num_filter_ops = static_cast<int32_t>(wu.filter_ops.size());
  for (size_t i = 0; i < wu.filter_ops.size(); ++i) p.filter_ops[i] = wu.filter_ops[i];
  p.filter_after_joins = wu.filter_after_joins;
  p.num_joins = static_cast<int32_t>(wu.joins.size());
  for (size_t i = 0; i < wu.joins.size(); ++i) p.joins[i] = wu.joins[i];
  const bool grouped = p.query_kind == HDK_Q_PERFECT_HASH || p.query_kind == HDK_Q_BASELINE_HASH;
  p.key_count = grouped ? static_cast<int32_t>(wu.keys.size()) : 0;
  for (int k = 0; k < p.key_count; ++k) {
    p.keys[k] = wu.keys[k];
    if (p.query_kind == HDK_Q_PERFECT_HASH && !wu.key_ranges.empty()) {  // several keys: every key's own range
      const HipWorkUnit::KeyRange& r = wu.key_ranges.at(k);
      const int64_t step = r.bucket ? r.bucket : 1;
      p.key_min[k] = r.min;
      p.key_bucket[k] = r.bucket;
      p.key_has_nulls[k] = r.has_nulls ? 1 : 0;
      p.key_null_translated[k] = r.max + step;  // RowFuncBuilder.cpp:456-461
      p.key_card[k] = (r.max - r.min) / step + 1 + (r.has_nulls ? 1 : 0);  // getBucketedCardinality
    } else if (p.query_kind == HDK_Q_PERFECT_HASH) {  // ColRangeInfo of the descriptor (single column) / per-key cardinalities
      p.key_min[k] = qmd.getMinVal();
      p.key_bucket[k] = qmd.getBucket();
      p.key_has_nulls[k] = qmd.hasNulls() ? 1 : 0;
      p.key_null_translated[k] = qmd.getMaxVal() + (qmd.getBucket() ? qmd.getBucket() : 1);  // RowFuncBuilder.cpp:456-461
      p.key_card[k] = k < static_cast<int>(wu.key_card.size()) ? wu.key_card[k] : static_cast<int64_t>(qmd.getEntryCount());
    }
  }
  p.entry_count = static_cast<uint32_t>(grouped || p.query_kind == HDK_Q_PROJECTION ? qmd.getEntryCount() : 1);
  p.key_width = p.query_kind == HDK_Q_BASELINE_HASH ? static_cast<int32_t>(qmd.getEffectiveKeyWidth()) : 8;
  p.keyless = qmd.hasKeylessHash() ? 1 : 0;
  p.idx_target_as_key = p.keyless ? static_cast<int32_t>(qmd.getTargetIdxForKey()) : -1;
  p.output_columnar = qmd.didOutputColumnar() ? 1 : 0;
  p.row_size_quad = (grouped || p.query_kind == HDK_Q_PROJECTION) && !p.output_columnar
                        ? static_cast<uint32_t>(qmd.getRowSize() / 8)
                        : 0;
  p.num_targets = static_cast<int32_t>(wu.targets.size());
  size_t slot = 0;
  for (size_t t = 0; t < wu.targets.size(); ++t) {
    const HipTargetDesc& d = wu.targets[t];
    hdk_hip_target& tg = p.targets[t];
    tg.agg = d.agg;
    tg.has_arg = d.has_arg ? 1 : 0;
    tg.arg = d.arg;
    tg.skip_null = d.skip_null ? 1 : 0;
    tg.arg_is_fp = d.arg_is_fp;
    tg.key_idx = d.key_idx;
    tg.null_val = d.null_val;
    tg.slot_width = static_cast<int32_t>(qmd.getPaddedSlotWidthBytes(slot));
    tg.slot_off = static_cast<int32_t>(qmd.getColOffInBytes(slot));
    if (tg.slot_width == 4 && d.arg_is_fp == HDK_FP_SLOT_NONE && d.null_val == INT64_MIN) {
      tg.null_val = INT32_MIN;  // the skip value is slot-typed: inline_int_null_val of the compacted slot (TargetExprBuilder.cpp:437-452)
    }
    ++slot;
    if (d.agg == HDK_AGG_AVG) {  // the count slot (ColSlotContext: two slots for AVG)
      tg.slot2_width = static_cast<int32_t>(qmd.getPaddedSlotWidthBytes(slot));
      tg.slot2_off = static_cast<int32_t>(qmd.getColOffInBytes(slot));
      ++slot;
    } else {
      tg.slot2_width = tg.slot_width;
      tg.slot2_off = tg.slot_off;
    }
  }
  return p;
}

}  // namespace hip_rt
