// A descriptor with the accessors of ResultSet/QueryMemoryDescriptor.h that hip_rt::make_plan reads
// (:131-144,:165,:209-216,:229,:239,:278; offsets: QueryMemoryDescriptor.cpp:244-256 getRowSize, :314-338
// getColOffInBytes; slot packing: ColSlotContext::getColOnlyOffInBytes).  The real class needs Config, DataMgr and
// the executor; this one is filled by hand in the harness the way MemoryLayoutBuilder would for the two queries.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

class QmdStandIn {
 public:
  QueryDescriptionType query_desc_type_{QueryDescriptionType::NonGroupedAggregate};
  bool keyless_hash_{false};
  int32_t idx_target_as_key_{0};
  std::vector<int8_t> group_col_widths_;
  int8_t group_col_compact_width_{0};
  std::vector<int8_t> padded_slot_widths_;
  size_t entry_count_{1};
  int64_t min_val_{0}, max_val_{0}, bucket_{0};
  bool has_nulls_{false};
  bool output_columnar_{false};

  QueryDescriptionType getQueryDescriptionType() const { return query_desc_type_; }
  bool hasKeylessHash() const { return keyless_hash_; }
  int32_t getTargetIdxForKey() const { return idx_target_as_key_; }
  int8_t getPaddedSlotWidthBytes(const size_t slot_idx) const { return padded_slot_widths_[slot_idx]; }
  size_t getEntryCount() const { return entry_count_; }
  int64_t getMinVal() const { return min_val_; }
  int64_t getMaxVal() const { return max_val_; }
  int64_t getBucket() const { return bucket_; }
  bool hasNulls() const { return has_nulls_; }
  bool didOutputColumnar() const { return output_columnar_; }
  size_t getGroupbyColCount() const { return group_col_widths_.size(); }
  size_t getSlotCount() const { return padded_slot_widths_.size(); }
  size_t getEffectiveKeyWidth() const { return group_col_compact_width_ ? group_col_compact_width_ : sizeof(int64_t); }

  static size_t align8(size_t v) { return (v + 7) & ~size_t(7); }
  size_t getColOnlyOffInBytes(const size_t col_idx) const {
    size_t off = 0;
    for (size_t i = 0; i <= col_idx; ++i) {
      const size_t w = static_cast<size_t>(padded_slot_widths_[i]);
      if (w == 8) off = align8(off);
      if (i == col_idx) return off;
      off += w;
    }
    return off;
  }
  size_t keyBytes() const { return keyless_hash_ ? 0 : align8(group_col_widths_.size() * getEffectiveKeyWidth()); }
  size_t getRowSize() const {
    size_t cols = 0;
    if (!padded_slot_widths_.empty()) {
      cols = getColOnlyOffInBytes(padded_slot_widths_.size() - 1) + static_cast<size_t>(padded_slot_widths_.back());
    }
    return align8(keyBytes() + cols);
  }
  size_t getColOffInBytes(const size_t col_idx) const {
    if (output_columnar_) {
      size_t off = keyless_hash_ ? 0 : group_col_widths_.size() * align8(8 * entry_count_);
      for (size_t i = 0; i < col_idx; ++i) off += align8(static_cast<size_t>(padded_slot_widths_[i]) * entry_count_);
      return off;
    }
    return keyBytes() + getColOnlyOffInBytes(col_idx);
  }
  size_t getBufferSizeBytes() const {
    if (query_desc_type_ == QueryDescriptionType::NonGroupedAggregate) return 8 * padded_slot_widths_.size();
    if (output_columnar_) return getColOffInBytes(padded_slot_widths_.size());
    return getRowSize() * entry_count_;
  }
};
