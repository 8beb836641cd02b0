// device_common.h -- per-row primitives shared by the gfx950 kernels.
//
// Device restatement of the reference's per-row runtime (the functions HDK's JIT inlines into its
// row function): decoders (QE/DecodersImpl.h:30-150), nullable arithmetic / comparisons
// (QE/RuntimeFunctions.cpp:49-230), scalar helpers (:240-280, omniscidb/Utils/ExtractFromTime.cpp),
// join probes (QE/GroupByRuntime.cpp:274-366) and MurmurHash3 (QE/MurmurHash3Inl.h:11-76).
// The plan lives in device memory and is wave-uniform: every branch on it is a scalar branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hdk_hip.h"

#define HDK_DEV __device__ __forceinline__

namespace hdk {

constexpr int kWave = 64;

HDK_DEV double bits_to_double(int64_t b) { return __longlong_as_double(b); }
HDK_DEV int64_t double_to_bits(double d) { return __double_as_longlong(d); }

// ---------------------------------------------------------------------------------------------
// kernel parameters: the reference's 12 device pointers (QE/QueryExecutionContext.h:111-125)
// ---------------------------------------------------------------------------------------------
struct KernParams {
  const int8_t* const* const* col_buffers;  // COL_BUFFERS
  const uint64_t* num_fragments;            // NUM_FRAGMENTS
  const int8_t* literals;                   // LITERALS (unused)
  const int64_t* num_rows;                  // NUM_ROWS
  const uint64_t* frag_row_offsets;         // FRAG_ROW_OFFSETS
  const int32_t* max_matched;               // MAX_MATCHED
  int32_t* total_matched;                   // TOTAL_MATCHED
  const int64_t* init_agg_vals;             // INIT_AGG_VALS
  int64_t** groupby_buf;                    // GROUPBY_BUF
  int32_t* error_code;                      // ERROR_CODE
  const uint32_t* num_tables;               // NUM_TABLES
  const int64_t* join_hash_tables;          // JOIN_HASH_TABLES
  const struct LaunchWatch* watch;          // this launch's interrupt / watchdog words (workspace head, watch.h)
  const int32_t* interrupt;                 // the device's interrupt word (hdk_hip_set_interrupt)
};

// two 64-bit words moved by ONE 16-byte load / store (global_load_dwordx4, ds_read_b128)
typedef long long __attribute__((ext_vector_type(2))) bf_i64x2;

// record_error_code (QE/RuntimeFunctions.cpp:1123-1135): positive codes are sticky.
HDK_DEV void record_error(int32_t* error_code, int32_t err) {
  if (err > 0) {
    atomicMax(error_code, err);
  }
}

// ---------------------------------------------------------------------------------------------
// decoders
// ---------------------------------------------------------------------------------------------
// the narrow NULL of a DATE-in-days column (FixedWidthSmallDate, QE/Codec.cpp:86-87)
HDK_DEV int64_t small_date_null(int32_t width) { return width == 4 ? static_cast<int64_t>(INT32_MIN) : static_cast<int64_t>(INT16_MIN); }

HDK_DEV int64_t decode_col(const int8_t* __restrict__ buf, int32_t width, int32_t kind, int64_t row) {
  switch (kind) {
    case HDK_COL_DOUBLE:
      return reinterpret_cast<const int64_t*>(buf)[row];
    case HDK_COL_FLOAT:
      return double_to_bits(static_cast<double>(reinterpret_cast<const float*>(buf)[row]));
    case HDK_COL_SMALL_DATE: {  // fixed_width_small_date_decode (QE/DecodersImpl.h:151-159): days -> epoch seconds
      const int64_t v = width == 4 ? static_cast<int64_t>(reinterpret_cast<const int32_t*>(buf)[row])
                                   : static_cast<int64_t>(reinterpret_cast<const int16_t*>(buf)[row]);
      return v == small_date_null(width) ? HDK_NULL_BIGINT : v * 86400;
    }
    case HDK_COL_UNSIGNED:
      switch (width) {
        case 1:
          return reinterpret_cast<const uint8_t*>(buf)[row];
        case 2:
          return reinterpret_cast<const uint16_t*>(buf)[row];
        case 4:
          return reinterpret_cast<const uint32_t*>(buf)[row];
        default:
          return reinterpret_cast<const int64_t*>(buf)[row];
      }
    default:
      switch (width) {
        case 1:
          return reinterpret_cast<const int8_t*>(buf)[row];
        case 2:
          return reinterpret_cast<const int16_t*>(buf)[row];
        case 4:
          return reinterpret_cast<const int32_t*>(buf)[row];
        default:
          return reinterpret_cast<const int64_t*>(buf)[row];
      }
  }
}

// NOTE on `nt`: with a run-time flag the two loads of `nt ? nontemporal : plain` are merged by the optimiser into
// ONE plain load (the hint is metadata and does not survive the merge) -- the generated ISA of the interpreter
// kernels carries no `nt` bit.  Kernels that depend on the hint call __builtin_nontemporal_load directly
// (scan_agg_keys.h, scan_agg_partitioned.h).
// Same decode through an explicit GLOBAL address-space pointer (column buffers are hipMalloc'ed:
// a pointer loaded from COL_BUFFERS is otherwise generic and compiles to flat_load), optionally
// non-temporal: streamed outer-table columns should not evict the join tables from the
// Infinity Cache.
template <typename T>
HDK_DEV T gload(const int8_t* buf, int64_t row, bool nt) {
  const __attribute__((address_space(1))) T* p =
      reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(buf)) + row;
  return nt ? __builtin_nontemporal_load(p) : *p;
}

HDK_DEV int64_t decode_col_g(const int8_t* buf, int32_t width, int32_t kind, int64_t row, bool nt) {
  switch (kind) {
    case HDK_COL_DOUBLE:
      return gload<int64_t>(buf, row, nt);
    case HDK_COL_FLOAT:
      return double_to_bits(static_cast<double>(gload<float>(buf, row, nt)));
    case HDK_COL_SMALL_DATE: {
      const int64_t v = width == 4 ? static_cast<int64_t>(gload<int32_t>(buf, row, nt)) : static_cast<int64_t>(gload<int16_t>(buf, row, nt));
      return v == small_date_null(width) ? HDK_NULL_BIGINT : v * 86400;
    }
    case HDK_COL_UNSIGNED:
      switch (width) {
        case 1: return gload<uint8_t>(buf, row, nt);
        case 2: return gload<uint16_t>(buf, row, nt);
        case 4: return gload<uint32_t>(buf, row, nt);
        default: return gload<int64_t>(buf, row, nt);
      }
    default:
      switch (width) {
        case 1: return gload<int8_t>(buf, row, nt);
        case 2: return gload<int16_t>(buf, row, nt);
        case 4: return gload<int32_t>(buf, row, nt);
        default: return gload<int64_t>(buf, row, nt);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------
HDK_DEV int64_t floor_div_lhs(int64_t dividend, int64_t divisor) {
  return (dividend < 0 ? dividend - (divisor - 1) : dividend) / divisor;
}

HDK_DEV int64_t scale_decimal_down(int64_t operand, int64_t scale) {
  int64_t tmp = scale >> 1;
  tmp = operand >= 0 ? operand + tmp : operand - tmp;
  return tmp / scale;
}

HDK_DEV int64_t extract_year(int64_t timeval) {
  constexpr uint32_t kEpochOffsetYear1900 = 2208988800u;
  constexpr uint32_t kSecsJanToMar1900 = 5097600u;
  constexpr uint32_t kSecondsPer4YearCycle = 126230400u;
  constexpr uint32_t kUSecsPerDay = 86400u;
  constexpr uint32_t kSecondsPerNonLeapYear = 31536000u;
  if (timeval >= 0LL && timeval <= static_cast<int64_t>(UINT32_MAX - kEpochOffsetYear1900)) {
    const uint32_t seconds_1900 = static_cast<uint32_t>(timeval) + kEpochOffsetYear1900;
    const uint32_t leap_years = (seconds_1900 - kSecsJanToMar1900) / kSecondsPer4YearCycle;
    const uint32_t year = (seconds_1900 - leap_years * kUSecsPerDay) / kSecondsPerNonLeapYear + 1900;
    return static_cast<int32_t>(year);
  }
  constexpr int64_t kSecsPerDay = 86400;
  constexpr int64_t kEpochAdjustedDays = 11017;
  constexpr int64_t kDaysPer400Years = 146097;
  constexpr unsigned MARJAN = 31 + 30 + 31 + 30 + 31 + 31 + 30 + 31 + 30 + 31;
  const int64_t day = floor_div_lhs(timeval, kSecsPerDay);
  const int64_t era = floor_div_lhs(day - kEpochAdjustedDays, kDaysPer400Years);
  const unsigned doe = static_cast<unsigned>(day - kEpochAdjustedDays - era * kDaysPer400Years);
  const unsigned yoe = (doe - doe / 1460 + doe / 36524 - (doe == 146096)) / 365;
  const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  return 2000 + era * 400 + yoe + (MARJAN <= doy);
}

HDK_DEV uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// MurmurHash3_x86_32 over `nwords` 32-bit words (key bytes are always a multiple of 4 here).
HDK_DEV uint32_t murmur_hash3_words(const uint32_t* k, int nwords, uint32_t seed) {
  uint32_t h1 = seed;
  const uint32_t c1 = 0xcc9e2d51;
  const uint32_t c2 = 0x1b873593;
  for (int i = 0; i < nwords; ++i) {
    uint32_t k1 = k[i];
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64;
  }
  h1 ^= static_cast<uint32_t>(nwords * 4);
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6b;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35;
  h1 ^= h1 >> 16;
  return h1;
}

// ---------------------------------------------------------------------------------------------
// plan interpretation
// ---------------------------------------------------------------------------------------------
struct RowCtx {
  const hdk_hip_plan* plan;        // device copy, uniform
  const int8_t* const* cols;       // col_buffers[frag], uniform
  int64_t pos;                     // outer row of this lane
  int64_t join_row[HDK_HIP_MAX_JOINS];
};

HDK_DEV bool col_is_fp(const hdk_hip_plan* p, int32_t col) {
  const int32_t k = p->cols[col].kind;
  return k == HDK_COL_FLOAT || k == HDK_COL_DOUBLE;
}

HDK_DEV int64_t load_leaf(const RowCtx& c, const hdk_hip_leaf& l) {
  if (l.kind == HDK_LEAF_COL) {
    const hdk_hip_col& col = c.plan->cols[l.col];
    int64_t row = c.pos;
    if (col.table == 1) {
      row = c.join_row[0];
    } else if (col.table == 2) {
      row = c.join_row[1];
    }
    if (row < 0) {
      // LEFT join without a match: the inner table's columns are NULL
      // (codegenOuterJoinNullPlaceholder, QE/ColumnIR.cpp)
      return l.null_val;
    }
    return decode_col(c.cols[col.buf_idx], col.width, col.kind, row);
  }
  return l.ival;
}

HDK_DEV bool leaf_is_fp(const hdk_hip_plan* p, const hdk_hip_leaf& l) {
  return l.kind == HDK_LEAF_FP || (l.kind == HDK_LEAF_COL && col_is_fp(p, l.col));
}

HDK_DEV bool is_null_val(int64_t v, int64_t null_val, int32_t nullable, bool fp) {
  if (!nullable) {
    return false;
  }
  return fp ? (bits_to_double(v) == bits_to_double(null_val)) : (v == null_val);
}

// + - * with the reference's overflow check (QE/ArithmeticIR.cpp:277-520: the operation's SQL type is `width` bytes
// wide; a result outside that type's range is an error).  Values are carried as int64, so for widths below 8 the
// exact result is at hand; width 8 uses the overflow-detecting builtins.  Returns true on overflow; *r wraps.
HDK_DEV bool checked_arith(int op, int64_t a, int64_t b, int32_t width, int64_t* r) {
  long long res;
  bool ovf;
  switch (op) {
    case HDK_OP_ADD: ovf = __builtin_saddll_overflow(a, b, &res); break;
    case HDK_OP_SUB: ovf = __builtin_ssubll_overflow(a, b, &res); break;
    default: ovf = __builtin_smulll_overflow(a, b, &res); break;
  }
  *r = res;
  if (width <= 0) {
    return false;
  }
  if (width < 8) {
    const int64_t lim = int64_t(1) << (8 * width - 1);
    return ovf || res > lim - 1 || res < -lim;
  }
  return ovf;
}

// expression chain; err receives ERR_DIV_BY_ZERO like the reference's division guard.
HDK_DEV int64_t eval_expr(const RowCtx& c, const hdk_hip_expr& e, int32_t& err) {
  const hdk_hip_plan* p = c.plan;
  int64_t acc = load_leaf(c, e.leaf0);
  bool acc_fp = leaf_is_fp(p, e.leaf0);
  int64_t acc_null = e.leaf0.null_val;
  int32_t acc_nullable = e.leaf0.nullable;
  const int nsteps = e.nsteps;
  for (int s = 0; s < nsteps; ++s) {
    const hdk_hip_step& st = e.steps[s];
    const bool lhs_null = is_null_val(acc, acc_null, acc_nullable, acc_fp);
    int64_t r = 0;
    bool r_null = false;
    const int op = st.op;
    if (op <= HDK_OP_MOD) {
      const int64_t rhs = load_leaf(c, st.rhs);
      const bool rhs_fp = leaf_is_fp(p, st.rhs);
      const bool rhs_null = is_null_val(rhs, st.rhs.null_val, st.rhs.nullable, rhs_fp);
      if (lhs_null || rhs_null) {
        r_null = true;
      } else if (st.out_class == HDK_VC_FP) {
        const double a = acc_fp ? bits_to_double(acc) : static_cast<double>(acc);
        const double b = rhs_fp ? bits_to_double(rhs) : static_cast<double>(rhs);
        double d = 0;
        switch (op) {
          case HDK_OP_ADD: d = a + b; break;
          case HDK_OP_SUB: d = a - b; break;
          case HDK_OP_MUL: d = a * b; break;
          case HDK_OP_DIV:
            if (b == 0.0) { err = HDK_HIP_ERR_DIV_BY_ZERO; r_null = true; } else { d = a / b; }
            break;
          default: r_null = true; break;
        }
        r = double_to_bits(d);
      } else {
        const int64_t a = acc, b = rhs;
        switch (op) {
          case HDK_OP_ADD:
          case HDK_OP_SUB:
          case HDK_OP_MUL:
            if (checked_arith(op, a, b, st.check_width, &r)) {
              err = HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW;
            }
            break;
          case HDK_OP_DIV:
            if (b == 0) { err = HDK_HIP_ERR_DIV_BY_ZERO; r_null = true; }
            else if (a == INT64_MIN && b == -1) { r = INT64_MIN; }
            else { r = a / b; }
            break;
          default:  // MOD
            if (b == 0) { err = HDK_HIP_ERR_DIV_BY_ZERO; r_null = true; }
            else if (b == -1) { r = 0; }
            else { r = a % b; }
            break;
        }
      }
    } else {
      if (lhs_null) {
        r_null = true;
      } else {
        switch (op) {
          case HDK_OP_EXTRACT_YEAR: r = extract_year(acc); break;
          case HDK_OP_SCALE_DOWN: r = scale_decimal_down(acc, st.rhs.ival); break;
          case HDK_OP_FLOOR_DIV: r = floor_div_lhs(acc, st.rhs.ival); break;
          case HDK_OP_CAST_INT_TO_FP: r = double_to_bits(static_cast<double>(acc)); break;
          case HDK_OP_CAST_FP_TO_INT: {
            const double d = bits_to_double(acc);
            r = static_cast<int64_t>(d + (d < 0.0 ? -0.5 : 0.5));
            break;
          }
          default: r_null = true; break;
        }
      }
    }
    acc = r_null ? st.null_out : r;
    acc_fp = st.out_class == HDK_VC_FP;
    acc_null = st.null_out;
    acc_nullable = 1;
  }
  return acc;
}

// one filter conjunct: true iff the three-valued comparison is TRUE; *is_null (when asked for) tells NULL from FALSE
HDK_DEV bool eval_qual(const RowCtx& c, const hdk_hip_qual& q, int32_t& err, bool* is_null = nullptr) {
  const int64_t lhs = eval_expr(c, q.lhs, err);
  const int64_t rhs = load_leaf(c, q.rhs);
  const bool lhs_fp = q.lhs.vclass == HDK_VC_FP;
  const bool rhs_fp = leaf_is_fp(c.plan, q.rhs);
  if (is_null) {
    *is_null = false;
  }
  if (is_null_val(lhs, q.lhs.null_val, q.lhs.nullable, lhs_fp) ||
      is_null_val(rhs, q.rhs.null_val, q.rhs.nullable, rhs_fp)) {
    if (is_null) {
      *is_null = true;
    }
    return false;
  }
  if (lhs_fp || rhs_fp) {
    const double a = lhs_fp ? bits_to_double(lhs) : static_cast<double>(lhs);
    const double b = rhs_fp ? bits_to_double(rhs) : static_cast<double>(rhs);
    switch (q.cmp) {
      case HDK_CMP_EQ: return a == b;
      case HDK_CMP_NE: return a != b;
      case HDK_CMP_LT: return a < b;
      case HDK_CMP_GT: return a > b;
      case HDK_CMP_LE: return a <= b;
      default: return a >= b;
    }
  }
  switch (q.cmp) {
    case HDK_CMP_EQ: return lhs == rhs;
    case HDK_CMP_NE: return lhs != rhs;
    case HDK_CMP_LT: return lhs < rhs;
    case HDK_CMP_GT: return lhs > rhs;
    case HDK_CMP_LE: return lhs <= rhs;
    default: return lhs >= rhs;
  }
}

// join probe: hash_join_idx[_nullable/_bitwise] and the bucketized forms
HDK_DEV int64_t probe_join(const hdk_hip_join& jn, const int32_t* __restrict__ table, int64_t key) {
  int64_t k = key;
  int64_t maxk = jn.max_key;
  if (jn.null_mode != HDK_JOIN_NULL_NONE && key == jn.null_val) {
    if (jn.null_mode == HDK_JOIN_NULL_NULLABLE) {
      return -1;
    }
    k = jn.translated_null;
    maxk = jn.translated_null;
  }
  if (k >= jn.min_key && k <= maxk) {
    int64_t off = k - jn.min_key;
    if (jn.bucket > 1) {
      off /= jn.bucket;
    }
    return table[off];
  }
  return -1;
}

// same probe, table read through a global-address-space pointer; *slot_out = the slot index
template <class JoinT>
HDK_DEV int64_t probe_join_g(const JoinT& jn, const int32_t* table, int64_t key, int64_t* slot_out) {
  int64_t k = key;
  int64_t maxk = jn.max_key;
  *slot_out = 0;
  if (jn.null_mode != HDK_JOIN_NULL_NONE && key == jn.null_val) {
    if (jn.null_mode == HDK_JOIN_NULL_NULLABLE) {
      return -1;
    }
    k = jn.translated_null;
    maxk = jn.translated_null;
  }
  if (k >= jn.min_key && k <= maxk) {
    int64_t off = k - jn.min_key;
    if (jn.bucket > 1) {
      off /= jn.bucket;
    }
    *slot_out = off;
    return gload<int32_t>(reinterpret_cast<const int8_t*>(table), off, false);
  }
  return -1;
}

// ---- matching sets ------------------------------------------------------------------------------
// MurmurHash1 (QE/MurmurHash1Inl.h:6-52) over a key of `n` 32-bit words
HDK_DEV uint32_t murmur_hash1_words(const uint32_t* w, int n) {
  const uint32_t m = 0xc6a4a793;
  uint32_t h = 0u ^ (static_cast<uint32_t>(n * 4) * m);
  for (int i = 0; i < n; ++i) {
    h += w[i];
    h *= m;
    h ^= h >> 16;
  }
  h *= m;
  h ^= h >> 10;
  h *= m;
  h ^= h >> 17;
  return h;
}

// The inner rows matching the current outer row at one join level (HashJoin::codegenMatchingSet,
// QE/JoinHashTable/HashJoin.cpp:149-197; keyed tables: BaselineJoinHashTable.cpp:769-811 with
// baseline_hash_join_idx / get_composite_key_index, JoinHashTableQueryRuntime.cpp:42-172).
struct MatchSet {
  const int32_t* ids;  // one-to-many: the bin's row ids; nullptr for a one-to-one match
  int64_t single;      // one-to-one: the row id
  int32_t n;
};

template <typename T>
HDK_DEV MatchSet keyed_matching_set(const hdk_hip_join& jn, const int8_t* table, const int64_t* key) {
  MatchSet ms;
  ms.ids = nullptr;
  ms.single = -1;
  ms.n = 0;
  const int kc = jn.key_component_count;
  const uint32_t entries = static_cast<uint32_t>(jn.entry_count);
  if (entries == 0) {
    return ms;
  }
  uint32_t words[2 * HDK_HIP_MAX_JOIN_KEYS];
  int nw = 0;
#pragma unroll
  for (int i = 0; i < HDK_HIP_MAX_JOIN_KEYS; ++i) {
    if (i < kc) {
      if constexpr (sizeof(T) == 8) {
        words[nw++] = static_cast<uint32_t>(static_cast<uint64_t>(key[i]));
        words[nw++] = static_cast<uint32_t>(static_cast<uint64_t>(key[i]) >> 32);
      } else {
        words[nw++] = static_cast<uint32_t>(key[i]);
      }
    }
  }
  const T invalid = sizeof(T) == 8 ? static_cast<T>(HDK_EMPTY_KEY_64) : static_cast<T>(HDK_EMPTY_KEY_32);
  const bool many = jn.kind == HDK_JOIN_KEYED_ONE_TO_MANY;
  const int comps = kc + (many ? 0 : 1);
  const T* dict = reinterpret_cast<const T*>(table);
  const uint32_t h = murmur_hash1_words(words, nw) % entries;
  uint32_t hp = h;
  int64_t slot = -1;
  do {
    const T* e = dict + static_cast<size_t>(hp) * comps;
    bool eq = true;
    for (int i = 0; i < kc; ++i) {
      eq = eq && e[i] == static_cast<T>(key[i]);
    }
    if (eq) {
      slot = hp;
      break;
    }
    if (e[0] == invalid) {
      break;  // kNotPresent / -1
    }
    hp = hp + 1 == entries ? 0 : hp + 1;
  } while (hp != h);
  if (slot < 0) {
    return ms;
  }
  if (!many) {
    ms.single = static_cast<int64_t>(dict[static_cast<size_t>(slot) * comps + kc]);
    ms.n = ms.single >= 0 ? 1 : 0;
    return ms;
  }
  const int32_t* otm = reinterpret_cast<const int32_t*>(table + static_cast<size_t>(entries) * kc * sizeof(T));
  const int32_t pos = otm[slot];
  if (pos < 0) {
    return ms;
  }
  ms.ids = otm + 2 * static_cast<size_t>(entries) + pos;
  ms.n = otm[static_cast<size_t>(entries) + slot];
  return ms;
}

HDK_DEV MatchSet matching_set(const RowCtx& c, const hdk_hip_join& jn, const int64_t* join_hash_tables, int nj,
                              int32_t& err) {
  const int8_t* table = (nj == 1 && jn.table_idx == 0) ? reinterpret_cast<const int8_t*>(join_hash_tables)
                                                       : reinterpret_cast<const int8_t*>(join_hash_tables[jn.table_idx]);
  if (jn.kind == HDK_JOIN_KEYED_ONE_TO_ONE || jn.kind == HDK_JOIN_KEYED_ONE_TO_MANY) {
    int64_t key[HDK_HIP_MAX_JOIN_KEYS];
    const int kc = jn.key_component_count;
#pragma unroll
    for (int i = 0; i < HDK_HIP_MAX_JOIN_KEYS; ++i) {
      key[i] = i < kc ? eval_expr(c, i == 0 ? jn.outer_key : jn.extra_keys[i > 0 ? i - 1 : 0], err) : 0;
    }
    return jn.key_component_width == 4 ? keyed_matching_set<int32_t>(jn, table, key)
                                       : keyed_matching_set<int64_t>(jn, table, key);
  }
  MatchSet ms;
  ms.ids = nullptr;
  ms.single = -1;
  ms.n = 0;
  const int64_t key = eval_expr(c, jn.outer_key, err);
  const int32_t* t = reinterpret_cast<const int32_t*>(table);
  if (jn.kind == HDK_JOIN_ONE_TO_MANY) {
    const int64_t pos = probe_join(jn, t, key);
    if (pos >= 0) {
      ms.ids = t + 2 * jn.entry_count + pos;
      ms.n = static_cast<int32_t>(probe_join(jn, t + jn.entry_count, key));
    }
    return ms;
  }
  ms.single = probe_join(jn, t, key);
  ms.n = ms.single >= 0 ? 1 : 0;
  return ms;
}

// The filter as a postfix program over the conjuncts (hdk_hip_plan::filter_ops) with the reference's three-valued
// logical_and / logical_or / logical_not (QE/RuntimeFunctions.cpp:357-384).  The value stack lives in two bit masks
// (bit i of `t`: entry i is TRUE; of `n`: entry i is NULL), indexed by the wave-uniform stack pointer.
HDK_DEV bool filter_program_pass(const RowCtx& c, int32_t& err) {
  const hdk_hip_plan* p = c.plan;
  uint32_t t = 0, n = 0;
  int sp = 0;
  const int nops = p->num_filter_ops;
  for (int i = 0; i < nops; ++i) {
    const uint32_t op = p->filter_ops[i];
    if (op < HDK_F_AND) {
      bool isnull;
      const bool v = eval_qual(c, p->quals[op], err, &isnull);
      t = (t & ~(1u << sp)) | (static_cast<uint32_t>(v) << sp);
      n = (n & ~(1u << sp)) | (static_cast<uint32_t>(isnull) << sp);
      ++sp;
    } else if (op == HDK_F_NOT) {
      const uint32_t m = 1u << (sp - 1);
      t = (t & ~m) | (~(t | n) & m);  // NULL stays NULL, TRUE <-> FALSE
    } else {
      const uint32_t ta = (t >> (sp - 2)) & 1u, tb = (t >> (sp - 1)) & 1u;
      const uint32_t na = (n >> (sp - 2)) & 1u, nb = (n >> (sp - 1)) & 1u;
      uint32_t tr, nr;
      if (op == HDK_F_AND) {
        const uint32_t fa = (ta | na) ^ 1u, fb = (tb | nb) ^ 1u;  // FALSE operands
        tr = ta & tb;
        nr = (tr | fa | fb) ^ 1u;
      } else {
        tr = ta | tb;
        nr = (tr ^ 1u) & (na | nb);
      }
      sp -= 1;
      const uint32_t m = 1u << (sp - 1);
      t = (t & ~m) | (tr << (sp - 1));
      n = (n & ~m) | (nr << (sp - 1));
    }
  }
  return sp == 1 && (t & 1u);
}

HDK_DEV bool quals_pass(const RowCtx& c, int stage, int32_t& err) {
  const hdk_hip_plan* p = c.plan;
  if (p->num_filter_ops) {
    return (p->filter_after_joins != 0) == (stage != 0) ? filter_program_pass(c, err) : true;
  }
  const int nq = p->num_quals;
  for (int q = 0; q < nq; ++q) {
    if ((p->quals[q].after_joins != 0) == (stage != 0) && !eval_qual(c, p->quals[q], err)) {
      return false;
    }
  }
  return true;
}

// The join loop nest of one outer row (Executor::buildJoinLoops, QE/IRCodegen.cpp:497-667): filters
// on the outer table, one loop level per join (a LEFT join without a match runs once with the inner
// row = -1, whose columns read as NULL), the filters that read joined columns, then `body()`.
template <typename Body>
HDK_DEV void for_each_row_match(RowCtx& c, const int64_t* join_hash_tables, int32_t& err, Body&& body) {
  const hdk_hip_plan* p = c.plan;
  if (!quals_pass(c, 0, err)) {
    return;
  }
  const int nj = p->num_joins;
  if (nj == 0) {
    if (quals_pass(c, 1, err)) {
      body();
    }
    return;
  }
  const hdk_hip_join& j0 = p->joins[0];
  const MatchSet m0 = matching_set(c, j0, join_hash_tables, nj, err);
  // ANTI: the rest of the row runs when the probe found no slot (JoinLoop.cpp:258-262), with the inner row "not found"
  const bool anti0 = j0.type == HDK_JOIN_ANTI;
  const bool left0 = m0.n <= 0 && (j0.type == HDK_JOIN_LEFT || anti0);
  const int n0 = left0 ? 1 : (anti0 ? 0 : m0.n);
  for (int i0 = 0; i0 < n0; ++i0) {
    c.join_row[0] = left0 ? -1 : (m0.ids ? static_cast<int64_t>(m0.ids[i0]) : m0.single);
    if (nj == 1) {
      if (quals_pass(c, 1, err)) {
        body();
      }
      continue;
    }
    const hdk_hip_join& j1 = p->joins[1];
    const MatchSet m1 = matching_set(c, j1, join_hash_tables, nj, err);
    const bool anti1 = j1.type == HDK_JOIN_ANTI;
    const bool left1 = m1.n <= 0 && (j1.type == HDK_JOIN_LEFT || anti1);
    const int n1 = left1 ? 1 : (anti1 ? 0 : m1.n);
    for (int i1 = 0; i1 < n1; ++i1) {
      c.join_row[1] = left1 ? -1 : (m1.ids ? static_cast<int64_t>(m1.ids[i1]) : m1.single);
      if (quals_pass(c, 1, err)) {
        body();
      }
    }
  }
}

// group key #k with the perfect-hash NULL translation (translate_null_key_*)
HDK_DEV int64_t eval_key(const RowCtx& c, int k, int32_t& err) {
  const hdk_hip_plan* p = c.plan;
  int64_t kv = eval_expr(c, p->keys[k], err);
  if (p->query_kind == HDK_Q_PERFECT_HASH && p->key_has_nulls[k] && p->keys[k].nullable &&
      kv == p->keys[k].null_val) {
    kv = p->key_null_translated[k];
  }
  return kv;
}

// perfect-hash entry index: single column (get_group_value_fast) or perfect_key_hash
// (QE/RowFuncBuilder.cpp:748-801).
HDK_DEV int64_t perfect_hash_entry(const RowCtx& c, int32_t& err) {
  const hdk_hip_plan* p = c.plan;
  const int nk = p->key_count;
  int64_t h = 0;
  int64_t stride = 1;
  for (int k = 0; k < nk; ++k) {
    const int64_t kv = eval_key(c, k, err);
    int64_t term = kv - p->key_min[k];
    if (p->key_bucket[k]) {
      term /= p->key_bucket[k];
    }
    h += term * stride;
    stride *= p->key_card[k];
  }
  return h;
}

// target argument with the arg-type NULL rewritten to the slot-type NULL (convertNullIfAny) and
// int->fp promotion (Executor::castToFP).  *is_null tells whether the value is the skip value.
HDK_DEV int64_t eval_target_arg(const RowCtx& c, const hdk_hip_target& tg, bool& is_null, int32_t& err) {
  is_null = false;
  if (!tg.has_arg) {
    return 0;
  }
  int64_t v = eval_expr(c, tg.arg, err);
  if (tg.agg == HDK_AGG_ID) {
    return v;
  }
  const bool arg_fp = tg.arg.vclass == HDK_VC_FP;
  if (tg.skip_null && is_null_val(v, tg.arg.null_val, tg.arg.nullable, arg_fp)) {
    is_null = true;
    return tg.null_val;
  }
  if (tg.arg_is_fp && !arg_fp) {
    v = double_to_bits(static_cast<double>(v));
  }
  if (tg.skip_null) {
    // a computed value that collides with the skip value is skipped, exactly as `val != skip_val`
    is_null = tg.arg_is_fp ? (bits_to_double(v) == bits_to_double(tg.null_val)) : (v == tg.null_val);
  }
  return v;
}

// 8- and 16-bit MIN / MAX slots of a columnar buffer with logical-sized columns: agg_{min,max}_int{8,16}[_skip_val]
// (QE/RuntimeFunctions.cpp:540-560,670-704) for a slot with ONE writer at a time (hdk_finalize, the reductions)
HDK_DEV void small_min_max(int agg, bool skip, int w, int8_t* slot, int64_t val, int64_t nullv) {
  if (w == 2) {
    int16_t* s = reinterpret_cast<int16_t*>(slot);
    const int16_t v = static_cast<int16_t>(val), n = static_cast<int16_t>(nullv);
    if (skip && v == n) return;
    if (skip && *s == n) {
      *s = v;
      return;
    }
    *s = agg == HDK_AGG_MIN ? (*s < v ? *s : v) : (*s > v ? *s : v);
  } else {
    const int8_t v = static_cast<int8_t>(val), n = static_cast<int8_t>(nullv);
    if (skip && v == n) return;
    if (skip && *slot == n) {
      *slot = v;
      return;
    }
    *slot = agg == HDK_AGG_MIN ? (*slot < v ? *slot : v) : (*slot > v ? *slot : v);
  }
}

// Quad q of an initialised row-wise group-by row: [keys: key_count x key_width, padded to 8][init_vals...]
// (init_group_by_buffer_gpu, QE/GpuInitGroups.cu:110-160): the one definition the init kernel and the fused
// initialisation of the partitioned group-by share.
HDK_DEV int64_t init_row_quad(uint32_t q, uint32_t keys_quads, uint32_t key_count, uint32_t key_width,
                              const int64_t* __restrict__ init_vals) {
  if (q >= keys_quads) {
    return init_vals[q - keys_quads];
  }
  if (key_width == 8) {
    return HDK_EMPTY_KEY_64;
  }
  // two int32 key components per quad; components past key_count keep zero bits
  const uint32_t c0 = q * 2;
  const uint32_t lo = c0 < key_count ? static_cast<uint32_t>(HDK_EMPTY_KEY_32) : 0u;
  const uint32_t hi = (c0 + 1) < key_count ? static_cast<uint32_t>(HDK_EMPTY_KEY_32) : 0u;
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

// the float sentinel of a float-accumulator target: tg.null_val carries it widened to double (what the scan compares)
HDK_DEV int32_t float_slot_null(const hdk_hip_target& tg) {
  return __float_as_int(static_cast<float>(bits_to_double(tg.null_val)));
}

// ---------------------------------------------------------------------------------------------
// columnar layout helpers (RS/QueryMemoryDescriptor.cpp getColOffInBytes)
// ---------------------------------------------------------------------------------------------
HDK_DEV size_t align8(size_t x) { return (x + 7) & ~static_cast<size_t>(7); }

HDK_DEV size_t columnar_slot_off(const hdk_hip_plan* p, uint32_t entry_count, int slot) {
  size_t off = p->keyless ? 0 : static_cast<size_t>(p->key_count) * align8(static_cast<size_t>(entry_count) * 8);
  if (p->query_kind == HDK_Q_PROJECTION) {
    off = align8(static_cast<size_t>(entry_count) * 8);  // the row-position column
  }
  int s = 0;
  const int nt = p->num_targets;
  for (int t = 0; t < nt; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    off = align8(off);
    if (s == slot) {
      return off;
    }
    off += static_cast<size_t>(entry_count) * tg.slot_width;
    ++s;
    if (tg.agg == HDK_AGG_AVG) {
      off = align8(off);
      if (s == slot) {
        return off;
      }
      off += static_cast<size_t>(entry_count) * tg.slot2_width;
      ++s;
    }
  }
  return off;
}

}  // namespace hdk
