// plain_quals.h -- the filter form the specialised kernels take: a conjunction of
// `outer column <cmp> literal` with the reference's three-valued semantics (DEF_CMP_NULLABLE,
// QE/RuntimeFunctions.cpp:83-117: a NULL operand fails the conjunct), evaluated for VR rows per lane
// with every wave-uniform decision (decoder, operator, fp/int) outside the row loops.
#pragma once
#include "device_common.h"

namespace hdk {

constexpr int kMaxPlainQuals = 3;

struct ProjFastCol {
  int32_t buf_idx;
  int32_t width;
  int32_t kind;
  int32_t pad_;
};
struct ProjFastQual {
  ProjFastCol col;
  int32_t cmp;       // hdk_hip_cmp
  int32_t nullable;
  int64_t null_val;  // in-band NULL of the column (int64-widened or double bits)
  int64_t rhs;       // literal: int64, or double bits when the comparison is fp
  int32_t fp;        // compare as double (column or literal is fp)
  int32_t col_fp;    // the column holds fp values
};

// VR rows (explicit row numbers, only where live) of one column
template <int VR>
HDK_DEV void plain_load_rows(const int8_t* buf, int width, int kind, const int64_t (&row)[VR], const bool (&live)[VR],
                             bool nt, int64_t (&out)[VR]) {
#define HDK_PQ_ROWS(T, CONV)                          \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {    \
    out[r] = 0;                                       \
    if (live[r]) {                                    \
      const T x = gload<T>(buf, row[r], nt);          \
      out[r] = CONV;                                  \
    }                                                 \
  }
  if (kind == HDK_COL_DOUBLE) {
    HDK_PQ_ROWS(int64_t, x)
  } else if (kind == HDK_COL_FLOAT) {
    HDK_PQ_ROWS(float, double_to_bits(static_cast<double>(x)))
  } else if (kind == HDK_COL_UNSIGNED) {
    switch (width) {
      case 1: HDK_PQ_ROWS(uint8_t, static_cast<int64_t>(x)) break;
      case 2: HDK_PQ_ROWS(uint16_t, static_cast<int64_t>(x)) break;
      case 4: HDK_PQ_ROWS(uint32_t, static_cast<int64_t>(x)) break;
      default: HDK_PQ_ROWS(int64_t, x) break;
    }
  } else {
    switch (width) {
      case 1: HDK_PQ_ROWS(int8_t, static_cast<int64_t>(x)) break;
      case 2: HDK_PQ_ROWS(int16_t, static_cast<int64_t>(x)) break;
      case 4: HDK_PQ_ROWS(int32_t, static_cast<int64_t>(x)) break;
      default: HDK_PQ_ROWS(int64_t, x) break;
    }
  }
#undef HDK_PQ_ROWS
}

// pass[r] &= every conjunct is TRUE for row[r]
template <int VR>
HDK_DEV void plain_quals_pass(const ProjFastQual* quals, int nquals, const int8_t* const* cols, const int64_t (&row)[VR],
                              bool (&pass)[VR], bool nt) {
  for (int qi = 0; qi < nquals; ++qi) {
    const ProjFastQual q = quals[qi];
    int64_t v[VR];
    plain_load_rows<VR>(cols[q.col.buf_idx], q.col.width, q.col.kind, row, pass, nt, v);
    const bool fpc = q.fp != 0;
    const bool col_fp = q.col_fp != 0;
    const bool nullable = q.nullable != 0;
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const bool isnull = nullable && (col_fp ? bits_to_double(v[r]) == bits_to_double(q.null_val) : v[r] == q.null_val);
      pass[r] = pass[r] && !isnull;
      if (fpc && !col_fp) {
        v[r] = double_to_bits(static_cast<double>(v[r]));
      }
    }
#define HDK_PQ_CMP(OP)                                                                                   \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                                                       \
    pass[r] = pass[r] && (fpc ? (bits_to_double(v[r]) OP bits_to_double(q.rhs)) : (v[r] OP q.rhs));      \
  }
    switch (q.cmp) {
      case HDK_CMP_EQ: HDK_PQ_CMP(==) break;
      case HDK_CMP_NE: HDK_PQ_CMP(!=) break;
      case HDK_CMP_LT: HDK_PQ_CMP(<) break;
      case HDK_CMP_GT: HDK_PQ_CMP(>) break;
      case HDK_CMP_LE: HDK_PQ_CMP(<=) break;
      default: HDK_PQ_CMP(>=) break;
    }
#undef HDK_PQ_CMP
  }
}

}  // namespace hdk
