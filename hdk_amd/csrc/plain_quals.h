// plain_quals.h -- the filter form the specialised kernels take: a conjunction of
// `outer column <cmp> literal` with the reference's three-valued semantics (DEF_CMP_NULLABLE,
// QE/RuntimeFunctions.cpp:83-117: a NULL operand fails the conjunct), evaluated for VR rows per lane
// with every wave-uniform decision (decoder, operator, fp/int) outside the row loops.
// Round 5: the same leaves under a postfix AND / OR / NOT program (hdk_hip_plan::filter_ops; logical_and / logical_or /
// logical_not, QE/RuntimeFunctions.cpp:357-384) -- `WHERE a < 0 OR k = 3` no longer sends a plan to the interpreter.  Every
// leaf is evaluated for every live row into two bit masks per lane (bit r: TRUE / NULL for row r), the program combines
// masks: no per-row control flow, a stack of three values in registers (at most kMaxPlainQuals leaves).
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
  // the filter's postfix program, carried by quals[0] (nprog == 0: the plain conjunction of all leaves)
  int32_t nprog;
  uint8_t prog[12];
};
constexpr int kMaxPlainProg = 12;

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

// pass[r] &= the program over the leaves is TRUE for row r.  load(qi, q, live, v): leaf qi's column values of the VR rows,
// int64-widened (double bits for an fp column) -- from memory by row number (plain_load_rows), or from registers a kernel
// already holds (scan_bhm.h: the filter columns ride in the tile's batch of 16-byte loads).
template <int VR, class LoadFn>
HDK_DEV void plain_quals_program_with(const ProjFastQual* quals, int nquals, LoadFn load, bool (&pass)[VR]) {
  constexpr int NW = (VR + 31) / 32;  // mask words per value: one bit per row
  uint32_t lt[kMaxPlainQuals][NW], ln[kMaxPlainQuals][NW];  // leaf qi: bit r = TRUE / NULL for row r
#pragma unroll
  for (int qi = 0; qi < kMaxPlainQuals; ++qi) {
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      lt[qi][w] = 0;
      ln[qi][w] = 0;
    }
    if (qi < nquals) {
      const ProjFastQual q = quals[qi];
      int64_t v[VR];
      load(qi, q, pass, v);
      const bool fpc = q.fp != 0, col_fp = q.col_fp != 0, nullable = q.nullable != 0;
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const bool isnull = nullable && (col_fp ? bits_to_double(v[r]) == bits_to_double(q.null_val) : v[r] == q.null_val);
        ln[qi][r / 32] |= isnull ? 1u << (r % 32) : 0u;
        if (fpc && !col_fp) {
          v[r] = double_to_bits(static_cast<double>(v[r]));
        }
      }
#define HDK_PQ_BITS(OP)                                                                                                     \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                                                                          \
    lt[qi][r / 32] |= (fpc ? (bits_to_double(v[r]) OP bits_to_double(q.rhs)) : (v[r] OP q.rhs)) ? 1u << (r % 32) : 0u;       \
  }
      switch (q.cmp) {
        case HDK_CMP_EQ: HDK_PQ_BITS(==) break;
        case HDK_CMP_NE: HDK_PQ_BITS(!=) break;
        case HDK_CMP_LT: HDK_PQ_BITS(<) break;
        case HDK_CMP_GT: HDK_PQ_BITS(>) break;
        case HDK_CMP_LE: HDK_PQ_BITS(<=) break;
        default: HDK_PQ_BITS(>=) break;
      }
#undef HDK_PQ_BITS
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        lt[qi][w] &= ~ln[qi][w];  // (a NULL operand: the comparison is NULL, not TRUE)
      }
    }
  }
  // the value stack: a = top, b, c below it (wave-uniform program, per-lane masks)
  uint32_t at[NW], an[NW], bt[NW], bn[NW], ct[NW], cn[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    at[w] = an[w] = bt[w] = bn[w] = ct[w] = cn[w] = 0;
  }
  const int nprog = quals[0].nprog;
  for (int i = 0; i < nprog; ++i) {
    const uint32_t op = quals[0].prog[i];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (op < HDK_F_AND) {
        ct[w] = bt[w]; cn[w] = bn[w];
        bt[w] = at[w]; bn[w] = an[w];
        at[w] = op == 0 ? lt[0][w] : (op == 1 ? lt[1][w] : lt[2][w]);
        an[w] = op == 0 ? ln[0][w] : (op == 1 ? ln[1][w] : ln[2][w]);
      } else if (op == HDK_F_NOT) {
        at[w] = ~(at[w] | an[w]);  // NULL stays NULL, TRUE <-> FALSE
      } else {
        uint32_t rt, rn;
        if (op == HDK_F_AND) {
          const uint32_t fa = ~(at[w] | an[w]), fb = ~(bt[w] | bn[w]);  // FALSE operands
          rt = at[w] & bt[w];
          rn = ~(rt | fa | fb);
        } else {
          rt = at[w] | bt[w];
          rn = ~rt & (an[w] | bn[w]);
        }
        at[w] = rt; an[w] = rn;
        bt[w] = ct[w]; bn[w] = cn[w];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    pass[r] = pass[r] && ((at[r / 32] >> (r % 32)) & 1u);
  }
}

// pass[r] &= every conjunct is TRUE for row[r].  PROG: the kernel's matcher accepts AND / OR / NOT programs
// (match_plain_quals(..., allow_program = true)) -- only those kernels carry the program's code: compiled into the radix
// scatters too it cost them registers and a resident block (C5 12.5 -> 15.1 ms per 1 B rows, the perfect-hash partitions 4.1 ->
// 4.9 ms per 256 M) although no plan of theirs has one
template <int VR, bool PROG, class LoadFn>
HDK_DEV void plain_quals_pass_with(const ProjFastQual* quals, int nquals, LoadFn load, bool (&pass)[VR]) {
  if (PROG && nquals > 0 && quals[0].nprog != 0) {  // (wave-uniform)
    plain_quals_program_with<VR>(quals, nquals, load, pass);
    return;
  }
  for (int qi = 0; qi < nquals; ++qi) {
    const ProjFastQual q = quals[qi];
    int64_t v[VR];
    load(qi, q, pass, v);
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

// the leaves' columns read from memory by row number
template <int VR, bool PROG = false>
HDK_DEV void plain_quals_pass(const ProjFastQual* quals, int nquals, const int8_t* const* cols, const int64_t (&row)[VR],
                              bool (&pass)[VR], bool nt) {
  plain_quals_pass_with<VR, PROG>(
      quals, nquals,
      [&](int, const ProjFastQual& q, const bool (&live)[VR], int64_t (&v)[VR]) {
        plain_load_rows<VR>(cols[q.col.buf_idx], q.col.width, q.col.kind, row, live, nt, v);
      },
      pass);
}

}  // namespace hdk
