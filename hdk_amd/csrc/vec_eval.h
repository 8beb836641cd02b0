// vec_eval.h -- the plan interpreter, VR rows per lane at a time.
//
// Same semantics as device_common.h (decoders, *_nullable arithmetic, three-valued filters, join
// probes: reference QE/DecodersImpl.h:30-150, QE/RuntimeFunctions.cpp:49-384,
// QE/GroupByRuntime.cpp:274-366), but every interpreter step is executed once per BATCH of VR rows
// per lane: the wave-uniform work (scalar loads of plan fields, branches on op codes) is amortised
// over VR x 64 rows, per-row differences are handled by selects, and the VR column loads of a leaf
// are issued back to back (lane i of step r reads row base + r*blockDim + i: coalesced).
// All row loops are fully unrolled so the per-row values live in registers.
#pragma once
#include "device_common.h"

namespace hdk {

constexpr int VR = 8;

// The plan is read through the CONSTANT address space: the compiler then knows that none of the
// kernel's stores and atomics can change it, so plan fields are loaded with scalar loads once and kept
// (or re-materialised) instead of being re-read after every LDS atomic -- the batched interpreter is
// bound by exactly those dependent scalar loads otherwise.
#define HDK_CAS __attribute__((address_space(4)))
typedef const HDK_CAS hdk_hip_plan* cplan_t;
typedef const HDK_CAS hdk_hip_expr& cexpr_t;
typedef const HDK_CAS hdk_hip_leaf& cleaf_t;
typedef const HDK_CAS hdk_hip_step& cstep_t;
typedef const HDK_CAS hdk_hip_qual& cqual_t;
typedef const HDK_CAS hdk_hip_join& cjoin_t;
typedef const HDK_CAS hdk_hip_target& ctarget_t;

HDK_DEV bool leaf_is_fp_c(cplan_t p, cleaf_t l) {
  if (l.kind == HDK_LEAF_FP) {
    return true;
  }
  if (l.kind != HDK_LEAF_COL) {
    return false;
  }
  const int32_t k = p->cols[l.col].kind;
  return k == HDK_COL_FLOAT || k == HDK_COL_DOUBLE;
}

HDK_DEV cplan_t to_const_as(const hdk_hip_plan* p) {
  return reinterpret_cast<cplan_t>(reinterpret_cast<uintptr_t>(p));
}
typedef long long __attribute__((ext_vector_type(2))) i64x2;

template <bool J, bool KEYED = false, bool MANY = false>
struct VecCtxT {
  static constexpr bool kJoins = J;  // false: the plan has no joins; all probe state compiles away
  static constexpr bool kKeyed = KEYED;  // some join probes a keyed ("baseline") one-to-one table: a kernel of its own
  // MANY: the plan's ONE join probes a one-to-many perfect-hash table (matching sets, HashJoin::codegenMatchingSet,
  // QE/JoinHashTable/HashJoin.cpp:149-197): the batch is replayed once per match -- round i takes every row's i-th
  // partner (vec_round_v, called by the kernel body); the filters on joined columns run per round
  static constexpr bool kMany = MANY;
  int32_t mpos[MANY ? VR : 1];  // the row's matching set: first position in the table's row-id area, number of partners
  int32_t mcnt[MANY ? VR : 1];
  const int32_t* mids;          // the table's row-id area
  cplan_t plan;
  const int8_t* const* cols;  // col_buffers[frag]
  int64_t row0;               // first row of the tile; slot r of lane `tid` is row0 + r*blk + tid
  int32_t nlive;              // rows of the tile inside the fragment (dead slots re-read row0)
  int32_t tid;
  int32_t blk;
  int32_t jref0[VR];          // join 0 / 1: matched inner row (reference table) or probed slot (fused table)
  int32_t jref1[VR];
  int64_t jpay0[VR];          // payload word 1 of a fused entry: arrives with the row id in one 16-B gather
  int64_t jpay1[VR];
  uint32_t jmiss0, jmiss1;    // bit r: slot r found no partner at join 0 / 1 of a LEFT (or ANTI) join: its inner columns read as NULL
  const int64_t* fused0;      // fused tables (HDK_JOIN_ONE_TO_ONE_FUSED) or nullptr
  const int64_t* fused1;
  int32_t fstride0;
  int32_t fstride1;
};

// outer row of batch slot r (kept as three scalars + the lane id instead of VR 64-bit registers)
template <class VecCtx>
HDK_DEV int64_t vrow(const VecCtx& c, int r) {
  const int32_t o = r * c.blk + c.tid;
  return c.row0 + (o < c.nlive ? o : 0);
}

template <class VecCtx>
HDK_DEV void vec_ctx_init(VecCtx& c, cplan_t p, int tid, int blk) {
  c.plan = p;
  c.tid = tid;
  c.blk = blk;
  c.row0 = 0;
  c.nlive = 0;
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    c.jref0[r] = 0;
    c.jref1[r] = 0;
    c.jpay0[r] = 0;
    c.jpay1[r] = 0;
  }
  c.jmiss0 = 0;
  c.jmiss1 = 0;
  c.mids = nullptr;
  c.fused0 = nullptr;
  c.fused1 = nullptr;
  c.fstride0 = 0;
  c.fstride1 = 0;
}

// start a tile: sets pass[r] = "slot r is a real row"
template <class VecCtx>
HDK_DEV void vec_ctx_tile(VecCtx& c, int64_t row0, int64_t nrows, bool (&pass)[VR]) {
  const int64_t left = nrows - row0;
  c.row0 = row0;
  c.nlive = left > static_cast<int64_t>(VR) * c.blk ? VR * c.blk : static_cast<int32_t>(left);
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    pass[r] = r * c.blk + c.tid < c.nlive;
  }
}

// row of batch slot r in the table a column lives in (0 = outer; 1/2 = matched row of join 0/1)
template <class VecCtx>
HDK_DEV int64_t leaf_row(const VecCtx& c, int table, int r) {
  if (VecCtx::kJoins && table == 1) {
    return c.jref0[r];
  }
  if (VecCtx::kJoins && table == 2) {
    return c.jref1[r];
  }
  return vrow(c, r);
}

// VR rows of one column, decoded; the decoder switch is wave-uniform and sits OUTSIDE the row loop (one scalar branch
// per batch, not per row)
template <class VecCtx, bool NT>
HDK_DEV void load_rows_decoded(const VecCtx& c, const int8_t* __restrict__ buf, int kind, int width, int table, int64_t (&out)[VR]) {
#define HDK_LOAD_ROWS(T, CONV)                  \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) { \
    const T x = gload<T>(buf, leaf_row(c, table, r), NT); \
    out[r] = CONV;                               \
  }
  if (kind == HDK_COL_DOUBLE) {
    HDK_LOAD_ROWS(int64_t, x)
  } else if (kind == HDK_COL_FLOAT) {
    HDK_LOAD_ROWS(float, double_to_bits(static_cast<double>(x)))
  } else if (kind == HDK_COL_SMALL_DATE) {  // fixed_width_small_date_decode: days -> epoch seconds, narrow NULL -> NULL_BIGINT
    if (width == 4) {
      HDK_LOAD_ROWS(int32_t, x == INT32_MIN ? HDK_NULL_BIGINT : static_cast<int64_t>(x) * 86400)
    } else {
      HDK_LOAD_ROWS(int16_t, x == INT16_MIN ? HDK_NULL_BIGINT : static_cast<int64_t>(x) * 86400)
    }
  } else if (kind == HDK_COL_UNSIGNED) {
    switch (width) {
      case 1: HDK_LOAD_ROWS(uint8_t, static_cast<int64_t>(x)) break;
      case 2: HDK_LOAD_ROWS(uint16_t, static_cast<int64_t>(x)) break;
      case 4: HDK_LOAD_ROWS(uint32_t, static_cast<int64_t>(x)) break;
      default: HDK_LOAD_ROWS(int64_t, x) break;
    }
  } else {
    switch (width) {
      case 1: HDK_LOAD_ROWS(int8_t, static_cast<int64_t>(x)) break;
      case 2: HDK_LOAD_ROWS(int16_t, static_cast<int64_t>(x)) break;
      case 4: HDK_LOAD_ROWS(int32_t, static_cast<int64_t>(x)) break;
      default: HDK_LOAD_ROWS(int64_t, x) break;
    }
  }
#undef HDK_LOAD_ROWS
}

template <class VecCtx>
HDK_DEV void load_leaf_raw_v(const VecCtx& c, cleaf_t l, int64_t (&out)[VR]) {
  if (l.kind == HDK_LEAF_COL) {
    const int32_t ci = l.col;
    const int width = c.plan->cols[ci].width;
    const int kind = c.plan->cols[ci].kind;
    const int table = c.plan->cols[ci].table;
    const int col_buf_idx = c.plan->cols[ci].buf_idx;
    const int8_t* __restrict__ buf = table >= 0 ? c.cols[col_buf_idx] : nullptr;
    if (VecCtx::kJoins && table < 0) {  // payload word of a fused join table: same cache line as the probed row id
      const int64_t* __restrict__ ft = table == -1 ? c.fused0 : c.fused1;
      const int64_t stride = table == -1 ? c.fstride0 : c.fstride1;
      const int word = col_buf_idx;
      if (word == 1) {  // already in registers (rows_pass_v)
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          out[r] = table == -1 ? c.jpay0[r] : c.jpay1[r];
        }
        return;
      }
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const int64_t slot = table == -1 ? c.jref0[r] : c.jref1[r];
        out[r] = gload<int64_t>(reinterpret_cast<const int8_t*>(ft), slot * stride + word, false);
      }
      return;
    }
    // outer-table columns are streamed exactly once: non-temporal; inner (joined) columns are gathered
    // repeatedly and should stay cached.  The hint has to be a compile-time constant: with a run-time flag the
    // optimiser merges `nt ? nontemporal : plain` into ONE plain load and the streamed columns evict the join table
    // from L2 and the Infinity Cache (device_common.h: gload).
    if (table == 0) {
      load_rows_decoded<VecCtx, true>(c, buf, kind, width, table, out);
    } else {
      load_rows_decoded<VecCtx, false>(c, buf, kind, width, table, out);
    }
  } else {
    const int64_t v = l.ival;
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      out[r] = v;
    }
  }
}

// A column of an inner table whose row found no partner (LEFT join): NULL, as load_leaf reads join_row == -1
// (codegenOuterJoinNullPlaceholder, QE/ColumnIR.cpp).
template <class VecCtx>
HDK_DEV void load_leaf_v(const VecCtx& c, cleaf_t l, int64_t (&out)[VR]) {
  load_leaf_raw_v(c, l, out);
  if (VecCtx::kJoins && l.kind == HDK_LEAF_COL) {
    const int table = c.plan->cols[l.col].table;
    const uint32_t miss = (table == 1 || table == -1) ? c.jmiss0 : ((table == 2 || table == -2) ? c.jmiss1 : 0u);
    if (__any(miss != 0)) {
      const int64_t nullv = l.null_val;
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        out[r] = (miss >> r) & 1u ? nullv : out[r];
      }
    }
  }
}

// `live[r]`: only live rows may raise ERR_DIV_BY_ZERO (dead rows are evaluated speculatively)
template <class VecCtx>
HDK_DEV void eval_expr_v(const VecCtx& c, cexpr_t e, int64_t (&acc)[VR], const bool (&live)[VR],
                         int32_t& err) {
  cplan_t p = c.plan;
  load_leaf_v(c, e.leaf0, acc);
  bool acc_fp = leaf_is_fp_c(p, e.leaf0);
  int64_t acc_null = e.leaf0.null_val;
  int32_t acc_nullable = e.leaf0.nullable;
  const int nsteps = e.nsteps;
  for (int s = 0; s < nsteps; ++s) {
    cstep_t st = e.steps[s];
    const int op = st.op;
    const bool out_fp = st.out_class == HDK_VC_FP;
    const int64_t null_out = st.null_out;
    if (op <= HDK_OP_MOD) {
      int64_t rhs[VR];
      load_leaf_v(c, st.rhs, rhs);
      const bool rhs_fp = leaf_is_fp_c(p, st.rhs);
      const int64_t rhs_null = st.rhs.null_val;
      const int32_t rhs_nullable = st.rhs.nullable;
      bool isnull[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        isnull[r] = is_null_val(acc[r], acc_null, acc_nullable, acc_fp) ||
                    is_null_val(rhs[r], rhs_null, rhs_nullable, rhs_fp);
      }
      // the op switch is wave-uniform: one scalar dispatch per batch, the row loop inside each case
      if (out_fp) {
        double a[VR], b[VR];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          a[r] = acc_fp ? bits_to_double(acc[r]) : static_cast<double>(acc[r]);
          b[r] = rhs_fp ? bits_to_double(rhs[r]) : static_cast<double>(rhs[r]);
        }
#define HDK_FP_ROWS(EXPR)                                             \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                    \
    acc[r] = isnull[r] ? null_out : double_to_bits(EXPR);             \
  }
        switch (op) {
          case HDK_OP_ADD: HDK_FP_ROWS(a[r] + b[r]) break;
          case HDK_OP_SUB: HDK_FP_ROWS(a[r] - b[r]) break;
          case HDK_OP_MUL: HDK_FP_ROWS(a[r] * b[r]) break;
          default:
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const bool bad = b[r] == 0.0;
              if (bad && !isnull[r] && live[r]) {
                err = HDK_HIP_ERR_DIV_BY_ZERO;
              }
              acc[r] = (isnull[r] || bad) ? null_out : double_to_bits(a[r] / b[r]);
            }
            break;
        }
#undef HDK_FP_ROWS
      } else {
#define HDK_INT_ROWS(EXPR)                                                                   \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                                           \
    const uint64_t a = static_cast<uint64_t>(acc[r]), b = static_cast<uint64_t>(rhs[r]);    \
    acc[r] = isnull[r] ? null_out : static_cast<int64_t>(EXPR);                              \
  }
        const int32_t cw = st.check_width;
        if (cw > 0 && op <= HDK_OP_MUL) {  // the reference's checked + - * (checked_arith, device_common.h)
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            int64_t res;
            const bool ovf = checked_arith(op, acc[r], rhs[r], cw, &res);
            if (ovf && !isnull[r] && live[r]) {
              err = HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW;
            }
            acc[r] = isnull[r] ? null_out : res;
          }
        } else
        switch (op) {
          case HDK_OP_ADD: HDK_INT_ROWS(a + b) break;
          case HDK_OP_SUB: HDK_INT_ROWS(a - b) break;
          case HDK_OP_MUL: HDK_INT_ROWS(a * b) break;
          case HDK_OP_DIV:
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t a = acc[r], b = rhs[r];
              const bool bad = b == 0;
              const int64_t bb = (bad || isnull[r]) ? 1 : b;
              const int64_t res = (a == INT64_MIN && bb == -1) ? INT64_MIN : a / bb;
              if (bad && !isnull[r] && live[r]) {
                err = HDK_HIP_ERR_DIV_BY_ZERO;
              }
              acc[r] = (isnull[r] || bad) ? null_out : res;
            }
            break;
          default:
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t a = acc[r], b = rhs[r];
              const bool bad = b == 0;
              const int64_t bb = (bad || isnull[r]) ? 1 : b;
              const int64_t res = bb == -1 ? 0 : a % bb;
              if (bad && !isnull[r] && live[r]) {
                err = HDK_HIP_ERR_DIV_BY_ZERO;
              }
              acc[r] = (isnull[r] || bad) ? null_out : res;
            }
            break;
        }
#undef HDK_INT_ROWS
      }
    } else {
      const int64_t param = st.rhs.ival;
      bool isnull[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        isnull[r] = is_null_val(acc[r], acc_null, acc_nullable, acc_fp);
      }
#define HDK_UNARY_ROWS(EXPR)                           \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {     \
    acc[r] = isnull[r] ? null_out : (EXPR);            \
  }
      switch (op) {
        case HDK_OP_EXTRACT_YEAR: HDK_UNARY_ROWS(extract_year(acc[r])) break;
        case HDK_OP_SCALE_DOWN: HDK_UNARY_ROWS(scale_decimal_down(acc[r], param)) break;
        case HDK_OP_FLOOR_DIV: HDK_UNARY_ROWS(floor_div_lhs(acc[r], param)) break;
        case HDK_OP_CAST_INT_TO_FP: HDK_UNARY_ROWS(double_to_bits(static_cast<double>(acc[r]))) break;
        case HDK_OP_CAST_FP_TO_INT:
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            const double d = bits_to_double(acc[r]);
            acc[r] = isnull[r] ? null_out : static_cast<int64_t>(d + (d < 0.0 ? -0.5 : 0.5));
          }
          break;
        default: HDK_UNARY_ROWS(null_out) break;
      }
#undef HDK_UNARY_ROWS
    }
    acc_fp = out_fp;
    acc_null = null_out;
    acc_nullable = 1;
  }
}

// pass[r] &= (conjunct is TRUE)
template <class VecCtx>
HDK_DEV void eval_qual3_v(const VecCtx& c, cqual_t q, const bool (&live)[VR], bool (&tv)[VR], bool (&nv)[VR], int32_t& err) {
  int64_t lhs[VR];
  int64_t rhs[VR];
  eval_expr_v(c, q.lhs, lhs, live, err);
  load_leaf_v(c, q.rhs, rhs);
  const bool lhs_fp = q.lhs.vclass == HDK_VC_FP;
  const bool rhs_fp = leaf_is_fp_c(c.plan, q.rhs);
  const int cmp = q.cmp;
  const int64_t lnull = q.lhs.null_val, rnull = q.rhs.null_val;
  const int32_t lnullable = q.lhs.nullable, rnullable = q.rhs.nullable;
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    const bool isnull = is_null_val(lhs[r], lnull, lnullable, lhs_fp) || is_null_val(rhs[r], rnull, rnullable, rhs_fp);
    bool t;
    if (lhs_fp || rhs_fp) {
      const double a = lhs_fp ? bits_to_double(lhs[r]) : static_cast<double>(lhs[r]);
      const double b = rhs_fp ? bits_to_double(rhs[r]) : static_cast<double>(rhs[r]);
      switch (cmp) {
        case HDK_CMP_EQ: t = a == b; break;
        case HDK_CMP_NE: t = a != b; break;
        case HDK_CMP_LT: t = a < b; break;
        case HDK_CMP_GT: t = a > b; break;
        case HDK_CMP_LE: t = a <= b; break;
        default: t = a >= b; break;
      }
    } else {
      const int64_t a = lhs[r], b = rhs[r];
      switch (cmp) {
        case HDK_CMP_EQ: t = a == b; break;
        case HDK_CMP_NE: t = a != b; break;
        case HDK_CMP_LT: t = a < b; break;
        case HDK_CMP_GT: t = a > b; break;
        case HDK_CMP_LE: t = a <= b; break;
        default: t = a >= b; break;
      }
    }
    tv[r] = !isnull && t;  // TRUE; a NULL operand makes the comparison NULL (DEF_CMP_NULLABLE)
    nv[r] = isnull;
  }
}

// one conjunct: the row stays when the comparison is TRUE
template <class VecCtx>
HDK_DEV void eval_qual_v(const VecCtx& c, cqual_t q, bool (&pass)[VR], int32_t& err) {
  bool tv[VR], nv[VR];
  eval_qual3_v(c, q, pass, tv, nv, err);
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    pass[r] = pass[r] && tv[r];
  }
}

// The filter as a postfix program over the comparisons (hdk_hip_plan::filter_ops: OR / NOT trees) with the reference's
// three-valued logical_and / logical_or / logical_not (QE/RuntimeFunctions.cpp:357-384) -- filter_program_pass
// (device_common.h) for the VR rows of a batch: per row two bit masks hold the value stack (bit i of t: entry i is TRUE; of
// n: NULL), the stack pointer and the operator are wave-uniform.
template <class VecCtx>
HDK_DEV void filter_program_pass_v(const VecCtx& c, bool (&pass)[VR], int32_t& err) {
  cplan_t p = c.plan;
  uint32_t t[VR], n[VR];
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    t[r] = 0;
    n[r] = 0;
  }
  int sp = 0;
  const int nops = p->num_filter_ops;
  for (int i = 0; i < nops; ++i) {
    const uint32_t op = p->filter_ops[i];
    if (op < HDK_F_AND) {
      bool tv[VR], nv[VR];
      eval_qual3_v(c, p->quals[op], pass, tv, nv, err);
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        t[r] = (t[r] & ~(1u << sp)) | (static_cast<uint32_t>(tv[r]) << sp);
        n[r] = (n[r] & ~(1u << sp)) | (static_cast<uint32_t>(nv[r]) << sp);
      }
      ++sp;
    } else if (op == HDK_F_NOT) {
      const uint32_t m = 1u << (sp - 1);
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        t[r] = (t[r] & ~m) | (~(t[r] | n[r]) & m);  // NULL stays NULL, TRUE <-> FALSE
      }
    } else {
      const bool is_and = op == HDK_F_AND;
      const uint32_t m = 1u << (sp - 2);
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint32_t ta = (t[r] >> (sp - 2)) & 1u, tb = (t[r] >> (sp - 1)) & 1u;
        const uint32_t na = (n[r] >> (sp - 2)) & 1u, nb = (n[r] >> (sp - 1)) & 1u;
        uint32_t tr, nr;
        if (is_and) {
          const uint32_t fa = (ta | na) ^ 1u, fb = (tb | nb) ^ 1u;  // FALSE operands
          tr = ta & tb;
          nr = (tr | fa | fb) ^ 1u;
        } else {
          tr = ta | tb;
          nr = (tr ^ 1u) & (na | nb);
        }
        t[r] = (t[r] & ~m) | (tr << (sp - 2));
        n[r] = (n[r] & ~m) | (nr << (sp - 2));
      }
      sp -= 1;
    }
  }
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    pass[r] = pass[r] && sp == 1 && (t[r] & 1u);
  }
}

// filter + join probes for the batch; dead slots stay dead
// Probes of a keyed one-to-one table for the VR rows of a batch: baseline_hash_join_idx_{32,64}
// (JoinHashTableQueryRuntime.cpp:42-98) -- MurmurHash1 over the key components, linear probing over
// [components | row id] entries; idx = -1: not present.  The probe sequences of the VR rows advance TOGETHER: every trip
// issues the (independent) entry loads of all rows still searching, so a batch costs as many memory round trips as its
// longest chain -- one row after the other it was the SUM of the chains' lengths, each a dependent load
// (64 M probes: 6.7 ms that way, no faster than the row-at-a-time interpreter).
template <typename T>
HDK_DEV void keyed_probe_batch(cjoin_t jn, const int8_t* table, const int64_t (&k0)[VR], const int64_t (&k1)[VR],
                               const int64_t (&k2)[VR], const bool (&pass)[VR], int64_t (&idx)[VR]) {
  const int kc = jn.key_component_count;
  const uint32_t entries = static_cast<uint32_t>(jn.entry_count);
  const T invalid = sizeof(T) == 8 ? static_cast<T>(HDK_EMPTY_KEY_64) : static_cast<T>(HDK_EMPTY_KEY_32);
  const int comps = kc + 1;
  const T* dict = reinterpret_cast<const T*>(table);
  uint32_t hp[VR], h0[VR];
  bool act[VR];
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    idx[r] = -1;
    act[r] = pass[r] && entries != 0;
    uint32_t words[2 * HDK_HIP_MAX_JOIN_KEYS];
    int nw = 0;
    const int64_t kk[HDK_HIP_MAX_JOIN_KEYS] = {k0[r], k1[r], k2[r]};
#pragma unroll
    for (int i = 0; i < HDK_HIP_MAX_JOIN_KEYS; ++i) {
      if (i < kc) {
        if constexpr (sizeof(T) == 8) {
          words[nw++] = static_cast<uint32_t>(static_cast<uint64_t>(kk[i]));
          words[nw++] = static_cast<uint32_t>(static_cast<uint64_t>(kk[i]) >> 32);
        } else {
          words[nw++] = static_cast<uint32_t>(kk[i]);
        }
      }
    }
    h0[r] = act[r] ? murmur_hash1_words(words, nw) % entries : 0u;
    hp[r] = h0[r];
  }
  for (;;) {
    bool any = false;
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      any = any || act[r];
    }
    if (!__any(any)) {
      break;
    }
    // (two consecutive entries per row and trip: 4.47 ms per 64 M probes against 4.16 with one)
    T e0[VR], e1[VR], e2[VR], er[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {  // all loads of the trip first
      const T* e = dict + static_cast<size_t>(act[r] ? hp[r] : 0u) * comps;
      e0[r] = e[0];
      e1[r] = kc > 1 ? e[1] : static_cast<T>(0);
      e2[r] = kc > 2 ? e[2] : static_cast<T>(0);
      er[r] = e[kc];
    }
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      if (act[r]) {
        const bool eq = e0[r] == static_cast<T>(k0[r]) && (kc < 2 || e1[r] == static_cast<T>(k1[r])) &&
                        (kc < 3 || e2[r] == static_cast<T>(k2[r]));
        if (eq) {
          idx[r] = static_cast<int64_t>(er[r]);
          act[r] = false;
        } else if (e0[r] == invalid) {
          act[r] = false;  // kNotPresent
        } else {
          hp[r] = hp[r] + 1 == entries ? 0u : hp[r] + 1;
          act[r] = hp[r] != h0[r];
        }
      }
    }
  }
}

template <class VecCtx>
HDK_DEV void rows_pass_v(VecCtx& c, const int64_t* join_hash_tables, bool (&pass)[VR], int32_t& err) {
  cplan_t p = c.plan;
  const int nq = p->num_quals;
  const bool program = p->num_filter_ops != 0;  // OR / NOT: one program, before the joins or (it reads a joined column) after
  if (program) {
    if (!VecCtx::kJoins || !p->filter_after_joins) {
      filter_program_pass_v(c, pass, err);
    }
  } else {
    for (int q = 0; q < nq; ++q) {
      if (!VecCtx::kJoins || !p->quals[q].after_joins) {
        eval_qual_v(c, p->quals[q], pass, err);
      }
    }
  }
  const int nj = VecCtx::kJoins ? p->num_joins : 0;
  if (VecCtx::kJoins) {
    c.jmiss0 = 0;
    c.jmiss1 = 0;
  }
  for (int j = 0; j < nj; ++j) {
    cjoin_t jn = p->joins[j];
    const bool anti = jn.type == HDK_JOIN_ANTI;  // the row goes on when the probe finds NO partner (JoinLoop.cpp:258-262)
    uint32_t miss = 0;
    int64_t key[VR];
    eval_expr_v(c, jn.outer_key, key, pass, err);
    const int32_t* __restrict__ table = (nj == 1 && jn.table_idx == 0)
                                            ? reinterpret_cast<const int32_t*>(join_hash_tables)
                                            : reinterpret_cast<const int32_t*>(join_hash_tables[jn.table_idx]);
    const bool inner = jn.type == HDK_JOIN_INNER || jn.type == HDK_JOIN_SEMI;  // (SEMI: INNER over a first-row-wins table)
    if constexpr (VecCtx::kMany) {
      // [first position | count | row ids] (fill_one_to_many_hash_table): two probes of the slot, no row id yet
      const int64_t ec = jn.entry_count;
      c.mids = table + 2 * ec;
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        int64_t slot;
        const int64_t pos = probe_join_g(jn, table, pass[r] ? key[r] : jn.min_key, &slot);
        int32_t n = 0;
        if (pos >= 0) {
          n = gload<int32_t>(reinterpret_cast<const int8_t*>(table), ec + slot, false);
        }
        c.mpos[r] = pos < 0 ? 0 : static_cast<int32_t>(pos);
        c.mcnt[r] = pass[r] ? n : 0;
        if (inner) {
          pass[r] = pass[r] && n > 0;
        }  // (LEFT: a row without partners takes ONE round with its inner columns NULL; ANTI: only such rows go on)
        c.jref0[r] = 0;
      }
      return;  // the rounds (and the filters on joined columns) are the caller's: vec_round_v
    }
    if constexpr (VecCtx::kKeyed) {
      if (jn.kind == HDK_JOIN_KEYED_ONE_TO_ONE) {
        // composite / wide keys: the key expressions for the batch, then one probe per live row (a rejected row
        // does not probe at all: the loop is divergent anyway)
        int64_t k1[VR], k2[VR], idx[VR];
        const int kc = jn.key_component_count;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          k1[r] = 0;
          k2[r] = 0;
        }
        if (kc > 1) {
          eval_expr_v(c, jn.extra_keys[0], k1, pass, err);
        }
        if (kc > 2) {
          eval_expr_v(c, jn.extra_keys[1], k2, pass, err);
        }
        if (jn.key_component_width == 4) {
          keyed_probe_batch<int32_t>(jn, reinterpret_cast<const int8_t*>(table), key, k1, k2, pass, idx);
        } else {
          keyed_probe_batch<int64_t>(jn, reinterpret_cast<const int8_t*>(table), key, k1, k2, pass, idx);
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (inner) {
            pass[r] = pass[r] && idx[r] >= 0;
          } else {
            pass[r] = pass[r] && !(anti && idx[r] >= 0);
            miss |= idx[r] < 0 ? 1u << r : 0u;
          }
          const int32_t ref = idx[r] < 0 ? 0 : static_cast<int32_t>(idx[r]);
          if (j == 0) {
            c.jref0[r] = ref;
          } else {
            c.jref1[r] = ref;
          }
        }
        if (j == 0) {
          c.jmiss0 = miss;
        } else {
          c.jmiss1 = miss;
        }
        continue;
      }
    }
    const bool fused = jn.kind == HDK_JOIN_ONE_TO_ONE_FUSED;
    if (fused) {
      if (j == 0) {
        c.fused0 = reinterpret_cast<const int64_t*>(table);
        c.fstride0 = jn.fused_stride;
      } else {
        c.fused1 = reinterpret_cast<const int64_t*>(table);
        c.fstride1 = jn.fused_stride;
      }
    }
    const int64_t fstride = jn.fused_stride;
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      // the loads stay unconditional, but a row the outer filters already rejected probes SLOT 0 (one cached line
      // for all of them) instead of fetching its own 128-byte line from memory: with a 5 % filter in front of the join
      // (bench_configs pj) the probes, not the scan, were the whole cost
      int64_t slot;
      int64_t idx;
      if (fused) {
        // the slot computation of probe_join_g, but the row id is word 0 of the 8*(1+k)-byte entry
        int64_t k = key[r];
        int64_t maxk = jn.max_key;
        bool in_range = true;
        if (jn.null_mode != HDK_JOIN_NULL_NONE && k == jn.null_val) {
          if (jn.null_mode == HDK_JOIN_NULL_NULLABLE) {
            in_range = false;
          }
          k = jn.translated_null;
          maxk = jn.translated_null;
        }
        in_range = in_range && k >= jn.min_key && k <= maxk;
        slot = (in_range && pass[r]) ? (jn.bucket > 1 ? (k - jn.min_key) / jn.bucket : (k - jn.min_key)) : 0;
        // one line fetch per probe: the row id and the first payload word come in together (a later,
        // separate payload gather would find the line evicted again -- 16 waves x 512 probes in flight
        // per CU dwarf the L1)
        int64_t rid;
        int64_t pay = 0;
        if ((fstride & 1) == 0) {
          const i64x2 e = gload<i64x2>(reinterpret_cast<const int8_t*>(table), slot * (fstride >> 1), false);
          rid = e.x;
          pay = e.y;
        } else {
          rid = gload<int64_t>(reinterpret_cast<const int8_t*>(table), slot * fstride, false);
          if (fstride > 1) {
            pay = gload<int64_t>(reinterpret_cast<const int8_t*>(table), slot * fstride + 1, false);
          }
        }
        if (j == 0) {
          c.jpay0[r] = pay;
        } else {
          c.jpay1[r] = pay;
        }
        idx = in_range ? rid : -1;
      } else {
        idx = probe_join_g(jn, table, pass[r] ? key[r] : jn.min_key, &slot);
      }
      if (inner) {
        pass[r] = pass[r] && idx >= 0;
      } else {  // LEFT: the row stays, its inner columns are NULL; ANTI: only such rows stay
        pass[r] = pass[r] && !(anti && idx >= 0);
        miss |= idx < 0 ? 1u << r : 0u;
      }
      const int32_t ref = fused ? static_cast<int32_t>(slot) : (idx < 0 ? 0 : static_cast<int32_t>(idx));
      if (j == 0) {
        c.jref0[r] = ref;
      } else {
        c.jref1[r] = ref;
      }
    }
    if (j == 0) {
      c.jmiss0 = miss;
    } else {
      c.jmiss1 = miss;
    }
  }
  if (VecCtx::kJoins) {  // filters that read joined columns
    if (program) {
      if (p->filter_after_joins) {
        filter_program_pass_v(c, pass, err);
      }
    } else {
      for (int q = 0; q < nq; ++q) {
        if (p->quals[q].after_joins) {
          eval_qual_v(c, p->quals[q], pass, err);
        }
      }
    }
  }
}

// Round `round` of a batch whose join is a matching-set join (VecCtx::kMany; rows_pass_v has run): live[r] = row r takes part
// in this round -- it has a `round`-th partner, or it is a LEFT / ANTI row without any taking its single round -- and the
// join's row references / miss mask are set for it; then the filters on joined columns.  Returns whether any row of the wave
// is still in (JoinLoop's Set form, QE/IRCodegen.cpp:497-667: the loop body once per match).
template <class VecCtx>
HDK_DEV bool vec_round_v(VecCtx& c, int round, const bool (&pass)[VR], bool (&live)[VR], int32_t& err) {
  cplan_t p = c.plan;
  cjoin_t jn = p->joins[0];
  const bool left = jn.type == HDK_JOIN_LEFT, anti = jn.type == HDK_JOIN_ANTI;
  uint32_t miss = 0;
  bool any = false;
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    const int32_t n = c.mcnt[r];
    const bool lone = (left || anti) && n == 0 && round == 0;  // no partner: one round, inner columns NULL
    live[r] = pass[r] && ((!anti && round < n) || lone);
    miss |= lone ? 1u << r : 0u;
    int32_t rid = 0;
    if (live[r] && !lone) {
      rid = gload<int32_t>(reinterpret_cast<const int8_t*>(c.mids), static_cast<int64_t>(c.mpos[r]) + round, false);
    }
    c.jref0[r] = rid;
    any = any || live[r];
  }
  c.jmiss0 = miss;
  if (!__any(any)) {
    return false;
  }
  if (p->num_filter_ops) {
    if (p->filter_after_joins) {
      filter_program_pass_v(c, live, err);
    }
  } else {
    const int nq = p->num_quals;
    for (int q = 0; q < nq; ++q) {
      if (p->quals[q].after_joins) {
        eval_qual_v(c, p->quals[q], live, err);
      }
    }
  }
  return true;
}

// group key #k for the batch (perfect hash: NULL translated)
template <class VecCtx>
HDK_DEV void eval_key_v(const VecCtx& c, int k, int64_t (&out)[VR], const bool (&live)[VR], int32_t& err) {
  cplan_t p = c.plan;
  eval_expr_v(c, p->keys[k], out, live, err);
  if (p->query_kind == HDK_Q_PERFECT_HASH && p->key_has_nulls[k] && p->keys[k].nullable) {
    const int64_t nullv = p->keys[k].null_val;
    const int64_t tr = p->key_null_translated[k];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      out[r] = out[r] == nullv ? tr : out[r];
    }
  }
}

template <class VecCtx>
HDK_DEV void perfect_hash_entry_v(const VecCtx& c, int64_t (&entry)[VR], const bool (&live)[VR], int32_t& err) {
  cplan_t p = c.plan;
  const int nk = p->key_count;
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    entry[r] = 0;
  }
  int64_t stride = 1;
  for (int k = 0; k < nk; ++k) {
    int64_t kv[VR];
    eval_key_v(c, k, kv, live, err);
    const int64_t kmin = p->key_min[k];
    const int64_t bucket = p->key_bucket[k];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      int64_t term = kv[r] - kmin;
      if (bucket) {
        term /= bucket;
      }
      entry[r] += term * stride;
    }
    stride *= p->key_card[k];
  }
}

// target argument for the batch; is_null[r] = the value is the skip value
template <class VecCtx>
HDK_DEV void eval_target_arg_v(const VecCtx& c, ctarget_t tg, int64_t (&v)[VR], bool (&is_null)[VR],
                               const bool (&live)[VR], int32_t& err) {
  if (!tg.has_arg) {
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      v[r] = 0;
      is_null[r] = false;
    }
    return;
  }
  eval_expr_v(c, tg.arg, v, live, err);
  const bool arg_fp = tg.arg.vclass == HDK_VC_FP;
  const bool skip = tg.skip_null && tg.agg != HDK_AGG_ID;
  const bool promote = tg.arg_is_fp && !arg_fp && tg.agg != HDK_AGG_ID;
  const int64_t anull = tg.arg.null_val;
  const int32_t anullable = tg.arg.nullable;
  const int64_t tnull = tg.null_val;
  const bool slot_fp = tg.arg_is_fp;
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    bool n = false;
    int64_t x = v[r];
    if (skip && is_null_val(x, anull, anullable, arg_fp)) {
      n = true;
      x = tnull;
    } else {
      if (promote) {
        x = double_to_bits(static_cast<double>(x));
      }
      if (skip) {
        n = slot_fp ? (bits_to_double(x) == bits_to_double(tnull)) : (x == tnull);
      }
    }
    v[r] = x;
    is_null[r] = n;
  }
}

}  // namespace hdk
