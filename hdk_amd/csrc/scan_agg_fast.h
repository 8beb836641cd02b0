// scan_agg_fast.h -- specialised scan/aggregate kernels for the shapes the headline configs use.
//
// Shape ("direct columns"): no filter, no join; zero or one group-by key that is a plain integer
// column (perfect hash, with or without NULL translation); every aggregate argument is the same
// plain column (or COUNT(*) / the projected key).  That is BASELINE C1 (SELECT SUM(a)), C2
// (GROUP BY key, SUM(val)), taxi Q1 (GROUP BY cab_type, COUNT(*)) and Q2 (GROUP BY passenger_count,
// AVG(total_amount)).  Everything data-dependent per row is template/SGPR resident:
//   * column widths are template parameters, so each lane reads 16 B (R rows) of the widest column
//     per load instruction, fully coalesced (lane i -> bytes [16 i, 16 i + 16) of the step), U steps
//     issued back to back before the first use (8 x 16 B in flight per lane for two int64 columns);
//   * the per-row LDS update list (which words get +1 / +val / min / max) is a handful of SGPRs;
//   * for <= 64 entries the "entry touched" / "saw a non-NULL" facts are two 64-bit register masks
//     per lane instead of LDS counters, leaving ONE ds_add_u64 per row for C2;
//   * non-grouped plans accumulate in registers and touch LDS once per lane.
// The slab format written at the end is the generic one (agg_common.h), so hdk_finalize is shared.
#pragma once
#include <type_traits>
#include "watch.h"
#include "agg_common.h"
#include "plain_quals.h"

namespace hdk {

constexpr int kFastBlock = 256;
constexpr int kFastMaxOps = 8;

enum FastOpKind : int32_t {
  FOP_ADD_ONE = 0,        // rowcount: +1 for every row
  FOP_ADD_ONE_IF_NULL,    // NULL count: +1 when the value IS NULL (the flush stores rowcount - NULLs)
  FOP_ADD_U64,            // += value (non-NULL rows)
  FOP_ADD_F64,
  FOP_MIN_I64,
  FOP_MAX_I64,
  FOP_MIN_F64,
  FOP_MAX_F64
};

struct FastArgs {
  KernParams kp;
  int64_t* slabs;
  int64_t key_min;
  int64_t key_null;             // in-band NULL of the key column (widened)
  int64_t key_null_translated;
  int64_t val_null;             // in-band NULL of the value column (widened / double bits)
  uint32_t entry_count;
  uint32_t rep;
  int32_t wpe;
  int32_t key_buf_idx;
  int32_t val_buf_idx;
  int32_t key_translate_null;   // perfect hash with NULL keys
  int32_t val_nullable;         // skip NULL values
  int32_t val_is_fp;            // value column is double
  int32_t mask_mode;            // entry_count <= 64 and no real counts needed
  int32_t rowcount_word_is_flag;  // (mask mode) word 0 receives 0/1
  int32_t nops;
  int32_t op_kind[kFastMaxOps];
  int32_t op_word[kFastMaxOps];
  int32_t wop[kMaxWordsPerEntry];  // combine op per word (flush)
  int32_t nn_words[HDK_HIP_MAX_TARGETS];  // (mask mode) words that receive the non-null flag
  int32_t n_nn_words;
  uint32_t nword_mask;  // (counting mode) bit w: word w holds a NULL count
  int32_t nquals;       // plain filters `outer column cmp literal` (Q instantiations only)
  ProjFastQual q[kMaxPlainQuals];
  // ---- X instantiations (round 4): up to two EXTRA 8-byte integer columns streamed like the value column (16-byte loads,
  // issued with the tile's other loads), so that
  //   * the aggregate argument may be an expression over the value column a:  a op e  or  a op literal, op in + - *, with the
  //     reference's NULL propagation and overflow check (DEF_ARITH_NULLABLE, QE/ArithmeticIR.cpp:277-520);
  //   * filters may compare any two of {a, the extra columns, the key column (8-byte keys)} or one of them with a literal
  //     (DEF_CMP_NULLABLE: a NULL on either side fails the conjunct).
  // Operands are named by source: 0 = a, 1 / 2 = extra column 0 / 1, 3 = the key column, 4 = a literal.
  int32_t nx;               // extra streamed columns
  int32_t x_buf_idx[2];
  int32_t nxq;              // filters in operand form (nquals is 0 then)
  struct XQual {
    int32_t lhs_src, rhs_src;
    int32_t cmp;            // hdk_hip_cmp
    int32_t lhs_nullable, rhs_nullable;
    int32_t pad_;
    int64_t lhs_null, rhs_null, lit;
  } xq[kMaxPlainQuals];
  int32_t nxprog;           // > 0: the filters are the leaves of this postfix AND / OR / NOT program (hdk_hip_plan::filter_ops)
  uint8_t xprog[kMaxPlainProg];
  int32_t vform;            // 0: the value column itself; 1: a op extra column b_src; 2: a op literal
  int32_t vop;              // HDK_OP_ADD / SUB / MUL
  int32_t v_check_width;
  int32_t b_src;            // 1 / 2
  int32_t a_nullable, b_nullable;
  int64_t a_null, b_null;
  int64_t v_lit;
  int64_t v_null_out;       // NULL of the expression's result (= val_null: what the row body skips)
};

// operand of an X-mode filter / expression
HDK_DEV int64_t fast_x_src(int32_t src, int64_t a, int64_t e0, int64_t e1, int64_t key, int64_t lit) {
  return src == 0 ? a : (src == 1 ? e0 : (src == 2 ? e1 : (src == 3 ? key : lit)));
}

// X-mode filters under an AND / OR / NOT program (FastArgs::xprog), for the N rows of a lane's tile at once: every leaf into
// two bit masks (bit j: TRUE / NULL for row j), the program once over masks -- three-valued logical_and / logical_or /
// logical_not (QE/RuntimeFunctions.cpp:357-384), a stack of three values.  (Row by row the wave-uniform program loop ran once
// per row: `WHERE val < 0 OR key = 3` took 2.8 ms per 256 M rows, twice the gathered form.)  Returns the rows that pass.
template <int N>
HDK_DEV uint32_t fast_x_program(const FastArgs& a, const int64_t (&va)[N], const int64_t (&e0)[N], const int64_t (&e1)[N],
                                const int64_t (&key)[N]) {
  static_assert(N <= 32, "one bit per row");
  uint32_t lt[kMaxPlainQuals], ln[kMaxPlainQuals];
#pragma unroll
  for (int q = 0; q < kMaxPlainQuals; ++q) {
    lt[q] = 0;
    ln[q] = 0;
    if (q < a.nxq) {
      const FastArgs::XQual xq = a.xq[q];
      uint32_t t = 0, n = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int64_t l = fast_x_src(xq.lhs_src, va[j], e0[j], e1[j], key[j], 0);
        const int64_t r = fast_x_src(xq.rhs_src, va[j], e0[j], e1[j], key[j], xq.lit);
        const bool isnull = (xq.lhs_nullable && l == xq.lhs_null) || (xq.rhs_nullable && r == xq.rhs_null);
        const bool c = xq.cmp == HDK_CMP_EQ ? l == r
                     : (xq.cmp == HDK_CMP_NE ? l != r
                     : (xq.cmp == HDK_CMP_LT ? l < r : (xq.cmp == HDK_CMP_GT ? l > r : (xq.cmp == HDK_CMP_LE ? l <= r : l >= r))));
        t |= c ? 1u << j : 0u;
        n |= isnull ? 1u << j : 0u;
      }
      lt[q] = t & ~n;
      ln[q] = n;
    }
  }
  uint32_t at = 0, an = 0, bt = 0, bn = 0, ct = 0, cn = 0;
  for (int i = 0; i < a.nxprog; ++i) {
    const uint32_t op = a.xprog[i];
    if (op < HDK_F_AND) {
      ct = bt; cn = bn;
      bt = at; bn = an;
      at = op == 0 ? lt[0] : (op == 1 ? lt[1] : lt[2]);
      an = op == 0 ? ln[0] : (op == 1 ? ln[1] : ln[2]);
    } else if (op == HDK_F_NOT) {
      at = ~(at | an);
    } else {
      uint32_t rt, rn;
      if (op == HDK_F_AND) {
        const uint32_t fa = ~(at | an), fb = ~(bt | bn);
        rt = at & bt;
        rn = ~(rt | fa | fb);
      } else {
        rt = at | bt;
        rn = ~rt & (an | bn);
      }
      at = rt; an = rn;
      bt = ct; bn = cn;
    }
  }
  return at;
}

// the X-mode filters of one row: true when every conjunct is TRUE (programs: fast_x_program)
HDK_DEV bool fast_x_pass(const FastArgs& a, int64_t va, int64_t e0, int64_t e1, int64_t key) {
  bool pass = true;
  const int n = a.nxq;
  for (int q = 0; q < n; ++q) {
    const FastArgs::XQual& xq = a.xq[q];
    const int64_t l = fast_x_src(xq.lhs_src, va, e0, e1, key, 0);
    const int64_t r = fast_x_src(xq.rhs_src, va, e0, e1, key, xq.lit);
    const bool isnull = (xq.lhs_nullable && l == xq.lhs_null) || (xq.rhs_nullable && r == xq.rhs_null);
    bool c;
    switch (xq.cmp) {
      case HDK_CMP_EQ: c = l == r; break;
      case HDK_CMP_NE: c = l != r; break;
      case HDK_CMP_LT: c = l < r; break;
      case HDK_CMP_GT: c = l > r; break;
      case HDK_CMP_LE: c = l <= r; break;
      default: c = l >= r; break;
    }
    pass = pass && c && !isnull;
  }
  return pass;
}

// the X-mode aggregate argument of one row (eval_expr's one integer step)
HDK_DEV int64_t fast_x_value(const FastArgs& a, int64_t va, int64_t e0, int64_t e1, int32_t& err) {
  if (!a.vform) {
    return va;
  }
  const int64_t b = a.vform == 1 ? (a.b_src == 1 ? e0 : e1) : a.v_lit;
  if ((a.a_nullable && va == a.a_null) || (a.vform == 1 && a.b_nullable && b == a.b_null)) {
    return a.v_null_out;
  }
  int64_t r;
  const bool ovf = a.vop == HDK_OP_ADD ? checked_arith(HDK_OP_ADD, va, b, a.v_check_width, &r)
                                       : (a.vop == HDK_OP_SUB ? checked_arith(HDK_OP_SUB, va, b, a.v_check_width, &r)
                                                              : checked_arith(HDK_OP_MUL, va, b, a.v_check_width, &r));
  if (ovf) {
    err = HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW;
  }
  return r;
}

// column buffers are plain hipMalloc'ed global memory: say so, or the pointers loaded from
// COL_BUFFERS are generic and every access becomes a flat_load
typedef const __attribute__((address_space(1))) int8_t* gcol_t;

// sign-extending element extraction from a 16-B (or narrower) register image
template <int W>
HDK_DEV int64_t extract_elem(const uint32_t* regs, int i) {
  if (W == 8) {
    return static_cast<int64_t>((static_cast<uint64_t>(regs[2 * i + 1]) << 32) | regs[2 * i]);
  } else if (W == 4) {
    return static_cast<int32_t>(regs[i]);
  } else if (W == 2) {
    return static_cast<int16_t>((regs[i / 2] >> (16 * (i & 1))) & 0xffff);
  } else {
    return static_cast<int8_t>((regs[i / 4] >> (8 * (i & 3))) & 0xff);
  }
}

// load NB bytes (4, 8 or 16; or 2/1 for the narrowest cases) for this lane.  NT: non-temporal -- a column is read once
// and 16 GB of it streaming through L2 and the Infinity Cache as ordinary (allocating) loads cost C2 9 %
// (6.1 -> 6.7 TB/s at 1 B rows with the hint, same kernel otherwise)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int NB, bool NT = false>
HDK_DEV void load_bytes(gcol_t p, uint32_t* regs) {
  if (NB == 16) {
    const __attribute__((address_space(1))) u32x4* q = reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(p);
    const u32x4 v = NT ? __builtin_nontemporal_load(q) : *q;
    regs[0] = v.x; regs[1] = v.y; regs[2] = v.z; regs[3] = v.w;
  } else if (NB == 8) {
    const __attribute__((address_space(1))) u32x2* q = reinterpret_cast<const __attribute__((address_space(1))) u32x2*>(p);
    const u32x2 v = NT ? __builtin_nontemporal_load(q) : *q;
    regs[0] = v.x; regs[1] = v.y;
  } else if (NB == 4) {
    const __attribute__((address_space(1))) uint32_t* q = reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(p);
    regs[0] = NT ? __builtin_nontemporal_load(q) : *q;
  } else if (NB == 2) {
    const __attribute__((address_space(1))) uint16_t* q = reinterpret_cast<const __attribute__((address_space(1))) uint16_t*>(p);
    regs[0] = NT ? __builtin_nontemporal_load(q) : *q;
  } else {
    const __attribute__((address_space(1))) uint8_t* q = reinterpret_cast<const __attribute__((address_space(1))) uint8_t*>(p);
    regs[0] = NT ? __builtin_nontemporal_load(q) : *q;
  }
}

template <int W>
HDK_DEV int64_t load_elem(gcol_t p, int64_t row) {
  if (W == 8) return reinterpret_cast<const __attribute__((address_space(1))) int64_t*>(p)[row];
  if (W == 4) return reinterpret_cast<const __attribute__((address_space(1))) int32_t*>(p)[row];
  if (W == 2) return reinterpret_cast<const __attribute__((address_space(1))) int16_t*>(p)[row];
  return reinterpret_cast<const __attribute__((address_space(1))) int8_t*>(p)[row];
}

HDK_DEV void fast_lds_op(int32_t kind, int64_t* wp, int64_t v) {
  switch (kind) {
    case FOP_ADD_ONE:
    case FOP_ADD_ONE_IF_NULL:
      atomicAdd(reinterpret_cast<unsigned long long*>(wp), 1ull);
      break;
    case FOP_ADD_U64:
      atomicAdd(reinterpret_cast<unsigned long long*>(wp), static_cast<unsigned long long>(v));
      break;
    case FOP_ADD_F64:
      atomicAdd(reinterpret_cast<double*>(wp), bits_to_double(v));
      break;
    case FOP_MIN_I64:
      atomicMin(reinterpret_cast<long long*>(wp), static_cast<long long>(v));
      break;
    case FOP_MAX_I64:
      atomicMax(reinterpret_cast<long long*>(wp), static_cast<long long>(v));
      break;
    case FOP_MIN_F64: {
      unsigned long long* a = reinterpret_cast<unsigned long long*>(wp);
      unsigned long long old = *a;
      const double d = bits_to_double(v);
      while (d < bits_to_double(static_cast<int64_t>(old))) {
        const unsigned long long assumed = old;
        old = atomicCAS(a, assumed, static_cast<unsigned long long>(v));
        if (old == assumed) break;
      }
      break;
    }
    default: {
      unsigned long long* a = reinterpret_cast<unsigned long long*>(wp);
      unsigned long long old = *a;
      const double d = bits_to_double(v);
      while (bits_to_double(static_cast<int64_t>(old)) < d) {
        const unsigned long long assumed = old;
        old = atomicCAS(a, assumed, static_cast<unsigned long long>(v));
        if (old == assumed) break;
      }
      break;
    }
  }
}

struct OpList {  // the per-row LDS update list, hoisted into scalar registers
  int32_t kind[kFastMaxOps];
  int32_t word[kFastMaxOps];
  int32_t n;
};

// One row: entry lookup + the LDS update list.  KW == 0 => non-grouped (entry 0).
// FIXED >= 0: the list is exactly one op of that kind (compile-time), at word ops.word[0].
template <int KW, int FIXED, bool MASK>
HDK_DEV void fast_row(const FastArgs& a, const OpList& ops, int64_t* lds, uint32_t my_rep, uint32_t estride, int64_t key,
                      int64_t val, bool has_val, uint64_t& touched, uint64_t& nonnull, int32_t& err) {
  uint32_t entry = 0;
  if (KW != 0 && KW <= 4) {
    // narrow key column: everything in 32 bits.  match_fast only lets a plan in when |key_min| <= 2^30 and the
    // translated NULL fits 32 bits; the table fits LDS, so entry_count is tiny: k - min cannot wrap onto a valid
    // entry (k in [-2^31, 2^31): the difference is >= -3 * 2^30, i.e. >= 2^30 as an unsigned 32-bit number).
    // (The 64-bit form costs 9 vector instructions more per row: taxi Q1 is 4 bytes per row.)
    int32_t k = static_cast<int32_t>(key);
    if (a.key_translate_null && k == static_cast<int32_t>(a.key_null)) {
      k = static_cast<int32_t>(a.key_null_translated);
    }
    const uint32_t e = static_cast<uint32_t>(k) - static_cast<uint32_t>(a.key_min);
    if (e >= a.entry_count) {
      err = HDK_HIP_ERR_OUT_OF_SLOTS;
      return;
    }
    entry = e;
  } else if (KW != 0) {
    if (a.key_translate_null && key == a.key_null) {
      key = a.key_null_translated;
    }
    const uint64_t e = static_cast<uint64_t>(key - a.key_min);
    if (e >= a.entry_count) {
      err = HDK_HIP_ERR_OUT_OF_SLOTS;
      return;
    }
    entry = static_cast<uint32_t>(e);
  }
  bool is_null = false;
  if (has_val && a.val_nullable) {
    // (the compile-time op lists fix the argument's class: as a run-time flag both compares and a select ran per row)
    constexpr int kFpKnown = FIXED >= 100 ? (FIXED & 1) : FIXED == FOP_ADD_U64 ? 0 : FIXED == FOP_ADD_F64 ? 1 : -1;
    const bool fp = kFpKnown >= 0 ? kFpKnown != 0 : a.val_is_fp != 0;
    is_null = fp ? (bits_to_double(val) == bits_to_double(a.val_null)) : (val == a.val_null);
  }
  if (MASK) {  // (compile time: as a run-time flag this was seven predicated vector instructions on every row)
    const uint64_t bit = 1ull << entry;
    touched |= bit;
    if (!is_null) {
      nonnull |= bit;
    }
  }
  // (entry * words per entry * replicas < 2^24: the table lives in LDS -- one full-rate 24-bit multiply)
  int64_t* base = lds + __umul24(entry, estride) + my_rep;
  if (FIXED >= 100) {
    // compile-time list "row count, sum, [NULL count]" (AVG / SUM+COUNT over one column; taxi Q2):
    // 100/101 = integer/fp sum with a NULL-count word, 102/103 = without (argument cannot be NULL)
    constexpr bool kFp = (FIXED & 1) != 0;
    constexpr bool kHasNullWord = FIXED < 102;
    fast_lds_op(FOP_ADD_ONE, base + ops.word[0] * a.rep, val);
    if (!is_null) {
      fast_lds_op(kFp ? FOP_ADD_F64 : FOP_ADD_U64, base + ops.word[1] * a.rep, val);
    } else if (kHasNullWord) {
      fast_lds_op(FOP_ADD_ONE_IF_NULL, base + ops.word[2] * a.rep, val);
    }
    return;
  }
  if (FIXED >= 0) {
    if (FIXED == FOP_ADD_ONE || !is_null) {
      fast_lds_op(FIXED, base + ops.word[0] * a.rep, val);
    }
    return;
  }
#pragma unroll
  for (int o = 0; o < kFastMaxOps; ++o) {
    if (o < ops.n) {
      const int32_t kind = ops.kind[o];
      if (kind == FOP_ADD_ONE || (kind == FOP_ADD_ONE_IF_NULL ? is_null : !is_null)) {
        fast_lds_op(kind, base + ops.word[o] * a.rep, val);
      }
    }
  }
}

// R = rows per lane per step (16 B of the widest column); U = steps per tile.
// Q: the plan has plain filters (plain_quals.h); rows that fail them are skipped before the LDS update.  A
// separate instantiation so that the unfiltered kernels (C1/C2) stay exactly as they were.
// XM > 0: X mode with XM - 1 extra 8-byte columns streamed beside the value column (filters between columns, expression
// arguments); Q is false then (the gathered `column cmp literal` filters are the other way to filter).
template <int KW, int VW, int U, int FIXED, bool Q = false, int XM = 0>
__global__ __launch_bounds__(kFastBlock) void hdk_scan_agg_direct(FastArgs a) {
  constexpr bool XMODE = XM > 0;
  constexpr bool XPROG = XM == 4;  // X mode, no extra column, the filters under an AND / OR / NOT program (its own instantiation:
                                   // with the program behind a run-time test the other X kernels grew by 10 registers, c2cc + 11 %)
  constexpr int X = (XM > 0 && XM < 4) ? XM - 1 : 0;
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  __shared__ unsigned long long s_masks[2];
  constexpr int WMAX = (KW > VW ? KW : VW) == 0 ? 8 : (KW > VW ? KW : VW);
  constexpr int R = 16 / WMAX;
  constexpr int KB = KW * R;  // bytes per lane per step, key column
  constexpr int VB = VW * R;
  constexpr int KREGS = KB >= 4 ? KB / 4 : 1;
  constexpr int VREGS = VB >= 4 ? VB / 4 : 1;
  const int tid = threadIdx.x;
  const uint32_t rep = a.rep;
  const int wpe = a.wpe;
  const uint32_t ew = a.entry_count * wpe;
  const uint32_t total_words = ew * rep;
  for (uint32_t i = tid; i < total_words; i += kFastBlock) {
    lds[i] = word_identity(a.wop[(i / rep) % wpe]);
  }
  if (tid < 2) {
    s_masks[tid] = 0;
  }
  __syncthreads();

  OpList ops;
  ops.n = a.nops;
#pragma unroll
  for (int o = 0; o < kFastMaxOps; ++o) {
    ops.kind[o] = a.op_kind[o];
    ops.word[o] = a.op_word[o];
  }
  const uint32_t my_rep = tid & (rep - 1);
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(kFastBlock) * R * U;
  uint64_t touched = 0, nonnull = 0;
  int32_t err = 0;

  const uint32_t estride = static_cast<uint32_t>(wpe) * rep;
  const Watch watch = watch_begin(a.kp);
  // the whole scan twice in the code, once per value of mask_mode: inside the row body the flag is a compile-time constant
  auto scan = [&](auto mask_tag) {
  constexpr bool MASK = decltype(mask_tag)::value;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    const gcol_t kcol = KW ? (gcol_t)cols[a.key_buf_idx] : nullptr;
    const gcol_t vcol = VW ? (gcol_t)cols[a.val_buf_idx] : nullptr;
    const gcol_t xcol0 = X > 0 ? (gcol_t)cols[a.x_buf_idx[0]] : nullptr;
    const gcol_t xcol1 = X > 1 ? (gcol_t)cols[a.x_buf_idx[1]] : nullptr;
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      if (row0 + kTileRows <= nrows) {
        // full tile: U coalesced 16-B steps per column, all issued before the first use.  (Issuing the NEXT tile's
        // loads before applying this one -- a second register set -- measured 2-3 % slower: 6.6 against 6.75 TB/s.)
        uint32_t kr[U][KREGS];
        uint32_t vr[U][VREGS];
        uint32_t xr0[X > 0 ? U : 1][4], xr1[X > 1 ? U : 1][4];  // (X mode: VW == 8, R == 2: 16 bytes per lane and step)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = row0 + (static_cast<int64_t>(u) * kFastBlock + tid) * R;
          if (KW) load_bytes<(KB > 0 ? KB : 4), true>(kcol + r * KW, kr[u]);
          if (VW) load_bytes<(VB > 0 ? VB : 4), true>(vcol + r * VW, vr[u]);
          if constexpr (X > 0) load_bytes<16, true>(xcol0 + r * 8, xr0[u]);
          if constexpr (X > 1) load_bytes<16, true>(xcol1 + r * 8, xr1[u]);
        }
        bool pass[U * R];
#pragma unroll
        for (int j = 0; j < U * R; ++j) {
          pass[j] = true;
        }
        if (Q) {
          int64_t rows[U * R];
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
              rows[u * R + i] = row0 + (static_cast<int64_t>(u) * kFastBlock + tid) * R + i;
            }
          }
          plain_quals_pass<U * R, true>(a.q, a.nquals, cols, rows, pass, true);
        }
        uint32_t xmask = ~0u;  // (X mode under a filter program: the tile's rows that pass)
        if constexpr (XPROG) {
          static_assert(U * R <= 32, "one bit per row");
          {
            int64_t va[U * R], ea0[U * R], ea1[U * R], ka[U * R];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
              for (int i = 0; i < R; ++i) {
                ka[u * R + i] = KW ? extract_elem<(KW ? KW : 8)>(kr[u], i) : 0;
                va[u * R + i] = VW ? extract_elem<(VW ? VW : 8)>(vr[u], i) : 0;
                ea0[u * R + i] = X > 0 ? extract_elem<8>(xr0[X > 0 ? u : 0], i) : 0;
                ea1[u * R + i] = X > 1 ? extract_elem<8>(xr1[X > 1 ? u : 0], i) : 0;
              }
            }
            xmask = fast_x_program<U * R>(a, va, ea0, ea1, ka);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const int64_t key = KW ? extract_elem<(KW ? KW : 8)>(kr[u], i) : 0;
            int64_t val = VW ? extract_elem<(VW ? VW : 8)>(vr[u], i) : 0;
            if constexpr (XMODE) {
              const int64_t e0 = X > 0 ? extract_elem<8>(xr0[X > 0 ? u : 0], i) : 0;
              const int64_t e1 = X > 1 ? extract_elem<8>(xr1[X > 1 ? u : 0], i) : 0;
              if (XPROG ? !((xmask >> ((u * R + i) & 31)) & 1u) : !fast_x_pass(a, val, e0, e1, key)) {
                continue;
              }
              val = fast_x_value(a, val, e0, e1, err);
            }
            if (!Q || pass[u * R + i]) {
              fast_row<KW, FIXED, MASK>(a, ops, lds, my_rep, estride, key, val, VW != 0, touched, nonnull, err);
            }
          }
        }
      } else {
        // ragged tail of a fragment: one row per lane per pass
        for (int64_t r = row0 + tid; r < nrows; r += kFastBlock) {
          if (Q) {
            const int64_t rows1[1] = {r};
            bool pass1[1] = {true};
            plain_quals_pass<1, true>(a.q, a.nquals, cols, rows1, pass1, true);
            if (!pass1[0]) {
              continue;
            }
          }
          const int64_t key = KW ? load_elem<(KW ? KW : 8)>(kcol, r) : 0;
          int64_t val = VW ? load_elem<(VW ? VW : 8)>(vcol, r) : 0;
          if constexpr (XMODE) {
            const int64_t e0 = X > 0 ? load_elem<8>(xcol0, r) : 0;
            const int64_t e1 = X > 1 ? load_elem<8>(xcol1, r) : 0;
            bool xp;
            if (XPROG) {
              const int64_t v1[1] = {val}, e01[1] = {e0}, e11[1] = {e1}, k1[1] = {key};
              xp = (fast_x_program<1>(a, v1, e01, e11, k1) & 1u) != 0;
            } else {
              xp = fast_x_pass(a, val, e0, e1, key);
            }
            if (!xp) {
              continue;
            }
            val = fast_x_value(a, val, e0, e1, err);
          }
          fast_row<KW, FIXED, MASK>(a, ops, lds, my_rep, estride, key, val, VW != 0, touched, nonnull, err);
        }
      }
    }
    frag_tile_begin += ntiles;
  }
  };
  if (a.mask_mode) {
    scan(std::true_type{});
  } else {
    scan(std::false_type{});
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
  if (a.mask_mode) {
    if (touched) atomicOr(&s_masks[0], static_cast<unsigned long long>(touched));
    if (nonnull) atomicOr(&s_masks[1], static_cast<unsigned long long>(nonnull));
  }
  __syncthreads();
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * ew;
  for (uint32_t i = tid; i < ew; i += kFastBlock) {
    const uint32_t w = i % wpe;
    const uint32_t e = i / wpe;
    const int32_t op = a.wop[w];
    int64_t acc = lds[i * rep];
    for (uint32_t r = 1; r < rep; ++r) {
      acc = word_combine(op, acc, lds[i * rep + r]);
    }
    if (!a.mask_mode && ((a.nword_mask >> w) & 1u)) {  // NULL count -> non-null count
      int64_t rows = 0;
      for (uint32_t r = 0; r < rep; ++r) {
        rows += lds[(e * wpe) * rep + r];
      }
      acc = rows - acc;
    }
    if (a.mask_mode) {
      if (w == 0) {
        acc = (s_masks[0] >> e) & 1;  // "row count" degenerates to touched/not touched
      } else {
        for (int k = 0; k < a.n_nn_words; ++k) {
          if (static_cast<uint32_t>(a.nn_words[k]) == w) {
            acc = (s_masks[1] >> e) & 1;
          }
        }
      }
    }
    slab[i] = acc;
  }
}

}  // namespace hdk
