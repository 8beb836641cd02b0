// scan_bhm.h -- group-by on chip for aggregates over SEVERAL argument columns / expressions and for several key columns.
//
// The packed kernels of scan_bh_packed.h take ONE argument column; one shape further the reference's own benchmark suite fell
// off the chip: MultiStep/MSBS001 / MSPHS001 (count(*), max(x100), max(x10), max(x10 + 1), sum(x100), sum(x10 + 1) by 1 000
// groups) ran on global atomics, PerfectHashMultiCol/PHM001-002 (five aggregates by two key columns) on the interpreter or the
// perfect-partitioned passes.  Reference being replaced: get_group_value[_fast] + agg_*_shared on the final table for every row
// (QE/GroupByRuntime.cpp:31-55,198-246, QE/cuda_mapd_rt.cu:167-203,424-478); the multi-column perfect-hash index is
// perfect_key_hash (QE/RowFuncBuilder.cpp:647-689,748-801).
//
// Same pricing as scan_bh_packed.h -- an LDS atomic is the expensive thing, a read is cheap:
//   * a block keeps a DENSE table in LDS: entry = sum_k (key_k - min_k) * stride_k by the key columns' statistics (a
//     GroupByPerfectHash plan: exactly its own entry index, NULLs under their translated value), no tags, no probe;
//   * every argument that is counted, summed or averaged has ONE 64-bit word [non-NULL rows : 24 | sum : 40] per entry -- one
//     ds_add_u64 a row gives COUNT, SUM and AVG; the row count comes from an argument that cannot be NULL (else a 32-bit
//     counter);
//   * every MIN / MAX of the row lives in a FIELD of one shared 64-bit word per entry, sized from the argument's statistics
//     (x100: 7 bits + a guard bit), all of them "bigger is better" codes (MAX: v - min + 1, MIN: max - v + 1, 0 = none): the
//     row reads the word (ds_read_b64), compares all fields at once ((cur | guards) - candidates keeps every guard bit iff no
//     field improves) and only then -- after a group's first rows, never -- runs a compare-and-swap loop;
//   * arguments may be `column op literal` (+ - *), several arguments may read the same column: it is streamed once.
// MSBS001's six aggregates are two ds_add_u64 and one ds_read_b32 a row, 20 bytes of LDS per group.
// Two forms of the same row body: SHAPES the suite uses (which argument has a packed word, a MAX / MIN field, an `op literal`
// step; no NULLs announced; all fields inside 32 bits) are compile-time constants of their own instantiations (BhmStatic: the
// per-row code has no branch and no select left); every other shape runs the same body with the descriptors read at run time
// (BhmDynamic: uniform branches, slower, any combination).
// At the end a block decodes its table into a slab of the generic partial-aggregate words (agg_common.h); a GroupByPerfectHash
// plan's slabs are folded by hdk_finalize (the internal index IS the plan's), an open-addressing plan's by hdk_bhm_fold
// (find_or_claim on the reference's probe sequence, once per group).
// Statistics that do not hold (a key or an argument outside them, a NULL where none was announced, more rows in a block
// than the packed fields were sized for) raise a flag instead of an exact per-row path: the folds then skip, and the
// global-atomics kernel armed behind them redoes the launch -- never a wrong result from stale metadata, only a slower one.
#pragma once
#include <cstddef>

#include "watch.h"
#include "agg_common.h"
#include "plain_quals.h"
#include "scan_agg_fast.h"
#include "scan_bh.h"

namespace hdk {

constexpr int kBhmMaxKeys = 3;
constexpr int kBhmMaxSrc = 3;     // streamed argument columns
constexpr int kBhmMaxDer = 4;     // arguments in all: a column itself, or `column op literal`
constexpr int kBhmMaxFields = 2 * kBhmMaxDer;
constexpr int kBhmSumBits = 40;

enum BhmWordKind : int32_t { BMW_ROWS = 0, BMW_SUM = 1, BMW_NN = 2, BMW_MIN = 3, BMW_MAX = 4 };

struct BhmKey {
  int32_t buf_idx;
  int32_t min;        // entry term = key - min ...
  uint32_t n;         // ... for key - min < n (else the statistics do not hold)
  uint32_t stride;
  int32_t nullable;   // the NULL key has a term of its own
  int32_t null32;
  uint32_t null_d;
  uint32_t pad_;
};

// a MIN / MAX field of the entry's 64-bit word (two 32-bit halves; a field lies inside one of them)
struct BhmField {
  uint32_t shift;       // inside its half
  uint32_t mask_lo;     // the code's bits (without the guard bit), unshifted, when the field is in the low half, else 0
  uint32_t mask_hi;     // ... in the high half
  int32_t bias;         // MAX: code = v - bias (bias = smallest value - 1);  MIN: code = bias - v (bias = largest value + 1)
};

struct BhmDer {
  int32_t src;          // argument column it reads
  int32_t mul, add;     // value = column * mul + add (the column itself: 1, 0)
  int32_t packed;       // index of its [non-NULL rows : sum] word, or -1
  int32_t has_mx, has_mn;
  BhmField mx, mn;
};

struct BhmSrc {
  int32_t buf_idx;
  int32_t nullable;     // the column's in-band NULL is announced by the statistics and skipped by every target
  int32_t null32;
  int32_t raw_min;      // statistics of the column: raw - raw_min <= raw_span
  uint32_t raw_span;
  uint32_t pad_;
};

// one word of the slab a block writes at the end (agg_common.h's layout)
struct BhmSlabWord {
  int32_t kind;         // BhmWordKind
  int32_t packed;       // BMW_SUM / BMW_NN / BMW_ROWS: the packed word (ROWS: -1 = the 32-bit row counter)
  int32_t in_mm;        // BMW_MIN / BMW_MAX (and BMW_NN of an argument without a packed word: non-NULL iff its field is not 0)
  int32_t half;
  uint32_t shift, mask;
  int32_t bias;
  int32_t pad_;
};

struct BhmArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  int64_t* slabs;          // [grid][entries][wpe]
  uint32_t* flag;          // != 0: the statistics did not hold somewhere -- the folds skip, the armed fallback runs
  uint32_t entries;        // internal (dense) entries = slab rows
  uint32_t e1;             // entries + the dummy, rounded up to even
  uint32_t rep;            // replicas (power of two)
  uint32_t rep_bytes;      // bytes between replicas
  uint32_t off_mm;         // byte offsets inside a replica: packed words at 0, then the MIN / MAX words, then the row counters
  uint32_t off_rows;
  uint32_t lds_bytes;
  uint32_t max_rows_per_block;  // what the packed fields were sized for
  int32_t nkeys, nsrc, nder;
  int32_t npacked;
  int32_t mm_bytes;        // 0: no MIN / MAX; 4: all fields inside 32 bits; 8
  int32_t rows_packed;     // packed word whose count is the row count, or -1: the 32-bit counters
  int32_t wpe;
  int32_t any_nullable;    // some key or argument column has a NULL to look for
  BhmKey key[kBhmMaxKeys];
  BhmSrc src[kBhmMaxSrc];
  BhmDer der[kBhmMaxDer];
  uint32_t guard_lo, guard_hi;   // guard bits of the fields
  int32_t nfields;
  int32_t pad_;
  uint32_t fshift[kBhmMaxFields], fmask[kBhmMaxFields], fhalf[kBhmMaxFields];
  BhmSlabWord sw[kMaxWordsPerEntry];
  // open-addressing plans: the key word of internal entry i (hdk_bhm_fold)
  int32_t key_form;        // 0: the key column's value; 1: cast(integer AS double)
  int32_t col_width;       // bytes of every streamed column: 4, or 8 (BIGINT columns inside 32 bits by their statistics; their
                           // in-band NULL is mapped to INT32_MIN: key / src `null32` say INT32_MIN then)
  int64_t key_null_word;   // key word of the NULL key's entry
  // plain filters `column cmp literal` / AND-OR-NOT programs over them (plain_quals.h): the kernels' Q instantiations
  // pass B of the two-pass form works on the tuples' CODES (value - min + 1): the sum of a plain column's codes is turned into
  // the sum of its values when the table is decoded (+= non-NULL rows x psum_k)
  int32_t psum_k[kBhmMaxDer];
  int32_t nquals;
  int32_t qvec;            // every filter column is an integer column of the streamed width: loaded 16 bytes a lane with the tile
                           // (2: and the filter is a plain conjunction of integer comparisons -- bhm_quals_lean)
  ProjFastQual q[kMaxPlainQuals];
  // qvec == 2: where leaf qi's column comes from -- key column kk (kk), argument column s (kBhmMaxKeys + s): already in the
  // tile's registers; kBhmQualOwnLoad: its own 16-byte load
  int32_t qsrc[kMaxPlainQuals];
  uint32_t plain_flags;    // BhmPlain: bits 3 i .. 3 i + 2 = argument i has a packed word / a MAX field / a MIN field
};
constexpr int32_t kBhmQualOwnLoad = 255;

// ---- the shape of a plan's arguments, as compile-time constants or read from the descriptors ------------------------------
// code of argument i: bits 0-1 the column, bit 2 a packed word, bit 3 a MAX field, bit 4 a MIN field, bits 5-6 the step
// (0 none, 1 + literal (also -), 2 * literal); 0xFF: no such argument
constexpr uint32_t kBhmNone = 0xFF;
constexpr uint32_t bhm_code(int src, bool packed, bool mx, bool mn, int step) {
  return static_cast<uint32_t>(src) | (packed ? 4u : 0u) | (mx ? 8u : 0u) | (mn ? 16u : 0u) | (static_cast<uint32_t>(step) << 5);
}
// NULLS: some key or argument column announces NULLs (a NULL argument is skipped by its targets, a NULL key has its own entry)
template <uint32_t D0, uint32_t D1 = kBhmNone, uint32_t D2 = kBhmNone, uint32_t D3 = kBhmNone, bool NULLS = false>
struct BhmStatic {
  static constexpr bool kSkips = NULLS;  // rows / arguments that take no part exist
  static constexpr uint32_t code(int i) { return i == 0 ? D0 : (i == 1 ? D1 : (i == 2 ? D2 : D3)); }
  HDK_DEV static bool used(const BhmArgs&, int i) { return code(i) != kBhmNone; }
  HDK_DEV static int src(const BhmArgs&, int i) { return static_cast<int>(code(i) & 3u); }
  HDK_DEV static bool packed(const BhmArgs&, int i) { return (code(i) & 4u) != 0; }
  HDK_DEV static bool mx(const BhmArgs&, int i) { return (code(i) & 8u) != 0; }
  HDK_DEV static bool mn(const BhmArgs&, int i) { return (code(i) & 16u) != 0; }
  HDK_DEV static int step(const BhmArgs&, int i) { return static_cast<int>((code(i) >> 5) & 3u); }
  HDK_DEV static bool nulls(const BhmArgs&) { return NULLS; }
  HDK_DEV static int mm_bytes(const BhmArgs&) { return ((D0 | (D1 == kBhmNone ? 0 : D1) | (D2 == kBhmNone ? 0 : D2) | (D3 == kBhmNone ? 0 : D3)) & 24u) ? 4 : 0; }
};
// Between the two: AGGREGATES OF PLAIN COLUMNS -- argument i is column i as it is (no literal step), which of count-and-sum /
// MAX / MIN it feeds is read at run time (BhmArgs::plain_flags, three bits an argument).  Every mix of count / sum / avg / min /
// max over up to NS columns without the run-time form's scalar-register spills (its descriptors are a dozen words here).
template <int NS_>
struct BhmPlain {
  static constexpr bool kSkips = true;
  HDK_DEV static bool used(const BhmArgs&, int i) { return i < NS_; }
  HDK_DEV static int src(const BhmArgs&, int i) { return i; }
  HDK_DEV static bool packed(const BhmArgs& a, int i) { return ((a.plain_flags >> (3 * i)) & 1u) != 0; }
  HDK_DEV static bool mx(const BhmArgs& a, int i) { return ((a.plain_flags >> (3 * i)) & 2u) != 0; }
  HDK_DEV static bool mn(const BhmArgs& a, int i) { return ((a.plain_flags >> (3 * i)) & 4u) != 0; }
  HDK_DEV static int step(const BhmArgs&, int) { return 0; }
  HDK_DEV static bool nulls(const BhmArgs& a) { return a.any_nullable != 0; }
  HDK_DEV static int mm_bytes(const BhmArgs& a) { return a.mm_bytes; }
};
struct BhmDynamic {
  static constexpr bool kSkips = true;
  HDK_DEV static bool used(const BhmArgs& a, int i) { return i < a.nder; }
  HDK_DEV static int src(const BhmArgs& a, int i) { return a.der[i].src; }
  HDK_DEV static bool packed(const BhmArgs& a, int i) { return a.der[i].packed >= 0; }
  HDK_DEV static bool mx(const BhmArgs& a, int i) { return a.der[i].has_mx != 0; }
  HDK_DEV static bool mn(const BhmArgs& a, int i) { return a.der[i].has_mn != 0; }
  HDK_DEV static int step(const BhmArgs& a, int i) { return a.der[i].mul != 1 ? 2 : (a.der[i].add != 0 ? 1 : 0); }
  HDK_DEV static bool nulls(const BhmArgs& a) { return a.any_nullable != 0; }
  HDK_DEV static int mm_bytes(const BhmArgs& a) { return a.mm_bytes; }
};

// a row improves some MIN / MAX field (after a group's first rows: never): compare-and-swap until every field of the word is at
// least the row's.  ONE inlined copy per row body (the caller walks its pending rows through it one after another).
HDK_DEV void bhm_mm_improve(const BhmArgs& ar, uint32_t* word, uint32_t cand_lo, uint32_t cand_hi, int32_t mm_bytes) {
  const BhmArgs* a = &ar;
  const int nf = a->nfields;
  if (mm_bytes == 4) {
    uint32_t old = *word;
    while (true) {
      uint32_t merged = 0;
      for (int f = 0; f < nf; ++f) {
        const uint32_t m = a->fmask[f] << a->fshift[f];
        const uint32_t x = old & m, y = cand_lo & m;
        merged |= x > y ? x : y;
      }
      if (merged == old) {
        return;
      }
      const uint32_t seen = atomicCAS(word, old, merged);
      if (seen == old) {
        return;
      }
      old = seen;
    }
  }
  unsigned long long* w64 = reinterpret_cast<unsigned long long*>(word);
  unsigned long long old = *w64;
  const unsigned long long cand = (static_cast<unsigned long long>(cand_hi) << 32) | cand_lo;
  while (true) {
    unsigned long long merged = 0;
    for (int f = 0; f < nf; ++f) {
      const unsigned long long m = static_cast<unsigned long long>(a->fmask[f]) << (a->fshift[f] + 32u * a->fhalf[f]);
      const unsigned long long x = old & m, y = cand & m;
      merged |= x > y ? x : y;
    }
    if (merged == old) {
      return;
    }
    const unsigned long long seen = atomicCAS(w64, old, merged);
    if (seen == old) {
      return;
    }
    old = seen;
  }
}

// NR rows of one lane.  k[kk][j]: key column kk of row j; x[s][j]: argument column s.  `stale` collects "the statistics do
// not hold" (the caller raises the launch's flag).
// One element of a streamed column as the 32-bit value the row body works on.  W == 8: a BIGINT column whose statistics lie
// inside 32 bits -- the in-band NULL (INT64_MIN) becomes INT32_MIN (what the descriptors' null32 says for such columns); a value
// that does not fit 32 bits, or is INT32_MIN itself, sets `wide` (the statistics do not hold for it).
template <int W>
HDK_DEV int32_t bhm_narrow(const uint32_t* regs, int i, bool& wide) {
  if (W == 4) {
    wide = false;
    return static_cast<int32_t>(regs[i]);
  }
  const int64_t v = extract_elem<8>(regs, i);
  const int32_t lo = static_cast<int32_t>(v);
  const bool isnull = v == INT64_MIN;
  wide = !isnull & ((v != static_cast<int64_t>(lo)) | (lo == INT32_MIN));
  return isnull ? INT32_MIN : lo;
}

// `integer column cmp integer literal` leaves -- a plain conjunction or an AND / OR / NOT program over them -- on columns that
// rode in with the tile (BhmArgs::qvec == 2): the leaves unrolled, their descriptors at constant offsets, one compare and one NULL test per row and leaf.  plain_quals.h's
// general evaluator (any column type, programs, a run-time loop over the leaves with their 64-byte descriptors re-read for
// every 16-byte step, and -- with the columns in registers -- three extracts and two selects per row to pick a leaf's column)
// made MSPHS001 WHERE x10 < 8 run at 4.1 ms per 1 B rows, twice the unfiltered time.  Returns the rows that pass.
template <int R, int W, int NK, int NS>
HDK_DEV uint32_t bhm_quals_lean(const BhmArgs& a, const uint32_t (&kr)[NK][4], const uint32_t (&xr)[NS][4], const uint32_t (&qr)[kMaxPlainQuals][4]) {
  uint32_t lt[kMaxPlainQuals], ln[kMaxPlainQuals];  // leaf qi: bit i = TRUE / NULL for row i
#pragma unroll
  for (int qi = 0; qi < kMaxPlainQuals; ++qi) {
    lt[qi] = 0;
    ln[qi] = 0;
    if (qi < a.nquals) {  // (wave-uniform)
      const int64_t rhs = a.q[qi].rhs, nullv = a.q[qi].null_val;
      const bool nullable = a.q[qi].nullable != 0;
      // the leaf's column: one the tile streams anyway (a key, an argument: wave-uniform branches over the register sets), or
      // its own load
      int64_t val[R];
      const int32_t from = a.qsrc[qi];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        val[i] = extract_elem<W>(qr[qi], i);
      }
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        if (from == kk) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            val[i] = extract_elem<W>(kr[kk], i);
          }
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        if (from == kBhmMaxKeys + s2) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            val[i] = extract_elem<W>(xr[s2], i);
          }
        }
      }
      uint32_t t = 0, n = 0;
#define HDK_BHM_QROWS(OP)                                                                  \
  _Pragma("unroll") for (int i = 0; i < R; ++i) {                                         \
    const int64_t v = val[i];                                                             \
    t |= (v OP rhs) ? 1u << i : 0u;                                                       \
    n |= (nullable && v == nullv) ? 1u << i : 0u;                                         \
  }
      switch (a.q[qi].cmp) {
        case HDK_CMP_EQ: HDK_BHM_QROWS(==) break;
        case HDK_CMP_NE: HDK_BHM_QROWS(!=) break;
        case HDK_CMP_LT: HDK_BHM_QROWS(<) break;
        case HDK_CMP_GT: HDK_BHM_QROWS(>) break;
        case HDK_CMP_LE: HDK_BHM_QROWS(<=) break;
        default: HDK_BHM_QROWS(>=) break;
      }
#undef HDK_BHM_QROWS
      lt[qi] = t & ~n;  // (a NULL operand: the comparison is NULL, not TRUE)
      ln[qi] = n;
    }
  }
  const uint32_t all = (1u << R) - 1u;
  const int nprog = a.q[0].nprog;
  if (nprog == 0) {  // the plain conjunction
    uint32_t okm = all;
#pragma unroll
    for (int qi = 0; qi < kMaxPlainQuals; ++qi) {
      okm &= qi < a.nquals ? lt[qi] : all;
    }
    return okm;
  }
  // the filter's postfix AND / OR / NOT program over the leaves' masks (plain_quals.h's three-valued combiner: a = top, b, c)
  // (the twelve program bytes as three kernel-argument words read once: a byte load per step is a scalar-memory round trip in
  // every 16-byte step of every wave -- MSPHS001 under a four-step OR / NOT program ran at 5.2 ms per 1 B rows with them)
  static_assert(kMaxPlainProg == 12 && offsetof(ProjFastQual, prog) % 4 == 0, "three aligned words");
  const uint32_t* pw = reinterpret_cast<const uint32_t*>(a.q[0].prog);
  const uint32_t p0 = pw[0], p1 = pw[1], p2 = pw[2];
  uint32_t at = 0, an = 0, bt = 0, bn = 0, ct = 0, cn = 0;
  for (int i = 0; i < nprog; ++i) {
    const uint32_t op = ((i < 4 ? p0 : (i < 8 ? p1 : p2)) >> (8 * (i & 3))) & 0xFFu;
    if (op < HDK_F_AND) {
      ct = bt; cn = bn;
      bt = at; bn = an;
      at = op == 0 ? lt[0] : (op == 1 ? lt[1] : lt[2]);
      an = op == 0 ? ln[0] : (op == 1 ? ln[1] : ln[2]);
    } else if (op == HDK_F_NOT) {
      at = ~(at | an);  // NULL stays NULL, TRUE <-> FALSE
    } else {
      uint32_t rt, rn;
      if (op == HDK_F_AND) {
        const uint32_t fa = ~(at | an), fb = ~(bt | bn);  // FALSE operands
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
  return at & all;
}

// the dense entry of NR rows from their key columns (dummy = the entry behind the table for a key outside the statistics);
// returns the rows whose key lies outside them
template <int NK, int NR>
HDK_DEV uint32_t bhm_key_entries(const BhmArgs& a, bool nulls, uint32_t dummy, const int32_t (&k)[NK][NR], uint32_t (&e)[NR]) {
  uint32_t badm = 0;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    uint32_t idx = 0;
    bool bad = false;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const BhmKey& key = a.key[kk];
      const int32_t kv = k[kk][j];
      uint32_t d = static_cast<uint32_t>(kv) - static_cast<uint32_t>(key.min);
      if (nulls) {
        const bool isnull = (key.nullable != 0) & (kv == key.null32);
        bad = bad | (!isnull & (d >= key.n));
        d = isnull ? key.null_d : d;
      } else {
        bad = bad | (d >= key.n);
      }
      idx += NK == 1 ? d : __umul24(d, key.stride);
    }
    badm |= bad ? 1u << j : 0u;
    e[j] = bad ? dummy : idx;
  }
  return badm;
}

// the argument columns against their statistics; live: not NULL (and inside them); returns the rows with a value outside them
template <int NS, int NR>
HDK_DEV uint32_t bhm_src_live(const BhmArgs& a, bool nulls, const int32_t (&x)[NS][NR], bool (&live)[NS][NR]) {
  uint32_t badm = 0;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const BhmSrc& src = a.src[s];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int32_t raw = x[s][j];
      const bool out = (static_cast<uint32_t>(raw) - static_cast<uint32_t>(src.raw_min)) > src.raw_span;
      if (nulls) {
        const bool isnull = (src.nullable != 0) & (raw == src.null32);
        badm |= (out & !isnull) ? 1u << j : 0u;
        live[s][j] = !isnull & !out;
      } else {
        badm |= out ? 1u << j : 0u;
        live[s][j] = true;  // (a row outside the statistics raises the flag: what it adds is thrown away with the launch)
      }
    }
  }
  return badm;
}

template <class C, int NS, int NR, bool Q = false>
HDK_DEV void bhm_update(const BhmArgs& a, uint8_t* rp, uint32_t dummy, const uint32_t (&e)[NR], const int32_t (&x)[NS][NR], const bool (&live)[NS][NR]);

// NR rows of one lane.  k[kk][j]: key column kk of row j; x[s][j]: argument column s; okm: rows that take part (a ragged tile's
// end, filters); `stale` collects "the statistics do not hold" for rows that take part (the caller raises the launch's flag).
template <class C, int NK, int NS, int NR, bool Q = false>
HDK_DEV void bhm_rows(const BhmArgs& a, uint8_t* rp, const int32_t (&k)[NK][NR], const int32_t (&x)[NS][NR], uint32_t okm, uint32_t widem,
                      uint32_t& stale) {
  const bool nulls = C::nulls(a);
  uint32_t e[NR];
  uint32_t badm = bhm_key_entries<NK, NR>(a, nulls, a.entries, k, e) | widem;
  bool live[NS][NR];
  badm |= bhm_src_live<NS, NR>(a, nulls, x, live);
  stale |= badm & okm;
  if (C::kSkips || Q) {  // (a compile-time shape without NULLs or a filter: every row takes part -- a bad one raises `stale`)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      e[j] = ((okm & ~badm) >> j) & 1u ? e[j] : a.entries;
    }
  }
  int32_t code[NS][NR];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      code[s][j] = x[s][j] - a.src[s].raw_min + 1;
    }
  }
  bhm_update<C, NS, NR, Q>(a, rp, a.entries, e, code, live);
}

// the LDS updates of NR rows whose entries are known (e[j]; `dummy` for rows that do not take part).
// x holds CODES: column value - (its minimum - 1), so 1 .. span + 1 inside the statistics (the one-pass kernel subtracts once per
// column and row, bhm_rows; pass B of the two-pass form finds them in its tuples).  The host hands the kernels descriptors
// shifted to codes (bhm_shift_to_codes, scan_bhm.hip): for `column` and `column +- literal` arguments the MAX field code IS the
// code, the MIN code one subtraction away (der.mn.bias = span + 2), the packed word adds the code itself (the decode adds non-NULL
// rows x (smallest value - 1) to the sum, BhmArgs::psum_k) -- no bias subtraction, no masks, no sign extension; `column x
// literal` arguments get their value from the code (der.add shifted).  A code outside the statistics may spill into the next
// field: the flag is up then and the table is thrown away.  Both kernels were bound by vector instructions, not by LDS atomics
// (pass B: 28 a row, SQ_ACTIVE_INST_VALU 78 % of the cycles; MSPHS001 in one pass: 45 a row, 67 %).
template <class C, int NS, int NR, bool Q>
HDK_DEV void bhm_update(const BhmArgs& a, uint8_t* rp, uint32_t dummy, const uint32_t (&e)[NR], const int32_t (&x)[NS][NR], const bool (&live)[NS][NR]) {
  const bool nulls = C::nulls(a);
  constexpr bool skips = C::kSkips || Q;  // rows that take no part exist
  uint32_t cand_lo[NR], cand_hi[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    cand_lo[j] = 0;
    cand_hi[j] = 0;
  }
  const int mmb = C::mm_bytes(a);
#pragma unroll
  for (int i = 0; i < kBhmMaxDer; ++i) {
    if (C::used(a, i)) {
      const BhmDer& der = a.der[i];
      const int s = C::src(a, i);
      int32_t v[NR];
      bool lv[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        int32_t raw = x[0][j];
        lv[j] = live[0][j];
        if (NS > 1 && s == 1) {
          raw = x[NS > 1 ? 1 : 0][j];
          lv[j] = live[NS > 1 ? 1 : 0][j];
        }
        if (NS > 2 && s == 2) {
          raw = x[NS > 2 ? 2 : 0][j];
          lv[j] = live[NS > 2 ? 2 : 0][j];
        }
        const int st = C::step(a, i);
        v[j] = st <= 1 ? raw : __mul24(raw, der.mul) + der.add;  // (the code; column x literal: the value)
      }
      const bool plain_code = C::step(a, i) <= 1;  // column, column +- literal: the same code
      if (C::packed(a, i)) {
        unsigned long long* pk = reinterpret_cast<unsigned long long*>(rp) + static_cast<uint32_t>(der.packed) * a.e1;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          // (a NULL argument is not counted and not added; nor is a row the filter dropped: many lanes adding to the one
          // dummy entry would serialise -- 30 % of the rows dropped: 5.2 ms per 1 B rows with the adds, X without)
          const uint32_t ej = nulls ? (lv[j] ? e[j] : dummy) : e[j];
          if (!skips || ej != dummy) {
            atomicAdd(pk + ej, (1ull << kBhmSumBits) + (plain_code ? static_cast<unsigned long long>(static_cast<uint32_t>(v[j]))
                                                                   : static_cast<unsigned long long>(static_cast<long long>(v[j]))));
          }
        }
      }
      if (C::mx(a, i)) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const uint32_t code = plain_code ? static_cast<uint32_t>(v[j]) : static_cast<uint32_t>(v[j] - der.mx.bias);
          const uint32_t c = nulls ? (lv[j] ? code : 0u) : code;
          if (plain_code && mmb != 8) {
            cand_lo[j] |= c << der.mx.shift;
          } else {
            cand_lo[j] |= (c & der.mx.mask_lo) << der.mx.shift;
            if (mmb == 8) cand_hi[j] |= (c & der.mx.mask_hi) << der.mx.shift;
          }
        }
      }
      if (C::mn(a, i)) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const uint32_t code = static_cast<uint32_t>(der.mn.bias - v[j]);
          const uint32_t c = nulls ? (lv[j] ? code : 0u) : code;
          if (plain_code && mmb != 8) {
            cand_lo[j] |= c << der.mn.shift;
          } else {
            cand_lo[j] |= (c & der.mn.mask_lo) << der.mn.shift;
            if (mmb == 8) cand_hi[j] |= (c & der.mn.mask_hi) << der.mn.shift;
          }
        }
      }
    }
  }
  // MIN / MAX: look before touching
  if (mmb != 0) {
    uint32_t better = 0;
    if (mmb == 4) {
      uint32_t* mm = reinterpret_cast<uint32_t*>(rp + a.off_mm);
      const uint32_t H = a.guard_lo;
      uint32_t cur[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        cur[j] = mm[e[j]];
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        better |= ((((cur[j] | H) - cand_lo[j]) & H) != H) ? 1u << j : 0u;
      }
    } else {
      const uint2* mm = reinterpret_cast<const uint2*>(rp + a.off_mm);
      const uint32_t Hl = a.guard_lo, Hh = a.guard_hi;
      uint2 cur[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        cur[j] = mm[e[j]];
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const bool b = ((((cur[j].x | Hl) - cand_lo[j]) & Hl) != Hl) | ((((cur[j].y | Hh) - cand_hi[j]) & Hh) != Hh);
        better |= b ? 1u << j : 0u;
      }
    }
    while (__builtin_amdgcn_ballot_w64(better != 0)) {
      if (better) {
        const int j = __ffs(better) - 1;
        better &= better - 1;
        uint32_t ej = e[0], cl = cand_lo[0], ch = cand_hi[0];
#pragma unroll
        for (int i = 1; i < NR; ++i) {
          ej = i == j ? e[i] : ej;
          cl = i == j ? cand_lo[i] : cl;
          ch = i == j ? cand_hi[i] : ch;
        }
        bhm_mm_improve(a, reinterpret_cast<uint32_t*>(rp + a.off_mm + static_cast<size_t>(ej) * static_cast<uint32_t>(mmb)), cl, ch, mmb);
      }
    }
  }
  if (a.rows_packed < 0) {
    uint32_t* rows = reinterpret_cast<uint32_t*>(rp + a.off_rows);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (!skips || e[j] != dummy) {
        atomicAdd(rows + e[j], 1u);
      }
    }
  }
}

// entry `ei` of the block's table as slab words (agg_common.h): the replicas' partials decoded and combined
HDK_DEV void bhm_slab_entry(const BhmArgs& a, const uint8_t* lds8, uint32_t ei, int64_t* out) {
  // the raw LDS words of the entry, summed / merged over the replicas
  int64_t psum[kBhmMaxDer], pcnt[kBhmMaxDer];
#pragma unroll
  for (int pi = 0; pi < kBhmMaxDer; ++pi) {
    psum[pi] = 0;
    pcnt[pi] = 0;
  }
  uint32_t mm_lo = 0, mm_hi = 0;
  int64_t rows = 0;
  for (uint32_t r = 0; r < a.rep; ++r) {
    const uint8_t* rp = lds8 + static_cast<size_t>(r) * a.rep_bytes;
#pragma unroll
    for (int pi = 0; pi < kBhmMaxDer; ++pi) {
      if (pi < a.npacked) {
        const uint64_t pk = reinterpret_cast<const uint64_t*>(rp)[static_cast<uint32_t>(pi) * a.e1 + ei];
        const int64_t sum = static_cast<int64_t>(pk << (64 - kBhmSumBits)) >> (64 - kBhmSumBits);
        psum[pi] += sum;
        pcnt[pi] += static_cast<int64_t>((pk - static_cast<uint64_t>(sum)) >> kBhmSumBits);
      }
    }
    if (a.mm_bytes == 4) {
      const uint32_t w = reinterpret_cast<const uint32_t*>(rp + a.off_mm)[ei];
      // per-field maximum over the replicas
      for (int f = 0; f < a.nfields; ++f) {
        const uint32_t m = a.fmask[f] << a.fshift[f];
        mm_lo = (mm_lo & ~m) | ((w & m) > (mm_lo & m) ? (w & m) : (mm_lo & m));
      }
    } else if (a.mm_bytes == 8) {
      const uint32_t wl = reinterpret_cast<const uint32_t*>(rp + a.off_mm)[2 * ei], wh = reinterpret_cast<const uint32_t*>(rp + a.off_mm)[2 * ei + 1];
      for (int f = 0; f < a.nfields; ++f) {
        const uint32_t m = a.fmask[f] << a.fshift[f];
        if (a.fhalf[f] == 0) {
          mm_lo = (mm_lo & ~m) | ((wl & m) > (mm_lo & m) ? (wl & m) : (mm_lo & m));
        } else {
          mm_hi = (mm_hi & ~m) | ((wh & m) > (mm_hi & m) ? (wh & m) : (mm_hi & m));
        }
      }
    }
    if (a.rows_packed < 0) {
      rows += reinterpret_cast<const uint32_t*>(rp + a.off_rows)[ei];
    }
  }
#pragma unroll
  for (int pi = 0; pi < kBhmMaxDer; ++pi) {
    psum[pi] += pcnt[pi] * a.psum_k[pi];  // (0 except in pass B of the two-pass form)
  }
  for (int w = 0; w < a.wpe; ++w) {
    const BhmSlabWord sw = a.sw[w];
    int64_t v;
    if (sw.kind == BMW_ROWS && sw.packed < 0) {
      v = rows;
    } else if (sw.kind == BMW_ROWS || sw.kind == BMW_SUM || (sw.kind == BMW_NN && sw.packed >= 0)) {
      int64_t sm = psum[0], ct = pcnt[0];
#pragma unroll
      for (int pi = 1; pi < kBhmMaxDer; ++pi) {
        sm = sw.packed == pi ? psum[pi] : sm;
        ct = sw.packed == pi ? pcnt[pi] : ct;
      }
      v = sw.kind == BMW_SUM ? sm : ct;
    } else {
      const uint32_t code = ((sw.half ? mm_hi : mm_lo) >> sw.shift) & sw.mask;
      if (sw.kind == BMW_NN) {
        v = code ? 1 : 0;  // (only "none / some" is read from the count of a MIN / MAX-only argument)
      } else if (sw.kind == BMW_MAX) {
        v = code ? static_cast<int64_t>(sw.bias) + code : INT64_MIN;
      } else {
        v = code ? static_cast<int64_t>(sw.bias) - code : INT64_MAX;
      }
    }
    out[w] = v;
  }
}

// C: the arguments' shape (BhmStatic<...> or BhmDynamic); NK key columns, NS argument columns, all 4 bytes wide; U steps of
// 16 bytes per lane, column and tile
// (round-6 note: issuing the NEXT full tile's loads before this tile's rows go through the table -- a second register set --
// measured WORSE: msbs1 / msphs1 at three blocks per CU 0.69 / 0.63 ms against 0.63 / 0.60 per 256 M rows; not kept)
// W: bytes of every streamed column (4, or 8: BIGINT columns inside 32 bits); U steps of 16 bytes per lane, column and tile
template <class C, int NK, int NS, int BLOCK, int U, int W = 4, bool Q = false>
__global__ __launch_bounds__(BLOCK) void hdk_scan_agg_bhm(BhmArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
  constexpr int R = 16 / W;
  const int tid = threadIdx.x;
  {
    uint4* z = reinterpret_cast<uint4*>(lds8);
    const uint32_t n16 = a.lds_bytes / 16;
    for (uint32_t i = tid; i < n16; i += BLOCK) {
      z[i] = make_uint4(0, 0, 0, 0);  // (every code is "0 = nothing yet")
    }
  }
  __syncthreads();
  uint8_t* rp = lds8 + static_cast<size_t>(tid & (a.rep - 1)) * a.rep_bytes;
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(BLOCK) * R * U;
  int32_t err = 0;
  uint32_t stale = 0;
  uint32_t rows_seen = 0;
  const Watch watch = watch_begin(a.kp);
  constexpr bool filtered = Q;  // (the matcher picks the Q kernels for plans with a filter, and only for them)
  const bool qvec = Q && a.qvec != 0;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    gcol_t kcol[NK], xcol[NS];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      kcol[kk] = (gcol_t)cols[a.key[kk].buf_idx];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      xcol[s] = (gcol_t)cols[a.src[s].buf_idx];
    }
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      rows_seen += static_cast<uint32_t>(kTileRows);
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      if (row0 + kTileRows <= nrows) {
        uint32_t kr[U][NK][4], xr[U][NS][4], qr[Q ? U : 1][kMaxPlainQuals][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R;
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            load_bytes<16, true>(kcol[kk] + r * W, kr[u][kk]);
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            load_bytes<16, true>(xcol[s] + r * W, xr[u][s]);
          }
          // the filter's columns in the same batch (integers of the streamed width: BhmArgs::qvec): read by row number after
          // the batch has landed, their latency stood alone -- 5.2 ms per 1 B rows for MSPHS001 WHERE x10 < 8, X this way
#pragma unroll
          for (int qi = 0; qi < kMaxPlainQuals; ++qi) {
            if (Q && qvec && qi < a.nquals && (a.qvec != 2 || a.qsrc[qi] == kBhmQualOwnLoad)) {
              load_bytes<16, true>((gcol_t)cols[a.q[qi].col.buf_idx] + r * W, qr[Q ? u : 0][qi]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int32_t kv[NK][R], xv[NS][R];
          uint32_t widem = 0, okm = (1u << R) - 1u;
#pragma unroll
          for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
              bool wide;
              kv[kk][i] = bhm_narrow<W>(kr[u][kk], i, wide);
              widem |= wide ? 1u << i : 0u;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              bool wide;
              xv[s][i] = bhm_narrow<W>(xr[u][s], i, wide);
              widem |= wide ? 1u << i : 0u;
            }
          }
          if (filtered) {
            int64_t rows[R];
            bool ok[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
              rows[i] = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R + i;
              ok[i] = true;
            }
            if (a.qvec == 2) {
              const uint32_t lean = bhm_quals_lean<R, W, NK, NS>(a, kr[u], xr[u], qr[Q ? u : 0]);
#pragma unroll
              for (int i = 0; i < R; ++i) {
                ok[i] = ((lean >> i) & 1u) != 0;
              }
            } else if (qvec) {
              plain_quals_pass_with<R, true>(
                  a.q, a.nquals,
                  [&](int qi, const ProjFastQual&, const bool (&)[R], int64_t (&v)[R]) {
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                      // (qi is wave-uniform: a select over the three register sets' values, not an indexed array -- that would live in scratch)
                      const int64_t v0 = extract_elem<W>(qr[Q ? u : 0][0], i), v1 = extract_elem<W>(qr[Q ? u : 0][1], i), v2 = extract_elem<W>(qr[Q ? u : 0][2], i);
                      v[i] = qi == 0 ? v0 : (qi == 1 ? v1 : v2);
                    }
                  },
                  ok);
            } else {
              plain_quals_pass<R, true>(a.q, a.nquals, cols, rows, ok, true);
            }
            okm = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
              okm |= ok[i] ? 1u << i : 0u;
            }
          }
          bhm_rows<C, NK, NS, R, Q>(a, rp, kv, xv, okm, widem, stale);
        }
      } else {
        // the ragged tail of a fragment, a row per lane and trip
        for (int64_t r = row0 + tid; r < nrows; r += BLOCK) {
          int32_t kv[NK][1], xv[NS][1];
          uint32_t widem = 0, okm = 1u;
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            const int64_t v = load_elem<W>(kcol[kk], r);
            const uint32_t two[2] = {static_cast<uint32_t>(v), static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32)};
            bool wide;
            kv[kk][0] = bhm_narrow<W>(two, 0, wide);
            widem |= wide ? 1u : 0u;
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const int64_t v = load_elem<W>(xcol[s], r);
            const uint32_t two[2] = {static_cast<uint32_t>(v), static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32)};
            bool wide;
            xv[s][0] = bhm_narrow<W>(two, 0, wide);
            widem |= wide ? 1u : 0u;
          }
          if (filtered) {
            const int64_t rows1[1] = {r};
            bool ok1[1] = {true};
            plain_quals_pass<1, true>(a.q, a.nquals, cols, rows1, ok1, true);
            okm = ok1[0] ? 1u : 0u;
          }
          bhm_rows<C, NK, NS, 1, Q>(a, rp, kv, xv, okm, widem, stale);
        }
      }
    }
    frag_tile_begin += ntiles;
  }
  // (more rows than the packed fields were sized for can only come from a row bound that did not hold)
  if (stale || rows_seen > a.max_rows_per_block) {
    atomicOr(a.flag, 1u);
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
  __syncthreads();
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * a.entries * a.wpe;
  for (uint32_t ei = tid; ei < a.entries; ei += BLOCK) {
    bhm_slab_entry(a, lds8, ei, slab + static_cast<size_t>(ei) * a.wpe);
  }
}

// (tests, HDK_HIP_BHM_FLAG_IS_ERROR: instead of arming the fallback)
template <int DUMMY = 0>
__global__ void hdk_bhm_flag_is_error(const uint32_t* flag, int32_t* error_code) {
  if (*flag) {
    record_error(error_code, HDK_HIP_ERR_OUT_OF_SLOTS);
  }
}

// ---- slabs [num_slabs][words] -> [groups][words]: output slab g combines the input slabs g, g + groups, ... -------------------------
// Thread per word, coalesced over the slab; what makes the folds cheap (a wave per entry over `groups` slabs instead of over
// hundreds, each of whose words sits in a different memory line: 77-97 us for 512 slabs of 1 000 groups, against ~20 us here).
struct BhmReduceArgs {
  const int64_t* in;
  int64_t* out;
  const uint32_t* flag;
  uint32_t num_slabs, groups, words;
  int32_t wpe;
  int32_t wop[kMaxWordsPerEntry];
};
template <int DUMMY = 0>
__global__ __launch_bounds__(256) void hdk_bhm_reduce_slabs(BhmReduceArgs a) {
  if (*a.flag) {
    return;
  }
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t g = blockIdx.y;
  if (i >= a.words) {
    return;
  }
  const int32_t op = a.wop[i % static_cast<uint32_t>(a.wpe)];
  int64_t acc = word_identity(op);
  constexpr int kInFlight = 8;
  uint32_t b = g;
  for (; b + (kInFlight - 1) * a.groups < a.num_slabs; b += kInFlight * a.groups) {
    int64_t v[kInFlight];
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) {
      v[j] = a.in[static_cast<size_t>(b + j * a.groups) * a.words + i];
    }
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) {
      acc = word_combine(op, acc, v[j]);
    }
  }
  for (; b < a.num_slabs; b += a.groups) {
    acc = word_combine(op, acc, a.in[static_cast<size_t>(b) * a.words + i]);
  }
  a.out[static_cast<size_t>(g) * a.words + i] = acc;
}

// ---- the fold of an open-addressing plan's slabs: one wave per internal entry ---------------------------------------------------
// lanes reduce the entry's words over the slabs (a fixed shuffle tree: deterministic), lane 0 finds or claims the group's
// entry on the reference's probe sequence (bh_fold_group_fn, scan_bh.h) and adds the partial in -- every group is folded into
// the output table ONCE.
struct BhmFoldArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  const int64_t* slabs;
  const uint32_t* flag;
  uint32_t num_slabs, entries, out_entry_count;
  int32_t wpe;
  int32_t wop[kMaxWordsPerEntry];
  uint32_t nword_mask;     // words that hold non-NULL counts (the fold wants NULL counts: rows - word)
  int32_t key_form;
  int64_t key_lo;          // internal entry i is the group of key key_lo + i ...
  uint32_t null_entry;     // ... but for this one (0xFFFFFFFF: none), the NULL key's
  uint32_t pad_;
  int64_t key_null_word;
};

template <int FLAT = 0>
__global__ __launch_bounds__(256) void hdk_bhm_fold(BhmFoldArgs a) {
  __shared__ WordLayout wl;
  __shared__ uint64_t s_col_off[2 * HDK_HIP_MAX_TARGETS];
  __shared__ int64_t s_words[FLAT ? kMaxWordsPerEntry : 4][FLAT ? 256 : kMaxWordsPerEntry];
  if (*a.flag) {
    return;  // the statistics did not hold: the slabs are not to be trusted (the armed fallback redoes the launch)
  }
  if (threadIdx.x == 0) {
    make_word_layout(a.plan, &wl);
  }
  if (threadIdx.x < 2 * HDK_HIP_MAX_TARGETS) {
    s_col_off[threadIdx.x] = a.plan->output_columnar ? columnar_slot_off(a.plan, a.out_entry_count, threadIdx.x) : 0;
  }
  __syncthreads();
  const int wpe = a.wpe;
  const size_t ew = static_cast<size_t>(a.entries) * wpe;
  uint32_t entry;
  const int64_t* words;
  int ws;
  if (FLAT) {
    // few slabs of a large table (the two-pass form's eight): a THREAD per internal entry
    entry = blockIdx.x * 256 + threadIdx.x;
    if (entry >= a.entries) {
      return;
    }
    for (int w = 0; w < wpe; ++w) {
      const int32_t op = a.wop[w];
      int64_t acc = word_identity(op);
      for (uint32_t b = 0; b < a.num_slabs; ++b) {
        acc = word_combine(op, acc, a.slabs[b * ew + static_cast<size_t>(entry) * wpe + w]);
      }
      s_words[FLAT ? w : 0][FLAT ? threadIdx.x : 0] = acc;
    }
    words = &s_words[0][FLAT ? threadIdx.x : 0];
    ws = 256;
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    entry = blockIdx.x * 4 + wave;
    if (entry >= a.entries) {
      return;
    }
    int64_t* wv = s_words[FLAT ? 0 : wave];
    for (int w = 0; w < wpe; ++w) {
      const int32_t op = a.wop[w];
      int64_t acc = word_identity(op);
      for (uint32_t b = lane; b < a.num_slabs; b += 64) {
        acc = word_combine(op, acc, a.slabs[b * ew + static_cast<size_t>(entry) * wpe + w]);
      }
      for (int d = 32; d > 0; d >>= 1) {
        const int lo = __shfl_down(static_cast<int>(static_cast<uint32_t>(acc)), d, 64);
        const int hi = __shfl_down(static_cast<int>(static_cast<uint64_t>(acc) >> 32), d, 64);
        acc = word_combine(op, acc, static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo)));
      }
      if (lane == 0) {
        wv[w] = acc;
      }
    }
    if (lane != 0) {
      return;
    }
    words = wv;
    ws = 1;
  }
  const int64_t rows = words[0];
  if (rows == 0) {
    return;  // (no row of this group: nothing to claim)
  }
  int32_t err = 0;
  int64_t key;
  if (entry == a.null_entry) {
    key = a.key_null_word;
  } else {
    const int64_t kv = a.key_lo + static_cast<int64_t>(entry);
    key = a.key_form == 1 ? double_to_bits(static_cast<double>(kv)) : kv;
  }
  bh_fold_group_fn(a.plan, table_shape(a.plan), wl, a.kp.groupby_buf[0], a.out_entry_count, s_col_off, key,
                   [&](int w) -> int64_t { return ((a.nword_mask >> w) & 1u) ? rows - words[w * ws] : words[w * ws]; }, err);
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
