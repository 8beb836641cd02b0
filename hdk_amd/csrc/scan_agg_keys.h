// scan_agg_keys.h -- perfect-hash group-by on up to three TRANSFORMED keys: row counts (hdk_scan_agg_keys<false>)
// or row counts plus SUM / AVG / MIN / MAX / COUNT over up to two plain outer columns (hdk_scan_agg_keys<true>).
//
// Shape: GroupByPerfectHash, no join, filters of the form `outer column cmp literal` (plain_quals.h),
// 1-3 group keys each of which is an integer column of the outer table with at most one unary step
// (EXTRACT(YEAR FROM ts) or the decimal scale-down of a CAST).  The counting form takes targets that need no value
// word: projected keys and COUNT(*).  That is taxi Q3 (GROUP BY passenger_count, year(pickup_datetime)) and Q4
// (... , cast(trip_distance as int)) of BASELINE config 4 (reference omniscidb/Benchmarks/taxi/
// taxi_reduced_bench.cpp).  Same arithmetic as the batched interpreter (vec_eval.h: eval_key_v,
// perfect_hash_entry_v; reference key computation QE/RuntimeFunctions.cpp get_columnar_perfect_hash*,
// extract_year Utils/ExtractFromTime.cpp:150-190), same LDS table / slab / finalize protocol
// (agg_common.h) -- what goes away is the interpreter: no expression walk, no 64-bit entry arithmetic
// (the key terms are range-checked against their cardinalities and combined in 32 bits), one LDS atomic
// per row.  The interpreter spends ~116 VALU instructions per row on these plans and is issue-bound at
// 28-31 % of the HBM roofline; this kernel needs about 50.
#pragma once
#include "watch.h"
#include "agg_common.h"
#include "plain_quals.h"
#include "scan_agg_fast.h"  // FastOpKind, fast_lds_op: the per-row LDS update list of the value form

// AND / OR / NOT filter programs (plain_quals.h) are NOT compiled into this kernel: measured in one call at 256 M rows, their
// code costs the unfiltered taxi queries 3-4 % (Q3 0.445 -> 0.431 ms, Q4 0.79 -> 0.77, Q3 + AVG 0.83 -> 0.80 without it);
// such plans stay with the batched interpreter (A/B: -DHDK_KEYS_PROG=true and allow_program in match_keys)
#ifndef HDK_KEYS_PROG
#define HDK_KEYS_PROG false
#endif

namespace hdk {

constexpr int kKeysBlock = 256;
// The value form with a table too large to replicate (Q4's 2 499 entries x 3 words = 60 KB) fits two blocks on a CU:
// with 256 threads that is two waves per SIMD, too few to cover the loads.  Same kernel with 512 threads per block:
// two blocks = four waves per SIMD sharing the same two tables.
constexpr int kKeysWideBlock = 512;
constexpr int kKeysMaxVals = 2;
constexpr int kKeysMaxOps = 8;
constexpr int kKeysVR = 8;   // rows per lane and tile (taxi Q3/Q4 at 256 M rows: 4 -> 0.79/1.23 ms, 8 -> 0.60/1.00, 12 -> 0.66/1.02)
// Rows are dealt to lanes R at a time (R = 16 bytes of the widest column the kernel reads, 2 or 4): in a full tile a
// lane's R rows of a column are ONE naturally aligned load of R x width bytes (16 B for the widest column), lane after
// lane contiguous, at a 32-bit lane offset from a tile base the scalar unit keeps -- half (a quarter) of the load
// instructions of a row-per-lane deal and no vector address arithmetic.
// Two paths per block: keys_tile, the straight-line body for FULL tiles whose rows all lie in the transforms' 32-bit
// fast ranges, and keys_row, a row-at-a-time evaluation with the 64-bit forms that takes (a) the rows of a fragment's
// ragged last tile and (b) the rare rows a full tile set aside (timestamps outside 1970..2036, decimals beyond 32
// bits).  Keeping the 64-bit forms and the bounds checks out of the tile body is what keeps it at ~70 VGPRs (with
// them inlined eight rows wide it needed 139 and lost two waves per SIMD).
constexpr int kKeysMax = 3;

enum KeyTransform : int32_t { KXF_NONE = 0, KXF_YEAR = 1, KXF_SCALE_DOWN = 2 };

struct KeysKey {
  int32_t buf_idx;
  int32_t width;
  int32_t is_unsigned;
  int32_t xf;              // KeyTransform
  int64_t param;           // KXF_SCALE_DOWN: the scale (2 .. INT32_MAX)
  int64_t col_null;        // in-band NULL of the column (widened)
  int64_t key_null;        // NULL of the key expression (plain keys: replaced by `translated` when translate != 0)
  int64_t kmin;
  int32_t col_nullable;
  int32_t translate;
  uint32_t card;
  uint32_t stride;         // product of the cardinalities of the keys before this one
  // what a NULL key contributes: (translated or untranslated NULL) - kmin, worked out by the matcher
  uint32_t null_term;
  int32_t null_ok;         // that term is inside [0, card)
  uint32_t null_term_p;    // null_ok ? null_term : kKeysPoison (what the tile body adds for a NULL)
  int32_t narrow;          // plain key whose column, NULL and kmin all fit 32 bits: the whole term is 32-bit
  int32_t kmin32;
  uint32_t year_base;      // KXF_YEAR: 1900 - kmin32
  uint32_t div_magic;      // KXF_SCALE_DOWN: n / scale == (((n - mulhi(magic, n)) >> 1) + mulhi(magic, n)) >> div_shift
  int32_t div_shift;       //   for every 32-bit unsigned n (round-up method, branch-free form)
};

struct KeysVal {           // an aggregate argument: a plain column of the outer table
  int32_t buf_idx;
  int32_t width;
  int32_t kind;            // hdk_hip_col_kind (FLOAT is widened to double, like the decoder of the interpreter)
  int32_t nullable;        // some target skips its NULLs
  int64_t null_val;        // widened (int) / double bits (fp)
};

struct KeysArgs {
  KernParams kp;
  int64_t* slabs;
  uint32_t entry_count;
  uint32_t rep;
  uint32_t rep_shift;      // rep == 1 << rep_shift
  int32_t nkeys;
  int32_t nquals;
  KeysKey key[kKeysMax];
  ProjFastQual q[kMaxPlainQuals];
  // value form only: words per entry, the argument columns and the per-row update list (word 0 = row count)
  int32_t wpe;
  int32_t nvals;
  KeysVal val[kKeysMaxVals];
  int32_t nops;
  int32_t op_kind[kKeysMaxOps];  // FastOpKind
  int32_t op_word[kKeysMaxOps];
  int32_t op_val[kKeysMaxOps];   // which argument column
  int32_t wop[kMaxWordsPerEntry];
  uint32_t nword_mask;           // bit w: word w counts NULLs (the flush stores row count - NULLs)
};

typedef const __attribute__((address_space(1))) int8_t* keys_gptr;

// VR values of a column for this lane in a full tile: dst[u * R + i] = row  row0 + (u * kKeysBlock + tid) * R + i
template <typename T, int R, int BS, typename D>
HDK_DEV void keys_load(const int8_t* buf, int64_t row0, int tid, D* dst) {
  constexpr int VR = kKeysVR;
  typedef T vec_t __attribute__((ext_vector_type(R)));
  // the tile base is block-uniform but was read through COL_BUFFERS with a vector load: move it to scalar registers
  const uint64_t b = reinterpret_cast<uintptr_t>(buf) + static_cast<uint64_t>(row0) * sizeof(T);
  const uint32_t b_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b));
  const uint32_t b_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b >> 32));
  const keys_gptr base = reinterpret_cast<keys_gptr>((static_cast<uint64_t>(b_hi) << 32) | b_lo);
#pragma unroll
  for (int u = 0; u < VR / R; ++u) {
    const uint32_t off = static_cast<uint32_t>(u * BS + tid) * static_cast<uint32_t>(R * sizeof(T));
    const vec_t v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) vec_t*>(base + off));
#pragma unroll
    for (int i = 0; i < R; ++i) {
      dst[u * R + i] = static_cast<D>(v[i]);
    }
  }
}

// One row with the 64-bit forms of everything: filters, keys (NULL translation, transforms), entry, LDS update.
template <bool VALS>
HDK_DEV void keys_row(const KeysArgs& a, int64_t* lds, const int8_t* const* cols, int64_t rr, uint32_t my_rep, uint32_t wpe,
                      int32_t& err) {
  if (a.nquals) {
    const int64_t rows1[1] = {rr};
    bool pass1[1] = {true};
    plain_quals_pass<1, HDK_KEYS_PROG>(a.q, a.nquals, cols, rows1, pass1, true);
    if (!pass1[0]) {
      return;
    }
  }
  uint32_t entry = 0;
  for (int k = 0; k < a.nkeys; ++k) {
    const KeysKey kk = a.key[k];
    const int64_t v = decode_col_g(cols[kk.buf_idx], kk.width, kk.is_unsigned ? HDK_COL_UNSIGNED : HDK_COL_INT, rr, true);
    uint64_t d;
    bool isnull;
    if (kk.xf == KXF_NONE) {
      isnull = kk.translate && v == kk.key_null;
      d = static_cast<uint64_t>(v) - static_cast<uint64_t>(kk.kmin);
    } else {
      isnull = kk.col_nullable && v == kk.col_null;
      const int64_t val = kk.xf == KXF_YEAR ? extract_year(v) : scale_decimal_down(v, kk.param);
      d = static_cast<uint64_t>(val) - static_cast<uint64_t>(kk.kmin);
    }
    if (isnull ? kk.null_ok == 0 : d >= kk.card) {
      err = HDK_HIP_ERR_OUT_OF_SLOTS;
      return;
    }
    entry += (isnull ? kk.null_term : static_cast<uint32_t>(d)) * kk.stride;
  }
  if (!VALS) {
    atomicAdd(reinterpret_cast<uint32_t*>(lds) + ((entry << a.rep_shift) + my_rep), 1u);
    return;
  }
  const uint32_t base = ((entry * wpe) << a.rep_shift) + my_rep;
  atomicAdd(reinterpret_cast<unsigned long long*>(lds + base), 1ull);
  for (int vi = 0; vi < a.nvals; ++vi) {
    const KeysVal kv = a.val[vi];
    const int64_t v = decode_col_g(cols[kv.buf_idx], kv.width, kv.kind, rr, true);
    const bool fp = kv.kind == HDK_COL_DOUBLE || kv.kind == HDK_COL_FLOAT;
    const bool isnull = kv.nullable && (fp ? bits_to_double(v) == bits_to_double(kv.null_val) : v == kv.null_val);
    for (int o = 0; o < a.nops; ++o) {
      if (a.op_val[o] == vi && isnull == (a.op_kind[o] == FOP_ADD_ONE_IF_NULL)) {
        fast_lds_op(a.op_kind[o], lds + base + (static_cast<uint32_t>(a.op_word[o]) << a.rep_shift), v);
      }
    }
  }
}

#define HDK_KEYS_LOAD(DST, T) keys_load<T, R, BS>(buf, row0, tid, DST);

// One FULL tile of kKeysBlock * VR rows starting at row `row0` of a fragment.  Returns a per-lane bit mask of the rows
// (bit r = the r-th row of this lane) that took no part in the tile's LDS updates and are the caller's to evaluate
// with keys_row: rows outside a transform's 32-bit fast range, and rows whose key fell outside the table (keys_row
// raises the error for those).
//
// No boolean survives a branch or the key loop here (the compiler keeps such values as 0/1 bytes in VGPRs and spends
// ~8 instructions per row converting them): a row's state is its entry number alone.  A key component that is NULL
// with no slot, out of range, or outside the fast range contributes kKeysPoison instead of its term -- more than any
// table this kernel takes has entries, so `entry >= entry_count` at the end says "not in this tile's update" -- and a
// row the filters rejected starts at kKeysFiltered (bit 31, which no sum of terms reaches).
constexpr uint32_t kKeysPoison = 8192;  // > kLdsMaxTableWords >= entry_count >= every cardinality; 3 x 8192 x 7680 < 2^31
constexpr uint32_t kKeysFiltered = 0x80000000u;

template <bool VALS, int R, int BS>
HDK_DEV uint32_t keys_tile(const KeysArgs& a, int64_t* lds, const int8_t* const* cols, int64_t row0, int tid,
                           uint32_t my_rep, uint32_t wpe, int32_t& err) {
  constexpr int VR = kKeysVR;
  const int nk = a.nkeys;
  uint32_t entry[VR];
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    entry[r] = 0;
  }
  if (a.nquals) {
    int64_t row[VR];
    bool pass[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      row[r] = row0 + static_cast<int64_t>((r / R) * BS + tid) * R + (r % R);
      pass[r] = true;
    }
    plain_quals_pass<VR, HDK_KEYS_PROG>(a.q, a.nquals, cols, row, pass, true);
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      entry[r] = pass[r] ? 0u : kKeysFiltered;
    }
  }
#pragma unroll 1  // one copy of the decoders and transforms: three unrolled copies spill SGPRs and triple the code
  for (int k = 0; k < nk; ++k) {
    const KeysKey kk = a.key[k];
    const int8_t* buf = cols[kk.buf_idx];
    if (kk.xf == KXF_NONE && kk.narrow) {
      int32_t v[VR];
      if (kk.is_unsigned) {
        if (kk.width == 1) { HDK_KEYS_LOAD(v, uint8_t) } else { HDK_KEYS_LOAD(v, uint16_t) }
      } else if (kk.width == 1) {
        HDK_KEYS_LOAD(v, int8_t)
      } else if (kk.width == 2) {
        HDK_KEYS_LOAD(v, int16_t)
      } else {
        HDK_KEYS_LOAD(v, int32_t)
      }
      const int32_t null32 = static_cast<int32_t>(kk.key_null);
      if (kk.translate) {  // perfect hash: the NULL key owns the slot past the range
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const uint32_t d = static_cast<uint32_t>(v[r]) - static_cast<uint32_t>(kk.kmin32);
          const uint32_t t = v[r] == null32 ? kk.null_term_p : d;
          entry[r] = __umul24(t < kk.card ? t : kKeysPoison, kk.stride) + entry[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const uint32_t d = static_cast<uint32_t>(v[r]) - static_cast<uint32_t>(kk.kmin32);
          entry[r] = __umul24(d < kk.card ? d : kKeysPoison, kk.stride) + entry[r];
        }
      }
    } else {
      int64_t v[VR];
      if (kk.is_unsigned) {
        switch (kk.width) {
          case 1: HDK_KEYS_LOAD(v, uint8_t) break;
          case 2: HDK_KEYS_LOAD(v, uint16_t) break;
          case 4: HDK_KEYS_LOAD(v, uint32_t) break;
          default: HDK_KEYS_LOAD(v, int64_t) break;
        }
      } else {
        switch (kk.width) {
          case 1: HDK_KEYS_LOAD(v, int8_t) break;
          case 2: HDK_KEYS_LOAD(v, int16_t) break;
          case 4: HDK_KEYS_LOAD(v, int32_t) break;
          default: HDK_KEYS_LOAD(v, int64_t) break;
        }
      }
      uint32_t t[VR];
      int check_null = 0;  // integer or-accumulation: a bool assigned under `if` compiles to an EXEC branch per row
      if (kk.xf == KXF_NONE) {  // wide plain key
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const uint64_t d = static_cast<uint64_t>(v[r]) - static_cast<uint64_t>(kk.kmin);
          const bool in = d < kk.card;
          t[r] = in ? static_cast<uint32_t>(d) : kKeysPoison;
          check_null |= static_cast<int>(!in);
        }
        check_null &= kk.translate;
      } else {
        // unary step in its 32-bit form; a row outside the form's range (a NULL is: every NULL sentinel of a 4- or
        // 8-byte column lies outside, and the matcher keeps narrower transformed columns away) gets the poison term
        if (kk.xf == KXF_YEAR) {
          // extract_year's fast range (device_common.h; reference Utils/ExtractFromTime.cpp:150-166)
          constexpr uint32_t kEpochOffsetYear1900 = 2208988800u;
          constexpr uint32_t kSecsJanToMar1900 = 5097600u;
          constexpr uint32_t kSecondsPer4YearCycle = 126230400u;
          constexpr uint32_t kSecsPerDay = 86400u;
          constexpr uint32_t kSecondsPerNonLeapYear = 31536000u;
          const uint32_t year_base = kk.year_base;  // 1900 - kmin, folded by the matcher (the compiler would not)
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            const bool fast = static_cast<uint64_t>(v[r]) <= static_cast<uint64_t>(UINT32_MAX - kEpochOffsetYear1900);
            const uint32_t seconds_1900 = static_cast<uint32_t>(v[r]) + kEpochOffsetYear1900;
            const uint32_t leap_years = (seconds_1900 - kSecsJanToMar1900) / kSecondsPer4YearCycle;
            const uint32_t d = (seconds_1900 - leap_years * kSecsPerDay) / kSecondsPerNonLeapYear + year_base;
            t[r] = (static_cast<int>(fast) & static_cast<int>(d < kk.card)) ? d : kKeysPoison;  // `&&` would branch per row
            check_null |= static_cast<int>(!fast);
          }
        } else {
          // scale_decimal_down (device_common.h; reference QE/RuntimeFunctions.cpp:245-262): round half away
          // from zero, then divide by the scale, truncating
          const int64_t half = kk.param >> 1;
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            const int64_t tmp = v[r] >= 0 ? v[r] + half : v[r] - half;
            const int32_t t32 = static_cast<int32_t>(tmp);
            const bool fast = t32 == tmp && t32 != INT32_MIN;
            const uint32_t n = static_cast<uint32_t>(t32 < 0 ? -t32 : t32);
            const uint32_t hi = __umulhi(kk.div_magic, n);
            const uint32_t q = (((n - hi) >> 1) + hi) >> kk.div_shift;
            const uint32_t sq = t32 < 0 ? 0u - q : q;
            const uint32_t d = sq - static_cast<uint32_t>(kk.kmin32);
            t[r] = (static_cast<int>(fast) & static_cast<int>(d < kk.card)) ? d : kKeysPoison;  // `&&` would branch per row
            check_null |= static_cast<int>(!fast);
          }
        }
        check_null &= kk.col_nullable;
      }
      if (check_null) {  // some row of the wave is NULL or beyond the fast range: give the NULLs their slot
        const int64_t null_in = kk.xf == KXF_NONE ? kk.key_null : kk.col_null;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          t[r] = v[r] == null_in ? kk.null_term_p : t[r];
        }
      }
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        entry[r] = __umul24(t[r], kk.stride) + entry[r];
      }
    }
  }
  // entry < entry_count: the row's slot.  Anything else below bit 31: not decided here.
  uint32_t slow = 0;
  int any_slow = 0;
  bool pass[VR];
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    pass[r] = entry[r] < a.entry_count;
    any_slow |= static_cast<int>(entry[r] - a.entry_count < kKeysFiltered - a.entry_count);
  }
  if (any_slow) {
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      slow |= (entry[r] - a.entry_count < kKeysFiltered - a.entry_count) ? (1u << r) : 0u;
    }
  }
  if (!VALS) {
    uint32_t* cnt = reinterpret_cast<uint32_t*>(lds);  // 32-bit row counters, folded into the slab before they can wrap
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      if (pass[r]) {
        atomicAdd(cnt + ((entry[r] << a.rep_shift) + my_rep), 1u);
      }
    }
  } else {
    uint32_t base[VR];  // word 0 of the row's entry, this lane's replica
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      base[r] = ((entry[r] * wpe) << a.rep_shift) + my_rep;
      if (pass[r]) {
        atomicAdd(reinterpret_cast<unsigned long long*>(lds + base[r]), 1ull);
      }
    }
#pragma unroll 1
    for (int vi = 0; vi < a.nvals; ++vi) {
      const KeysVal kv = a.val[vi];
      const int8_t* buf = cols[kv.buf_idx];
      int64_t v[VR];
      bool isnull[VR];
      if (kv.kind == HDK_COL_DOUBLE || kv.kind == HDK_COL_FLOAT) {
        if (kv.kind == HDK_COL_DOUBLE) {
          HDK_KEYS_LOAD(v, int64_t)
        } else {
          float f[VR];
          HDK_KEYS_LOAD(f, float)
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            v[r] = double_to_bits(static_cast<double>(f[r]));
          }
        }
        const double dn = bits_to_double(kv.null_val);
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          isnull[r] = kv.nullable && bits_to_double(v[r]) == dn;  // `val != skip_val`: a value compare
        }
      } else {
        if (kv.kind == HDK_COL_UNSIGNED) {
          switch (kv.width) {
            case 1: HDK_KEYS_LOAD(v, uint8_t) break;
            case 2: HDK_KEYS_LOAD(v, uint16_t) break;
            default: HDK_KEYS_LOAD(v, uint32_t) break;
          }
        } else {
          switch (kv.width) {
            case 1: HDK_KEYS_LOAD(v, int8_t) break;
            case 2: HDK_KEYS_LOAD(v, int16_t) break;
            case 4: HDK_KEYS_LOAD(v, int32_t) break;
            default: HDK_KEYS_LOAD(v, int64_t) break;
          }
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          isnull[r] = kv.nullable && v[r] == kv.null_val;
        }
      }
#pragma unroll 1
      for (int o = 0; o < a.nops; ++o) {
        if (a.op_val[o] != vi) {
          continue;
        }
        const int32_t kind = a.op_kind[o];
        const uint32_t woff = static_cast<uint32_t>(a.op_word[o]) << a.rep_shift;
        const bool on_null = kind == FOP_ADD_ONE_IF_NULL;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (pass[r] && isnull[r] == on_null) {
            fast_lds_op(kind, lds + base[r] + woff, v[r]);
          }
        }
      }
    }
  }
  return slow;
}
#undef HDK_KEYS_LOAD

// counting form: fold the 32-bit LDS counters into this block's slab (and clear them)
HDK_DEV void keys_fold_counts(const KeysArgs& a, int64_t* lds, int64_t* slab, int tid, bool first) {
  uint32_t* cnt = reinterpret_cast<uint32_t*>(lds);
  const uint32_t rep = a.rep;
  for (uint32_t i = tid; i < a.entry_count; i += blockDim.x) {
    int64_t acc = first ? 0 : slab[i];
    for (uint32_t r = 0; r < rep; ++r) {
      acc += cnt[i * rep + r];
      cnt[i * rep + r] = 0;
    }
    slab[i] = acc;
  }
}

template <bool VALS, int R, int BS = kKeysBlock>
// waves per SIMD the register budget is held to: the tile body needs 81 / 87 VGPRs (counting / value form)
__global__ __launch_bounds__(BS, VALS ? 5 : 6) void hdk_scan_agg_keys(KeysArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  constexpr int VR = kKeysVR;
  const int tid = threadIdx.x;
  const uint32_t rep = a.rep;
  const uint32_t wpe = VALS ? static_cast<uint32_t>(a.wpe) : 1u;  // counting form: one word (the row count) per entry
  const uint32_t total_words = a.entry_count * wpe * rep;
  if (VALS) {
    for (uint32_t i = tid; i < total_words; i += BS) {
      lds[i] = word_identity(a.wop[(i >> a.rep_shift) % wpe]);
    }
  } else {
    for (uint32_t i = tid; i < total_words; i += BS) {
      reinterpret_cast<uint32_t*>(lds)[i] = 0;
    }
  }
  __syncthreads();

  const uint32_t my_rep = tid & (rep - 1);
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(BS) * VR;
  int32_t err = 0;
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * a.entry_count * wpe;
  // counting form: a 32-bit counter can take 2^32 - 1 rows; the block folds its counters into the slab well before.
  // (A lane that the watch stopped leaves the tile loops early; its block's results are discarded with the error.)
  uint32_t tiles_since_fold = 0;
  bool folded = false;
  constexpr uint32_t kFoldEvery = (1u << 31) / static_cast<uint32_t>(kTileRows);

  int64_t tile = blockIdx.x;
  const Watch watch = watch_begin(a.kp);
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      if (row0 + kTileRows <= nrows) {
        uint32_t slow = keys_tile<VALS, R, BS>(a, lds, cols, row0, tid, my_rep, wpe, err);
        while (slow) {  // rare: rows outside the 32-bit ranges of the transforms
          const int r = __ffs(slow) - 1;
          slow &= slow - 1;
          keys_row<VALS>(a, lds, cols, row0 + static_cast<int64_t>((r / R) * BS + tid) * R + (r % R), my_rep, wpe, err);
        }
      } else {
        for (int64_t rr = row0 + tid; rr < nrows; rr += BS) {  // the ragged last tile of the fragment
          keys_row<VALS>(a, lds, cols, rr, my_rep, wpe, err);
        }
      }
      if (!VALS && ++tiles_since_fold == kFoldEvery) {
        __syncthreads();
        keys_fold_counts(a, lds, slab, tid, !folded);
        __syncthreads();
        folded = true;
        tiles_since_fold = 0;
      }
    }
    frag_tile_begin += ntiles;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
  __syncthreads();
  if (!VALS) {
    keys_fold_counts(a, lds, slab, tid, !folded);
    return;
  }
  const uint32_t ew = a.entry_count * wpe;
  for (uint32_t i = tid; i < ew; i += BS) {
    const uint32_t w = i % wpe;
    const int32_t op = a.wop[w];
    int64_t acc = lds[i * rep];
    for (uint32_t r = 1; r < rep; ++r) {
      acc = word_combine(op, acc, lds[i * rep + r]);
    }
    if ((a.nword_mask >> w) & 1u) {  // NULL count -> non-null count
      int64_t rows = 0;
      for (uint32_t r = 0; r < rep; ++r) {
        rows += lds[(i - w) * rep + r];
      }
      acc = rows - acc;
    }
    slab[i] = acc;
  }
}

}  // namespace hdk
