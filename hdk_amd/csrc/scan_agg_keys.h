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

namespace hdk {

constexpr int kKeysBlock = 256;
constexpr int kKeysMaxVals = 2;
constexpr int kKeysMaxOps = 8;
constexpr int kKeysVR = 8;   // rows per lane and tile (taxi Q3/Q4 at 256 M rows: 4 -> 0.79/1.23 ms, 8 -> 0.60/1.00, 12 -> 0.66/1.02)
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
  int32_t narrow;          // plain key whose column, NULL and kmin all fit 32 bits: the whole term is 32-bit
  int32_t kmin32;
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

#define HDK_KEYS_LOAD(DST, T)                              \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {         \
    DST[r] = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) T*>( \
                                            reinterpret_cast<uintptr_t>(buf)) + row[r]);               \
  }

template <bool VALS>
__global__ __launch_bounds__(kKeysBlock) void hdk_scan_agg_keys(KeysArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  constexpr int VR = kKeysVR;
  const int tid = threadIdx.x;
  const uint32_t rep = a.rep;
  const uint32_t wpe = VALS ? static_cast<uint32_t>(a.wpe) : 1u;  // counting form: one word (the row count) per entry
  const uint32_t total_words = a.entry_count * wpe * rep;
  for (uint32_t i = tid; i < total_words; i += kKeysBlock) {
    lds[i] = VALS ? word_identity(a.wop[(i >> a.rep_shift) % wpe]) : 0;
  }
  __syncthreads();

  const uint32_t my_rep = tid & (rep - 1);
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(kKeysBlock) * VR;
  const int nk = a.nkeys;
  int32_t err = 0;

  int64_t tile = blockIdx.x;
  const Watch watch = watch_begin();
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows + tid;
      bool pass[VR];
      int64_t row[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const int64_t rr = row0 + static_cast<int64_t>(r) * kKeysBlock;
        pass[r] = rr < nrows;
        row[r] = pass[r] ? rr : row0 - tid;  // dead slots re-read the tile's first row; their results are dropped
      }
      if (a.nquals) {
        plain_quals_pass<VR>(a.q, a.nquals, cols, row, pass, true);
      }
      uint32_t entry[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        entry[r] = 0;
      }
      int bad = 0;  // some live row fell outside the range the table was sized for
#pragma unroll 1  // one copy of the decoders and transforms: three unrolled copies spill SGPRs and triple the code
      for (int k = 0; k < nk; ++k) {
        const KeysKey kk = a.key[k];
        const int8_t* buf = cols[kk.buf_idx];
        // term[r] = key - kmin as a 32-bit number, ok[r] = it lies in [0, card).  Everything that can be decided
        // in 32 bits is: the 64-bit forms are kept for wide plain keys and for rows outside the fast ranges.
        uint32_t term[VR];
        bool ok[VR];
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
          const bool translate = kk.translate != 0;
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            const bool isnull = translate && v[r] == null32;  // perfect hash: the NULL key owns the slot past the range
            const uint32_t d = static_cast<uint32_t>(v[r]) - static_cast<uint32_t>(kk.kmin32);
            term[r] = isnull ? kk.null_term : d;
            ok[r] = isnull ? kk.null_ok != 0 : d < kk.card;
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
          bool isnull[VR];
          if (kk.xf == KXF_NONE) {  // wide plain key
            const bool translate = kk.translate != 0;
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              isnull[r] = translate && v[r] == kk.key_null;
              const uint64_t d = static_cast<uint64_t>(v[r]) - static_cast<uint64_t>(kk.kmin);
              term[r] = static_cast<uint32_t>(d);
              ok[r] = d < kk.card;
            }
          } else {
            // unary step with the *_nullable convention (NULL in, the step's NULL out; the matcher folded what
            // that NULL contributes into null_term / null_ok).  Both steps have a 32-bit form that covers
            // practically every row and a 64-bit form that is 5-10x longer: the short form runs straight-line
            // for all VR rows, the long form is entered only by lanes that hold a row outside the fast range.
            const bool nullable = kk.col_nullable != 0;
            int need_long = 0;  // integer or-accumulation: a bool assigned under `if` compiles to an EXEC branch per row
            bool fast[VR];
            if (kk.xf == KXF_YEAR) {
              // extract_year's fast range (device_common.h; reference Utils/ExtractFromTime.cpp:150-166)
              constexpr uint32_t kEpochOffsetYear1900 = 2208988800u;
              constexpr uint32_t kSecsJanToMar1900 = 5097600u;
              constexpr uint32_t kSecondsPer4YearCycle = 126230400u;
              constexpr uint32_t kSecsPerDay = 86400u;
              constexpr uint32_t kSecondsPerNonLeapYear = 31536000u;
#pragma unroll
              for (int r = 0; r < VR; ++r) {
                isnull[r] = nullable && v[r] == kk.col_null;
                fast[r] = static_cast<uint64_t>(v[r]) <= static_cast<uint64_t>(UINT32_MAX - kEpochOffsetYear1900);
                const uint32_t seconds_1900 = static_cast<uint32_t>(v[r]) + kEpochOffsetYear1900;
                const uint32_t leap_years = (seconds_1900 - kSecsJanToMar1900) / kSecondsPer4YearCycle;
                const uint32_t year = (seconds_1900 - leap_years * kSecsPerDay) / kSecondsPerNonLeapYear + 1900;
                const uint32_t d = year - static_cast<uint32_t>(kk.kmin32);
                term[r] = d;
                ok[r] = d < kk.card;
                need_long |= static_cast<int>(!fast[r]) & static_cast<int>(!isnull[r]);
              }
              if (need_long) {
#pragma unroll
                for (int r = 0; r < VR; ++r) {
                  if (!fast[r] && !isnull[r]) {
                    const uint64_t d = static_cast<uint64_t>(extract_year(v[r])) - static_cast<uint64_t>(kk.kmin);
                    term[r] = static_cast<uint32_t>(d);
                    ok[r] = d < kk.card;
                  }
                }
              }
            } else {
              // scale_decimal_down (device_common.h; reference QE/RuntimeFunctions.cpp:245-262): round half away
              // from zero, then divide by the scale, truncating
              const int64_t half = kk.param >> 1;
              int64_t tmp[VR];
#pragma unroll
              for (int r = 0; r < VR; ++r) {
                isnull[r] = nullable && v[r] == kk.col_null;
                tmp[r] = v[r] >= 0 ? v[r] + half : v[r] - half;
                const int32_t t32 = static_cast<int32_t>(tmp[r]);
                fast[r] = t32 == tmp[r] && t32 != INT32_MIN;
                const uint32_t n = static_cast<uint32_t>(t32 < 0 ? -t32 : t32);
                const uint32_t hi = __umulhi(kk.div_magic, n);
                const uint32_t q = (((n - hi) >> 1) + hi) >> kk.div_shift;
                const uint32_t sq = t32 < 0 ? 0u - q : q;
                const uint32_t d = sq - static_cast<uint32_t>(kk.kmin32);
                term[r] = d;
                ok[r] = d < kk.card;
                need_long |= static_cast<int>(!fast[r]) & static_cast<int>(!isnull[r]);
              }
              if (need_long) {
#pragma unroll
                for (int r = 0; r < VR; ++r) {
                  if (!fast[r] && !isnull[r]) {
                    const uint64_t d = static_cast<uint64_t>(tmp[r] / kk.param) - static_cast<uint64_t>(kk.kmin);
                    term[r] = static_cast<uint32_t>(d);
                    ok[r] = d < kk.card;
                  }
                }
              }
            }
          }
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            term[r] = isnull[r] ? kk.null_term : term[r];
            ok[r] = isnull[r] ? kk.null_ok != 0 : ok[r];
          }
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          bad |= static_cast<int>(pass[r]) & static_cast<int>(!ok[r]);
          pass[r] = pass[r] && ok[r];
          entry[r] += __umul24(term[r], kk.stride);  // term < card, stride <= entry_count: both far below 2^24
        }
      }
      if (bad) {
        err = HDK_HIP_ERR_OUT_OF_SLOTS;
      }
      if (!VALS) {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (pass[r]) {
            atomicAdd(reinterpret_cast<unsigned long long*>(lds + ((entry[r] << a.rep_shift) + my_rep)), 1ull);
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
    }
    frag_tile_begin += ntiles;
  }
#undef HDK_KEYS_LOAD
  if (err) {
    record_error(a.kp.error_code, err);
  }
  __syncthreads();
  const uint32_t ew = a.entry_count * wpe;
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * ew;
  for (uint32_t i = tid; i < ew; i += kKeysBlock) {
    if (!VALS) {
      int64_t acc = 0;
      for (uint32_t r = 0; r < rep; ++r) {
        acc += lds[i * rep + r];
      }
      slab[i] = acc;
    } else {
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
}

}  // namespace hdk
