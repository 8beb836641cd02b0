// agg_common.h -- the LDS-privatised partial-aggregate representation shared by the scan kernels
// and the finalise kernel.
//
// Per output entry the scan keeps `WPE` 64-bit words:
//   word 0                      row count of the entry (rows that passed the filter / join)
//   per target, in order        [value word]   SUM/AVG: running sum (u64 wrap or f64)
//                                               MIN/MAX: running min/max (i64 or f64)
//                               [non-null word] count of non-NULL arguments (skip_null targets,
//                                               incl. COUNT(x))
// COUNT(*) and projected keys (HDK_AGG_ID) need no words: the row count / the entry index carry them.
// The words are summed over lanes in LDS (replicated REP times to dodge same-address
// serialisation), flushed once per block to a [block][entry][word] slab in the workspace, and the
// finalise kernel folds the slabs into the reference's output buffer with the exact
// agg_*[_skip_val] semantics (QE/RuntimeFunctions.cpp:387-875) -- so repeated launches accumulate
// into one buffer exactly like repeated row-function calls would.
#pragma once
#include "device_common.h"

namespace hdk {

enum WordOp : int32_t {
  WOP_ADD_U64 = 0,
  WOP_ADD_F64 = 1,
  WOP_MIN_I64 = 2,
  WOP_MAX_I64 = 3,
  WOP_MIN_F64 = 4,
  WOP_MAX_F64 = 5
};

constexpr int kMaxWordsPerEntry = 1 + 2 * HDK_HIP_MAX_TARGETS;

struct WordLayout {
  int32_t wpe;                             // words per entry
  int32_t vword[HDK_HIP_MAX_TARGETS];      // value word of target t, or -1
  int32_t nword[HDK_HIP_MAX_TARGETS];      // non-null-count word of target t, or -1.  In LDS the word counts
                                           // the NULL rows instead (rare, so it costs an atomic only on a NULL);
                                           // the slab flush stores rowcount - nulls
  int32_t wop[kMaxWordsPerEntry];          // combine op per word
  int32_t is_nword[kMaxWordsPerEntry];     // word w is some target's non-null-count word
};

__host__ __device__ inline bool target_has_value_word(const hdk_hip_target& tg) {
  return tg.agg == HDK_AGG_SUM || tg.agg == HDK_AGG_AVG || tg.agg == HDK_AGG_MIN || tg.agg == HDK_AGG_MAX;
}
__host__ __device__ inline bool target_has_nn_word(const hdk_hip_target& tg) {
  return tg.agg != HDK_AGG_ID && tg.has_arg && tg.skip_null;
}

__host__ __device__ inline void make_word_layout(const hdk_hip_plan* p, WordLayout* wl) {
  int w = 1;
  wl->wop[0] = WOP_ADD_U64;
  for (int i = 0; i < kMaxWordsPerEntry; ++i) {
    wl->is_nword[i] = 0;
  }
  for (int t = 0; t < HDK_HIP_MAX_TARGETS; ++t) {
    wl->vword[t] = -1;
    wl->nword[t] = -1;
  }
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (target_has_value_word(tg)) {
      wl->vword[t] = w;
      if (tg.agg == HDK_AGG_SUM || tg.agg == HDK_AGG_AVG) {
        wl->wop[w] = tg.arg_is_fp ? WOP_ADD_F64 : WOP_ADD_U64;
      } else if (tg.agg == HDK_AGG_MIN) {
        wl->wop[w] = tg.arg_is_fp ? WOP_MIN_F64 : WOP_MIN_I64;
      } else {
        wl->wop[w] = tg.arg_is_fp ? WOP_MAX_F64 : WOP_MAX_I64;
      }
      ++w;
    }
    if (target_has_nn_word(tg)) {
      wl->nword[t] = w;
      wl->wop[w] = WOP_ADD_U64;
      wl->is_nword[w] = 1;
      ++w;
    }
  }
  wl->wpe = w;
}

HDK_DEV int64_t word_identity(int32_t op) {
  switch (op) {
    case WOP_MIN_I64: return INT64_MAX;
    case WOP_MAX_I64: return INT64_MIN;
    case WOP_MIN_F64: return 0x7FF0000000000000LL;                        // +inf
    case WOP_MAX_F64: return static_cast<int64_t>(0xFFF0000000000000ULL); // -inf
    default: return 0;  // 0 and +0.0
  }
}

HDK_DEV int64_t word_combine(int32_t op, int64_t a, int64_t b) {
  switch (op) {
    case WOP_ADD_F64: return double_to_bits(bits_to_double(a) + bits_to_double(b));
    case WOP_MIN_I64: return a < b ? a : b;
    case WOP_MAX_I64: return a > b ? a : b;
    case WOP_MIN_F64: return bits_to_double(b) < bits_to_double(a) ? b : a;
    case WOP_MAX_F64: return bits_to_double(a) < bits_to_double(b) ? b : a;
    default: return static_cast<int64_t>(static_cast<uint64_t>(a) + static_cast<uint64_t>(b));
  }
}

}  // namespace hdk
