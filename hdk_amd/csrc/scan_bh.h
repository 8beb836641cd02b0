// scan_bh.h -- open-addressing (GroupByBaselineHash) group-by ON CHIP, for tables that are small.
//
// The reference's GPU path for a baseline-hash layout is get_group_value + agg_*_shared on the final table for every row
// (QE/GroupByRuntime.cpp:31-55, QE/cuda_mapd_rt.cu:167-203,424-478): with ten or a thousand groups that is a billion
// memory-side atomics onto a handful of addresses.  The shapes are real: cast(x as double) keys
// (Benchmarks/synthetic_benchmark/queries/BaselineHash/BH001-005.sql), keys without an expression range such as
// `dval % 64` (QE/ExpressionRange.cpp:391-397), the 16 384-entry default guess (Shared/Config.h:42).
//
// Here every block keeps a PRIVATE open-addressing table in LDS -- capacity C = the plan's entry count rounded up to a
// power of two, so it can never fill before the output table does -- with the partial-aggregate words of agg_common.h
// behind each key: a row costs one ds_read of its home key (a ds_cmpswap the first time a key is seen) and the same LDS
// atomics as a perfect-hash plan.  At the end of the scan each block re-inserts its groups into the output table through
// the reference's probe sequence (find_or_claim, baseline_table.h) and folds its partial words in with global atomics --
// reduceOneEntryBaseline's job (QE/ResultSetReduction.cpp:694-731), one partial per (block, group) instead of per row.
// The LDS table's own hash is free to choose (it never leaves the block): a multiplicative hash of the 64-bit key word.
//
// LDS words: entry e, word w, replica r at ((e * W + w) * rep + r), W = wpe + 1, word `wpe` = the key word.  Replicas are
// independent tables (a lane works in replica tid % rep): few groups mean few LDS addresses, and same-address atomics of
// one wave serialise.  Before the flush the replicas are merged into replica 0 inside LDS.
// Keys: ONE 64-bit key word -- one key column of either width, or two 4-byte keys packed the way the table packs them.
#pragma once
#include "agg_common.h"
#include "baseline_table.h"
#include "scan_agg_global.h"
#include "scan_bh_decl.h"

namespace hdk {

constexpr int64_t kBhEmpty = HDK_EMPTY_KEY_64;  // never a valid key word (see the planner's pick_baseline_key_width)

HDK_DEV void bh_layout_identity(const WordLayout& wl, BhLdsLayout* ll) {
  ll->nlw = wl.wpe;
  for (int w = 0; w < kMaxWordsPerEntry; ++w) {
    ll->lwop[w] = w < wl.wpe ? wl.wop[w] : 0;
    ll->lmap[w] = w;
  }
}

HDK_DEV uint32_t bh_lds_home(int64_t key, uint32_t cap_log2) {
  return static_cast<uint32_t>((static_cast<uint64_t>(key) * 0x9E3779B97F4A7C15ull) >> (64 - cap_log2));
}

// entry of `key` in the lane's replica, claiming it when new; -1: the table is full (more groups than the plan's table holds)
HDK_DEV int32_t bh_lds_find_or_claim(int64_t* lds, int64_t key, uint32_t cap_log2, uint32_t estride, uint32_t key_off) {
  const uint32_t mask = (1u << cap_log2) - 1;
  uint32_t e = cap_log2 ? bh_lds_home(key, cap_log2) : 0u;
  int32_t found = -2;
  uint32_t steps = 0;
  while (found == -2) {
    unsigned long long* kp = reinterpret_cast<unsigned long long*>(lds + __umul24(e, estride) + key_off);
    unsigned long long cur = *kp;
    if (cur == static_cast<unsigned long long>(kBhEmpty)) {
      cur = atomicCAS(kp, static_cast<unsigned long long>(kBhEmpty), static_cast<unsigned long long>(key));
      if (cur == static_cast<unsigned long long>(kBhEmpty)) {
        cur = static_cast<unsigned long long>(key);
      }
    }
    if (cur == static_cast<unsigned long long>(key)) {
      found = static_cast<int32_t>(e);
    } else {
      e = (e + 1) & mask;
      if (++steps > mask) {
        found = -1;
      }
    }
  }
  return found;
}

HDK_DEV void bh_lds_word_op(int32_t wop, int64_t* wp, int64_t v) {
  switch (wop) {
    case WOP_ADD_U64: atomicAdd(reinterpret_cast<unsigned long long*>(wp), static_cast<unsigned long long>(v)); break;
    case WOP_ADD_F64: atomicAdd(reinterpret_cast<double*>(wp), bits_to_double(v)); break;
    case WOP_MIN_I64: atomicMin(reinterpret_cast<long long*>(wp), static_cast<long long>(v)); break;
    case WOP_MAX_I64: atomicMax(reinterpret_cast<long long*>(wp), static_cast<long long>(v)); break;
    case WOP_MIN_F64: {
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

// the LDS table before the scan: empty keys, identity words
HDK_DEV void bh_lds_init(int64_t* lds, const BhLdsLayout& ll, uint32_t cap_log2, uint32_t rep, int tid, int block) {
  const uint32_t W = static_cast<uint32_t>(ll.nlw) + 1;
  const uint32_t total = (W << cap_log2) * rep;
  for (uint32_t i = tid; i < total; i += block) {
    const uint32_t w = (i / rep) % W;
    lds[i] = w == static_cast<uint32_t>(ll.nlw) ? kBhEmpty : word_identity(ll.lwop[w]);
  }
}

// One group's partial folded into the output table: the key goes through the reference's probe sequence, the words
// through agg_*_shared's atomics with the partial in place of a row's value (hdk_finalize's apply_* with atomics).
// word(w): word w of the layout (agg_common.h): word 0 = rows of the group, value words the partial sum / min / max,
// NULL-count words the NULL rows (as the scans count them).
template <class WordFn>
HDK_DEV void bh_fold_group_fn(const hdk_hip_plan* p, const TableShape shape, const WordLayout& wl, int64_t* buf, uint32_t entry_count,
                              const uint64_t* col_off, int64_t keyword, WordFn word, int32_t& err) {
  int64_t entry;
  bool fresh = false;
  int64_t kv[2] = {keyword, 0};
  if (p->query_kind == HDK_Q_PERFECT_HASH) {
    // a GroupByPerfectHash table fed from the same LDS tables (one key column, no bucket): get_group_value_fast[_keyless]
    // (QE/GroupByRuntime.cpp:198-246) -- entry = key - min with a NULL key under its translated value, the key stored when
    // the layout keeps one; `keyword` is the key column's value
    int64_t k = keyword;
    if (p->key_has_nulls[0] && p->keys[0].nullable && k == p->keys[0].null_val) {
      k = p->key_null_translated[0];
    }
    const uint64_t e = static_cast<uint64_t>(k - p->key_min[0]);
    if (e >= entry_count) {
      err = HDK_HIP_ERR_OUT_OF_SLOTS;  // a key outside the range the layout was sized for
      return;
    }
    entry = static_cast<int64_t>(e);
    fresh = true;  // (a projected key is stored by every writer: the same value)
    if (!p->keyless) {
      int64_t* kp = p->output_columnar ? buf + e : buf + e * p->row_size_quad;
      g_store_i64(kp, k);
    }
  } else if (p->key_width == 4) {
    const int32_t key[HDK_HIP_MAX_KEYS] = {static_cast<int32_t>(keyword), static_cast<int32_t>(static_cast<uint64_t>(keyword) >> 32), 0, 0};
    kv[0] = key[0];
    kv[1] = key[1];
    entry = find_or_claim<int32_t>(shape, buf, entry_count, key, &fresh);
  } else {
    const int64_t key[HDK_HIP_MAX_KEYS] = {keyword, 0, 0, 0};
    entry = find_or_claim<int64_t>(shape, buf, entry_count, key, &fresh);
  }
  if (entry < 0) {
    err = HDK_HIP_ERR_OUT_OF_SLOTS;  // get_group_value returned NULL
    return;
  }
  const bool columnar = p->output_columnar;
  int8_t* rowb = reinterpret_cast<int8_t*>(buf + static_cast<size_t>(entry) * p->row_size_quad);
  const int64_t rowcount = word(0);
  const int nt = p->num_targets;
  int slot_idx = 0;
  for (int t = 0; t < nt; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    int8_t* s1;
    int8_t* s2 = nullptr;
    if (columnar) {
      s1 = reinterpret_cast<int8_t*>(buf) + col_off[slot_idx] + static_cast<size_t>(entry) * tg.slot_width;
      if (tg.agg == HDK_AGG_AVG) {
        s2 = reinterpret_cast<int8_t*>(buf) + col_off[slot_idx + 1] + static_cast<size_t>(entry) * tg.slot2_width;
      }
    } else {
      s1 = rowb + tg.slot_off;
      s2 = rowb + tg.slot2_off;
    }
    slot_idx += tg.agg == HDK_AGG_AVG ? 2 : 1;
    const int vw = wl.vword[t];
    const int nw = wl.nword[t];
    const int64_t nn = nw >= 0 ? rowcount - word(nw) : rowcount;
    if (tg.agg == HDK_AGG_ID) {
      if (tg.slot_width && fresh) {  // (a row-wise baseline table keeps no slot for a projected key)
        const int64_t k = kv[tg.key_idx ? 1 : 0];
        if (tg.slot_width == 4) {
          g_store_i32(reinterpret_cast<int32_t*>(s1), static_cast<int32_t>(k));
        } else {
          g_store_i64(reinterpret_cast<int64_t*>(s1), k);
        }
      }
      continue;
    }
    if (tg.agg == HDK_AGG_COUNT || tg.agg == HDK_AGG_AVG) {
      int8_t* cs = tg.agg == HDK_AGG_COUNT ? s1 : s2;
      const int cw = tg.agg == HDK_AGG_COUNT ? tg.slot_width : tg.slot2_width;
      if (nn) {
        if (cw == 4) {
          atomicAdd(reinterpret_cast<unsigned int*>(cs), static_cast<unsigned int>(nn));
        } else {
          atomicAdd(reinterpret_cast<unsigned long long*>(cs), static_cast<unsigned long long>(nn));
        }
      }
      if (tg.agg == HDK_AGG_COUNT) {
        continue;
      }
    }
    if (vw < 0 || (tg.skip_null && nn == 0)) {
      continue;  // nothing but NULLs: the slot keeps its value
    }
    const int64_t partial = word(vw);
    if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) {  // float accumulator in the slot's low 4 bytes; the block partial is a double
      g_aggf32(tg.agg, tg.skip_null, float_slot_null(tg), reinterpret_cast<int32_t*>(s1), static_cast<float>(bits_to_double(partial)));
    } else if (tg.slot_width == 4) {
      g_agg32(tg.agg, tg.skip_null, static_cast<int32_t>(tg.null_val), reinterpret_cast<int32_t*>(s1), static_cast<int32_t>(partial));
    } else {
      g_agg64(tg.agg, tg.arg_is_fp != 0, tg.skip_null, tg.null_val, reinterpret_cast<int64_t*>(s1), partial);
    }
  }
}

// ew: the group's LDS words, word w of the layout at ew[ll.lmap[w] * wstride]
HDK_DEV void bh_fold_group(const hdk_hip_plan* p, const TableShape shape, const WordLayout& wl, const BhLdsLayout& ll, int64_t* buf,
                           uint32_t entry_count, const uint64_t* col_off, int64_t keyword, const int64_t* ew, uint32_t wstride,
                           int32_t& err) {
  bh_fold_group_fn(p, shape, wl, buf, entry_count, col_off, keyword,
                   [&](int w) -> int64_t { return ew[static_cast<uint32_t>(ll.lmap[w]) * wstride]; }, err);
}

// End of a block's scan: merge replicas 1.. into replica 0, then either fold replica 0's groups into the output table, or
// (slab != nullptr) leave replica 0 -- cap x (nlw + 1) words, entry-major -- in the block's slab for hdk_bh_fold_words.
// s_col_off: [2 * HDK_HIP_MAX_TARGETS] LDS words for the columnar slot offsets (filled here).
template <int BLOCK>
HDK_DEV void bh_flush_block(const hdk_hip_plan* p, const WordLayout& wl, const BhLdsLayout& ll, int64_t* lds, const BhGeom& g,
                            int64_t* const* groupby_buf, uint64_t* s_col_off, int tid, int32_t& err, int64_t* slab) {
  const uint32_t W = static_cast<uint32_t>(ll.nlw) + 1;
  const uint32_t rep = g.rep;
  const uint32_t cap = 1u << g.cap_log2;
  const uint32_t estride = W * rep;
  const uint32_t key_off = static_cast<uint32_t>(ll.nlw) * rep;
  if (p->output_columnar && tid < 2 * HDK_HIP_MAX_TARGETS) {
    s_col_off[tid] = columnar_slot_off(p, g.out_entry_count, tid);
  }
  __syncthreads();
  // ---- replicas r >= 1 into replica 0 (LDS atomics; the claim is the scan's) ---------------------------------------
  if (rep > 1) {
    const uint32_t n = cap * (rep - 1);
    for (uint32_t i = tid; i < n; i += BLOCK) {
      const uint32_t e = i / (rep - 1), r = 1 + i % (rep - 1);
      const int64_t key = lds[e * estride + key_off + r];
      if (key == kBhEmpty) {
        continue;
      }
      const int32_t e0 = bh_lds_find_or_claim(lds, key, g.cap_log2, estride, key_off);  // replica 0: offset 0
      if (e0 < 0) {
        err = HDK_HIP_ERR_OUT_OF_SLOTS;
        continue;
      }
      for (int w = 0; w < ll.nlw; ++w) {
        const int64_t v = lds[(e * W + w) * rep + r];
        if (v != word_identity(ll.lwop[w])) {
          bh_lds_word_op(ll.lwop[w], lds + (static_cast<uint32_t>(e0) * W + w) * rep, v);
        }
      }
    }
    __syncthreads();
  }
  if (slab) {
    const uint32_t n = cap * W;
    for (uint32_t i = tid; i < n; i += BLOCK) {
      slab[i] = lds[i * rep];
    }
    return;
  }
  // ---- replica 0 into the output table ------------------------------------------------------------------------------
  const TableShape shape = table_shape(p);
  int64_t* buf = groupby_buf[0];
  for (uint32_t e = tid; e < cap; e += BLOCK) {
    const int64_t key = lds[e * estride + key_off];
    if (key == kBhEmpty) {
      continue;
    }
    bh_fold_group(p, shape, wl, ll, buf, g.out_entry_count, s_col_off, key, lds + e * estride, rep, err);
  }
}

// ---- the fold of the scan blocks' slabs (two-level fold) ------------------------------------------------------------------
// Folding from every scan block costs (blocks x groups) contended memory-side atomics: 0.3 ms for 10 groups, 3.6 ms for
// 1000 (measured with the packed form, profiles/r05_bh_configs.txt).  Instead the blocks leave their tables in slabs and block
// (slice, group) of this kernel merges entry range `slice` of slabs group, group + fold_groups, ... in LDS (same hash, same
// positions as the scan: a key sits in the same range of every slab, give or take a probe over the range's end), then
// folds each group of its table into the output table.
struct BhFoldArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  BhGeom g;          // rep is 1 here
  BhLdsLayout ll;
  const int64_t* slabs;
  uint32_t num_slabs;
  uint32_t fold_slices;  // power of two, <= cap
  uint32_t fold_groups;
  uint32_t pad_;
};
constexpr int kBhFoldWordsBlock = 256;
HDK_DEV void bh_fold_words_body(const BhFoldArgs& a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  __shared__ WordLayout wl;
  __shared__ BhLdsLayout s_ll;
  __shared__ uint64_t s_col_off[2 * HDK_HIP_MAX_TARGETS];
  const int tid = threadIdx.x;
  if (tid == 0) {
    make_word_layout(a.plan, &wl);
    s_ll = a.ll;
  }
  __syncthreads();
  bh_lds_init(lds, s_ll, a.g.cap_log2, 1, tid, kBhFoldWordsBlock);
  __syncthreads();
  const uint32_t W = static_cast<uint32_t>(s_ll.nlw) + 1;
  const uint32_t cap = 1u << a.g.cap_log2;
  const uint32_t per_slice = cap / a.fold_slices;
  const uint32_t slice = blockIdx.x % a.fold_slices, group = blockIdx.x / a.fold_slices;
  int32_t err = 0;
  for (uint32_t sl = group; sl < a.num_slabs; sl += a.fold_groups) {
    const int64_t* slab = a.slabs + static_cast<size_t>(sl) * cap * W;
    for (uint32_t i = tid; i < per_slice; i += kBhFoldWordsBlock) {
      const uint32_t e = slice * per_slice + i;
      const int64_t key = slab[e * W + s_ll.nlw];
      if (key == kBhEmpty) {
        continue;
      }
      const int32_t e0 = bh_lds_find_or_claim(lds, key, a.g.cap_log2, W, static_cast<uint32_t>(s_ll.nlw));
      if (e0 < 0) {
        err = HDK_HIP_ERR_OUT_OF_SLOTS;
        continue;
      }
      for (int w = 0; w < s_ll.nlw; ++w) {
        const int64_t v = slab[e * W + w];
        if (v != word_identity(s_ll.lwop[w])) {
          bh_lds_word_op(s_ll.lwop[w], lds + static_cast<uint32_t>(e0) * W + w, v);
        }
      }
    }
  }
  BhGeom g = a.g;
  g.rep = 1;
  bh_flush_block<kBhFoldWordsBlock>(a.plan, wl, s_ll, lds, g, a.kp.groupby_buf, s_col_off, tid, err, nullptr);
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
