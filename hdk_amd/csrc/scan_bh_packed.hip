// scan_bh_packed.hip -- instantiations, matcher and launcher of the packed on-chip open-addressing group-by
// (scan_bh_packed.h): the reference's BaselineHash benchmark shape at three LDS operations a row.
#include <string.h>

#include "host_match.h"
#include "scan_bh_packed.h"
#include "scan_bh_dense_part.h"
#include "scan_bh_host.h"
#include "scan_bhm_host.h"

namespace hdk {

constexpr uint32_t kBhPackedReplicatedBytes = 32u << 10;  // replicas while the table is tiny
constexpr uint32_t kBhPackedMaxBytes256 = 52u << 10;      // one replica, 256-thread blocks (three on a CU)
constexpr uint32_t kBhPackedMaxBytes512 = 100u << 10;     // one replica, one 512-thread block per CU
// |argument| below this keeps a block's flush interval at 2^20 rows or more (|sum| < 2^39)
constexpr int64_t kBhPackedMaxAbsVal = (1ll << 19) - 1;

static bool bh_packed_off() {
  const bool off = hdk_sw(SW_NO_BH_PACKED) != nullptr || hdk_sw(SW_NO_BH_LDS) != nullptr;  // (A/B measurements)
  return off;
}

// the part of the match that does not depend on the table's size: columns, statistics, the word kinds
static bool match_bh_packed_shape(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhPackedArgs* a, int* kw_out, int* vw_out) {
  // GroupByBaselineHash, or a GroupByPerfectHash layout of the same shape (one plain key column, no bucket): the LDS side is
  // the same, the fold addresses the table by key - min instead of the probe sequence (scan_bh.h: bh_fold_group_fn)
  const bool perfect = p->query_kind == HDK_Q_PERFECT_HASH;
  if (bh_packed_off() || (p->query_kind != HDK_Q_BASELINE_HASH && !perfect)) return false;
  if (perfect && (p->key_count != 1 || p->key_bucket[0] > 1 || p->keys[0].nsteps != 0)) return false;
  if (ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS | HDK_HIP_LAUNCH_FORCE_PARTITIONED))) return false;
  if (launch_forces_generic(ko)) return false;
  if (p->num_joins || p->key_count != 1 || (p->key_width != 4 && p->key_width != 8)) return false;
  memset(a, 0, sizeof(*a));
  if (!match_plain_quals(p, a->q, true)) return false;
  a->nquals = p->num_quals;
  const hdk_hip_expr& ke = p->keys[0];
  if (ke.leaf0.kind != HDK_LEAF_COL) return false;
  const hdk_hip_col& kc = p->cols[ke.leaf0.col];
  if (kc.table != 0 || kc.kind != HDK_COL_INT || (kc.width != 4 && kc.width != 8)) return false;
  if (ke.nsteps == 1) {
    const hdk_hip_step& sp = ke.steps[0];
    if (sp.op == HDK_OP_CAST_INT_TO_FP && p->key_width == 8) {
      a->key_form = 1;
      a->key_null_out = sp.null_out;
    } else if (sp.op == HDK_OP_MOD && sp.out_class == HDK_VC_INT && sp.rhs.kind == HDK_LEAF_INT && sp.rhs.ival >= 1 &&
               sp.rhs.ival <= (1 << 15) && p->query_kind == HDK_Q_BASELINE_HASH && !hdk_sw(SW_NO_BH_MOD_KEYS)) {
      // column % m: no expression range for the planner (QE/ExpressionRange.cpp:391-419: a baseline-hash layout), but the
      // kernel knows the key lies in (-m, m) -- a dense table over that range (match_bh_packed; the general kernels)
      a->key_form = 2;
      a->key_mod = static_cast<int32_t>(sp.rhs.ival);
      if (a->key_mod >= 2) magic_u32(static_cast<uint32_t>(a->key_mod), &a->mod_magic, &a->mod_shift);
      a->key_null_out = ke.null_val;
    } else {
      return false;
    }
  } else if (ke.nsteps != 0) {
    return false;
  }
  a->key_nullable = ke.leaf0.nullable;
  a->key_null = ke.leaf0.null_val;
  a->key_buf_idx = kc.buf_idx;
  a->key_width = kc.width;
  a->key_min = INT32_MIN;
  a->key_max = INT32_MAX;
  // the key column's statistics, for the dense form (match_bh_packed decides)
  if (kc.has_stats && kc.max_val >= kc.min_val && kc.min_val > INT32_MIN && kc.max_val < INT32_MAX &&
      static_cast<uint64_t>(kc.max_val - kc.min_val) < (1u << 20)) {
    a->dense_min = static_cast<int32_t>(kc.min_val);
    a->dense_n = static_cast<uint32_t>(kc.max_val - kc.min_val) + 1;
  }
  if (a->key_form == 2) {  // the remainder's own range, whatever the column's: [0, m) when the statistics say >= 0, else (-m, m)
    const bool nonneg = kc.has_stats && kc.min_val >= 0;
    a->dense_min = nonneg ? 0 : -(a->key_mod - 1);
    a->dense_n = static_cast<uint32_t>(nonneg ? a->key_mod : 2 * a->key_mod - 1);
  }
  if (kc.width == 8) {  // an 8-byte key column rides as 32 bits when the statistics say it fits (strangers take the exact path)
    if (!kc.has_stats || kc.min_val < INT32_MIN || kc.max_val > INT32_MAX) return false;
    a->key_min = static_cast<int32_t>(kc.min_val);
    a->key_max = static_cast<int32_t>(kc.max_val);
  }
  WordLayout wl;
  make_word_layout(p, &wl);
  for (int w = 0; w < kMaxWordsPerEntry; ++w) a->wkind[w] = BHW_ROWS;
  bool have_val = false;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_SINGLE_VALUE) return false;
    if (tg.agg == HDK_AGG_ID) {
      if (tg.key_idx != 0 || (tg.slot_width != 0 && tg.slot_width != 4 && tg.slot_width != 8)) return false;
      continue;
    }
    if (tg.slot_width != 4 && tg.slot_width != 8) return false;
    if (tg.agg == HDK_AGG_AVG && tg.slot2_width != 4 && tg.slot2_width != 8) return false;
    if (!tg.has_arg) {
      if (tg.agg != HDK_AGG_COUNT) return false;
      continue;
    }
    int c;
    if (!plain_outer_col(p, tg.arg, &c)) return false;
    const hdk_hip_col& col = p->cols[c];
    if (col.kind != HDK_COL_INT || (col.width != 4 && col.width != 8) || tg.arg_is_fp) return false;
    if (!col.has_stats || col.min_val < -kBhPackedMaxAbsVal || col.max_val > kBhPackedMaxAbsVal) return false;
    const int nullable = tg.skip_null && tg.arg.nullable;
    if (have_val && (a->val_buf_idx != col.buf_idx || a->val_nullable != nullable)) return false;
    have_val = true;
    a->val_buf_idx = col.buf_idx;
    a->val_width = col.width;
    a->val_nullable = nullable;
    a->val_null = tg.arg.null_val;
    a->val_min = static_cast<int32_t>(col.min_val);
    a->val_max = static_cast<int32_t>(col.max_val);
    if (wl.vword[t] >= 0) {
      a->wkind[wl.vword[t]] = tg.agg == HDK_AGG_MIN ? BHW_MIN : (tg.agg == HDK_AGG_MAX ? BHW_MAX : BHW_SUM);
      a->want_minmax = a->want_minmax || tg.agg == HDK_AGG_MIN || tg.agg == HDK_AGG_MAX;
    }
    if (wl.nword[t] >= 0) a->wkind[wl.nword[t]] = BHW_NULLS;
  }
  a->has_val = have_val;
  // rows a block may put into one entry before it folds its table: rows < 2^23, |sum| < 2^39
  const int64_t amax = have_val ? std::max<int64_t>(std::max<int64_t>(-static_cast<int64_t>(a->val_min), a->val_max), 1) : 1;
  const int64_t by_sum = ((1ll << 39) - 1) / amax;
  a->flush_rows = static_cast<uint32_t>(std::min<int64_t>(by_sum, 1ll << 23));
  a->out_entry_count = p->entry_count;
  *kw_out = kc.width;
  *vw_out = have_val ? a->val_width : 0;
  return true;
}

// LDS arrays of `cap` entries, `rep` replicas
static void bh_packed_geometry(BhPackedArgs* a, uint32_t cap_log2, uint32_t rep) {
  const uint32_t cap = (1u << cap_log2) + 4;  // (+ the dummy entry, padded to a bucket)
  a->cap_log2 = cap_log2;
  a->rep = rep;
  a->off_mm = cap;
  a->off_packed = 3 * cap;
  a->off_nulls = 5 * cap;
  uint32_t words = 6 * cap;
  // replica r starts 4 r banks further on: lanes that read the same bucket of different replicas do not collide
  if (rep > 1) {
    while (words % 64 != 4) words += 4;
  }
  a->rep_words = words;
}

static bool match_bh_packed(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhPackedArgs* a, int* kw, int* vw, int* block,
                            uint32_t* lds_bytes) {
  if (!match_bh_packed_shape(p, ko, a, kw, vw)) return false;
  uint32_t cap_log2 = std::max<uint32_t>(pow2_ceil_log2(p->entry_count < 4 ? 4 : p->entry_count), 2);
  // (a perfect-hash layout has one entry per possible key: every one may be in use, and tags at a load of 1 probe for ever --
  // twice the entries where LDS allows; an open-addressing table is sized at twice its groups already)
  if (p->query_kind == HDK_Q_PERFECT_HASH && 24ull * ((2ull << cap_log2) + 4) <= kBhPackedMaxBytes512) ++cap_log2;
  const uint64_t one = 24ull * ((1ull << cap_log2) + 4);
  // dense: one entry per value of the key column's statistics (+ the NULL key's) when that costs at most twice the tags' LDS
  // and fits a 256-thread block's share -- no tags to read, no probe (bh_dense_rows)
  if (a->dense_n && !hdk_sw(SW_NO_BH_DENSE)) {
    const uint32_t dlog2 = std::max<uint32_t>(pow2_ceil_log2(a->dense_n + 1), 2);
    // (to 52 KB with 256-thread blocks; to 100 KB -- 4 096 entries -- with one 512-thread block per CU: slower per row, but one
    // pass where the alternative is two)
    if (dlog2 <= cap_log2 + 1 && 24ull * ((1ull << dlog2) + 4) <= kBhPackedMaxBytes512) {
      a->dense = 1;
      cap_log2 = dlog2;
    }
  }
  if (a->key_form == 2 && !a->dense) return false;  // (modulo keys: only as a dense table over the remainder's range)
  // (tags: the LDS table must hold as many entries as the output table; a dense table only its keys' range -- the 16 384-entry
  // default guess of a small input does not keep a 64-key plan off the chip)
  if (!a->dense && one > kBhPackedMaxBytes512) return false;
  const uint64_t bytes = 24ull * ((1ull << cap_log2) + 4);
  uint32_t rep = 32;
  while (rep > 1 && (bytes + 16) * rep > kBhPackedReplicatedBytes) rep >>= 1;
  bh_packed_geometry(a, cap_log2, rep);
  *lds_bytes = a->rep_words * 4 * rep;
  *block = bytes > kBhPackedMaxBytes256 ? 512 : 256;
  return true;
}

template <int KW, int VW, int BLOCK>
static const void* bh_packed_kernel_of() {
#ifndef HDK_BH_PACKED_U
#define HDK_BH_PACKED_U 4  // 16-byte steps per lane and tile (A/B builds at 256 M rows: 2 -> bh3 0.86 ms, 4 -> 0.79, 8 -> 0.87)
#endif
#ifndef HDK_BH_GENERAL_U
#define HDK_BH_GENERAL_U 2  // the general (filtered) kernels: two steps per tile (measured at 256 M rows, filtered BH001 on 4- / 8-byte columns: four steps 1.31 / 1.55 ms, two 1.15 / 1.34, one 1.11 / 1.36)
#endif
  return reinterpret_cast<const void*>(hdk_scan_agg_bh_packed<KW, VW, HDK_BH_GENERAL_U, BLOCK>);
}
template <int KW, int VW>
static const void* bh_dense_kernel_of() {
  return reinterpret_cast<const void*>(hdk_scan_agg_bh_dense<KW, VW, HDK_BH_GENERAL_U>);
}
template <int KW, int VW>
static const void* bh_dense_kernel512_of() {
  return reinterpret_cast<const void*>(hdk_scan_agg_bh_dense<KW, VW, HDK_BH_GENERAL_U, 512>);
}
static const void* bh_dense_kernel(int kw, int vw, int block) {
  if (block == 512) {
    if (kw == 4) return vw == 0 ? bh_dense_kernel512_of<4, 0>() : (vw == 4 ? bh_dense_kernel512_of<4, 4>() : bh_dense_kernel512_of<4, 8>());
    return vw == 0 ? bh_dense_kernel512_of<8, 0>() : (vw == 4 ? bh_dense_kernel512_of<8, 4>() : bh_dense_kernel512_of<8, 8>());
  }
  if (kw == 4) return vw == 0 ? bh_dense_kernel_of<4, 0>() : (vw == 4 ? bh_dense_kernel_of<4, 4>() : bh_dense_kernel_of<4, 8>());
  return vw == 0 ? bh_dense_kernel_of<8, 0>() : (vw == 4 ? bh_dense_kernel_of<8, 4>() : bh_dense_kernel_of<8, 8>());
}
// unfiltered plans: the kernels that hold only the hot form of the steps (every column width)
static bool bh_plain(const BhPackedArgs& a, int kw, int vw) { return a.nquals == 0 && a.key_form != 2 && !hdk_sw(SW_NO_BH_PLAIN); }
template <int KW, int VW>
static const void* bh_plain_kernel_of(bool dense, int block) {
  if (dense) {
    return block == 512 ? reinterpret_cast<const void*>(hdk_scan_agg_bh_dense_plain<KW, VW, HDK_BH_PACKED_U, 512>)
                        : reinterpret_cast<const void*>(hdk_scan_agg_bh_dense_plain<KW, VW, HDK_BH_PACKED_U, 256>);
  }
  return block == 512 ? reinterpret_cast<const void*>(hdk_scan_agg_bh_packed_plain<KW, VW, HDK_BH_PACKED_U, 512>)
                      : reinterpret_cast<const void*>(hdk_scan_agg_bh_packed_plain<KW, VW, HDK_BH_PACKED_U, 256>);
}
static const void* bh_plain_kernel(bool dense, int kw, int vw, int block) {
  if (kw == 4) return vw == 0 ? bh_plain_kernel_of<4, 0>(dense, block) : (vw == 4 ? bh_plain_kernel_of<4, 4>(dense, block) : bh_plain_kernel_of<4, 8>(dense, block));
  return vw == 0 ? bh_plain_kernel_of<8, 0>(dense, block) : (vw == 4 ? bh_plain_kernel_of<8, 4>(dense, block) : bh_plain_kernel_of<8, 8>(dense, block));
}
template <int BLOCK>
static const void* bh_packed_kernel(int kw, int vw) {
  if (kw == 4) return vw == 0 ? bh_packed_kernel_of<4, 0, BLOCK>() : (vw == 4 ? bh_packed_kernel_of<4, 4, BLOCK>() : bh_packed_kernel_of<4, 8, BLOCK>());
  return vw == 0 ? bh_packed_kernel_of<8, 0, BLOCK>() : (vw == 4 ? bh_packed_kernel_of<8, 4, BLOCK>() : bh_packed_kernel_of<8, 8, BLOCK>());
}

// ---- tables beyond LDS (to 1 M entries): 256 bins by the key's hash, a bin's groups in the LDS of one block ----------------
constexpr uint32_t kBhBinsLog2 = 8;
constexpr int kBhScatterVR = 8;
struct BhPartLayout {
  size_t cursor_bytes, total;
  uint32_t lds_bytes;
};
static bool match_bh_partitioned(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhPackedArgs* a, BhPartLayout* l) {
  const bool off = hdk_sw(SW_NO_BH_PARTITIONS) != nullptr;  // (A/B measurements)
  const bool always = hdk_sw(SW_BH_PARTITIONS_ALWAYS) != nullptr;  // (tests: small inputs and tables)
  if (off || !ko || ko->total_rows == 0) return false;
  if (ko->flags & HDK_HIP_LAUNCH_INIT_OUTPUT) {
    // (fine: the table is initialised by the init kernel before the passes, launch_baseline)
  }
  int kw, vw;
  if (!match_bh_packed_shape(p, ko, a, &kw, &vw) || a->key_form == 2) return false;
  if (!always && (ko->total_rows < (4ull << 20) || p->entry_count > (1u << 20))) return false;
  if (ko->total_rows >= (1ull << 40)) return false;
  // a bin sees 1 / 256 of the groups: room for 2.5 x the bin's share of the table's entries (every entry in use and an
  // uneven hash still fit; what does not goes through the exact path), between 64 and 4096 tags
  const uint64_t share = (static_cast<uint64_t>(p->entry_count) + 255) / 256;
  uint32_t cap_log2 = pow2_ceil_log2(share * 5 / 2 + 1);
  if (cap_log2 < 6) cap_log2 = 6;
  if (cap_log2 > 12) cap_log2 = 12;
  bh_packed_geometry(a, cap_log2, 1);
  a->bins_log2 = kBhBinsLog2;
  l->lds_bytes = a->rep_words * 4;
  a->cap = ((ko->total_rows / (256 * kPbXcds)) * 5 / 4 + 2048 + 1) & ~1ull;
  l->cursor_bytes = static_cast<size_t>(256) * kPbXcds * kPbCursorStride * sizeof(uint32_t);
  l->total = l->cursor_bytes + static_cast<size_t>(256) * kPbXcds * a->cap * 8;
  return true;
}

// ---- dense keys beyond LDS: 256 bins by key range, 4-byte tuples (scan_bh_dense_part.h) ---------------------------------------
struct BhDensePartLayout {
  size_t cursor_bytes, total;
  uint32_t lds_bytes;
  int kw, vw;
};
static bool match_bh_dense_part(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhDensePartArgs* g, BhDensePartLayout* l) {
  const bool always = hdk_sw(SW_BH_PARTITIONS_ALWAYS) != nullptr;  // (tests: small inputs and tables)
  if (hdk_sw(SW_NO_BH_PARTITIONS) || hdk_sw(SW_NO_BH_DENSE_PARTITIONS) || !ko || ko->total_rows == 0) return false;
  memset(g, 0, sizeof(*g));
  BhPackedArgs& a = g->p;
  if (!match_bh_packed_shape(p, ko, &a, &l->kw, &l->vw) || a.key_form == 2) return false;
  if (!a.dense_n) return false;
  if (!always && ko->total_rows < (4ull << 20)) return false;
  if (ko->total_rows >= (1ull << 40)) return false;
  g->n_entries = a.dense_n + ((a.key_nullable && a.key_width == 4) ? 1u : 0u);
  g->width = (g->n_entries + 255) / 256;
  if (g->width < 2) g->width = 2;
  g->nbins = (g->n_entries + g->width - 1) / g->width;
  g->w = pow2_ceil_log2(g->width);
  if (g->w < 1) g->w = 1;
  if (g->w > 11) return false;  // (a bin's table: 2 048 entries of 24 bytes)
  magic_u32(g->width, &g->wmagic, &g->wshift);
  // the argument's code next to the offset inside the bin
  uint32_t vbits = 0;
  if (a.has_val) {
    const uint64_t codes = static_cast<uint64_t>(static_cast<int64_t>(a.val_max) - a.val_min) + 2;
    vbits = pow2_ceil_log2(codes);
  }
  if (g->w + vbits > 32) return false;
  // pass B's LDS: the bin's table, replicated while that stays below 48 KB
  uint32_t reps = 16;
  while (reps > 1 && (24ull * ((1ull << g->w) + 4) + 16) * reps > (48u << 10)) reps >>= 1;
  bh_packed_geometry(&a, g->w, reps);
  l->lds_bytes = a.rep_words * 4 * reps;
  g->cap4 = ((ko->total_rows / (static_cast<uint64_t>(g->nbins) * kPbXcds)) * 5 / 4 + 4096 + 3) & ~3ull;
  if (g->cap4 > 0xFFFFFFF0ull) return false;
  l->cursor_bytes = static_cast<size_t>(g->nbins) * kPbXcds * kPbCursorStride * sizeof(uint32_t);
  l->total = l->cursor_bytes + static_cast<size_t>(g->nbins) * kPbXcds * g->cap4 * 4;
  return true;
}

template <int KW>
static const void* bh_dscatter_kernel_kw(int vw) {
  return vw == 0 ? reinterpret_cast<const void*>(hdk_bh_dscatter<KW, 0>)
                 : (vw == 4 ? reinterpret_cast<const void*>(hdk_bh_dscatter<KW, 4>) : reinterpret_cast<const void*>(hdk_bh_dscatter<KW, 8>));
}

static int32_t launch_bh_dense_part(const hdk_hip_plan* d_plan, const KernParams& kp, BhDensePartArgs& g, const BhDensePartLayout& l,
                                    const hdk_hip_device_properties* props, hipStream_t s, bool* launched) {
  AsyncScratch scratch(s);
  if (hipMallocAsync(&scratch.p, l.total, s) != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return HDK_HIP_OK;  // no room for the tuples: the other strategies
  }
  int8_t* base = static_cast<int8_t*>(scratch.p);
  HDK_HIP_CHECK(hipMemsetAsync(base, 0, l.cursor_bytes, s));
  g.p.plan = d_plan;
  g.p.kp = kp;
  g.fill = reinterpret_cast<uint32_t*>(base);
  g.tuples4 = reinterpret_cast<uint32_t*>(base + l.cursor_bytes);
  const void* sk = l.kw == 4 ? bh_dscatter_kernel_kw<4>(l.vw) : bh_dscatter_kernel_kw<8>(l.vw);
  const unsigned g1 = scatter_grid(sk, kPbBlock, kBhDpScatterLds, props, 2);
  void* kargs[] = {&g};
  HDK_HIP_CHECK(hipLaunchKernel(sk, dim3(g1), dim3(kPbBlock), kargs, kBhDpScatterLds, s));
  const void* ak = reinterpret_cast<const void*>(hdk_bh_daggregate);
  if (l.lds_bytes > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(ak, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(l.lds_bytes)));
  }
  hipLaunchKernelGGL(hdk_bh_daggregate, dim3(g.nbins), dim3(kBhDpAggThreads), l.lds_bytes, s, g);
  HDK_HIP_CHECK(hipGetLastError());
  *launched = true;
  return HDK_HIP_OK;
}

const char* bh_packed_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko) {
  BhPackedArgs a;
  int kw, vw, block;
  uint32_t lds;
  if (match_bh_packed(p, ko, &a, &kw, &vw, &block, &lds)) {
    if (bh_plain(a, kw, vw)) return a.dense ? "hdk_scan_agg_bh_dense_plain,hdk_bh_fold_slabs" : "hdk_scan_agg_bh_packed_plain,hdk_bh_fold_slabs";
    return a.dense ? "hdk_scan_agg_bh_dense,hdk_bh_fold_slabs" : "hdk_scan_agg_bh_packed,hdk_bh_fold_slabs";
  }
  // several argument or key columns, and one-argument tables that fit LDS at 12 bytes an entry but not at 24: scan_bhm.hip
  if (const char* m = bhm_kernel_name(p, ko)) return m;
  BhDensePartArgs dg;
  BhDensePartLayout dl;
  if (match_bh_dense_part(p, ko, &dg, &dl)) return "hdk_bh_dscatter,hdk_bh_daggregate";
  BhPartLayout l;
  if (match_bh_partitioned(p, ko, &a, &l)) return "hdk_bh_scatter,hdk_bh_aggregate";
  return nullptr;
}

static int32_t launch_bh_partitioned(const hdk_hip_plan* d_plan, const KernParams& kp, BhPackedArgs& a, const BhPartLayout& l,
                                     const hdk_hip_device_properties* props, hipStream_t s, bool* launched) {
  AsyncScratch scratch(s);
  if (hipMallocAsync(&scratch.p, l.total, s) != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return HDK_HIP_OK;  // no room for the tuples: the other strategies
  }
  int8_t* base = static_cast<int8_t*>(scratch.p);
  HDK_HIP_CHECK(hipMemsetAsync(base, 0, l.cursor_bytes, s));
  a.plan = d_plan;
  a.kp = kp;
  a.fill = reinterpret_cast<uint32_t*>(base);
  a.tuples = reinterpret_cast<int64_t*>(base + l.cursor_bytes);
  const size_t lds_sc = PbStage<1, kBhScatterVR>::lds_bytes();
  const bool prog = a.nquals > 0 && a.q[0].nprog != 0;
  const void* sk = prog ? reinterpret_cast<const void*>(hdk_bh_scatter<kBhScatterVR, true>)
                        : reinterpret_cast<const void*>(hdk_bh_scatter<kBhScatterVR, false>);
  const unsigned g1 = scatter_grid(sk, kPbBlock, lds_sc, props, 2);
  void* sargs[] = {&a};
  HDK_HIP_CHECK(hipLaunchKernel(sk, dim3(g1), dim3(kPbBlock), sargs, lds_sc, s));
  const void* ak = reinterpret_cast<const void*>(hdk_bh_aggregate);
  if (l.lds_bytes > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(ak, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(l.lds_bytes)));
  }
  hipLaunchKernelGGL(hdk_bh_aggregate, dim3(256), dim3(kBhAggThreads), l.lds_bytes, s, a);
  HDK_HIP_CHECK(hipGetLastError());
  *launched = true;
  return HDK_HIP_OK;
}

int32_t launch_bh_packed(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                         const hdk_hip_device_properties* props, hipStream_t s, bool* launched) {
  *launched = false;
  BhPackedArgs a;
  int kw, vw, block;
  uint32_t lds;
  if (!match_bh_packed(plan, ko, &a, &kw, &vw, &block, &lds)) {
    int32_t st = launch_bhm(plan, d_plan, kp, ko, props, s, launched);
    if (st || *launched) return st;
    BhDensePartArgs dg;
    BhDensePartLayout dl;
    if (match_bh_dense_part(plan, ko, &dg, &dl)) return launch_bh_dense_part(d_plan, kp, dg, dl, props, s, launched);
    BhPartLayout l;
    if (match_bh_partitioned(plan, ko, &a, &l)) return launch_bh_partitioned(d_plan, kp, a, l, props, s, launched);
    return HDK_HIP_OK;
  }
  a.plan = d_plan;
  a.kp = kp;
  const void* k = bh_plain(a, kw, vw) ? bh_plain_kernel(a.dense != 0, kw, vw, block)
                                      : (a.dense ? bh_dense_kernel(kw, vw, block) : (block == 512 ? bh_packed_kernel<512>(kw, vw) : bh_packed_kernel<256>(kw, vw)));
  if (lds > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  }
  uint32_t grid = resident_grid(k, block, lds, props);
  const uint32_t cu = static_cast<uint32_t>(props->num_cu);
  const int per_cu_env = hdk_sw(SW_BH_BLOCKS_PER_CU) ? atoi(hdk_sw(SW_BH_BLOCKS_PER_CU)) : 0;  // (measurements)
  const uint32_t want = (per_cu_env > 0 ? static_cast<uint32_t>(per_cu_env) : (block == 512 ? 1u : 4u)) * cu;
  if (grid > want) grid = want;
  if (ko && ko->grid_dim_x) grid = ko->grid_dim_x;
  // Two-level fold: the blocks leave their tables in scratch slabs, a second kernel merges them in LDS and folds every group
  // into the output table `fold_groups` times -- folding from every scan block is (blocks x groups) contended memory-side
  // atomics: 0.3 ms for 10 groups, 3.6 ms for 1000 (profiles/r05_bh_fold.txt).  HDK_HIP_BH_DIRECT_FOLD=1: the one-level form.
  const bool direct_fold = hdk_sw(SW_BH_DIRECT_FOLD) != nullptr;
  AsyncScratch scratch(s);
  const size_t slab_bytes = static_cast<size_t>(6) * ((static_cast<size_t>(1) << a.cap_log2) + 4) * 4;
  const uint32_t cap = 1u << a.cap_log2;
  // bucket ranges of 8 buckets (measured at 1 024 slabs: ranges of 32 buckets left stage 2 with 8 blocks for 1 000 groups and
  // ONE for 100 -- 100 us either way; with these both stages take a trip or two per thread)
  const uint32_t slices = cap >= 64 ? cap / 32 : 1;
  // three levels when there are many slabs (hdk_bh_fold_slabs: fold_stage): about 256 stage-1 blocks, lists of twice a
  // slice's entries (probing may carry a key over a slice border)
  const bool staged = grid >= 64 && !hdk_sw(SW_BH_FOLD_GROUPS);
  const uint32_t groups1 = std::min<uint32_t>(std::max<uint32_t>(256u / slices, 2u), grid);
  const uint32_t list_cap = std::min<uint32_t>(cap, 2 * (cap / slices));
  const size_t list_bytes = staged ? static_cast<size_t>(slices) * groups1 * list_cap * 40 : 0;
  const size_t count_bytes = staged ? static_cast<size_t>(slices) * groups1 * 4 : 0;
  const size_t slabs_total = (slab_bytes * grid + 255) & ~static_cast<size_t>(255);
  if (!direct_fold && hipMallocAsync(&scratch.p, slabs_total + list_bytes + count_bytes, s) == hipSuccess) {
    a.slabs = static_cast<uint32_t*>(scratch.p);
    a.num_slabs = grid;
    a.fold_slices = slices;
    if (staged) {
      a.lists = reinterpret_cast<uint64_t*>(static_cast<int8_t*>(scratch.p) + slabs_total);
      a.list_counts = reinterpret_cast<uint32_t*>(static_cast<int8_t*>(scratch.p) + slabs_total + list_bytes);
      a.list_cap = list_cap;
    }
    // about 256 fold blocks: every (slice, group) folds its groups into the output table once -- `fold_groups` contended updates
    // per entry instead of one per scan block
    a.fold_groups = staged ? groups1 : 8;
    if (const char* e = hdk_sw(SW_BH_FOLD_GROUPS)) a.fold_groups = std::max(1, atoi(e));  // (measurements: the one-kernel fold)
    if (a.fold_groups > grid) a.fold_groups = grid;
  } else {
    (void)hipGetLastError();
    scratch.p = nullptr;
  }
  void* kargs[] = {&a};
  HDK_HIP_CHECK(hipLaunchKernel(k, dim3(grid), dim3(block), kargs, lds, s));
  if (a.slabs) {
    const size_t fold_lds = (static_cast<size_t>(9) << a.cap_log2) * 4;
    const void* fk = reinterpret_cast<const void*>(hdk_bh_fold_slabs);
    if (fold_lds > (48u << 10)) {
      HDK_HIP_CHECK(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(fold_lds)));
    }
    if (a.lists) {
      a.fold_stage = 1;
      hipLaunchKernelGGL(hdk_bh_fold_slabs, dim3(a.fold_slices * a.fold_groups), dim3(kBhFoldBlock), fold_lds, s, a);
      a.fold_stage = 2;
      hipLaunchKernelGGL(hdk_bh_fold_slabs, dim3(a.fold_slices), dim3(kBhFoldBlock), fold_lds, s, a);
    } else {
      hipLaunchKernelGGL(hdk_bh_fold_slabs, dim3(a.fold_slices * a.fold_groups), dim3(kBhFoldBlock), fold_lds, s, a);
    }
  }
  HDK_HIP_CHECK(hipGetLastError());
  *launched = true;
  return HDK_HIP_OK;
}

}  // namespace hdk
