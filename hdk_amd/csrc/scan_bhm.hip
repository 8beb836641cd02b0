// scan_bhm.hip -- matcher, launcher and instantiations of the multi-argument / multi-key on-chip group-by (scan_bhm.h).
#include <string.h>

#include <algorithm>

#include "host_match.h"
#include "scan_bhm.h"
#include "scan_bhm_host.h"

namespace hdk {

constexpr uint32_t kBhmMaxLdsBytes = 144u << 10;        // one 1024-thread block per CU
constexpr uint32_t kBhmSmallLdsBytes = 40u << 10;       // up to here: 256-thread blocks, four on a CU
constexpr uint32_t kBhmReplicatedBytes = 32u << 10;     // replicas while the table is tiny
constexpr int64_t kBhmMaxAbsVal = (1ll << 19) - 1;      // |argument| below this keeps a block's row budget at 2^20 or more
constexpr int kBhmU = 2;                                // 16-byte steps per lane, column and tile

static bool bhm_off() { return hdk_sw(SW_NO_BHM) != nullptr || hdk_sw(SW_NO_BH_LDS) != nullptr; }

static uint32_t bits_for(uint64_t codes) {  // bits that hold the values 0 .. codes - 1
  uint32_t b = 1;
  while ((1ull << b) < codes) ++b;
  return b;
}

struct BhmGeom {
  int nk, ns, block;
  uint32_t grid_per_cu;
  bool perfect;
  int64_t key_lo;
  uint32_t null_entry;
  uint32_t nword_mask;
  int32_t wop[kMaxWordsPerEntry];
};

static bool match_bhm(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhmArgs* a, BhmGeom* g) {
  const bool perfect = p->query_kind == HDK_Q_PERFECT_HASH;
  if (bhm_off() || (!perfect && p->query_kind != HDK_Q_BASELINE_HASH)) return false;
  if (!ko || ko->total_rows == 0 || ko->total_rows >= (1ull << 40)) return false;
  if (ko->flags & (HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS | HDK_HIP_LAUNCH_FORCE_PARTITIONED)) return false;
  if (launch_forces_generic(ko)) return false;
  if (p->num_joins || p->num_quals || p->num_filter_ops) return false;
  if (p->key_count < 1 || p->key_count > kBhmMaxKeys) return false;
  memset(a, 0, sizeof(*a));
  memset(g, 0, sizeof(*g));
  g->perfect = perfect;
  g->null_entry = 0xFFFFFFFFu;
  // ---- keys: plain 4-byte integer columns with statistics --------------------------------------------------------------
  uint64_t entries = 1;
  for (int k = 0; k < p->key_count; ++k) {
    const hdk_hip_expr& ke = p->keys[k];
    if (ke.leaf0.kind != HDK_LEAF_COL) return false;
    const hdk_hip_col& kc = p->cols[ke.leaf0.col];
    if (kc.table != 0 || kc.kind != HDK_COL_INT || kc.width != 4) return false;
    BhmKey& key = a->key[k];
    key.buf_idx = kc.buf_idx;
    key.null32 = static_cast<int32_t>(ke.leaf0.null_val);
    if (perfect) {
      // the plan's own index: (key - min) * stride, a NULL under its translated value (perfect_key_hash)
      if (ke.nsteps != 0 || p->key_bucket[k] > 1) return false;
      const bool translate = p->key_has_nulls[k] && ke.nullable;
      const int64_t card = p->key_count == 1 ? static_cast<int64_t>(p->entry_count) : p->key_card[k];
      if (card < 1 || card > (1 << 22) || p->key_min[k] < INT32_MIN / 2 || p->key_min[k] > INT32_MAX / 2) return false;
      key.min = static_cast<int32_t>(p->key_min[k]);
      key.n = static_cast<uint32_t>(card - (translate ? 1 : 0));
      key.nullable = translate;
      const int64_t nd = p->key_null_translated[k] - p->key_min[k];
      if (translate && (nd < 0 || nd >= card)) return false;
      key.null_d = translate ? static_cast<uint32_t>(nd) : 0u;
      key.stride = static_cast<uint32_t>(entries);
      entries *= static_cast<uint64_t>(card);
    } else {
      // open addressing: one key, as it is or cast to double; dense over the key column's statistics
      if (p->key_count != 1 || (p->key_width != 4 && p->key_width != 8)) return false;
      if (ke.nsteps == 1) {
        const hdk_hip_step& sp = ke.steps[0];
        if (sp.op != HDK_OP_CAST_INT_TO_FP || p->key_width != 8) return false;
        a->key_form = 1;
        a->key_null_word = sp.null_out;
      } else if (ke.nsteps != 0) {
        return false;
      } else {
        a->key_null_word = ke.leaf0.null_val;
      }
      if (!kc.has_stats || kc.max_val < kc.min_val || kc.min_val < INT32_MIN / 2 || kc.max_val > INT32_MAX / 2) return false;
      const uint64_t n = static_cast<uint64_t>(kc.max_val - kc.min_val) + 1;
      // (a dense table only pays while it is not much sparser than the groups the output table was sized for)
      if (n > (1u << 20) || n > 4ull * p->entry_count + 64) return false;
      key.min = static_cast<int32_t>(kc.min_val);
      key.n = static_cast<uint32_t>(n);
      key.nullable = ke.leaf0.nullable != 0;
      key.null_d = static_cast<uint32_t>(n);
      key.stride = 1;
      entries = n + (key.nullable ? 1 : 0);
      g->key_lo = kc.min_val;
      g->null_entry = key.nullable ? static_cast<uint32_t>(n) : 0xFFFFFFFFu;
    }
    if (entries > (1u << 22)) return false;
  }
  if (perfect && entries != p->entry_count) return false;
  a->nkeys = p->key_count;
  a->entries = static_cast<uint32_t>(entries);
  // ---- targets ------------------------------------------------------------------------------------------------------------
  WordLayout wl;
  make_word_layout(p, &wl);
  a->wpe = wl.wpe;
  for (int w = 0; w < wl.wpe; ++w) g->wop[w] = wl.wop[w];
  struct DerInfo {
    int64_t rmin, rmax;
    bool want_packed, want_max, want_min;
  } info[kBhmMaxSrc][kBhmMaxDer];
  memset(info, 0, sizeof(info));
  int word_src[kMaxWordsPerEntry], word_der[kMaxWordsPerEntry], word_kind[kMaxWordsPerEntry];
  for (int w = 0; w < kMaxWordsPerEntry; ++w) {
    word_src[w] = word_der[w] = -1;
    word_kind[w] = BMW_ROWS;
  }
  int64_t amax = 1;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_SINGLE_VALUE) return false;
    if (tg.agg == HDK_AGG_ID) {
      if (tg.key_idx < 0 || tg.key_idx >= p->key_count || (tg.slot_width != 0 && tg.slot_width != 4 && tg.slot_width != 8)) return false;
      continue;
    }
    if (tg.slot_width != 4 && tg.slot_width != 8) return false;
    if (tg.agg == HDK_AGG_AVG && tg.slot2_width != 4 && tg.slot2_width != 8) return false;
    if (!tg.has_arg) {
      if (tg.agg != HDK_AGG_COUNT) return false;
      continue;
    }
    const hdk_hip_expr& e = tg.arg;
    if (e.nsteps > 1 || e.leaf0.kind != HDK_LEAF_COL || tg.arg_is_fp || e.vclass != HDK_VC_INT) return false;
    const hdk_hip_col& col = p->cols[e.leaf0.col];
    if (col.table != 0 || col.kind != HDK_COL_INT || col.width != 4 || !col.has_stats || col.max_val < col.min_val) return false;
    if (col.min_val < -kBhmMaxAbsVal || col.max_val > kBhmMaxAbsVal) return false;
    int32_t op = 0;
    int64_t lit = 0, rmin = col.min_val, rmax = col.max_val;
    if (e.nsteps == 1) {
      const hdk_hip_step& sp = e.steps[0];
      if (sp.out_class != HDK_VC_INT || sp.rhs.kind != HDK_LEAF_INT) return false;
      if (sp.op != HDK_OP_ADD && sp.op != HDK_OP_SUB && sp.op != HDK_OP_MUL) return false;
      if (sp.rhs.ival < -kBhmMaxAbsVal || sp.rhs.ival > kBhmMaxAbsVal) return false;
      // a NULL operand gives the step's NULL, which must be what the target skips (as match_fast)
      if (tg.skip_null && (e.null_val != sp.null_out || tg.null_val != sp.null_out)) return false;
      if (!tg.skip_null && e.leaf0.nullable) return false;
      op = sp.op;
      lit = sp.rhs.ival;
      const int64_t c0 = op == HDK_OP_ADD ? rmin + lit : (op == HDK_OP_SUB ? rmin - lit : rmin * lit);
      const int64_t c1 = op == HDK_OP_ADD ? rmax + lit : (op == HDK_OP_SUB ? rmax - lit : rmax * lit);
      rmin = std::min(c0, c1);
      rmax = std::max(c0, c1);
      // (the result's range fits the operation's type: the checked arithmetic of QE/ArithmeticIR.cpp:277-520 cannot fire)
      const int cw = sp.check_width ? sp.check_width : 8;
      const int64_t tmax = cw >= 8 ? INT64_MAX : (1ll << (8 * cw - 1)) - 1;
      if (rmin < -tmax || rmax > tmax) return false;
    } else if (!tg.skip_null && e.leaf0.nullable && col.has_nulls) {
      return false;  // (a NULL aggregated as a value: the sentinel is far outside the packed fields)
    }
    if (rmin < -kBhmMaxAbsVal || rmax > kBhmMaxAbsVal) return false;
    // the column's NULL: skipped when the statistics announce NULLs; else it would be outside them (the flag)
    const int nullable = (tg.skip_null && e.leaf0.nullable && col.has_nulls) ? 1 : 0;
    if (!tg.skip_null && col.has_nulls && e.leaf0.nullable) return false;
    int s = -1;
    for (int i = 0; i < a->nsrc; ++i) {
      if (a->src[i].buf_idx == col.buf_idx) s = i;
    }
    if (s < 0) {
      if (a->nsrc == kBhmMaxSrc) return false;
      s = a->nsrc++;
      BhmSrc& src = a->src[s];
      src.buf_idx = col.buf_idx;
      src.nullable = nullable;
      src.null32 = static_cast<int32_t>(e.leaf0.null_val);
      src.raw_min = static_cast<int32_t>(col.min_val);
      src.raw_span = static_cast<uint32_t>(col.max_val - col.min_val);
    } else if (a->src[s].nullable != nullable) {
      return false;
    }
    BhmSrc& src = a->src[s];
    int d = -1;
    for (int i = 0; i < src.nder; ++i) {
      if (src.der[i].op == op && src.der[i].lit == static_cast<int32_t>(lit)) d = i;
    }
    if (d < 0) {
      if (src.nder == kBhmMaxDer) return false;
      d = src.nder++;
      src.der[d].op = op;
      src.der[d].lit = static_cast<int32_t>(lit);
      src.der[d].packed = src.der[d].mx_word = src.der[d].mn_word = -1;
      info[s][d].rmin = rmin;
      info[s][d].rmax = rmax;
    }
    amax = std::max<int64_t>(amax, std::max<int64_t>(-rmin, rmax));
    if (tg.agg == HDK_AGG_SUM || tg.agg == HDK_AGG_AVG || tg.agg == HDK_AGG_COUNT) info[s][d].want_packed = true;
    if (tg.agg == HDK_AGG_MAX) info[s][d].want_max = true;
    if (tg.agg == HDK_AGG_MIN) info[s][d].want_min = true;
    if (wl.vword[t] >= 0) {
      word_src[wl.vword[t]] = s;
      word_der[wl.vword[t]] = d;
      word_kind[wl.vword[t]] = tg.agg == HDK_AGG_MIN ? BMW_MIN : (tg.agg == HDK_AGG_MAX ? BMW_MAX : BMW_SUM);
    }
    if (wl.nword[t] >= 0) {
      word_src[wl.nword[t]] = s;
      word_der[wl.nword[t]] = d;
      word_kind[wl.nword[t]] = BMW_NN;
      g->nword_mask |= 1u << wl.nword[t];
    }
  }
  if (a->nsrc == 0) return false;  // (COUNT(*) alone: the keys kernel / the one-argument packed kernels)
  if (a->nkeys == 1 && a->nsrc == 1 && a->src[0].nder == 1 && a->src[0].der[0].op == 0) return false;  // scan_bh_packed.h's own shape
  // ---- LDS words: packed [rows : sum] words, MIN / MAX fields -------------------------------------------------------------
  a->rows_packed = -1;
  uint32_t bitpos[kBhmMaxMm] = {0, 0};
  auto place_field = [&](uint32_t codebits, int32_t* word, uint32_t* shift, uint32_t* mask) -> bool {
    const uint32_t need = codebits + 1;  // + the guard bit
    for (int w = 0; w < kBhmMaxMm; ++w) {
      if (bitpos[w] + need <= 64 && a->nfields[w] < kBhmMaxFields) {
        *word = w;
        *shift = bitpos[w];
        *mask = static_cast<uint32_t>((1ull << codebits) - 1);
        a->fshift[w][a->nfields[w]] = bitpos[w];
        a->fmask[w][a->nfields[w]] = *mask;
        a->nfields[w]++;
        a->guards[w] |= 1ull << (bitpos[w] + codebits);
        bitpos[w] += need;
        a->nmm = std::max(a->nmm, w + 1);
        return true;
      }
    }
    return false;
  };
  for (int s = 0; s < a->nsrc; ++s) {
    for (int d = 0; d < a->src[s].nder; ++d) {
      BhmDer& der = a->src[s].der[d];
      const DerInfo& di = info[s][d];
      if (di.want_packed) {
        der.packed = a->npacked++;
        if (!a->src[s].nullable && a->rows_packed < 0) a->rows_packed = der.packed;
      }
      const uint32_t codebits = bits_for(static_cast<uint64_t>(di.rmax - di.rmin) + 2);
      if (codebits > 31) return false;
      if (di.want_max) {
        der.mx_bias = static_cast<int32_t>(di.rmin);
        if (!place_field(codebits, &der.mx_word, &der.mx_shift, &der.mx_mask)) return false;
      }
      if (di.want_min) {
        der.mn_bias = static_cast<int32_t>(di.rmax);
        if (!place_field(codebits, &der.mn_word, &der.mn_shift, &der.mn_mask)) return false;
      }
    }
  }
  // the slab words
  for (int w = 0; w < wl.wpe; ++w) {
    BhmSlabWord& sw = a->sw[w];
    sw.kind = word_kind[w];
    sw.packed = -1;
    sw.mm_word = -1;
    if (w == 0 || word_src[w] < 0) {
      sw.kind = BMW_ROWS;
      sw.packed = a->rows_packed;
      continue;
    }
    const BhmSrc& src = a->src[word_src[w]];
    const BhmDer& der = src.der[word_der[w]];
    if (sw.kind == BMW_SUM) {
      sw.packed = der.packed;
    } else if (sw.kind == BMW_NN) {
      if (!src.nullable) {  // the argument is never NULL: its non-NULL count is the row count
        sw.kind = BMW_ROWS;
        sw.packed = a->rows_packed;
      } else if (der.packed >= 0) {
        sw.packed = der.packed;
      } else {  // MIN / MAX only: "some non-NULL row" is all its count says
        const bool mx = der.mx_word >= 0;
        sw.mm_word = mx ? der.mx_word : der.mn_word;
        sw.shift = mx ? der.mx_shift : der.mn_shift;
        sw.mask = mx ? der.mx_mask : der.mn_mask;
      }
    } else if (sw.kind == BMW_MAX) {
      sw.mm_word = der.mx_word;
      sw.shift = der.mx_shift;
      sw.mask = der.mx_mask;
      sw.bias = der.mx_bias;
    } else {
      sw.mm_word = der.mn_word;
      sw.shift = der.mn_shift;
      sw.mask = der.mn_mask;
      sw.bias = der.mn_bias;
    }
  }
  // ---- geometry -----------------------------------------------------------------------------------------------------------
  a->e1 = (a->entries + 1 + 1) & ~1u;
  const uint32_t per_entry = 8u * static_cast<uint32_t>(a->npacked) + 8u * static_cast<uint32_t>(a->nmm) + (a->rows_packed < 0 ? 4u : 0u);
  const uint64_t one = static_cast<uint64_t>(a->e1) * per_entry;
  if (one > kBhmMaxLdsBytes) return false;
  a->off_mm = 8u * static_cast<uint32_t>(a->npacked) * a->e1;
  a->off_rows = a->off_mm + 8u * static_cast<uint32_t>(a->nmm) * a->e1;
  uint32_t rep_bytes = (static_cast<uint32_t>(one) + 15u) & ~15u;
  uint32_t rep = 16;
  while (rep > 1 && static_cast<uint64_t>(rep_bytes + 16) * rep > kBhmReplicatedBytes) rep >>= 1;
  if (rep > 1) {
    while (rep_bytes % 256 != 16) rep_bytes += 16;  // replica r starts 4 r banks further on
  }
  a->rep = rep;
  a->rep_bytes = rep_bytes;
  a->lds_bytes = rep_bytes * rep;
  g->nk = a->nkeys;
  g->ns = a->nsrc;
  g->block = a->lds_bytes > kBhmSmallLdsBytes ? 1024 : 256;
  g->grid_per_cu = g->block == 1024 ? 1 : 4;
  // rows a block may put into one entry: rows < 2^23, |sum| < 2^39
  const int64_t by_sum = ((1ll << 39) - 1) / amax;
  a->max_rows_per_block = static_cast<uint32_t>(std::min<int64_t>(by_sum, 1ll << 23));
  return true;
}

template <int NK, int NS>
static const void* bhm_kernel_ns(int block) {
  return block == 1024 ? reinterpret_cast<const void*>(hdk_scan_agg_bhm<NK, NS, 1024, kBhmU>)
                       : reinterpret_cast<const void*>(hdk_scan_agg_bhm<NK, NS, 256, kBhmU>);
}
template <int NK>
static const void* bhm_kernel_nk(int ns, int block) {
  return ns == 1 ? bhm_kernel_ns<NK, 1>(block) : (ns == 2 ? bhm_kernel_ns<NK, 2>(block) : bhm_kernel_ns<NK, 3>(block));
}
static const void* bhm_kernel(const BhmGeom& g) {
  return g.nk == 1 ? bhm_kernel_nk<1>(g.ns, g.block) : (g.nk == 2 ? bhm_kernel_nk<2>(g.ns, g.block) : bhm_kernel_nk<3>(g.ns, g.block));
}

const char* bhm_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko) {
  BhmArgs a;
  BhmGeom g;
  if (!match_bhm(p, ko, &a, &g)) return nullptr;
  return g.perfect ? "hdk_scan_agg_bhm,hdk_finalize" : "hdk_scan_agg_bhm,hdk_bhm_fold";
}

int32_t launch_bhm(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                   const hdk_hip_device_properties* props, hipStream_t s, bool* launched) {
  *launched = false;
  BhmArgs a;
  BhmGeom g;
  if (!match_bhm(plan, ko, &a, &g)) return HDK_HIP_OK;
  const void* k = bhm_kernel(g);
  if (a.lds_bytes > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(a.lds_bytes)));
  }
  const uint32_t cu = static_cast<uint32_t>(props->num_cu);
  uint32_t grid = std::min(resident_grid(k, g.block, a.lds_bytes, props), g.grid_per_cu * cu);
  if (const char* e = hdk_sw(SW_BHM_BLOCKS_PER_CU)) grid = std::max(1, atoi(e)) * cu;  // (measurements)
  // a block's rows must stay inside what the packed fields were sized for (twice the even share: tiles are dealt round robin)
  constexpr uint64_t kTileRows256 = 256ull * 4 * kBhmU;
  const uint64_t tile_rows = kTileRows256 * (g.block / 256);
  const uint64_t budget = a.max_rows_per_block > 4 * tile_rows ? a.max_rows_per_block - 4 * tile_rows : tile_rows;
  const uint64_t need = (2 * ko->total_rows + budget - 1) / budget;
  if (need > grid) grid = static_cast<uint32_t>(std::min<uint64_t>(need, 1u << 16));
  if (ko->grid_dim_x) grid = ko->grid_dim_x;
  const size_t slab_bytes = static_cast<size_t>(grid) * a.entries * a.wpe * 8;
  AsyncScratch scratch(s);
  if (hipMallocAsync(&scratch.p, 256 + slab_bytes, s) != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return HDK_HIP_OK;  // no room for the slabs: the other strategies
  }
  HDK_HIP_CHECK(hipMemsetAsync(scratch.p, 0, 256, s));
  a.plan = d_plan;
  a.kp = kp;
  a.flag = static_cast<uint32_t*>(scratch.p);
  a.slabs = reinterpret_cast<int64_t*>(static_cast<int8_t*>(scratch.p) + 256);
  void* kargs[] = {&a};
  HDK_HIP_CHECK(hipLaunchKernel(k, dim3(grid), dim3(g.block), kargs, a.lds_bytes, s));
  int32_t st;
  if (g.perfect) {
    st = launch_finalize_slabs(d_plan, a.slabs, kp.groupby_buf, grid, plan->entry_count, a.flag, s);
    if (st) return st;
  } else {
    BhmFoldArgs f;
    memset(&f, 0, sizeof(f));
    f.plan = d_plan;
    f.kp = kp;
    f.slabs = a.slabs;
    f.flag = a.flag;
    f.num_slabs = grid;
    f.entries = a.entries;
    f.out_entry_count = plan->entry_count;
    f.wpe = a.wpe;
    for (int w = 0; w < a.wpe; ++w) f.wop[w] = g.wop[w];
    f.nword_mask = g.nword_mask;
    f.key_form = a.key_form;
    f.key_lo = g.key_lo;
    f.null_entry = g.null_entry;
    f.key_null_word = a.key_null_word;
    hipLaunchKernelGGL(hdk_bhm_fold<0>, dim3((a.entries + 3) / 4), dim3(256), 0, s, f);
    HDK_HIP_CHECK(hipGetLastError());
  }
  // armed: runs only when the flag says the statistics did not hold (the folds skipped then)
  st = launch_scan_global_armed(plan, d_plan, kp, ko, props, s, a.flag);
  if (st) return st;
  *launched = true;
  return HDK_HIP_OK;
}

}  // namespace hdk
