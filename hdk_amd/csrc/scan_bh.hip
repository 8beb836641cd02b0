// scan_bh.hip -- GroupByBaselineHash plans whose table is SMALL: the open-addressing group-by kept in LDS (scan_bh.h).
// Reference being replaced: get_group_value + agg_*_shared per row on the final table (QE/GroupByRuntime.cpp:31-55,
// QE/cuda_mapd_rt.cu:167-203,424-478).  Two kernels:
//   hdk_scan_agg_bh_vec[_join]   the batched plan interpreter (vec_eval.h) with the LDS table of scan_bh.h behind it: every
//                                plan the interpreter takes -- expression keys (cast(x as double), dval % 64), filters,
//                                one-to-one joins
//   hdk_scan_agg_bh_direct       the streaming form (scan_bh_fast.h): one plain key column (as it is or cast to double),
//                                one plain argument column, the shape of the reference's BH001-005 benchmark queries
#include <string.h>

#include "host_match.h"
#define HDK_VEC_BODY_ONLY
#include "scan_agg_vec.h"
#include "scan_bh.h"
#include "scan_bh_fast.h"
#include "scan_bh_host.h"

namespace hdk {

extern "C" __global__ __launch_bounds__(kVecBlock, 3) void hdk_scan_agg_bh_vec(VecArgs a) {
  scan_agg_vec_body<false, false, false, true>(a);
}
extern "C" __global__ __launch_bounds__(kVecBlock, 2) void hdk_scan_agg_bh_vec_join(VecArgs a) {
  scan_agg_vec_body<true, false, false, true>(a);
}

extern "C" __global__ __launch_bounds__(kBhFoldWordsBlock) void hdk_bh_fold_words(BhFoldArgs a) { bh_fold_words_body(a); }

// Two-level fold of a scan's per-block tables (scan_bh.h): scratch slabs + the fold kernel's geometry; false: the blocks fold
// into the output table themselves (tiny grids, no scratch, HDK_HIP_BH_DIRECT_FOLD=1)
static bool bh_two_level_fold(AsyncScratch& scratch, const BhGeom& g, const BhLdsLayout& ll, uint32_t grid, hipStream_t s, BhFoldArgs* fa) {
  const bool direct_fold = hdk_sw(SW_BH_DIRECT_FOLD) != nullptr;  // (A/B measurements)
  const uint32_t cap = 1u << g.cap_log2;
  const size_t slab_words = static_cast<size_t>(cap) * (static_cast<uint32_t>(ll.nlw) + 1);
  // worth it once the one-level fold would mean more than a few thousand contended group folds
  if (direct_fold || static_cast<uint64_t>(grid) * cap < 4096) return false;
  if (hipMallocAsync(&scratch.p, slab_words * 8 * grid, s) != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return false;
  }
  memset(fa, 0, sizeof(*fa));
  fa->g = g;
  fa->g.rep = 1;
  fa->ll = ll;
  fa->slabs = static_cast<const int64_t*>(scratch.p);
  fa->num_slabs = grid;
  fa->fold_slices = cap >= 256 ? cap / 128 : 1;
  fa->fold_groups = fa->fold_slices >= 32 ? 2 : (fa->fold_slices >= 8 ? 4 : 8);
  if (fa->fold_groups > grid) fa->fold_groups = grid;
  return true;
}
static int32_t launch_bh_fold_words(BhFoldArgs& fa, const hdk_hip_plan* d_plan, const KernParams& kp, hipStream_t s) {
  fa.plan = d_plan;
  fa.kp = kp;
  const size_t lds = (static_cast<size_t>(fa.ll.nlw) + 1) * 8 << fa.g.cap_log2;
  if (lds > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(hdk_bh_fold_words), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
  }
  hipLaunchKernelGGL(hdk_bh_fold_words, dim3(fa.fold_slices * fa.fold_groups), dim3(kBhFoldWordsBlock), lds, s, fa);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// ---- dense internal tables folded into the open-addressing table (scan_bh_host.h: BhDenseFold) -------------------------------
struct BhDenseFoldArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  BhDenseFold f;
};
// one wave per internal entry: lanes fold the slabs' words (a fixed shuffle tree: deterministic), lane 0 finds or claims the
// group's entry on the reference's probe sequence and adds the partial in
extern "C" __global__ __launch_bounds__(256) void hdk_bh_fold_dense(BhDenseFoldArgs a) {
  __shared__ WordLayout wl;
  __shared__ uint64_t s_col_off[2 * HDK_HIP_MAX_TARGETS];
  __shared__ int64_t s_words[4][kMaxWordsPerEntry];
  if (threadIdx.x == 0) {
    make_word_layout(a.plan, &wl);
  }
  if (threadIdx.x < 2 * HDK_HIP_MAX_TARGETS) {
    s_col_off[threadIdx.x] = a.plan->output_columnar ? columnar_slot_off(a.plan, a.f.out_entry_count, threadIdx.x) : 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t entry = blockIdx.x * 4 + wave;
  if (entry >= a.f.entries) {
    return;
  }
  const int wpe = a.f.wpe;
  const size_t ew = static_cast<size_t>(a.f.entries) * wpe;
  int64_t* words = s_words[wave];
  for (int w = 0; w < wpe; ++w) {
    const int32_t op = a.f.wop[w];
    int64_t acc = word_identity(op);
    for (uint32_t b = lane; b < a.f.num_slabs; b += 64) {
      acc = word_combine(op, acc, a.f.slabs[b * ew + static_cast<size_t>(entry) * wpe + w]);
    }
    for (int d = 32; d > 0; d >>= 1) {
      const int lo = __shfl_down(static_cast<int>(static_cast<uint32_t>(acc)), d, 64);
      const int hi = __shfl_down(static_cast<int>(static_cast<uint64_t>(acc) >> 32), d, 64);
      acc = word_combine(op, acc, static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo)));
    }
    if (lane == 0) {
      words[w] = acc;
    }
  }
  if (lane != 0 || words[0] == 0) {
    return;  // (no row of this group: nothing to claim)
  }
  const int64_t rows = words[0];
  int32_t err = 0;
  const int64_t key = entry == a.f.null_entry ? a.f.null_key : a.f.key_lo + static_cast<int64_t>(entry);
  bh_fold_group_fn(a.plan, table_shape(a.plan), wl, a.kp.groupby_buf[0], a.f.out_entry_count, s_col_off, key,
                   [&](int w) -> int64_t { return ((a.f.nword_mask >> w) & 1u) ? rows - words[w] : words[w]; }, err);
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

int32_t launch_bh_fold_dense(const BhDenseFold& fold, const hdk_hip_plan* d_plan, const KernParams& kp, hipStream_t s) {
  BhDenseFoldArgs a;
  a.plan = d_plan;
  a.kp = kp;
  a.f = fold;
  hipLaunchKernelGGL(hdk_bh_fold_dense, dim3((fold.entries + 3) / 4), dim3(256), 0, s, a);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// LDS for the table: up to 32 KiB keeps three or four interpreter blocks on a CU; an unreplicated table may take 64 KiB
// (two blocks) -- beyond that the plan goes to the partitioned or the global-atomics kernels
constexpr uint32_t kBhLdsReplicatedBytes = 32u << 10;
constexpr uint32_t kBhLdsMaxBytes = 64u << 10;

static bool bh_switch_off() {
  const bool off = hdk_sw(SW_NO_BH_LDS) != nullptr;  // (A/B measurements)
  return off;
}

bool match_bh_lds(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhGeom* g) {
  if (p->query_kind != HDK_Q_BASELINE_HASH || bh_switch_off()) return false;
  if (ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS | HDK_HIP_LAUNCH_FORCE_PARTITIONED))) return false;
  if (launch_forces_generic(ko)) return false;
  // one 64-bit key word: one key of either width, or two 4-byte keys of a row-wise table
  if (!(p->key_count == 1 || (p->key_count == 2 && p->key_width == 4 && !p->output_columnar))) return false;
  if (p->key_width != 4 && p->key_width != 8) return false;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_SINGLE_VALUE) return false;  // lives on the final table only
    if (tg.agg == HDK_AGG_ID) {
      if (tg.slot_width != 0 && tg.slot_width != 4 && tg.slot_width != 8) return false;
      continue;
    }
    if (tg.slot_width != 4 && tg.slot_width != 8) return false;
    if (tg.agg == HDK_AGG_AVG && tg.slot2_width != 4 && tg.slot2_width != 8) return false;
  }
  // joins: what the batched interpreter's one-to-one probes cover
  if (needs_join_loops(p) || plan_is_single_matching_set_join(p) || plan_has_keyed_join(p)) return false;
  WordLayout wl;
  make_word_layout(p, &wl);
  const uint32_t W = static_cast<uint32_t>(wl.wpe) + 1;
  const uint32_t cap_log2 = pow2_ceil_log2(p->entry_count < 2 ? 2 : p->entry_count);
  const uint64_t one = (static_cast<uint64_t>(W) << cap_log2) * 8;
  if (one > kBhLdsMaxBytes) return false;
  uint32_t rep = 32;
  while (rep > 1 && one * rep > kBhLdsReplicatedBytes) rep >>= 1;
  g->out_entry_count = p->entry_count;
  g->cap_log2 = cap_log2;
  g->rep = rep;
  g->lds_bytes = static_cast<uint32_t>(one * rep);
  return true;
}

// ---- the streaming form (scan_bh_fast.h) ---------------------------------------------------------------------------------
// an unreplicated table beyond 64 KiB takes one 512-thread block per CU, up to this much LDS
constexpr uint32_t kBhFastLdsMaxBytes = 144u << 10;

static bool match_bh_fast(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhFastArgs* fa, int* kw_out, int* vw_out, int* block_out) {
  const bool off = hdk_sw(SW_NO_BH_DIRECT) != nullptr;  // (A/B measurements)
  if (off || p->query_kind != HDK_Q_BASELINE_HASH || bh_switch_off()) return false;
  if (ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS | HDK_HIP_LAUNCH_FORCE_PARTITIONED))) return false;
  if (launch_forces_generic(ko)) return false;
  if (p->num_joins || p->key_count != 1 || (p->key_width != 4 && p->key_width != 8)) return false;
  memset(fa, 0, sizeof(*fa));
  if (!match_plain_quals(p, fa->q, true)) return false;
  fa->nquals = p->num_quals;
  // the key: a plain integer column, or cast(such a column AS double)
  const hdk_hip_expr& ke = p->keys[0];
  if (ke.leaf0.kind != HDK_LEAF_COL) return false;
  const hdk_hip_col& kc = p->cols[ke.leaf0.col];
  if (kc.table != 0 || kc.kind != HDK_COL_INT || (kc.width != 4 && kc.width != 8)) return false;
  if (ke.nsteps == 1) {
    const hdk_hip_step& sp = ke.steps[0];
    if (sp.op != HDK_OP_CAST_INT_TO_FP || p->key_width != 8) return false;
    fa->key_form = 1;
    fa->key_nullable = ke.leaf0.nullable;
    fa->key_null = ke.leaf0.null_val;
    fa->key_null_out = sp.null_out;
  } else if (ke.nsteps != 0) {
    return false;
  }
  fa->key_buf_idx = kc.buf_idx;
  // the argument column and the distinct updates
  WordLayout wl;
  make_word_layout(p, &wl);
  int vw = 0;
  bool have_val = false, want_sum = false, want_min = false, want_max = false, want_nulls = false;
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
    const bool fp = col.kind == HDK_COL_DOUBLE;
    if ((col.kind != HDK_COL_INT && !fp) || (col.width != 4 && col.width != 8) || (fp && col.width != 8)) return false;
    if (tg.agg != HDK_AGG_COUNT && ((tg.arg_is_fp != 0) != fp || tg.arg_is_fp == HDK_FP_SLOT_FLOAT)) return false;  // no promotion here
    const int nullable = tg.skip_null && tg.arg.nullable;
    if (have_val && (fa->val_buf_idx != col.buf_idx || fa->val_nullable != nullable)) return false;
    have_val = true;
    vw = col.width;
    fa->val_buf_idx = col.buf_idx;
    fa->val_nullable = nullable;
    fa->val_null = tg.arg.null_val;
    fa->val_is_fp = fp;
    if (tg.skip_null && !tg.arg.nullable) {
      // (a skip_val target over a column that cannot be NULL still has its non-NULL word: it stays equal to the rows)
    }
    want_nulls = want_nulls || wl.nword[t] >= 0;
    want_sum = want_sum || tg.agg == HDK_AGG_SUM || tg.agg == HDK_AGG_AVG;
    want_min = want_min || tg.agg == HDK_AGG_MIN;
    want_max = want_max || tg.agg == HDK_AGG_MAX;
  }
  BhLdsLayout& ll = fa->ll;
  int n = 0;
  ll.lwop[n++] = WOP_ADD_U64;  // rows
  fa->lw_nulls = fa->lw_sum = fa->lw_min = fa->lw_max = -1;
  const bool fp = fa->val_is_fp != 0;
  if (want_nulls) { fa->lw_nulls = n; ll.lwop[n++] = WOP_ADD_U64; }
  if (want_sum) { fa->lw_sum = n; ll.lwop[n++] = fp ? WOP_ADD_F64 : WOP_ADD_U64; }
  if (want_min) { fa->lw_min = n; ll.lwop[n++] = fp ? WOP_MIN_F64 : WOP_MIN_I64; }
  if (want_max) { fa->lw_max = n; ll.lwop[n++] = fp ? WOP_MAX_F64 : WOP_MAX_I64; }
  ll.nlw = n;
  ll.lmap[0] = 0;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (wl.vword[t] >= 0) ll.lmap[wl.vword[t]] = tg.agg == HDK_AGG_MIN ? fa->lw_min : (tg.agg == HDK_AGG_MAX ? fa->lw_max : fa->lw_sum);
    if (wl.nword[t] >= 0) ll.lmap[wl.nword[t]] = fa->lw_nulls;
  }
  if (!fa->val_nullable && fa->lw_nulls >= 0) {
    // the NULL word is never written: it stays 0 and rows - 0 is the non-NULL count
  }
  // geometry: replicas while the table is tiny (few groups = few LDS addresses), one 512-thread block per CU beyond 64 KiB
  const uint32_t W = static_cast<uint32_t>(n) + 1;
  const uint32_t cap_log2 = pow2_ceil_log2(p->entry_count < 2 ? 2 : p->entry_count);
  const uint64_t one = (static_cast<uint64_t>(W) << cap_log2) * 8;
  if (one > kBhFastLdsMaxBytes) return false;
  uint32_t rep = 32;
  while (rep > 1 && one * rep > kBhLdsReplicatedBytes) rep >>= 1;
  fa->g.out_entry_count = p->entry_count;
  fa->g.cap_log2 = cap_log2;
  fa->g.rep = rep;
  fa->g.lds_bytes = static_cast<uint32_t>(one * rep);
  *kw_out = kc.width;
  *vw_out = have_val ? vw : 0;
  *block_out = one > kBhLdsMaxBytes ? 512 : 256;
  return true;
}

template <int KW, int VW, int BLOCK>
static const void* bh_fast_kernel_of() {
  constexpr int U = 4;
  return reinterpret_cast<const void*>(hdk_scan_agg_bh_direct<KW, VW, U, BLOCK>);
}
template <int BLOCK>
static const void* bh_fast_kernel(int kw, int vw) {
  if (kw == 4) return vw == 0 ? bh_fast_kernel_of<4, 0, BLOCK>() : (vw == 4 ? bh_fast_kernel_of<4, 4, BLOCK>() : bh_fast_kernel_of<4, 8, BLOCK>());
  return vw == 0 ? bh_fast_kernel_of<8, 0, BLOCK>() : (vw == 4 ? bh_fast_kernel_of<8, 4, BLOCK>() : bh_fast_kernel_of<8, 8, BLOCK>());
}

static const void* bh_kernel(const hdk_hip_plan* p) {
  return p->num_joins ? reinterpret_cast<const void*>(hdk_scan_agg_bh_vec_join) : reinterpret_cast<const void*>(hdk_scan_agg_bh_vec);
}

const char* bh_lds_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko) {
  if (const char* pk = bh_packed_kernel_name(p, ko)) return pk;
  BhGeom g;
  BhFastArgs fa;
  int kw, vw, block;
  if (match_bh_fast(p, ko, &fa, &kw, &vw, &block)) return "hdk_scan_agg_bh_direct";
  if (!match_bh_lds(p, ko, &g)) return nullptr;
  return p->num_joins ? "hdk_scan_agg_bh_vec_join" : "hdk_scan_agg_bh_vec";
}

int32_t launch_bh_vec_armed(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                            const hdk_hip_device_properties* props, hipStream_t s, const uint32_t* run_if) {
  BhGeom g;
  if (!match_bh_lds(plan, ko, &g)) {
    set_error("the open-addressing interpreter does not take this plan");
    return HDK_HIP_ERR_UNSUPPORTED;
  }
  const void* k = bh_kernel(plan);
  if (g.lds_bytes > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(g.lds_bytes)));
  }
  VecArgs v;
  memset(&v, 0, sizeof(v));
  v.plan = d_plan;
  v.kp = kp;
  v.entry_count = g.out_entry_count;
  v.rep = g.rep;
  v.run_if = run_if;
  v.bh_cap_log2 = g.cap_log2;
  void* kargs[] = {&v};
  HDK_HIP_CHECK(hipLaunchKernel(k, dim3(resident_grid(k, kVecBlock, g.lds_bytes, props)), dim3(kVecBlock), kargs, g.lds_bytes, s));
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

int32_t launch_bh_lds(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                      const hdk_hip_device_properties* props, hipStream_t s, bool* launched) {
  *launched = false;
  {
    const int32_t st = launch_bh_packed(plan, d_plan, kp, ko, props, s, launched);
    if (st || *launched) return st;
  }
  {
    BhFastArgs fa;
    int kw, vw, block;
    if (match_bh_fast(plan, ko, &fa, &kw, &vw, &block)) {
      fa.plan = d_plan;
      fa.kp = kp;
      const void* k = block == 512 ? bh_fast_kernel<512>(kw, vw) : bh_fast_kernel<256>(kw, vw);
      if (fa.g.lds_bytes > (48u << 10)) {
        HDK_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(fa.g.lds_bytes)));
      }
      // streaming blocks: two per CU of 256 threads (as the perfect-hash kernel runs C2), more while the table is tiny --
      // narrow rows carry more LDS atomics per byte; one 512-thread block per CU when the table needs the LDS
      uint32_t grid = resident_grid(k, block, fa.g.lds_bytes, props);
      const uint32_t cu = static_cast<uint32_t>(props->num_cu);
      const int per_cu_env = hdk_sw(SW_BH_BLOCKS_PER_CU) ? atoi(hdk_sw(SW_BH_BLOCKS_PER_CU)) : 0;  // (measurements)
      const uint32_t want = (per_cu_env > 0 ? static_cast<uint32_t>(per_cu_env) : (block == 512 ? 1u : 4u)) * cu;
      if (grid > want) grid = want;
      if (ko && ko->grid_dim_x) grid = ko->grid_dim_x;
      AsyncScratch scratch(s);
      BhFoldArgs fold;
      const bool two_level = bh_two_level_fold(scratch, fa.g, fa.ll, grid, s, &fold);
      fa.slabs = two_level ? static_cast<int64_t*>(scratch.p) : nullptr;
      void* kargs[] = {&fa};
      HDK_HIP_CHECK(hipLaunchKernel(k, dim3(grid), dim3(block), kargs, fa.g.lds_bytes, s));
      HDK_HIP_CHECK(hipGetLastError());
      if (two_level) {
        const int32_t st = launch_bh_fold_words(fold, d_plan, kp, s);
        if (st) return st;
      }
      *launched = true;
      return HDK_HIP_OK;
    }
  }
  BhGeom g;
  if (!match_bh_lds(plan, ko, &g)) return HDK_HIP_OK;
  const void* k = bh_kernel(plan);
  if (g.lds_bytes > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(g.lds_bytes)));
  }
  const uint32_t grid = (ko && ko->grid_dim_x) ? ko->grid_dim_x : resident_grid(k, kVecBlock, g.lds_bytes, props);
  VecArgs v;
  memset(&v, 0, sizeof(v));
  v.plan = d_plan;
  v.kp = kp;
  v.slabs = nullptr;
  v.entry_count = g.out_entry_count;
  v.rep = g.rep;
  v.run_if = nullptr;
  v.bh_cap_log2 = g.cap_log2;
  AsyncScratch scratch(s);
  BhFoldArgs fold;
  WordLayout wl;
  make_word_layout(plan, &wl);
  BhLdsLayout ll;
  ll.nlw = wl.wpe;
  for (int w = 0; w < kMaxWordsPerEntry; ++w) {
    ll.lwop[w] = w < wl.wpe ? wl.wop[w] : 0;
    ll.lmap[w] = w;
  }
  const bool two_level = bh_two_level_fold(scratch, g, ll, grid, s, &fold);
  v.slabs = two_level ? static_cast<int64_t*>(scratch.p) : nullptr;
  void* kargs[] = {&v};
  HDK_HIP_CHECK(hipLaunchKernel(k, dim3(grid), dim3(kVecBlock), kargs, g.lds_bytes, s));
  HDK_HIP_CHECK(hipGetLastError());
  if (two_level) {
    const int32_t st = launch_bh_fold_words(fold, d_plan, kp, s);
    if (st) return st;
  }
  *launched = true;
  return HDK_HIP_OK;
}

}  // namespace hdk
