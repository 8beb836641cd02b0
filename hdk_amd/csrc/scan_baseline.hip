// scan_baseline.hip -- open-addressing group-by strategies: the radix-partitioned passes (scan_agg_partitioned.h), the
// specialised global-atomics kernel (scan_agg_baseline_fast.h) and the general one (scan_agg_global.h), with their
// matchers.  A translation unit of its own (scan_agg.hip holds the API and the LDS strategies): the kernels of one
// strategy family are compiled together and nothing else.
#include <algorithm>
#include <vector>
#define HDK_SCAN_AGG_GLOBAL_KERNEL
#include "host_match.h"
#include "scan_bh_host.h"
#include "scan_bhm_host.h"
#include "scan_agg_baseline_fast.h"
#include "scan_agg_global.h"
#include "scan_agg_partitioned.h"
#include "scan_agg_perfect_part.h"

namespace hdk {

static const void* baseline_direct_kernel(const hdk_hip_plan* p) {
  if (p->key_width == 4) {
    return p->key_count == 2 ? reinterpret_cast<const void*>(hdk_scan_agg_baseline_direct<int32_t, 2>)
                             : reinterpret_cast<const void*>(hdk_scan_agg_baseline_direct<int32_t, 1>);
  }
  return p->key_count == 2 ? reinterpret_cast<const void*>(hdk_scan_agg_baseline_direct<int64_t, 2>)
                           : reinterpret_cast<const void*>(hdk_scan_agg_baseline_direct<int64_t, 1>);
}

static void launch_baseline_direct(const hdk_hip_plan* p, const BaseFastArgs& fa, unsigned grid, hipStream_t s) {
  if (p->key_width == 4) {
    if (p->key_count == 2) {
      hipLaunchKernelGGL((hdk_scan_agg_baseline_direct<int32_t, 2>), dim3(grid), dim3(kBaseFastBlock), 0, s, fa);
    } else {
      hipLaunchKernelGGL((hdk_scan_agg_baseline_direct<int32_t, 1>), dim3(grid), dim3(kBaseFastBlock), 0, s, fa);
    }
  } else if (p->key_count == 2) {
    hipLaunchKernelGGL((hdk_scan_agg_baseline_direct<int64_t, 2>), dim3(grid), dim3(kBaseFastBlock), 0, s, fa);
  } else {
    hipLaunchKernelGGL((hdk_scan_agg_baseline_direct<int64_t, 1>), dim3(grid), dim3(kBaseFastBlock), 0, s, fa);
  }
}

// the shape hdk_scan_agg_baseline_direct takes (scan_agg_baseline_fast.h)
static bool match_baseline_fast(const hdk_hip_plan* p, BaseFastArgs* fa) {
  if (p->query_kind != HDK_Q_BASELINE_HASH || p->output_columnar || p->num_joins || p->key_count < 1 ||
      p->key_count > 2) {
    return false;
  }
  int kc;
  if (!plain_outer_col(p, p->keys[0], &kc) || p->cols[kc].kind != HDK_COL_INT) return false;
  memset(fa, 0, sizeof(*fa));
  fa->key_buf_idx = p->cols[kc].buf_idx;
  fa->key_width = p->cols[kc].width;
  fa->key_kind = p->cols[kc].kind;
  fa->nkeys = p->key_count;
  if (p->key_count == 2) {
    int kc2;
    if (!plain_outer_col(p, p->keys[1], &kc2) || p->cols[kc2].kind != HDK_COL_INT) return false;
    fa->key2_buf_idx = p->cols[kc2].buf_idx;
    fa->key2_width = p->cols[kc2].width;
    fa->key2_kind = p->cols[kc2].kind;
  }
  if (!match_plain_quals(p, fa->q)) return false;
  fa->nquals = p->num_quals;
  int n = 0;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_ID) {
      if (tg.slot_width != 0) return false;  // (perfect-hash style key slots: generic kernel)
      continue;
    }
    if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) return false;  // float accumulators: generic kernel
    if (tg.agg == HDK_AGG_SINGLE_VALUE) return false;      // checked_single_agg_id: generic kernel (scan_agg_global.h)
    BaseFastTarget ft;
    ft.target = t;
    ft.buf_idx = -1;
    ft.width = 8;
    ft.kind = HDK_COL_INT;
    if (tg.has_arg) {
      int c;
      if (!plain_outer_col(p, tg.arg, &c)) return false;
      ft.buf_idx = p->cols[c].buf_idx;
      ft.width = p->cols[c].width;
      ft.kind = p->cols[c].kind;
    } else if (tg.agg != HDK_AGG_COUNT) {
      return false;
    }
    fa->tg[n++] = ft;
  }
  fa->ntargets = n;
  return true;
}

// ---- radix-partitioned open-addressing group-by (scan_agg_partitioned.h) -------------------------------
// Taken for the hdk_scan_agg_baseline_direct shape when the table is large enough that the memory-side
// atomic rate is the bound (>= 2 M entries, >= 8 M rows) and the caller told us the row count.
// key_hash (QE/GroupByRuntime.cpp:24-29: MurmurHash3 of the packed key, seed 0) of the key (k, 0) on the host, for the
// padding keys of the partitioned group-by; same word order as key_hash_dev (baseline_table.h)
template <typename K>
static uint32_t host_key_hash(int64_t k, int nkeys) {
  uint32_t h1 = 0;
  auto rotl = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
  auto mix = [&](uint32_t k1) {
    k1 *= 0xcc9e2d51u;
    k1 = rotl(k1, 15);
    k1 *= 0x1b873593u;
    h1 ^= k1;
    h1 = rotl(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  };
  for (int i = 0; i < nkeys; ++i) {
    const uint64_t v = i == 0 ? static_cast<uint64_t>(k) : 0;
    mix(static_cast<uint32_t>(v));
    if (sizeof(K) == 8) mix(static_cast<uint32_t>(v >> 32));
  }
  h1 ^= static_cast<uint32_t>(nkeys * sizeof(K));
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}

// ---- what a tuple carries (independent of the table's geometry) ---------------------------------------------------
static const hdk_hip_col* outer_col_of_buf(const hdk_hip_plan* p, int32_t buf_idx) {
  for (int i = 0; i < p->num_cols && i < HDK_HIP_MAX_COLS; ++i) {
    if (p->cols[i].table == 0 && p->cols[i].buf_idx == buf_idx) return &p->cols[i];
  }
  return nullptr;
}

// does hdk_part_aggregate_simple apply?  rows of [key quad | one 8-byte integer slot]
static bool part_simple_shape(const hdk_hip_plan* p, PartArgs* pa) {
  pa->simple_agg = -1;
  const int ltw = pa->narrow ? 2 : pa->tw;  // tuple words as the readers see them
  if (p->row_size_quad != 2 || pa->nkeys != 1 || ltw > 2) return false;
  int found = -1;
  for (int i = 0; i < pa->ntargets; ++i) {
    const hdk_hip_target& tg = p->targets[pa->tgt_index[i]];
    if (tg.agg == HDK_AGG_ID && tg.slot_width == 0) continue;
    if (found >= 0) return false;
    found = i;
  }
  if (found < 0) return false;
  const hdk_hip_target& tg = p->targets[pa->tgt_index[found]];
  if (tg.slot_width != 8 || tg.slot_off != 8 || tg.arg_is_fp) return false;
  if (tg.agg != HDK_AGG_SUM && tg.agg != HDK_AGG_MIN && tg.agg != HDK_AGG_MAX && tg.agg != HDK_AGG_COUNT) return false;
  if (tg.has_arg && (pa->tgt_arg[found] != 1 || pa->arg[0].kind != HDK_COL_INT)) return false;
  if (!tg.has_arg && tg.agg != HDK_AGG_COUNT) return false;
  pa->simple_agg = tg.agg;
  pa->simple_has_arg = tg.has_arg;
  pa->simple_skip = tg.skip_null;
  pa->simple_arg_nullable = tg.arg.nullable;
  pa->simple_null = tg.null_val;
  pa->simple_arg_null = tg.arg.null_val;
  return true;
}

static bool part_tuple_shape(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, PartArgs* pa) {
  const bool keep_wide = (ko && (ko->flags & HDK_HIP_LAUNCH_WIDE_TUPLES)) || hdk_sw(SW_PART_WIDE);
  BaseFastArgs bf;
  if (!match_baseline_fast(p, &bf)) return false;
  if (p->row_size_quad == 0 || p->row_size_quad > 16 || p->entry_count < 128) return false;
  memset(pa, 0, sizeof(*pa));
  pa->key_buf_idx = bf.key_buf_idx;
  pa->key_width = bf.key_width;
  pa->key_kind = bf.key_kind;
  pa->nkeys = bf.nkeys;
  pa->key2_buf_idx = bf.key2_buf_idx;
  pa->key2_width = bf.key2_width;
  pa->key2_kind = bf.key2_kind;
  pa->nquals = bf.nquals;
  for (int i = 0; i < bf.nquals; ++i) pa->q[i] = bf.q[i];
  for (int t = 0; t < bf.ntargets; ++t) {
    const BaseFastTarget& ft = bf.tg[t];
    int word = 0;
    if (ft.buf_idx >= 0) {
      for (int k = 0; k < pa->nargs; ++k) {
        if (pa->arg[k].buf_idx == ft.buf_idx) word = 1 + k;
      }
      if (!word) {
        if (pa->nkeys + pa->nargs == kPartMaxTW) return false;  // the tuple holds 3 words: keys + argument columns
        pa->arg[pa->nargs] = ft;
        word = 1 + pa->nargs++;
      }
    }
    pa->tgt_index[t] = ft.target;
    pa->tgt_arg[t] = word ? word - 1 + pa->nkeys : 0;  // absolute tuple word of the argument
  }
  pa->ntargets = bf.ntargets;
  pa->tw = pa->nkeys + pa->nargs;
  pa->all_wide = pa->key_width == 8 && pa->key_kind == HDK_COL_INT && (pa->nkeys < 2 || (pa->key2_width == 8 && pa->key2_kind == HDK_COL_INT));
  for (int k = 0; k < pa->nargs; ++k) {
    if (pa->arg[k].width != 8 || (pa->arg[k].kind != HDK_COL_INT && pa->arg[k].kind != HDK_COL_DOUBLE)) pa->all_wide = 0;
  }
  // narrow tuples (scan_agg_partitioned.h): one 4-byte table key, one integer argument column that the statistics
  // put inside 32 bits -- a nullable one gives up INT32_MIN for its in-band NULL
  if (pa->nkeys == 1 && p->key_width == 4 && pa->nargs == 1 && pa->arg[0].kind == HDK_COL_INT && !keep_wide) {
    const hdk_hip_col* c = outer_col_of_buf(p, pa->arg[0].buf_idx);
    if (c && c->has_stats && c->min_val >= static_cast<int64_t>(INT32_MIN) + (c->has_nulls ? 1 : 0) &&
        c->max_val <= static_cast<int64_t>(INT32_MAX) && c->min_val <= c->max_val) {
      pa->narrow = 1;
      pa->narrow_null = c->has_nulls ? 1 : 0;
      for (int t = 0; t < pa->ntargets; ++t) {  // the column's in-band NULL, as the targets' argument leaf names it
        const hdk_hip_target& tg = p->targets[pa->tgt_index[t]];
        if (pa->tgt_arg[t]) {
          if (pa->narrow_null && !tg.arg.nullable) pa->narrow = 0;  // (statistics and type disagree: stay wide)
          pa->narrow_arg_null = tg.arg.null_val;
          // "no NULLs" by the statistics, nullable by the type: a NULL showing up anyway must not be summed as a value
          pa->narrow_null_is_stale = (!pa->narrow_null && tg.arg.nullable && tg.skip_null) ? 1 : 0;
        }
      }
      if (pa->narrow) pa->tw = 1;
    }
  }
  // pass 3 with a structure-of-arrays image (hdk_part_aggregate_soa): one 4-byte key, SUM over one integer column
  const bool simple = part_simple_shape(p, pa);
  pa->soa = simple && p->key_width == 4 && pa->simple_agg == HDK_AGG_SUM && pa->simple_has_arg && (pa->narrow || pa->tw == 2) &&
            !hdk_sw(SW_PART_AOS);
  return true;
}

// ---- geometry of the passes for a table of `entry_count` entries: the plan's (`owners` == 0, a one-GPU job), or an
// owner's in a tuple exchange among `owners` >= 1 ranks (one owner: a rank exchanging with itself, the shape RCCL tests
// on a one-GPU box run) ----------------------------------------------------------------------------------------------
static bool part_geometry(const hdk_hip_plan* p, uint64_t rows, uint32_t entry_count, uint32_t owners, PartArgs* pa) {
  if (entry_count < 128) return false;
  pa->entry_count = entry_count;
  pa->owners = owners;
  magic_u32(entry_count, &pa->mod_magic, &pa->mod_shift);
  // regions: as many entries as fit the LDS image
  pa->slots = pa->soa ? kPartSoaSlots : kPartLdsBytes / (p->row_size_quad * 8);
  if (pa->slots < 16 || pa->slots >= entry_count) return false;
  magic_u32(pa->slots, &pa->reg_magic, &pa->reg_shift);
  const uint64_t pf = (static_cast<uint64_t>(entry_count) + pa->slots - 1) / pa->slots;
  // two scatter levels of <= 256 bins each, as even as powers of two allow (longer runs per bin and batch); level 1
  // also separates the owners
  const uint64_t g = owners ? owners : 1;
  uint32_t p2_log2 = (pow2_ceil_log2(pf * g) + 1) / 2;
  if (p2_log2 > pow2_ceil_log2(pf)) p2_log2 = pow2_ceil_log2(pf);
  while (((pf + (1ull << p2_log2) - 1) >> p2_log2) * g > static_cast<uint64_t>(kPartMaxBins)) ++p2_log2;
  if ((1u << p2_log2) > static_cast<uint32_t>(kPartMaxBins)) return false;  // > 64 K regions: a third level would be needed
  pa->fine_count = static_cast<uint32_t>(pf);
  pa->p2_log2 = p2_log2;
  pa->p1 = static_cast<uint32_t>((pf + (1ull << p2_log2) - 1) >> p2_log2);
  // whole 128-byte lines: runs of G tuples (8 of 16 B; 16 of 8 or 24 B), padded with keys of another partition --
  // which needs two coarse partitions (a table this small gains nothing from alignment anyway)
  // Measured at the C5 shape (256 M rows): rounding runs up to whole lines costs more than it gains -- the padding
  // is 18 % of pass 1's output and compounds to 44 % of pass 3's input, and its staging slots cost the third block
  // per CU: 4.9 + 3.1 + 2.4 ms against 2.6 + 2.7 + 2.0 ms for runs as they come.  Runs as they come is the default;
  // HDK_HIP_PART_G_LOG2=3 turns the padding on for measurements (wide tuples on one GPU only).
  pa->g_log2 = 0;
  if (const char* e = hdk_sw(SW_PART_G_LOG2)) pa->g_log2 = (pa->p1 < 2 || owners || pa->narrow) ? 0 : atoi(e);
  if (pa->g_log2) {
    int found = 0;
    for (int64_t k = 1; k < 4096 && found < 2; ++k) {
      const uint32_t h = p->key_width == 4 ? host_key_hash<int32_t>(k, pa->nkeys) : host_key_hash<int64_t>(k, pa->nkeys);
      const uint32_t c = static_cast<uint32_t>((h % entry_count) / pa->slots) >> p2_log2;
      if (found == 0 || c != pa->pad_coarse[0]) {
        // (tuples of 4-byte keys carry their home in the upper half of word 0: part_pack_home)
        pa->pad_key[found] = p->key_width == 4 ? static_cast<int64_t>((static_cast<uint64_t>(h % entry_count) << 32) | static_cast<uint32_t>(k)) : k;
        pa->pad_coarse[found] = c;
        ++found;
      }
    }
    if (found < 2) pa->g_log2 = 0;
  }
  const uint64_t gr = 1ull << pa->g_log2;
  const uint64_t tile = static_cast<uint64_t>(part_tile(pa->narrow != 0));
  pa->total_rows = rows;
  auto round_g = [&](uint64_t x) { return (x + gr - 1) & ~(gr - 1); };
  // a coarse slab takes the rows of P2 regions out of PF (the last one fewer) -- of ONE owner out of G when the launch
  // scatters to owners: uniform hash, 6 % + 8 K slack, plus the padding: on average (G - 1) / 2 slots per bin and batch
  const uint64_t batches = rows / tile + 1;
  const uint64_t share1 = static_cast<uint64_t>((static_cast<unsigned __int128>(rows) << p2_log2) / (pf * g)) + 1;
  pa->cap1 = round_g(share1 + share1 / 16 + 8192 + batches * (gr - 1) * 5 / 8);
  // level 2 scatters what ALL ranks sent for the coarse slab: g x share1 tuples
  const uint64_t batches2 = share1 * g / tile + kPartG2X;  // batches a coarse slab is scattered in
  pa->cap2 = round_g(rows / pf + rows / (pf * 4) + 256 + batches2 * (gr - 1) * 5 / 8);  // 25 % + 256 + padding
  pa->cap2 = (pa->cap2 + 15) & ~15ull;  // (slabs of 8-byte tuples start 16-byte aligned; whole lines for every width)
  pa->cap_ovf = rows / 16 + 4096;
  pa->sub1 = ((pa->cap1 / kPartXcds + kPartXcds * 256) + 15) & ~15ull;  // per-XCD share of a coarse slab, with slack, whole lines for every tuple width
  pa->cap1 = pa->sub1 * kPartXcds;
  pa->cap_spill = owners ? rows / 16 + 4096 : static_cast<uint64_t>(pa->p1) * pa->cap1;  // one GPU: slab 1, reused
  if (pa->cap1 > 0xFFFFFFF0ull || pa->cap2 > 0xFFFF0000ull || pa->cap_ovf > 0xFFFFFFF0ull) return false;  // 32-bit cursors (and 32-bit tuple indices with look-ahead in pass 3)
  return true;
}

static bool match_partitioned(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, PartArgs* pa) {
  if (!ko || ko->total_rows == 0) return false;
  const bool forced = (ko->flags & HDK_HIP_LAUNCH_FORCE_PARTITIONED) != 0;
  if (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR)) return false;
  if (!forced && (p->entry_count < (2u << 20) || ko->total_rows < (8ull << 20))) return false;
  return part_tuple_shape(p, ko, pa) && part_geometry(p, ko->total_rows, p->entry_count, 0, pa);
}

constexpr int32_t kPartitionedNoScratch = -1000;  // internal: scratch for the slabs could not be allocated

// kernel of a scatter level for the tuple format of `pa`
template <int LEVEL>
static const void* part_scatter_kernel(const PartArgs& pa, bool k32) {
  if (pa.narrow) return reinterpret_cast<const void*>(hdk_part_scatter<LEVEL, int32_t, 1, true>);
  if (k32) {
    switch (pa.tw) {
      case 1: return reinterpret_cast<const void*>(hdk_part_scatter<LEVEL, int32_t, 1>);
      case 2: return reinterpret_cast<const void*>(hdk_part_scatter<LEVEL, int32_t, 2>);
      default: return reinterpret_cast<const void*>(hdk_part_scatter<LEVEL, int32_t, 3>);
    }
  }
  switch (pa.tw) {
    case 1: return reinterpret_cast<const void*>(hdk_part_scatter<LEVEL, int64_t, 1>);
    case 2: return reinterpret_cast<const void*>(hdk_part_scatter<LEVEL, int64_t, 2>);
    default: return reinterpret_cast<const void*>(hdk_part_scatter<LEVEL, int64_t, 3>);
  }
}

template <int LEVEL>
static void launch_part_scatter(const PartArgs& pa, bool k32, dim3 grid, size_t lds, hipStream_t s) {
  if (pa.narrow) {
    hipLaunchKernelGGL((hdk_part_scatter<LEVEL, int32_t, 1, true>), grid, dim3(kPartBlock), lds, s, pa);
  } else if (k32) {
    switch (pa.tw) {
      case 1: hipLaunchKernelGGL((hdk_part_scatter<LEVEL, int32_t, 1>), grid, dim3(kPartBlock), lds, s, pa); break;
      case 2: hipLaunchKernelGGL((hdk_part_scatter<LEVEL, int32_t, 2>), grid, dim3(kPartBlock), lds, s, pa); break;
      default: hipLaunchKernelGGL((hdk_part_scatter<LEVEL, int32_t, 3>), grid, dim3(kPartBlock), lds, s, pa); break;
    }
  } else {
    switch (pa.tw) {
      case 1: hipLaunchKernelGGL((hdk_part_scatter<LEVEL, int64_t, 1>), grid, dim3(kPartBlock), lds, s, pa); break;
      case 2: hipLaunchKernelGGL((hdk_part_scatter<LEVEL, int64_t, 2>), grid, dim3(kPartBlock), lds, s, pa); break;
      default: hipLaunchKernelGGL((hdk_part_scatter<LEVEL, int64_t, 3>), grid, dim3(kPartBlock), lds, s, pa); break;
    }
  }
}

// passes 2-4 (level-2 scatter, LDS aggregation, overflow) for the sources and scratch `pa` names
static void launch_part_tail(const hdk_hip_plan* plan, PartArgs& pa, const hdk_hip_device_properties* props, hipStream_t s) {
  const bool k32 = plan->key_width == 4;
  const uint32_t gmask = (1u << pa.g_log2) - 1;
  const size_t lds2 = part_scatter_lds_bytes(1u << pa.p2_log2, gmask, pa.tw, pa.narrow != 0);
  const unsigned go = static_cast<unsigned>(props->num_cu) * 4;
  // pass 2: kPartG2X blocks per coarse slab, all of them on one XCD (block id % 8 picks the slab inside a set of eight)
  const unsigned g2 = ((pa.p1 + kPartXcds - 1) / kPartXcds) * kPartXcds * kPartG2X;
  const size_t table_bytes = static_cast<size_t>(pa.slots) * plan->row_size_quad * 8;
  const bool simple = pa.simple_agg >= 0 && !hdk_sw(SW_PART_GENERAL);  // (part_simple_shape; env: A/B measurements)
  const bool sum2 = simple && (pa.narrow || pa.tw == 2) && pa.simple_agg == HDK_AGG_SUM;
  launch_part_scatter<2>(pa, k32, dim3(g2), lds2, s);
  const dim3 ga(pa.fine_count), ba(kPartAggBlock);
  if (pa.soa) {
    // (the bits are needed when a tuple may carry a NULL argument, or the table holds an earlier launch's groups)
    const bool nulls = pa.simple_skip && !(pa.narrow && !pa.narrow_null && pa.init_output);
    const size_t lds = part_soa_lds_bytes(pa.slots);
    if (pa.narrow) {
      if (nulls) {
        hipLaunchKernelGGL((hdk_part_aggregate_soa<true, true>), ga, ba, lds, s, pa);
      } else {
        hipLaunchKernelGGL((hdk_part_aggregate_soa<true, false>), ga, ba, lds, s, pa);
      }
    } else if (nulls) {
      hipLaunchKernelGGL((hdk_part_aggregate_soa<false, true>), ga, ba, lds, s, pa);
    } else {
      hipLaunchKernelGGL((hdk_part_aggregate_soa<false, false>), ga, ba, lds, s, pa);
    }
  } else if (pa.narrow && simple) {
    if (sum2 && pa.simple_skip) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int32_t, 1, HDK_AGG_SUM, 1, true>), ga, ba, table_bytes, s, pa);
    } else if (sum2) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int32_t, 1, HDK_AGG_SUM, 0, true>), ga, ba, table_bytes, s, pa);
    } else {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int32_t, 1, -1, -1, true>), ga, ba, table_bytes, s, pa);
    }
  } else if (k32) {
    if (sum2 && pa.simple_skip) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int32_t, 2, HDK_AGG_SUM, 1>), ga, ba, table_bytes, s, pa);
    } else if (sum2) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int32_t, 2, HDK_AGG_SUM, 0>), ga, ba, table_bytes, s, pa);
    } else if (simple) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int32_t>), ga, ba, table_bytes, s, pa);
    } else {
      hipLaunchKernelGGL(hdk_part_aggregate<int32_t>, ga, ba, table_bytes, s, pa);
    }
  } else {
    if (sum2 && pa.simple_skip) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int64_t, 2, HDK_AGG_SUM, 1>), ga, ba, table_bytes, s, pa);
    } else if (sum2) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int64_t, 2, HDK_AGG_SUM, 0>), ga, ba, table_bytes, s, pa);
    } else if (simple) {
      hipLaunchKernelGGL((hdk_part_aggregate_simple<int64_t>), ga, ba, table_bytes, s, pa);
    } else {
      hipLaunchKernelGGL(hdk_part_aggregate<int64_t>, ga, ba, table_bytes, s, pa);
    }
  }
  if (k32) {
    hipLaunchKernelGGL(hdk_part_overflow<int32_t>, dim3(go), dim3(kPartBlock), 0, s, pa);
  } else {
    hipLaunchKernelGGL(hdk_part_overflow<int64_t>, dim3(go), dim3(kPartBlock), 0, s, pa);
  }
}

// scratch of the passes behind level 1, carved from `q` (or just sized when q == nullptr): fine slabs, overflow area,
// spill segments, [spill list when level 1's slabs are not ours to reuse], cursors
static size_t part_carve_tail(PartArgs& pa, int8_t* q, bool own_spill_list, size_t* cursor_bytes, uint32_t** cursors) {
  auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t tw = static_cast<size_t>(pa.tw);
  const size_t b2 = static_cast<size_t>(pa.fine_count) * pa.cap2 * tw * 8;
  const size_t bo = static_cast<size_t>(pa.cap_ovf) * tw * 8;
  const size_t bs = static_cast<size_t>(pa.fine_count) * kPartSpillSeg * tw * 8;
  const size_t bl = own_spill_list ? static_cast<size_t>(pa.cap_spill) * tw * 8 : 0;
  const size_t nc = 2 * static_cast<size_t>(pa.fine_count) + 4;  // fill2 | nspill | fill_ovf, fill_spill, fallback
  const size_t bc = nc * sizeof(uint32_t);
  if (q) {
    pa.slab2 = reinterpret_cast<int64_t*>(q); q += up(b2);
    pa.ovf = reinterpret_cast<int64_t*>(q); q += up(bo);
    pa.spill_seg = reinterpret_cast<int64_t*>(q); q += up(bs);
    if (own_spill_list) {
      pa.slab1 = reinterpret_cast<int64_t*>(q);  // (pass 3 and 4 know the shared spill list as `slab1`)
      q += up(bl);
    }
    pa.fill2 = reinterpret_cast<uint32_t*>(q);
    pa.nspill = pa.fill2 + pa.fine_count;
    pa.fill_ovf = pa.nspill + pa.fine_count;
    pa.fill_spill = pa.fill_ovf + 1;
    pa.fallback = pa.fill_spill + 1;
    *cursors = pa.fill2;
  }
  *cursor_bytes = bc;
  return up(b2) + up(bo) + up(bs) + up(bl) + up(bc);
}

static int32_t launch_scan_partitioned(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                                       PartArgs pa, const LaunchShape& shape, const hdk_hip_device_properties* props,
                                       hipStream_t s) {
  pa.plan = d_plan;
  pa.kp = kp;
  const bool k32 = plan->key_width == 4;
  const size_t tw = static_cast<size_t>(pa.tw);
  const uint32_t gmask = (1u << pa.g_log2) - 1;
  const size_t lds1 = part_scatter_lds_bytes(pa.p1, gmask, pa.tw, pa.narrow != 0);
  // pass-1 grid: what is resident, at most one block per batch
  unsigned g1 = scatter_grid(part_scatter_kernel<1>(pa, k32), kPartBlock, lds1, props);
  const uint64_t tile = static_cast<uint64_t>(part_tile(pa.narrow != 0));
  const uint64_t tiles = (pa.total_rows + tile - 1) / tile;
  if (tiles < g1) g1 = static_cast<unsigned>(tiles ? tiles : 1);
  auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t b1 = static_cast<size_t>(pa.p1) * pa.cap1 * tw * 8;
  const size_t bc1 = static_cast<size_t>(pa.p1) * kPartXcds * kPartCursorStride * sizeof(uint32_t);
  size_t bc2 = 0;
  uint32_t* cur2 = nullptr;
  const size_t tail = part_carve_tail(pa, nullptr, false, &bc2, &cur2);
  AsyncScratch scratch(s);
  const hipError_t me = hipMallocAsync(&scratch.p, up(b1) + up(bc1) + tail, s);
  if (me != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return kPartitionedNoScratch;  // not an error: the caller takes the global-atomics kernel instead
  }
  int8_t* q = static_cast<int8_t*>(scratch.p);
  pa.slab1 = reinterpret_cast<int64_t*>(q); q += up(b1);
  pa.fill1 = reinterpret_cast<uint32_t*>(q); q += up(bc1);
  part_carve_tail(pa, q, false, &bc2, &cur2);
  HDK_HIP_CHECK(hipMemsetAsync(pa.fill1, 0, bc1, s));
  HDK_HIP_CHECK(hipMemsetAsync(cur2, 0, bc2, s));
  pa.nsrc = 1;
  pa.src_slab[0] = pa.slab1;
  pa.src_fill[0] = pa.fill1;
  pa.src_fill_stride = kPartCursorStride;
  if (hdk_sw(SW_PART_TRACE)) {
    fprintf(stderr, "part: scratch %p +%zu | slab1 %p (%zu) fill1 %p (%zu) slab2 %p ovf %p spill_seg %p fill2 %p | p1 %u p2_log2 %u fine %u "
            "cap1 %llu sub1 %llu cap2 %llu cap_ovf %llu cap_spill %llu tw %d slots %u rows %llu table %p\n",
            scratch.p, up(b1) + up(bc1) + tail, (void*)pa.slab1, b1, (void*)pa.fill1, bc1, (void*)pa.slab2, (void*)pa.ovf,
            (void*)pa.spill_seg, (void*)pa.fill2, pa.p1, pa.p2_log2, pa.fine_count, (unsigned long long)pa.cap1,
            (unsigned long long)pa.sub1, (unsigned long long)pa.cap2, (unsigned long long)pa.cap_ovf,
            (unsigned long long)pa.cap_spill, pa.tw, pa.slots, (unsigned long long)pa.total_rows, (void*)nullptr);
  }
  launch_part_scatter<1>(pa, k32, dim3(g1), lds1, s);
  launch_part_tail(plan, pa, props, s);
  // armed fallback: runs only if the scatter passes found the data too skewed for slabs (or the statistics stale)
  BaseFastArgs bf;
  match_baseline_fast(plan, &bf);
  bf.plan = d_plan;
  bf.kp = kp;
  bf.entry_count = shape.entry_count;
  bf.run_if = pa.fallback;
  launch_baseline_direct(plan, bf, shape.grid, s);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;  // (`scratch` goes back to the pool here, stream-ordered)
}

// ---- multi-GPU tuple exchange (include/hdk_hip.h: hdk_hip_exchange_shape_for / scatter_to_owners / aggregate_from_ranks) ----
static size_t exchange_cursor_bytes(const PartArgs& pa) {
  // level-1 cursors one per 128-byte line, then the flag word
  return (static_cast<size_t>(pa.owners) * pa.p1 * kPartXcds * kPartCursorStride + 64) * sizeof(uint32_t);
}

int32_t exchange_shape(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko, int32_t num_owners,
                       uint32_t owner_entry_count, PartArgs* pa, hdk_hip_exchange_shape* out) {
  HDK_REQUIRE(num_owners >= 1 && num_owners <= kPartMaxSrc, "num_owners must be in [1, %d]", kPartMaxSrc);
  HDK_REQUIRE(ko && ko->total_rows, "hdk_hip_kernel_options::total_rows (the per-rank row bound) is required");
  if (plan->query_kind != HDK_Q_BASELINE_HASH || !part_tuple_shape(plan, ko, pa) ||
      !part_geometry(plan, ko->total_rows, owner_entry_count, static_cast<uint32_t>(num_owners), pa)) {
    set_error("plan or table geometry outside the radix-partitioned group-by's shape");
    return HDK_HIP_ERR_UNSUPPORTED;
  }
  auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  pa->seg_header_bytes = up((static_cast<size_t>(pa->p1) * kPartXcds + 2) * sizeof(uint32_t));  // counts, flag, shape tag
  pa->seg_bytes = pa->seg_header_bytes + up(static_cast<size_t>(pa->p1) * pa->cap1 * pa->tw * 8);
  if (out) {
    memset(out, 0, sizeof(*out));
    out->num_owners = static_cast<uint32_t>(num_owners);
    out->owner_entry_count = owner_entry_count;
    out->tuple_bytes = static_cast<uint32_t>(pa->tw * 8);
    out->coarse_per_owner = pa->p1;
    out->regions_log2 = pa->p2_log2;
    out->sub_slab_tuples = pa->sub1;
    out->segment_header_bytes = pa->seg_header_bytes;
    out->segment_bytes = pa->seg_bytes;
    out->rows_bound = ko->total_rows;
    out->scatter_workspace_bytes = kPlanRegionBytes + up(exchange_cursor_bytes(*pa));
    size_t bc2 = 0;
    uint32_t* cur2 = nullptr;
    PartArgs tmp = *pa;
    out->aggregate_workspace_bytes = kPlanRegionBytes + part_carve_tail(tmp, nullptr, true, &bc2, &cur2);
  }
  return HDK_HIP_OK;
}

// the shape the caller hands back must be the one shape_for computes for the same arguments
static int32_t exchange_args(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape,
                             PartArgs* pa) {
  HDK_REQUIRE(shape, "shape is NULL");
  hdk_hip_exchange_shape want;
  const int32_t st = exchange_shape(plan, ko, static_cast<int32_t>(shape->num_owners), shape->owner_entry_count, pa, &want);
  if (st) return st;
  HDK_REQUIRE(memcmp(&want, shape, sizeof(want)) == 0, "exchange shape does not belong to this plan / row bound");
  return HDK_HIP_OK;
}

int32_t launch_scatter_to_owners(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                                 const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape, int8_t* send,
                                 int8_t* cursors, const hdk_hip_device_properties* props, hipStream_t s) {
  PartArgs pa;
  const int32_t st = exchange_args(plan, ko, shape, &pa);
  if (st) return st;
  pa.plan = d_plan;
  pa.kp = kp;
  pa.send = send;
  const bool k32 = plan->key_width == 4;
  const size_t bc = exchange_cursor_bytes(pa);
  pa.fill1 = reinterpret_cast<uint32_t*>(cursors);
  pa.fallback = pa.fill1 + static_cast<size_t>(pa.owners) * pa.p1 * kPartXcds * kPartCursorStride;
  pa.fill_ovf = pa.fallback + 1;  // (never used: a scatter to owners has no overflow area)
  HDK_HIP_CHECK(hipMemsetAsync(pa.fill1, 0, bc, s));
  const size_t lds1 = part_scatter_lds_bytes(pa.owners * pa.p1, 0, pa.tw, pa.narrow != 0);
  unsigned g1 = scatter_grid(part_scatter_kernel<1>(pa, k32), kPartBlock, lds1, props);
  const uint64_t tile = static_cast<uint64_t>(part_tile(pa.narrow != 0));
  const uint64_t tiles = (pa.total_rows + tile - 1) / tile;
  if (tiles < g1) g1 = static_cast<unsigned>(tiles ? tiles : 1);
  launch_part_scatter<1>(pa, k32, dim3(g1), lds1, s);
  hipLaunchKernelGGL(hdk_part_publish, dim3(pa.owners), dim3(256), 0, s, pa);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

int32_t launch_aggregate_from_ranks(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                                    const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape,
                                    const int8_t* recv, int8_t* scratch, const hdk_hip_device_properties* props,
                                    hipStream_t s) {
  PartArgs pa;
  const int32_t st = exchange_args(plan, ko, shape, &pa);
  if (st) return st;
  pa.plan = d_plan;
  pa.kp = kp;
  // pass 3 writes every region of the owner's table -- or merges into the table an earlier chunk left (ACCUMULATE)
  pa.init_output = (ko && (ko->flags & HDK_HIP_LAUNCH_ACCUMULATE)) ? 0 : 1;
  size_t bc2 = 0;
  uint32_t* cur2 = nullptr;
  part_carve_tail(pa, scratch, true, &bc2, &cur2);
  HDK_HIP_CHECK(hipMemsetAsync(cur2, 0, bc2, s));
  pa.nsrc = pa.owners;
  for (uint32_t r = 0; r < pa.owners; ++r) {
    const int8_t* seg = recv + static_cast<size_t>(r) * pa.seg_bytes;
    pa.src_fill[r] = reinterpret_cast<const uint32_t*>(seg);
    pa.src_slab[r] = reinterpret_cast<const int64_t*>(seg + pa.seg_header_bytes);
  }
  pa.src_fill_stride = 1;
  pa.owners = 0;  // from here on this is a one-table job: the owner's
  hipLaunchKernelGGL(hdk_part_collect_flags, dim3(1), dim3(64), 0, s, pa);
  launch_part_tail(plan, pa, props, s);
  // armed like the one-GPU path's atomics kernel: only if level 2 found the inbox too skewed for fine slabs
  const unsigned go = static_cast<unsigned>(props->num_cu) * 4;
  if (plan->key_width == 4) {
    hipLaunchKernelGGL(hdk_part_owner_fallback<int32_t>, dim3(go), dim3(kPartBlock), 0, s, pa);
  } else {
    hipLaunchKernelGGL(hdk_part_owner_fallback<int64_t>, dim3(go), dim3(kPartBlock), 0, s, pa);
  }
  HDK_HIP_CHECK(hipGetLastError());
  if (hdk_sw(SW_PART_TRACE)) {  // debugging aid: where the owner's tuples are after the passes (synchronises)
    HDK_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<uint32_t> c(bc2 / 4);
    HDK_HIP_CHECK(hipMemcpy(c.data(), cur2, bc2, hipMemcpyDeviceToHost));
    uint64_t in2 = 0, over2 = 0, segs = 0, maxfill = 0;
    for (uint32_t f = 0; f < pa.fine_count; ++f) {
      in2 += std::min<uint64_t>(c[f], pa.cap2);
      over2 += c[f] > pa.cap2 ? c[f] - pa.cap2 : 0;
      maxfill = std::max<uint64_t>(maxfill, c[f]);
      segs += c[pa.fine_count + f];
    }
    fprintf(stderr, "owner: fine %u cap2 %llu | in fine slabs %llu (+%llu past cap, max cursor %llu) ovf %u/%llu seg-spill %llu list-spill %u/%llu fallback %u\n",
            pa.fine_count, (unsigned long long)pa.cap2, (unsigned long long)in2, (unsigned long long)over2, (unsigned long long)maxfill,
            c[2 * pa.fine_count], (unsigned long long)pa.cap_ovf, (unsigned long long)segs, c[2 * pa.fine_count + 1],
            (unsigned long long)pa.cap_spill, c[2 * pa.fine_count + 2]);
  }
  return HDK_HIP_OK;
}

static int32_t launch_scan_global(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan,
                                  const KernParams& kp, const LaunchShape& shape, hipStream_t s, bool force_generic,
                                  const uint32_t* run_if = nullptr) {
  BaseFastArgs fa;
  if (!run_if && !force_generic && match_baseline_fast(plan, &fa)) {
    fa.plan = d_plan;
    fa.kp = kp;
    fa.entry_count = shape.entry_count;
    launch_baseline_direct(plan, fa, shape.grid, s);
    HDK_HIP_CHECK(hipGetLastError());
    return HDK_HIP_OK;
  }
  GlobalArgs a;
  a.plan = d_plan;
  a.kp = kp;
  a.entry_count = shape.entry_count;
  a.rows_per_tile = kGlobalBlock * 4;
  a.run_if = run_if;
  hipLaunchKernelGGL(hdk_scan_agg_global, dim3(shape.grid), dim3(kGlobalBlock), 0, s, a);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

int32_t launch_scan_global_armed(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                                 const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props, hipStream_t s,
                                 const uint32_t* run_if) {
  LaunchShape shape;
  memset(&shape, 0, sizeof(shape));
  shape.strategy = STRAT_GLOBAL;
  shape.entry_count = plan->entry_count;
  shape.grid = baseline_grid(plan, ko, props);
  return launch_scan_global(plan, d_plan, kp, shape, s, true, run_if);
}

uint32_t baseline_grid(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props) {
  BaseFastArgs bf;
  const void* k;
  int block;
  if (!launch_forces_generic(ko) && match_baseline_fast(p, &bf)) {
    k = baseline_direct_kernel(p);
    block = kBaseFastBlock;
  } else {
    k = reinterpret_cast<const void*>(hdk_scan_agg_global);
    block = kGlobalBlock;
  }
  // random atomics make block run times uneven: 4 waves of blocks rebalance the tail
  // (C5 shape: 1792 blocks 21.5 ms, 3584 18.8 ms, 7168 17.6 ms)
  return resident_grid(k, block, 0, props) * 4;
}

struct PpLayout;
static bool match_perfect_partitioned(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, PpArgs* a, PpLayout* l);
static bool perfect_partitioned_takes(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko);

void baseline_describe(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko, char* out, size_t out_len) {
  BaseFastArgs fa;
  PartArgs part;
  if (const char* bh = bh_lds_kernel_name(plan, ko)) {  // a small table: open addressing in LDS (scan_bh.hip)
    snprintf(out, out_len, "%s", bh);
  } else if (!(ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS)) && match_partitioned(plan, ko, &part)) {
    snprintf(out, out_len, "hdk_part_scatter,hdk_part_scatter,hdk_part_aggregate,hdk_part_overflow,hdk_scan_agg_baseline_direct");
  } else if (perfect_partitioned_takes(plan, ko)) {
    snprintf(out, out_len, "hdk_pp_scatter,hdk_pp_scatter2,hdk_pp_aggregate,hdk_scan_agg_global");
  } else {
    snprintf(out, out_len, "%s", !launch_forces_generic(ko) && match_baseline_fast(plan, &fa) ? "hdk_scan_agg_baseline_direct"
                                                                                               : "hdk_scan_agg_global");
  }
}

// HDK_HIP_LAUNCH_INIT_OUTPUT for the strategies that do not fuse it: the init kernel, on the launch stream
static int32_t init_row_wise_output(const hdk_hip_plan* plan, const KernParams& kp, const hdk_hip_device_properties* props,
                                    hipStream_t s) {
  const uint32_t key_count = plan->keyless ? 0u : static_cast<uint32_t>(plan->key_count);
  return launch_init_row_wise_indirect(kp.groupby_buf, kp.init_agg_vals, plan->entry_count, key_count,
                                       static_cast<uint32_t>(plan->key_width), plan->row_size_quad, plan->keyless, props, s);
}


// ---- perfect-hash tables beyond LDS: entry-range partitions (scan_agg_perfect_part.h) ---------------------------------------
struct PpLayout {
  size_t off_fill1, off_fill2, off_t1, off_t2, cursor_bytes, total;
  int tw;
};

static bool match_perfect_partitioned(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, PpArgs* a, PpLayout* l) {
  if (!ko || hdk_sw(SW_NO_PERFECT_PARTITIONS)) return false;
  const bool forced = hdk_sw(SW_PERFECT_PARTITIONS_ALWAYS) != nullptr;  // (tests: small inputs)
  if (ko->flags & (HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS | HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR |
                   HDK_HIP_LAUNCH_CHECK_INTERRUPT)) {
    return false;
  }
  if (ko->watchdog_ms || ko->total_rows == 0 || (!forced && ko->total_rows < (16ull << 20))) return false;
  if (p->query_kind != HDK_Q_PERFECT_HASH || p->key_count < 1 || p->key_count > kPpMaxKeys || p->output_columnar || p->num_joins) return false;
  if (p->entry_count < 2 || static_cast<uint64_t>(p->entry_count) >= 0x7FFFFFFFull) return false;
  memset(a, 0, sizeof(*a));
  if (!match_plain_quals(p, a->q)) return false;
  a->nquals = p->num_quals;
  a->nkeys = p->key_count;
  uint64_t stride = 1;
  for (int k = 0; k < p->key_count; ++k) {
    int kc;
    if (p->key_bucket[k] > 1 || !plain_outer_col(p, p->keys[k], &kc) ||
        (p->cols[kc].kind != HDK_COL_INT && p->cols[kc].kind != HDK_COL_UNSIGNED)) {
      return false;
    }
    // (one key: the entry count is its cardinality; several: getBucketedCardinality per key, perfect_key_hash's strides)
    const uint64_t card = p->key_count == 1 ? static_cast<uint64_t>(p->entry_count) : static_cast<uint64_t>(p->key_card[k]);
    if (card == 0 || card > 0x7FFFFFFFull || stride > 0x7FFFFFFFull) return false;
    a->key[k].buf_idx = p->cols[kc].buf_idx;
    a->key[k].width = p->cols[kc].width;
    a->key[k].kind = p->cols[kc].kind;
    a->key_nullable[k] = p->keys[k].nullable;
    a->key_null[k] = p->keys[k].null_val;
    a->key_min[k] = p->key_min[k];
    a->null_has_entry[k] = p->key_has_nulls[k] && p->keys[k].nullable;
    a->key_translated[k] = p->key_null_translated[k];
    a->key_card[k] = static_cast<uint32_t>(card);
    a->key_stride[k] = static_cast<uint32_t>(stride);
    stride *= card;
  }
  a->entry_count = p->entry_count;
  a->row_bytes = static_cast<uint32_t>(p->row_size_quad) * 8;
  if (a->row_bytes == 0 || a->row_bytes > 256) return false;
  if (!p->keyless) {  // the layout's key slots: quads 0 .. keys - 1 of the row (get_group_value_fast / perfect_key_hash rows)
    for (int k = 0; k < p->key_count; ++k) {
      a->keyslot_off[a->nkeyslots] = 8 * k;
      a->keyslot_width[a->nkeyslots] = 8;
      a->keyslot_key[a->nkeyslots] = k;
      a->keyslot_translated[a->nkeyslots] = 1;
      ++a->nkeyslots;
    }
  }
  int arg_col[kPpMaxArgs] = {-1, -1};
  bool any_expr = false;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_ID) {
      if (tg.key_idx < 0 || tg.key_idx >= p->key_count) return false;
      if (tg.slot_width == 0) continue;
      if (a->nkeyslots == kPpMaxKeySlots || (tg.slot_width != 4 && tg.slot_width != 8)) return false;
      a->keyslot_off[a->nkeyslots] = tg.slot_off;
      a->keyslot_width[a->nkeyslots] = tg.slot_width;
      a->keyslot_key[a->nkeyslots] = tg.key_idx;
      a->keyslot_translated[a->nkeyslots] = 0;
      ++a->nkeyslots;
      continue;
    }
    if (tg.agg == HDK_AGG_SINGLE_VALUE || tg.arg_is_fp == HDK_FP_SLOT_FLOAT) return false;
    if ((tg.slot_width != 4 && tg.slot_width != 8) || (tg.agg == HDK_AGG_AVG && tg.slot2_width != 4 && tg.slot2_width != 8)) return false;
    PartTarget d;
    memset(&d, 0, sizeof(d));
    d.agg = tg.agg;
    d.has_arg = tg.has_arg;
    d.skip_null = tg.skip_null;
    d.arg_is_fp = tg.arg_is_fp;
    d.slot_width = tg.slot_width;
    d.slot2_width = tg.slot2_width;
    d.slot_off = tg.slot_off;
    d.slot2_off = tg.slot2_off;
    d.null_val = tg.null_val;
    d.arg_null_val = tg.arg.null_val;
    d.arg_nullable = tg.arg.nullable;
    if (tg.has_arg) {
      // a plain column, or one checked integer step over plain columns: a op b / a op literal
      const hdk_hip_expr& e = tg.arg;
      if (e.nsteps > 1 || e.leaf0.kind != HDK_LEAF_COL || p->cols[e.leaf0.col].table != 0 || p->cols[e.leaf0.col].kind == HDK_COL_SMALL_DATE) return false;
      const int c = e.leaf0.col;
      const hdk_hip_col& col = p->cols[c];
      PpArgs::ArgExpr ex;
      memset(&ex, 0, sizeof(ex));
      if (e.nsteps == 1) {
        const hdk_hip_step& sp = e.steps[0];
        if (col.kind != HDK_COL_INT || tg.arg_is_fp || sp.out_class != HDK_VC_INT) return false;
        if (sp.op != HDK_OP_ADD && sp.op != HDK_OP_SUB && sp.op != HDK_OP_MUL) return false;
        if (tg.skip_null && (e.null_val != sp.null_out || tg.null_val != sp.null_out)) return false;
        if (!tg.skip_null && (e.leaf0.nullable || sp.rhs.nullable)) return false;
        ex.op = sp.op;
        ex.check_width = sp.check_width;
        ex.a_nullable = e.leaf0.nullable;
        ex.a_null = e.leaf0.null_val;
        ex.null_out = sp.null_out;
        if (sp.rhs.kind == HDK_LEAF_INT) {
          ex.form = 2;
          ex.lit = sp.rhs.ival;
        } else if (sp.rhs.kind == HDK_LEAF_COL && p->cols[sp.rhs.col].table == 0 && p->cols[sp.rhs.col].kind == HDK_COL_INT) {
          ex.form = 1;
          ex.b.buf_idx = p->cols[sp.rhs.col].buf_idx;
          ex.b.width = p->cols[sp.rhs.col].width;
          ex.b.kind = p->cols[sp.rhs.col].kind;
          ex.b_nullable = sp.rhs.nullable;
          ex.b_null = sp.rhs.null_val;
        } else {
          return false;
        }
      }
      int w = -1;
      for (int i = 0; i < a->nargs; ++i) {
        if (arg_col[i] == c && memcmp(&a->ax[i], &ex, sizeof(ex)) == 0) w = i;
      }
      if (w < 0) {
        if (a->nargs == kPpMaxArgs) return false;
        w = a->nargs++;
        arg_col[w] = c;
        a->ax[w] = ex;
        any_expr = any_expr || ex.form != 0;
        a->arg[w].buf_idx = col.buf_idx;
        a->arg[w].width = col.width;
        a->arg[w].kind = col.kind;
      }
      d.arg_word = 1 + w;
      d.arg_fp = col.kind == HDK_COL_FLOAT || col.kind == HDK_COL_DOUBLE;
    }
    a->tg[a->ntargets++] = d;
  }
  // ---- geometry --------------------------------------------------------------------------------------------------------------
  uint32_t slice_log2 = 0;
  while ((2ull << slice_log2) * a->row_bytes <= kPpLdsBytes) ++slice_log2;
  if (const char* e = hdk_sw(SW_PERFECT_SLICE_LOG2)) {  // (tests: smaller slices; never more than LDS holds)
    const uint32_t want = static_cast<uint32_t>(atoi(e));
    if (want < slice_log2) slice_log2 = want;
  }
  if (slice_log2 < 4 || slice_log2 > 20) return false;
  a->slice_log2 = slice_log2;
  a->nslices = static_cast<uint32_t>((static_cast<uint64_t>(a->entry_count) + (1ull << slice_log2) - 1) >> slice_log2);
  uint32_t fpc_log2 = 0;
  while (((a->nslices + (1u << fpc_log2) - 1) >> fpc_log2) > static_cast<uint32_t>(kPbMaxBins)) ++fpc_log2;
  if (fpc_log2 > 8) return false;
  a->fpc_log2 = fpc_log2;
  a->two_level = fpc_log2 ? 1u : 0u;
  a->nb1 = (a->nslices + (1u << fpc_log2) - 1) >> fpc_log2;
  const uint64_t rows = ko->total_rows;
  const uint64_t nsub = static_cast<uint64_t>(a->nb1) * kPbXcds;
  a->cap1 = ((rows / nsub) * 5 / 4 + 4096 + 15) & ~15ull;
  a->cap2 = ((rows / a->nslices) * 5 / 4 + 1024 + 15) & ~15ull;
  if (a->cap1 > 0xFFFFFFF0ull || a->cap2 > 0xFFFFFFF0ull) return false;
  auto up = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
  l->tw = 1 + a->nargs;
  if (a->nargs == 1 && arg_col[0] >= 0 && !any_expr && !(ko->flags & HDK_HIP_LAUNCH_WIDE_TUPLES)) {
    const hdk_hip_col& c = p->cols[arg_col[0]];
    // (nullable: some target skips this column's NULLs with the column's own sentinel -- the leaf of any of them says so)
    int64_t nullv = 0;
    bool nullable = false;
    for (int t = 0; t < p->num_targets; ++t) {
      const hdk_hip_target& tg = p->targets[t];
      if (tg.agg != HDK_AGG_ID && tg.has_arg && tg.arg.leaf0.nullable) nullable = true, nullv = tg.arg.leaf0.null_val;
    }
    if (c.kind == HDK_COL_INT && c.has_stats && c.min_val > static_cast<int64_t>(INT32_MIN) && c.max_val <= static_cast<int64_t>(INT32_MAX) &&
        (nullable || !c.has_nulls)) {
      a->packed = 1;
      a->packed_nullable = nullable ? 1 : 0;
      a->packed_null = nullv;
      l->tw = 1;
    }
  }
  l->off_fill1 = 256;
  l->off_fill2 = l->off_fill1 + up(nsub * kPbCursorStride * 4);
  l->cursor_bytes = l->off_fill2 + (a->two_level ? up(static_cast<size_t>(a->nslices) * kPbCursor2Stride * 4) : 0);
  l->off_t1 = l->cursor_bytes;
  l->off_t2 = l->off_t1 + up(nsub * a->cap1 * l->tw * 8);
  l->total = l->off_t2 + (a->two_level ? up(static_cast<size_t>(a->nslices) * a->cap2 * l->tw * 8) : 0);
  return true;
}

static bool perfect_partitioned_takes(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko) {
  PpArgs a;
  PpLayout l;
  return match_perfect_partitioned(p, ko, &a, &l);
}

template <int TW>
static void pp_launch(const PpArgs& a, const hdk_hip_device_properties* props, hipStream_t s) {
  constexpr int VR = TW == 1 ? 8 : (TW == 2 ? 4 : 2);
  const size_t lds_sc = PbStage<TW, VR>::lds_bytes();
  const unsigned cu = static_cast<unsigned>(props->num_cu);
  unsigned per_cu = 2;
  if (const char* e = hdk_sw(SW_PP_BLOCKS_PER_CU)) per_cu = static_cast<unsigned>(atoi(e)) ? static_cast<unsigned>(atoi(e)) : 2u;
  if (a.nkeys == 1) {
    hipLaunchKernelGGL((k_pp_scatter<TW, VR, 1>), dim3(per_cu * cu), dim3(kPbBlock), lds_sc, s, a);
  } else {
    hipLaunchKernelGGL((k_pp_scatter<TW, VR, kPpMaxKeys>), dim3(per_cu * cu), dim3(kPbBlock), lds_sc, s, a);
  }
  if (a.two_level) {
    // level 2: `members2` blocks per level-1 bin, a bin's blocks congruent modulo 8 (one XCD); about three resident blocks per CU
    PpArgs a2 = a;
    const unsigned bins8 = (a.nb1 + kPbXcds - 1) / kPbXcds;
    unsigned m2 = (3 * cu) / (bins8 * kPbXcds);
    if (m2 < 1) m2 = 1;
    if (m2 > 16) m2 = 16;
    a2.members2 = m2;
    hipLaunchKernelGGL((k_pp_scatter2<TW, VR>), dim3(bins8 * kPbXcds * m2), dim3(kPbBlock), lds_sc, s, a2);
  }
  const size_t lds_b = (static_cast<size_t>(1) << a.slice_log2) * a.row_bytes;  // (granted: launch_perfect_partitioned asked)
  unsigned gb = 2 * cu;
  if (gb > a.nslices) gb = a.nslices;
  hipLaunchKernelGGL((k_pp_aggregate<TW>), dim3(gb), dim3(kPbBlock), lds_b, s, a);
}

// true: the passes (and the armed global-atomics kernel) are on the stream
static int32_t launch_perfect_partitioned(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, PpArgs& a,
                                          const PpLayout& l, const LaunchShape& shape, const hdk_hip_device_properties* props,
                                          hipStream_t s, bool* launched) {
  *launched = false;
  // the aggregate pass's LDS first: a device that does not grant it runs the plan on global atomics, not into a launch error
  {
    const size_t lds_b = (static_cast<size_t>(1) << a.slice_log2) * a.row_bytes;
    const void* ak = l.tw == 1 ? reinterpret_cast<const void*>(k_pp_aggregate<1>)
                               : (l.tw == 2 ? reinterpret_cast<const void*>(k_pp_aggregate<2>) : reinterpret_cast<const void*>(k_pp_aggregate<3>));
    if (hipFuncSetAttribute(ak, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)) != hipSuccess) {
      (void)hipGetLastError();
      return HDK_HIP_OK;
    }
  }
  AsyncScratch scratch(s);
  if (hipMallocAsync(&scratch.p, l.total, s) != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return HDK_HIP_OK;  // no room for the tuples: global atomics
  }
  int8_t* base = static_cast<int8_t*>(scratch.p);
  HDK_HIP_CHECK(hipMemsetAsync(base, 0, l.cursor_bytes, s));
  a.kp = kp;
  a.flag = reinterpret_cast<uint32_t*>(base);
  a.fill1 = reinterpret_cast<uint32_t*>(base + l.off_fill1);
  a.fill2 = reinterpret_cast<uint32_t*>(base + l.off_fill2);
  a.tuples1 = reinterpret_cast<int64_t*>(base + l.off_t1);
  a.tuples2 = reinterpret_cast<int64_t*>(base + l.off_t2);
  switch (l.tw) {
    case 1: pp_launch<1>(a, props, s); break;
    case 2: pp_launch<2>(a, props, s); break;
    default: pp_launch<3>(a, props, s); break;
  }
  HDK_HIP_CHECK(hipGetLastError());
  const int32_t st = launch_scan_global(plan, d_plan, kp, shape, s, true, a.flag);  // armed: only when a slab overflowed
  if (st) return st;
  *launched = true;
  return HDK_HIP_OK;
}

int32_t launch_baseline(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                        const hdk_hip_kernel_options* ko, const LaunchShape& shape, bool init_output,
                        const hdk_hip_device_properties* props, hipStream_t s) {
  PartArgs part;
  if (!bh_lds_kernel_name(plan, ko) && !(ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS)) && match_partitioned(plan, ko, &part)) {
    part.init_output = init_output;
    int32_t st = launch_scan_partitioned(plan, d_plan, kp, part, shape, props, s);
    if (st == kPartitionedNoScratch) {
      if (init_output) {
        st = init_row_wise_output(plan, kp, props, s);
        if (st) return st;
      }
      st = launch_scan_global(plan, d_plan, kp, shape, s, false);
    }
    return st;
  }
  if (init_output) {
    const int32_t st = init_row_wise_output(plan, kp, props, s);
    if (st) return st;
  }
  {
    bool launched = false;
    int32_t st = launch_baseline_sliced_join(plan, d_plan, kp, ko, props, s, &launched);
    if (st || launched) return st;
    st = launch_bh_lds(plan, d_plan, kp, ko, props, s, &launched);
    if (st || launched) return st;
  }
  PpArgs pp;
  PpLayout pl;
  if (match_perfect_partitioned(plan, ko, &pp, &pl)) {
    bool launched = false;
    const int32_t st = launch_perfect_partitioned(plan, d_plan, kp, pp, pl, shape, props, s, &launched);
    if (st || launched) return st;
  }
  return launch_scan_global(plan, d_plan, kp, shape, s, launch_forces_generic(ko));
}

}  // namespace hdk

using namespace hdk;

// ---- multi-GPU tuple exchange: the C ABI (include/hdk_hip.h) ---------------------------------------------------------
extern "C" int32_t hdk_hip_exchange_shape_for(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko,
                                              int32_t num_owners, uint32_t owner_entry_count, int32_t device_id,
                                              hdk_hip_exchange_shape* shape) {
  const int32_t st = validate_plan(plan);
  if (st) return st;
  HDK_REQUIRE(shape, "shape is NULL");
  (void)device_id;  // (the geometry depends on the LDS budget of gfx950 only; kept for per-device tuning)
  PartArgs pa;
  return exchange_shape(plan, ko, num_owners, owner_entry_count, &pa, shape);
}

static int32_t exchange_enter(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT], const hdk_hip_exchange_shape* shape,
                              const void* buf, void* workspace, size_t workspace_bytes, size_t need, int32_t device_id,
                              void* stream, hipStream_t* s) {
  const int32_t st = validate_plan(plan);
  if (st) return st;
  HDK_REQUIRE(params && shape && buf, "NULL argument");
  HDK_REQUIRE(params[HDK_KP_ERROR_CODE], "ERROR_CODE is NULL");
  HDK_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 255) == 0, "exchange buffers must be 256-byte aligned");
  HDK_REQUIRE(workspace && workspace_bytes >= need, "workspace too small: %zu < %zu", workspace_bytes, need);
  HDK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  return device_enter(device_id, stream, s);
}

extern "C" int32_t hdk_hip_scatter_to_owners(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT],
                                             const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape,
                                             int8_t* send, int32_t device_id, void* stream, void* workspace,
                                             size_t workspace_bytes) {
  hipStream_t s;
  int32_t st = exchange_enter(plan, params, shape, send, workspace, workspace_bytes,
                              shape ? static_cast<size_t>(shape->scatter_workspace_bytes) : 0, device_id, stream, &s);
  if (st) return st;
  HDK_REQUIRE(params[HDK_KP_COL_BUFFERS] && params[HDK_KP_NUM_FRAGMENTS] && params[HDK_KP_NUM_ROWS] && params[HDK_KP_NUM_TABLES],
              "a required kernel parameter is NULL");
  hdk_hip_plan* d_plan = nullptr;
  KernParams kp;
  st = launch_head(plan, params, ko, device_id, workspace, s, &d_plan, &kp);
  if (st) return st;
  const bool timed = ko && (ko->flags & HDK_HIP_LAUNCH_RECORD_EVENTS);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    st = scan_events_begin(device_id, s, &e0, &e1);
    if (st) return st;
  }
  st = launch_scatter_to_owners(plan, d_plan, kp, ko, shape, send, static_cast<int8_t*>(workspace) + kPlanRegionBytes,
                                device_props(device_id), s);
  if (st) return st;
  if (timed) {
    HDK_HIP_CHECK(hipEventRecord(e1, s));
  }
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_aggregate_from_ranks(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT],
                                                const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape,
                                                const int8_t* recv, int32_t device_id, void* stream, void* workspace,
                                                size_t workspace_bytes) {
  hipStream_t s;
  int32_t st = exchange_enter(plan, params, shape, recv, workspace, workspace_bytes,
                              shape ? static_cast<size_t>(shape->aggregate_workspace_bytes) : 0, device_id, stream, &s);
  if (st) return st;
  HDK_REQUIRE(params[HDK_KP_GROUPBY_BUF] && params[HDK_KP_INIT_AGG_VALS], "GROUPBY_BUF / INIT_AGG_VALS is NULL");
  hdk_hip_plan* d_plan = nullptr;
  KernParams kp;
  st = launch_head(plan, params, ko, device_id, workspace, s, &d_plan, &kp);
  if (st) return st;
  const bool timed = ko && (ko->flags & HDK_HIP_LAUNCH_RECORD_EVENTS);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    st = scan_events_begin(device_id, s, &e0, &e1);
    if (st) return st;
  }
  st = launch_aggregate_from_ranks(plan, d_plan, kp, ko, shape, recv, static_cast<int8_t*>(workspace) + kPlanRegionBytes,
                                   device_props(device_id), s);
  if (st) return st;
  if (timed) {
    HDK_HIP_CHECK(hipEventRecord(e1, s));
  }
  return HDK_HIP_OK;
}
