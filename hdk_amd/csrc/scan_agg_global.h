// scan_agg_global.h -- scan/aggregate straight into the output buffer with global atomics.
//
// For GroupByBaselineHash (open addressing, any cardinality -- BASELINE C5) and for perfect-hash
// tables too large for LDS.  This is the closest relative of the reference's GPU path (row function
// + agg_*_shared, QE/cuda_mapd_rt.cu:167-261,424-478,886-957): one claim/lookup and one atomic per
// target per row, on the final table.  Differences: 64-wide waves, the plan interpreter instead of
// JIT'ed code, native global_atomic_add_f64 / 64-bit min-max instead of CAS loops, and a
// "CAS-the-sentinel-then-add" form for *_skip_val that needs no atomicExch spin.
#pragma once
#include "watch.h"
#include "agg_common.h"
#include "baseline_table.h"

namespace hdk {

constexpr int kGlobalBlock = 256;

struct GlobalArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  uint32_t entry_count;
  uint32_t rows_per_tile;
  const uint32_t* run_if;  // nullptr: always; else only when *run_if != 0 (armed behind the partitioned perfect-hash passes)
};

HDK_DEV void g_store_i64(int64_t* p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
HDK_DEV void g_store_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// agg_sum[_skip_val]_shared and friends on an 8-byte slot
HDK_DEV void g_agg64(int agg, bool fp, bool skip, int64_t nullv, int64_t* slot, int64_t v) {
  unsigned long long* us = reinterpret_cast<unsigned long long*>(slot);
  if (skip) {
    // the slot starts at the NULL sentinel: the first non-NULL value replaces it
    if (atomic_load_i64(slot) == nullv) {
      const unsigned long long prev =
          atomicCAS(us, static_cast<unsigned long long>(nullv), static_cast<unsigned long long>(v));
      if (prev == static_cast<unsigned long long>(nullv)) {
        return;
      }
    }
  }
  if (fp) {
    const double d = bits_to_double(v);
    if (agg == HDK_AGG_MIN || agg == HDK_AGG_MAX) {
      unsigned long long old = static_cast<unsigned long long>(atomic_load_i64(slot));
      while (agg == HDK_AGG_MIN ? (d < bits_to_double(static_cast<int64_t>(old)))
                                : (bits_to_double(static_cast<int64_t>(old)) < d)) {
        const unsigned long long assumed = old;
        old = atomicCAS(us, assumed, static_cast<unsigned long long>(v));
        if (old == assumed) break;
      }
    } else {
      atomicAdd(reinterpret_cast<double*>(slot), d);
    }
    return;
  }
  if (agg == HDK_AGG_MIN) {
    atomicMin(reinterpret_cast<long long*>(slot), static_cast<long long>(v));
  } else if (agg == HDK_AGG_MAX) {
    atomicMax(reinterpret_cast<long long*>(slot), static_cast<long long>(v));
  } else {
    atomicAdd(us, static_cast<unsigned long long>(v));
  }
}

// float accumulators (takes_float_argument, Shared/TargetInfo.h:170-179): the slot's low 4 bytes hold a float whatever
// the padded slot width (agg_{sum,min,max}_float[_skip_val]_shared, QE/cuda_mapd_rt.cu:448-480,666-680,981-1030)
HDK_DEV void g_aggf32(int agg, bool skip, int32_t nullbits, int32_t* slot, float v) {
  const int32_t vbits = __float_as_int(v);
  if (skip) {
    if (atomic_load_i32(slot) == nullbits) {
      if (atomicCAS(slot, nullbits, vbits) == nullbits) {
        return;
      }
    }
  }
  if (agg == HDK_AGG_MIN || agg == HDK_AGG_MAX) {
    int32_t old = atomic_load_i32(slot);
    while (agg == HDK_AGG_MIN ? (v < __int_as_float(old)) : (__int_as_float(old) < v)) {
      const int32_t assumed = old;
      old = atomicCAS(slot, assumed, vbits);
      if (old == assumed) break;
    }
  } else {
    atomicAdd(reinterpret_cast<float*>(slot), v);
  }
}

// 4-byte slots: integers
HDK_DEV void g_agg32(int agg, bool skip, int32_t nullv, int32_t* slot, int32_t v) {
  if (skip) {
    if (atomic_load_i32(slot) == nullv) {
      const int prev = atomicCAS(slot, nullv, v);
      if (prev == nullv) {
        return;
      }
    }
  }
  if (agg == HDK_AGG_MIN) {
    atomicMin(slot, v);
  } else if (agg == HDK_AGG_MAX) {
    atomicMax(slot, v);
  } else {
    atomicAdd(slot, v);
  }
}

// checked_single_agg_id[_int32|_double|_float]_shared (QE/cuda_mapd_rt.cu:670-782): the slot moves from the NULL pattern
// to the first value once; true = a different value is already there (ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES)
HDK_DEV bool g_single64(int64_t* slot, int64_t vbits, int64_t nullbits) {
  unsigned long long* us = reinterpret_cast<unsigned long long*>(slot);
  unsigned long long old = static_cast<unsigned long long>(atomic_load_i64(slot));
  for (;;) {
    if (static_cast<int64_t>(old) != nullbits) {
      return static_cast<int64_t>(old) != vbits;
    }
    const unsigned long long assumed = old;
    old = atomicCAS(us, assumed, static_cast<unsigned long long>(vbits));
    if (old == assumed) {
      return false;
    }
  }
}
HDK_DEV bool g_single32(int32_t* slot, int32_t vbits, int32_t nullbits) {
  int32_t old = atomic_load_i32(slot);
  for (;;) {
    if (old != nullbits) {
      return old != vbits;
    }
    const int32_t assumed = old;
    old = atomicCAS(slot, assumed, vbits);
    if (old == assumed) {
      return false;
    }
  }
}
// the NULL test is by value (`val == null_val`), the slot compare by bit pattern
HDK_DEV bool g_single_value(const hdk_hip_target& tg, int8_t* slot, int64_t v) {
  if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) {
    const float f = static_cast<float>(bits_to_double(v)), fn = static_cast<float>(bits_to_double(tg.null_val));
    return f == fn ? false : g_single32(reinterpret_cast<int32_t*>(slot), __float_as_int(f), __float_as_int(fn));
  }
  if (tg.arg_is_fp) {
    return bits_to_double(v) == bits_to_double(tg.null_val) ? false : g_single64(reinterpret_cast<int64_t*>(slot), v, tg.null_val);
  }
  if (v == tg.null_val) {
    return false;
  }
  return tg.slot_width == 4 ? g_single32(reinterpret_cast<int32_t*>(slot), static_cast<int32_t>(v), static_cast<int32_t>(tg.null_val))
                            : g_single64(reinterpret_cast<int64_t*>(slot), v, tg.null_val);
}

HDK_DEV void g_count(int8_t* slot, int width) {
  if (width == 4) {
    atomicAdd(reinterpret_cast<unsigned int*>(slot), 1u);
  } else {
    atomicAdd(reinterpret_cast<unsigned long long*>(slot), 1ull);
  }
}

// (the atomics above are shared with scan_bh.h; the kernel itself belongs to ONE translation unit, scan_baseline.hip)
#ifdef HDK_SCAN_AGG_GLOBAL_KERNEL
extern "C" __global__ __launch_bounds__(kGlobalBlock) void hdk_scan_agg_global(GlobalArgs a) {
  __shared__ uint64_t s_col_off[2 * HDK_HIP_MAX_TARGETS];  // columnar slot-column offsets
  const hdk_hip_plan* __restrict__ p = a.plan;
  const int tid = threadIdx.x;
  if (a.run_if && *a.run_if == 0) {
    return;
  }
  const bool columnar = p->output_columnar;
  if (columnar && tid < 2 * HDK_HIP_MAX_TARGETS) {
    s_col_off[tid] = columnar_slot_off(p, a.entry_count, tid);
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const int64_t tile_rows = a.rows_per_tile;
  const int nt = p->num_targets;
  const int nk = p->key_count;
  const bool baseline = p->query_kind == HDK_Q_BASELINE_HASH;
  const TableShape shape = table_shape(p);
  int64_t* buf = a.kp.groupby_buf[0];

  RowCtx c;
  c.plan = p;
  c.join_row[0] = 0;
  c.join_row[1] = 0;
  int32_t err = 0;

  int64_t tile = blockIdx.x;
  const Watch watch = watch_begin(a.kp);
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + tile_rows - 1) / tile_rows;
    c.cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * tile_rows;
      const int64_t row_end = min(row0 + tile_rows, nrows);
      for (int64_t row = row0 + tid; row < row_end; row += kGlobalBlock) {
        c.pos = row;
        for_each_row_match(c, a.kp.join_hash_tables, err, [&]() {
        // ---- keys -> entry --------------------------------------------------------------------
        int64_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        if (nk > 0) k0 = eval_key(c, 0, err);
        if (nk > 1) k1 = eval_key(c, 1, err);
        if (nk > 2) k2 = eval_key(c, 2, err);
        if (nk > 3) k3 = eval_key(c, 3, err);
        int64_t entry;
        bool fresh = true;  // baseline: this row created its group (perfect hash: every row writes)
        if (baseline) {
          if (p->key_width == 4) {
            const int32_t key[HDK_HIP_MAX_KEYS] = {static_cast<int32_t>(k0), static_cast<int32_t>(k1),
                                                   static_cast<int32_t>(k2), static_cast<int32_t>(k3)};
            entry = find_or_claim<int32_t>(shape, buf, a.entry_count, key, &fresh);
          } else {
            const int64_t key[HDK_HIP_MAX_KEYS] = {k0, k1, k2, k3};
            entry = find_or_claim<int64_t>(shape, buf, a.entry_count, key, &fresh);
          }
          if (entry < 0) {
            err = HDK_HIP_ERR_OUT_OF_SLOTS;  // get_group_value returned NULL
            return;
          }
        } else {
          // perfect hash: stride walk as perfect_key_hash does
          int64_t h = 0, stride = 1;
          const int64_t kv[HDK_HIP_MAX_KEYS] = {k0, k1, k2, k3};
#pragma unroll
          for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
            if (k < nk) {
              int64_t term = kv[k] - p->key_min[k];
              if (p->key_bucket[k]) term /= p->key_bucket[k];
              h += term * stride;
              stride *= p->key_card[k];
            }
          }
          if (static_cast<uint64_t>(h) >= a.entry_count) {
            err = HDK_HIP_ERR_OUT_OF_SLOTS;
            return;
          }
          entry = h;
          if (!p->keyless) {  // every writer stores the same key values: plain publication is enough
            if (columnar) {
#pragma unroll
              for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
                if (k < nk) g_store_i64(buf + static_cast<size_t>(k) * a.entry_count + entry, kv[k]);
              }
            } else {
              int64_t* rowp = buf + static_cast<size_t>(entry) * p->row_size_quad;
#pragma unroll
              for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
                if (k < nk) g_store_i64(rowp + k, kv[k]);
              }
            }
          }
        }
        // ---- aggregates ---------------------------------------------------------------------------
        int8_t* rowb = reinterpret_cast<int8_t*>(buf + static_cast<size_t>(entry) * p->row_size_quad);
        int slot_idx = 0;
        for (int t = 0; t < nt; ++t) {
          const hdk_hip_target& tg = p->targets[t];
          int8_t* s1;
          int8_t* s2 = nullptr;
          if (columnar) {
            s1 = reinterpret_cast<int8_t*>(buf) + s_col_off[slot_idx] + static_cast<size_t>(entry) * tg.slot_width;
            if (tg.agg == HDK_AGG_AVG) {
              s2 = reinterpret_cast<int8_t*>(buf) + s_col_off[slot_idx + 1] + static_cast<size_t>(entry) * tg.slot2_width;
            }
          } else {
            s1 = rowb + tg.slot_off;
            s2 = rowb + tg.slot2_off;
          }
          slot_idx += tg.agg == HDK_AGG_AVG ? 2 : 1;
          // agg_id stores the same value for every row of a group: in a baseline table the row that
          // created the group does it once (one write less per row on the group's cache line)
          if (tg.agg == HDK_AGG_ID && (!fresh || tg.slot_width == 0)) {
            continue;  // (a baseline table keeps no slot for a projected key at all)
          }
          bool is_null;
          const int64_t v = eval_target_arg(c, tg, is_null, err);
          if (tg.agg == HDK_AGG_ID) {
            if (tg.slot_width == 4) {
              g_store_i32(reinterpret_cast<int32_t*>(s1), static_cast<int32_t>(v));
            } else {
              g_store_i64(reinterpret_cast<int64_t*>(s1), v);
            }
            continue;
          }
          if (tg.agg == HDK_AGG_SINGLE_VALUE) {
            if (g_single_value(tg, s1, v)) {
              err = HDK_HIP_ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES;
            }
            continue;
          }
          if (is_null) {
            continue;
          }
          if (tg.agg == HDK_AGG_COUNT) {
            g_count(s1, tg.slot_width);
            continue;
          }
          if (tg.agg == HDK_AGG_AVG) {
            g_count(s2, tg.slot2_width);
          }
          if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) {
            g_aggf32(tg.agg, tg.skip_null, float_slot_null(tg), reinterpret_cast<int32_t*>(s1),
                     static_cast<float>(bits_to_double(v)));
          } else if (tg.slot_width == 4) {
            g_agg32(tg.agg, tg.skip_null, static_cast<int32_t>(tg.null_val), reinterpret_cast<int32_t*>(s1),
                    static_cast<int32_t>(v));
          } else {
            g_agg64(tg.agg, tg.arg_is_fp, tg.skip_null, tg.null_val, reinterpret_cast<int64_t*>(s1), v);
          }
        }
        });
      }
    }
    frag_tile_begin += ntiles;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

#endif  // HDK_SCAN_AGG_GLOBAL_KERNEL

}  // namespace hdk
