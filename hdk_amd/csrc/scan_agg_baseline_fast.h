// scan_agg_baseline_fast.h -- open-addressing group-by, specialised for the common shape
// (BASELINE C5): GroupByBaselineHash, row-wise, ONE group key that is a plain integer column of the
// outer table, filters of the form `column cmp literal` only, no join, targets = projected key / COUNT / SUM /
// MIN / MAX / AVG over plain
// outer columns.  Same table, same claim protocol, same agg_* semantics as hdk_scan_agg_global
// (reference get_group_value + agg_*_shared, QE/cuda_mapd_rt.cu:167-261,424-478); what changes is
// the shape of the memory traffic, priced against scripts/microbench/atomics.hip:
//   * VR rows per lane per step: key loads, hashes, home-entry loads and argument loads of VR rows
//     are issued back to back, so a lane has VR independent table accesses in flight instead of a
//     chain column -> hash -> key -> slot -> atomic per row;
//   * the home entry is read ONCE, speculatively and cacheably: 16-byte entries ([key | one 8-byte
//     slot]) come in with one global_load_dwordx4 that delivers both the key to compare and the slot
//     value for the NULL-sentinel check -- a found group then costs one load and one atomic, the
//     measured floor for this access pattern.  A stale line can only show EMPTY or the sentinel for a
//     slot that has since moved on; both fall into the atomic slow path (CAS), which is exact;
//   * everything else (probe collisions, fresh claims, wide rows) takes find_or_claim / g_agg* as is.
#pragma once
#include "watch.h"
#include "plain_quals.h"
#include "scan_agg_global.h"

namespace hdk {

constexpr int kBaseFastBlock = 256;
constexpr int kBaseFastVR = 4;

struct BaseFastTarget {
  int32_t buf_idx;  // argument column (outer table) or -1
  int32_t width;
  int32_t kind;     // hdk_hip_col_kind
  int32_t target;   // index into plan->targets
};

struct BaseFastArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  uint32_t entry_count;
  int32_t key_buf_idx;
  int32_t key_width;   // of the input column
  int32_t key_kind;
  int32_t nkeys;       // 1 or 2 group keys (plain outer columns)
  int32_t key2_buf_idx, key2_width, key2_kind;
  int32_t ntargets;
  BaseFastTarget tg[HDK_HIP_MAX_TARGETS];
  const uint32_t* run_if;  // nullptr: always run; else run only when *run_if != 0 (fallback of the partitioned path)
  int32_t nquals;          // plain filters `outer column cmp literal` (plain_quals.h)
  ProjFastQual q[kMaxPlainQuals];
};


// g_agg64 with the slot's value already observed (`seen`): skips the pre-check load
HDK_DEV void g_agg64_seen(int agg, bool fp, bool skip, int64_t nullv, int64_t* slot, int64_t v, int64_t seen) {
  unsigned long long* us = reinterpret_cast<unsigned long long*>(slot);
  if (skip && seen == nullv) {
    const unsigned long long prev = atomicCAS(us, static_cast<unsigned long long>(nullv), static_cast<unsigned long long>(v));
    if (prev == static_cast<unsigned long long>(nullv)) {
      return;
    }
  }
  g_agg64(agg, fp, false, nullv, slot, v);
}

template <typename K, int NK>  // key type of the TABLE (int32_t or int64_t); number of key columns (1 or 2)
__global__ __launch_bounds__(kBaseFastBlock) void hdk_scan_agg_baseline_direct(BaseFastArgs a) {
  if (a.run_if && *a.run_if != 1) {
    return;  // (armed fallback of the partitioned path: 1 = take over; 2 = the launch was interrupted)
  }
  const hdk_hip_plan* __restrict__ p = a.plan;
  const int tid = threadIdx.x;
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  int64_t* buf = a.kp.groupby_buf[0];
  const TableShape shape = table_shape(p);  // scalar once: four inlined find_or_claim copies follow each other
  const uint32_t rq = shape.row_quads;
  const bool entry16 = rq == 2;  // [key region 8 B | one 8-byte slot]
  const int nt = a.ntargets;
  constexpr int VR = kBaseFastVR;
  constexpr int64_t kTileRows = static_cast<int64_t>(kBaseFastBlock) * VR;
  int32_t err = 0;

  int64_t tile = blockIdx.x;
  const Watch watch = watch_begin(a.kp);
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    const int8_t* keybuf = cols[a.key_buf_idx];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      int64_t row[VR];
      bool live[VR];
      K key[VR][NK];
      uint32_t home[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const int64_t rr = row0 + static_cast<int64_t>(r) * kBaseFastBlock + tid;
        live[r] = rr < nrows;
        row[r] = live[r] ? rr : row0;
      }
      if (a.nquals) {
        plain_quals_pass<VR>(a.q, a.nquals, cols, row, live, true);
      }
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        key[r][0] = static_cast<K>(decode_col_g(keybuf, a.key_width, a.key_kind, row[r], true));
      }
      if (NK == 2) {
        const int8_t* key2buf = cols[a.key2_buf_idx];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          key[r][NK - 1] = static_cast<K>(decode_col_g(key2buf, a.key2_width, a.key2_kind, row[r], true));
        }
      }
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        home[r] = key_hash_dev<K>(key[r], NK) % a.entry_count;
      }
      // speculative read of the home entries (16 bytes each, all VR in flight): the whole key region and,
      // for 16-byte entries, the one slot
      constexpr bool kKeys16 = sizeof(K) == 8 && NK == 2;  // two 8-byte keys: the key region alone is 16 bytes
      bf_i64x2 e[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const int64_t* ep = buf + static_cast<size_t>(home[r]) * rq;
        if (entry16) {
          e[r] = *reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(reinterpret_cast<uintptr_t>(ep));
        } else {
          e[r].x = *reinterpret_cast<const __attribute__((address_space(1))) long long*>(reinterpret_cast<uintptr_t>(ep));
          e[r].y = kKeys16 ? *reinterpret_cast<const __attribute__((address_space(1))) long long*>(reinterpret_cast<uintptr_t>(ep + 1)) : 0;
        }
      }
      int64_t entry[VR];
      bool fresh[VR];
      bool have_slot[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        bool resident_is_mine;
        if (sizeof(K) == 8) {
          resident_is_mine = static_cast<K>(e[r].x) == key[r][0] && (NK == 1 || static_cast<K>(e[r].y) == key[r][NK - 1]);
        } else if (NK == 1) {
          resident_is_mine = static_cast<K>(static_cast<int32_t>(e[r].x)) == key[r][0];
        } else {  // two 4-byte keys share the first quad: low half = key 0, high half = key 1
          resident_is_mine = static_cast<K>(static_cast<int32_t>(e[r].x)) == key[r][0] &&
                             static_cast<K>(static_cast<int32_t>(e[r].x >> 32)) == key[r][NK - 1];
        }
        fresh[r] = false;
        have_slot[r] = false;
        entry[r] = -1;
        if (!live[r]) {
          continue;
        }
        if (resident_is_mine && key[r][0] != empty_key<K>()) {
          entry[r] = home[r];
          have_slot[r] = entry16 && !kKeys16;
        } else {
          entry[r] = find_or_claim<K>(shape, buf, a.entry_count, key[r], &fresh[r]);
          if (entry[r] < 0) {
            err = HDK_HIP_ERR_OUT_OF_SLOTS;
            live[r] = false;
          }
        }
      }
      // aggregates: argument loads of the VR rows first, then the atomics
      for (int t = 0; t < nt; ++t) {
        const BaseFastTarget ft = a.tg[t];
        const hdk_hip_target& tg = p->targets[ft.target];
        int64_t v[VR];
        if (ft.buf_idx >= 0) {
          const int8_t* ab = cols[ft.buf_idx];
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            v[r] = decode_col_g(ab, ft.width, ft.kind, row[r], true);
          }
        } else {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            v[r] = 0;
          }
        }
        const bool arg_fp = ft.kind == HDK_COL_FLOAT || ft.kind == HDK_COL_DOUBLE;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (!live[r]) {
            continue;
          }
          int8_t* rowb = reinterpret_cast<int8_t*>(buf + static_cast<size_t>(entry[r]) * rq);
          int8_t* s1 = rowb + tg.slot_off;
          int8_t* s2 = rowb + tg.slot2_off;
          // eval_target_arg for a plain column argument (device_common.h)
          bool is_null = false;
          if (tg.has_arg) {
            if (tg.skip_null && is_null_val(v[r], tg.arg.null_val, tg.arg.nullable, arg_fp)) {
              is_null = true;
            } else {
              if (tg.arg_is_fp && !arg_fp) {
                v[r] = double_to_bits(static_cast<double>(v[r]));
              }
              if (tg.skip_null) {
                is_null = tg.arg_is_fp ? (bits_to_double(v[r]) == bits_to_double(tg.null_val)) : (v[r] == tg.null_val);
              }
            }
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
          if (tg.slot_width == 4) {
            g_agg32(tg.agg, tg.skip_null, static_cast<int32_t>(tg.null_val), reinterpret_cast<int32_t*>(s1),
                    static_cast<int32_t>(v[r]));
          } else if (have_slot[r]) {
            g_agg64_seen(tg.agg, tg.arg_is_fp, tg.skip_null, tg.null_val, reinterpret_cast<int64_t*>(s1), v[r], e[r].y);
          } else {
            g_agg64(tg.agg, tg.arg_is_fp, tg.skip_null, tg.null_val, reinterpret_cast<int64_t*>(s1), v[r]);
          }
        }
      }
    }
    frag_tile_begin += ntiles;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
