// scan_bh_fast.h -- the streaming form of the LDS open-addressing group-by (scan_bh.h), for the shape of the reference's
// BaselineHash benchmark queries (Benchmarks/synthetic_benchmark/queries/BaselineHash/BH001-005.sql):
//     SELECT cast(x AS double) AS key0, count(y), sum(y), max(y), min(y), avg(y) FROM t GROUP BY key0
// one group key that is a plain integer column of the outer table, as it is or cast to double (the key word is the double's
// bit pattern, groupByColumnCodegen, QE/IRCodegen.cpp:1219-1221), every aggregate over ONE plain column (or COUNT(*)),
// filters of the form `column cmp literal`.  Loads are scan_agg_fast.h's: 16 bytes of the widest column per lane and step,
// non-temporal, U steps in flight.  Per row: the key word, one ds_read (ds_cmpswap the first time) to find its entry in the
// lane's replica, then ONE LDS atomic per distinct update -- rows, sum, min, max of the argument (NULLs only when one turns
// up): the five aggregates of BH001 are four atomics, the layout's nine words are mapped onto them at the flush
// (BhLdsLayout::lmap).
#pragma once
#include <type_traits>
#include "watch.h"
#include "plain_quals.h"
#include "scan_agg_fast.h"
#include "scan_bh.h"

namespace hdk {

struct BhFastArgs {
  const hdk_hip_plan* plan;  // device copy (the flush reads the targets)
  KernParams kp;
  BhGeom g;
  BhLdsLayout ll;
  int32_t key_buf_idx, val_buf_idx;
  int32_t key_form;      // 0: the column's value; 1: cast(integer AS double)
  int32_t key_nullable;  // key_form 1: the column's NULL becomes the cast's NULL (cast_int64_t_to_double_nullable)
  int64_t key_null, key_null_out;
  int32_t val_nullable, val_is_fp;
  int64_t val_null;
  int32_t lw_nulls, lw_sum, lw_min, lw_max;  // LDS word of each update of the argument, -1: none (LDS word 0 = rows)
  int32_t nquals;
  int32_t pad_;
  int64_t* slabs;  // two-level fold (scan_bh.h: hdk_bh_fold_words): block b leaves its table in slab b; nullptr: folds itself
  ProjFastQual q[kMaxPlainQuals];
};

HDK_DEV void bh_fast_row(const BhFastArgs& a, int64_t* lds_rep, uint32_t estride, uint32_t key_off, uint32_t rep, int64_t key,
                         int64_t val, bool has_val, int32_t& err) {
  int64_t kw = key;
  if (a.key_form == 1) {
    kw = (a.key_nullable && key == a.key_null) ? a.key_null_out : double_to_bits(static_cast<double>(key));
  }
  const int32_t e = bh_lds_find_or_claim(lds_rep, kw, a.g.cap_log2, estride, key_off);
  if (e < 0) {
    err = HDK_HIP_ERR_OUT_OF_SLOTS;  // more groups than the plan's table holds
    return;
  }
  int64_t* base = lds_rep + __umul24(static_cast<uint32_t>(e), estride);
  atomicAdd(reinterpret_cast<unsigned long long*>(base), 1ull);  // rows of the group
  if (!has_val) {
    return;
  }
  const bool fp = a.val_is_fp != 0;
  if (a.val_nullable && (fp ? bits_to_double(val) == bits_to_double(a.val_null) : val == a.val_null)) {
    atomicAdd(reinterpret_cast<unsigned long long*>(base + a.lw_nulls * rep), 1ull);
    return;
  }
  if (a.lw_sum >= 0) {
    if (fp) {
      atomicAdd(reinterpret_cast<double*>(base + a.lw_sum * rep), bits_to_double(val));
    } else {
      atomicAdd(reinterpret_cast<unsigned long long*>(base + a.lw_sum * rep), static_cast<unsigned long long>(val));
    }
  }
  if (a.lw_min >= 0) {
    if (fp) {
      fast_lds_op(FOP_MIN_F64, base + a.lw_min * rep, val);
    } else {
      atomicMin(reinterpret_cast<long long*>(base + a.lw_min * rep), static_cast<long long>(val));
    }
  }
  if (a.lw_max >= 0) {
    if (fp) {
      fast_lds_op(FOP_MAX_F64, base + a.lw_max * rep, val);
    } else {
      atomicMax(reinterpret_cast<long long*>(base + a.lw_max * rep), static_cast<long long>(val));
    }
  }
}

// KW / VW: byte width of the key / argument column (VW 0: no argument column); U steps of 16 bytes per lane and tile
template <int KW, int VW, int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hdk_scan_agg_bh_direct(BhFastArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  __shared__ WordLayout wl;
  __shared__ BhLdsLayout s_ll;
  __shared__ uint64_t s_col_off[2 * HDK_HIP_MAX_TARGETS];
  constexpr int WMAX = KW > VW ? KW : VW;
  constexpr int R = 16 / WMAX;
  constexpr int KB = KW * R;
  constexpr int VB = VW * R;
  constexpr int KREGS = KB >= 4 ? KB / 4 : 1;
  constexpr int VREGS = VB >= 4 ? VB / 4 : 1;
  const int tid = threadIdx.x;
  if (tid == 0) {
    make_word_layout(a.plan, &wl);
    s_ll = a.ll;
  }
  __syncthreads();
  const uint32_t rep = a.g.rep;
  bh_lds_init(lds, s_ll, a.g.cap_log2, rep, tid, BLOCK);
  __syncthreads();
  const uint32_t W = static_cast<uint32_t>(a.ll.nlw) + 1;
  const uint32_t estride = W * rep;
  const uint32_t key_off = static_cast<uint32_t>(a.ll.nlw) * rep;
  int64_t* lds_rep = lds + (tid & (rep - 1));
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(BLOCK) * R * U;
  int32_t err = 0;
  const Watch watch = watch_begin(a.kp);
  const bool filtered = a.nquals != 0;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    const gcol_t kcol = (gcol_t)cols[a.key_buf_idx];
    const gcol_t vcol = VW ? (gcol_t)cols[a.val_buf_idx] : nullptr;
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      if (row0 + kTileRows <= nrows) {
        uint32_t kr[U][KREGS];
        uint32_t vr[U][VREGS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R;
          load_bytes<KB, true>(kcol + r * KW, kr[u]);
          if (VW) load_bytes<(VB > 0 ? VB : 4), true>(vcol + r * VW, vr[u]);
        }
        bool pass[U * R];
#pragma unroll
        for (int j = 0; j < U * R; ++j) {
          pass[j] = true;
        }
        if (filtered) {
          int64_t rows[U * R];
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
              rows[u * R + i] = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R + i;
            }
          }
          plain_quals_pass<U * R, true>(a.q, a.nquals, cols, rows, pass, true);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            if (pass[u * R + i]) {
              const int64_t key = extract_elem<KW>(kr[u], i);
              const int64_t val = VW ? extract_elem<(VW ? VW : 8)>(vr[u], i) : 0;
              bh_fast_row(a, lds_rep, estride, key_off, rep, key, val, VW != 0, err);
            }
          }
        }
      } else {
        for (int64_t r = row0 + tid; r < nrows; r += BLOCK) {
          if (filtered) {
            const int64_t rows1[1] = {r};
            bool pass1[1] = {true};
            plain_quals_pass<1, true>(a.q, a.nquals, cols, rows1, pass1, true);
            if (!pass1[0]) {
              continue;
            }
          }
          const int64_t key = load_elem<KW>(kcol, r);
          const int64_t val = VW ? load_elem<(VW ? VW : 8)>(vcol, r) : 0;
          bh_fast_row(a, lds_rep, estride, key_off, rep, key, val, VW != 0, err);
        }
      }
    }
    frag_tile_begin += ntiles;
  }
  int64_t* slab = a.slabs ? a.slabs + (static_cast<size_t>(blockIdx.x) * (static_cast<uint32_t>(a.ll.nlw) + 1) << a.g.cap_log2) : nullptr;
  bh_flush_block<BLOCK>(a.plan, wl, s_ll, lds, a.g, a.kp.groupby_buf, s_col_off, tid, err, slab);
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
