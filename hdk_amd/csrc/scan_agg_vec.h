// scan_agg_vec.h -- the general LDS-privatised scan/aggregate kernel: the plan interpreter run on
// batches of VR rows per lane (vec_eval.h).  Handles every NonGroupedAggregate / GroupByPerfectHash
// plan the library accepts: filters, perfect-hash join probes, expression keys and arguments,
// multi-column perfect hash (taxi Q3/Q4, BASELINE C3).  Same LDS table, slab and finalize protocol as
// the other scan kernels (agg_common.h).
#pragma once
#include "watch.h"
#include "agg_common.h"
#include "vec_eval.h"
#include "scan_bh_decl.h"

namespace hdk {

constexpr int kVecBlock = 256;

struct VecArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  int64_t* slabs;
  uint32_t entry_count;
  uint32_t rep;
  const uint32_t* run_if;  // nullptr: always run; else only when *run_if == 1 (armed behind the sliced join, scan_join_sliced2.h)
  // BH instantiations (scan_bh.hip): the group table is an open-addressing table in LDS (scan_bh.h); `entry_count` is the
  // output table's, `rep` the replicas, `slabs` unused
  uint32_t bh_cap_log2;
  uint32_t bh_pad_;
};

HDK_DEV void vec_lds_op(int32_t wop, int64_t* wp, int64_t v) {
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

template <bool J, bool KEYED = false, bool MANY = false, bool BH = false>
HDK_DEV void scan_agg_vec_body(const VecArgs& a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  __shared__ WordLayout wl;
  __shared__ uint64_t s_col_off[BH ? 2 * HDK_HIP_MAX_TARGETS : 1];
  __shared__ BhLdsLayout s_ll;  // (BH only; an unused LDS object costs nothing)
  if (a.run_if && *a.run_if != 1) {
    return;  // (the sliced passes did the job -- or the launch was interrupted: 2)
  }
  const cplan_t p = to_const_as(a.plan);
  const int tid = threadIdx.x;
  if (tid == 0) {
    make_word_layout(a.plan, &wl);
    if constexpr (BH) {
      bh_layout_identity(wl, &s_ll);
    }
  }
  __syncthreads();
  // (BH: an entry is its wpe words and then the key word)
  const int wpe = BH ? wl.wpe + 1 : wl.wpe;
  const uint32_t rep = a.rep;
  const uint32_t ew = a.entry_count * wpe;
  const uint32_t total_words = ew * rep;
  if constexpr (BH) {
    bh_lds_init(lds, s_ll, a.bh_cap_log2, rep, tid, kVecBlock);
  } else {
    for (uint32_t i = tid; i < total_words; i += kVecBlock) {
      lds[i] = word_identity(wl.wop[(i / rep) % wpe]);
    }
  }
  __syncthreads();

  const uint32_t my_rep = tid & (rep - 1);
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(kVecBlock) * VR;
  const bool grouped = p->query_kind != HDK_Q_NON_GROUPED;
  const int nt = p->num_targets;

  VecCtxT<J, KEYED, MANY> c;
  vec_ctx_init(c, p, tid, kVecBlock);
  int32_t err = 0;

  int64_t tile = blockIdx.x;
  const Watch watch = watch_begin(a.kp);
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    c.cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      bool pass0[VR];
      vec_ctx_tile(c, row0, nrows, pass0);  // dead slots re-read a valid row; their results are dropped
      rows_pass_v(c, a.kp.join_hash_tables, pass0, err);
      // group entry + aggregate updates of the batch's rows that are still in
      auto aggregate = [&](bool (&pass)[VR]) {
      int64_t entry[VR];
      if constexpr (BH) {
        // key word of the batch's rows -> entry of the lane's replica (found or claimed)
        int64_t kw[VR];
        eval_key_v(c, 0, kw, pass, err);
        if (p->key_count == 2) {  // two 4-byte keys share the key word, as they share the table's first quad
          int64_t k1[VR];
          eval_key_v(c, 1, k1, pass, err);
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            kw[r] = static_cast<int64_t>((static_cast<uint64_t>(k1[r]) << 32) | static_cast<uint32_t>(kw[r]));
          }
        } else if (p->key_width == 4) {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            kw[r] = static_cast<int32_t>(kw[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          entry[r] = 0;
          if (pass[r]) {
            const int32_t e = bh_lds_find_or_claim(lds + my_rep, kw[r], a.bh_cap_log2, static_cast<uint32_t>(wpe) * rep,
                                                   static_cast<uint32_t>(wl.wpe) * rep);
            if (e < 0) {
              err = HDK_HIP_ERR_OUT_OF_SLOTS;  // more groups than the plan's table holds
              pass[r] = false;
            } else {
              entry[r] = e;
            }
          }
        }
      } else if (grouped) {
        perfect_hash_entry_v(c, entry, pass, err);
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (pass[r] && static_cast<uint64_t>(entry[r]) >= a.entry_count) {
            err = HDK_HIP_ERR_OUT_OF_SLOTS;
            pass[r] = false;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          entry[r] = 0;
        }
      }
      uint32_t base[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        base[r] = (pass[r] ? static_cast<uint32_t>(entry[r]) * wpe : 0u) * rep + my_rep;
        if (pass[r]) {
          atomicAdd(reinterpret_cast<unsigned long long*>(lds + base[r]), 1ull);  // row count
        }
      }
      for (int t = 0; t < nt; ++t) {
        const int vw = wl.vword[t];
        const int nw = wl.nword[t];
        if (vw < 0 && nw < 0) {
          continue;
        }
        ctarget_t tg = p->targets[t];
        int64_t v[VR];
        bool is_null[VR];
        eval_target_arg_v(c, tg, v, is_null, pass, err);
        const int32_t wop = vw >= 0 ? wl.wop[vw] : 0;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (pass[r]) {
            if (is_null[r]) {
              if (nw >= 0) {  // counts NULLs; the flush turns it into the non-null count
                atomicAdd(reinterpret_cast<unsigned long long*>(lds + base[r] + nw * rep), 1ull);
              }
            } else if (vw >= 0) {
              vec_lds_op(wop, lds + base[r] + vw * rep, v[r]);
            }
          }
        }
      }
          };
      if constexpr (MANY) {
        // a matching-set join: the batch once per match (round i: every row's i-th partner), vec_eval.h: vec_round_v
        for (int round = 0;; ++round) {
          bool live[VR];
          if (!vec_round_v(c, round, pass0, live, err)) {
            break;
          }
          aggregate(live);
        }
      } else {
        aggregate(pass0);
      }
    }
    frag_tile_begin += ntiles;
  }
  if constexpr (BH) {
    BhGeom g;
    g.out_entry_count = a.entry_count;
    g.cap_log2 = a.bh_cap_log2;
    g.rep = rep;
    g.lds_bytes = 0;
    // (a.slabs: the two-level fold -- this block's table goes to its slab, hdk_bh_fold_words folds the slabs)
    int64_t* slab = a.slabs ? a.slabs + (static_cast<size_t>(blockIdx.x) * (static_cast<uint32_t>(wl.wpe) + 1) << a.bh_cap_log2) : nullptr;
    bh_flush_block<kVecBlock>(a.plan, wl, s_ll, lds, g, a.kp.groupby_buf, s_col_off, tid, err, slab);
    if (err) {
      record_error(a.kp.error_code, err);
    }
    return;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
  __syncthreads();
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * ew;
  for (uint32_t i = tid; i < ew; i += kVecBlock) {
    const int32_t op = wl.wop[i % wpe];
    int64_t acc = lds[i * rep];
    for (uint32_t r = 1; r < rep; ++r) {
      acc = word_combine(op, acc, lds[i * rep + r]);
    }
    if (wl.is_nword[i % wpe]) {  // NULL count -> non-null count = rows of the entry - NULLs
      const uint32_t w0 = (i / wpe) * wpe;
      int64_t rows = 0;
      for (uint32_t r = 0; r < rep; ++r) {
        rows += lds[w0 * rep + r];
      }
      acc = rows - acc;
    }
    slab[i] = acc;
  }
}

#ifndef HDK_VEC_BODY_ONLY  // (scan_bh.hip instantiates the body with BH = true under its own kernel names)
// Two instantiations: plans without joins carry no probe state (fewer VGPRs, more waves for the
// compute-bound taxi Q3/Q4 shapes); plans with joins trade occupancy for 16-byte probe gathers --
// a random gather is bound by line fetches from HBM, not by occupancy (scripts/microbench/gather.hip).
#ifndef HDK_VEC_WAVES
#define HDK_VEC_WAVES 3  // (151 registers: three waves per SIMD as it is)
#endif
extern "C" __global__ __launch_bounds__(kVecBlock, HDK_VEC_WAVES) void hdk_scan_agg_vec(VecArgs a) {
  scan_agg_vec_body<false>(a);
}
#ifndef HDK_VEC_JOIN_WAVES
#define HDK_VEC_JOIN_WAVES 2
#endif
extern "C" __global__ __launch_bounds__(kVecBlock, HDK_VEC_JOIN_WAVES) void hdk_scan_agg_vec_join(VecArgs a) {
  scan_agg_vec_body<true>(a);
}
// ... and plans in which some inner-like join probes a KEYED one-to-one table (composite or wide keys): the probe is a
// hash and a short linear walk per row; a kernel of its own so that its registers are not the perfect-hash plans' problem
extern "C" __global__ __launch_bounds__(kVecBlock, 2) void hdk_scan_agg_vec_keyed(VecArgs a) {  // (two waves per SIMD: 256 registers; unbounded it took 296 and ran one)
  scan_agg_vec_body<true, true>(a);
}
// ... and plans whose ONE join probes a one-to-many perfect-hash table: the batch is replayed once per match
extern "C" __global__ __launch_bounds__(kVecBlock) void hdk_scan_agg_vec_many(VecArgs a) {
  scan_agg_vec_body<true, false, true>(a);
}

#endif  // HDK_VEC_BODY_ONLY

}  // namespace hdk
