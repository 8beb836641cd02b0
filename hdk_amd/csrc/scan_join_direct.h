// scan_join_direct.h -- probe-and-aggregate for the shape of BASELINE config 3, without the interpreter:
//   SELECT agg(f(x, payload)) ... FROM outer JOIN inner ON outer.key = inner.key          (NonGroupedAggregate)
// one INNER join on a fused one-to-one perfect-hash table with ONE payload word ([row id | payload], 16 bytes per key),
// a plain 8-byte outer key column, at most one other 8-byte integer outer column x, no filter; every aggregate argument is
// x, the payload, or `a op b` (+ - *) over x / payload / an integer literal -- evaluated with the interpreter's rules
// (device_common.h eval_expr / eval_target_arg: NULL in, NULL out; checked arithmetic; `val != skip_val`).
// The batched interpreter needs 3.1 ms per 256 M rows for such a plan even when the join table sits in L2 (112 vector +
// 145 scalar instructions per row); this kernel keeps the aggregates in registers and touches LDS once per lane.
// Same LDS table / slab / hdk_finalize protocol as the other LDS-strategy kernels (agg_common.h).
//
// Input: the plan's own columns, row order (tables that fit the caches), or -- for tables far larger than L2 -- (key, x)
// TUPLES that hdk_cluster_by_key (scan_cluster.h, AoS mode) scattered by key range: a sub-slab's probes then stay inside
// a slice of the table that L2 holds, instead of costing one 128-byte memory line each (profiles/r02_c3_pmc.json).
#pragma once
#include "agg_common.h"
#include "scan_agg_vec.h"   // vec_lds_op
#include "scan_cluster.h"
#include "watch.h"

namespace hdk {

constexpr int kJdBlock = 256;
constexpr int kJdVR = 8;          // rows per lane and tile
constexpr int kJdMaxTargets = 4;

enum JdLeafKind : int32_t { JD_X = 0, JD_PAYLOAD = 1, JD_LITERAL = 2 };
struct JdLeaf {
  int32_t kind;
  int32_t nullable;
  int64_t ival;       // JD_LITERAL
  int64_t null_val;
};
struct JdTarget {
  int32_t has_arg;    // 0: COUNT(*), served by the row count
  int32_t nsteps;     // 0: the argument is leaf `a`; 1: a op b
  int32_t op;         // HDK_OP_ADD / SUB / MUL
  int32_t check_width;
  JdLeaf a, b;
  int64_t step_null;  // NULL of the step's result
  int64_t arg_null;   // expr.null_val / expr.nullable: how eval_target_arg recognises a NULL argument
  int32_t arg_nullable;
  int32_t skip_null;
  int64_t slot_null;  // the target's skip value
  int32_t vword, nword;  // words of the entry (or -1)
  int32_t wop;        // combine op of vword (WOP_ADD_U64 / WOP_MIN_I64 / WOP_MAX_I64)
  int32_t pad_;
};

struct JoinDirectArgs {
  KernParams kp;
  int64_t* slabs;
  uint32_t rep;
  int32_t wpe;
  int32_t key_buf_idx;
  int32_t x_buf_idx;      // -1: no second outer column
  int64_t min_key, max_key, key_null;
  int32_t key_nullable;   // HDK_JOIN_NULL_NULLABLE: a NULL key has no partner
  int32_t ntargets;
  JdTarget t[kJdMaxTargets];
  int32_t wop[kMaxWordsPerEntry];
  uint32_t nword_mask;    // bit w: word w counts NULLs (the flush stores row count - NULLs)
  // clustered input (scan_cluster.h, AoS): sub-slab f = tuples[f * sub ...), fill[f * stride] of them; then the overflow area
  int32_t clustered;
  int32_t tw;             // words per tuple: 1 (key) or 2 (key, x)
  const int64_t* tuples;
  const uint32_t* fill;
  const uint32_t* fill_ovf;
  uint64_t sub;
  uint64_t cap_ovf;
  const uint32_t* run_if;  // nullptr: always run; else only when *run_if == 1 (armed by the sliced path, scan_join_sliced.h)
};

HDK_DEV int64_t jd_leaf(const JdLeaf& l, int64_t x, int64_t pay) {
  return l.kind == JD_X ? x : (l.kind == JD_PAYLOAD ? pay : l.ival);
}

// per-lane state: row count, and per target the combined value and the NULLs seen
struct JdAcc {
  uint64_t rows;
  int64_t val[kJdMaxTargets];
  uint64_t nulls[kJdMaxTargets];
};

// every target of one joined row (the row function of this shape behind the probe)
HDK_DEV void jd_eval(const JoinDirectArgs& a, int64_t x, int64_t pay, JdAcc& acc, int32_t& err) {
  acc.rows += 1;
#pragma unroll
  for (int t = 0; t < kJdMaxTargets; ++t) {
    if (t < a.ntargets && a.t[t].has_arg) {
      const JdTarget& tg = a.t[t];
      int64_t v = jd_leaf(tg.a, x, pay);
      if (tg.nsteps) {  // eval_expr, one integer step
        const int64_t b = jd_leaf(tg.b, x, pay);
        const bool a_null = tg.a.nullable && v == tg.a.null_val;
        const bool b_null = tg.b.nullable && b == tg.b.null_val;
        if (a_null || b_null) {
          v = tg.step_null;
        } else {
          int64_t r;
          if (checked_arith(tg.op, v, b, tg.check_width, &r)) {
            err = HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW;
          }
          v = r;
        }
      }
      // eval_target_arg: a NULL argument, or a value that collides with the skip value
      const bool is_null = tg.skip_null && ((tg.arg_nullable && v == tg.arg_null) || v == tg.slot_null);
      if (is_null) {
        acc.nulls[t] += 1;
      } else if (tg.vword >= 0) {
        acc.val[t] = word_combine(tg.wop, acc.val[t], v);
      }
    }
  }
}

// one outer row: probe the table in memory, then every target
HDK_DEV void jd_row(const JoinDirectArgs& a, const int64_t* __restrict__ table, int64_t key, int64_t x, JdAcc& acc, int32_t& err) {
  const bool in_range = !(a.key_nullable && key == a.key_null) && key >= a.min_key && key <= a.max_key;
  const int64_t slot = in_range ? key - a.min_key : 0;
  const bf_i64x2 e = *reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(reinterpret_cast<uintptr_t>(table) + static_cast<uint64_t>(slot) * 16);
  if (!(in_range && e.x >= 0)) {
    return;  // INNER join: no partner, no row
  }
  jd_eval(a, x, e.y, acc, err);
}

// end of a block: one LDS update per lane and word, then the block's slab
HDK_DEV void jd_flush(const JoinDirectArgs& a, const JdAcc& acc, int64_t* lds, uint32_t my_rep, int tid, int block) {
  const uint32_t rep = a.rep;
  const uint32_t ew = static_cast<uint32_t>(a.wpe);
  if (acc.rows) {
    atomicAdd(reinterpret_cast<unsigned long long*>(lds + my_rep), static_cast<unsigned long long>(acc.rows));
#pragma unroll
    for (int t = 0; t < kJdMaxTargets; ++t) {
      if (t < a.ntargets && a.t[t].has_arg) {
        if (a.t[t].vword >= 0) {
          vec_lds_op(a.t[t].wop, lds + static_cast<uint32_t>(a.t[t].vword) * rep + my_rep, acc.val[t]);
        }
        if (a.t[t].nword >= 0 && acc.nulls[t]) {
          atomicAdd(reinterpret_cast<unsigned long long*>(lds + static_cast<uint32_t>(a.t[t].nword) * rep + my_rep),
                    static_cast<unsigned long long>(acc.nulls[t]));
        }
      }
    }
  }
  __syncthreads();
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * ew;
  for (uint32_t i = tid; i < ew; i += block) {
    const int32_t op = a.wop[i];
    int64_t v = lds[i * rep];
    for (uint32_t r = 1; r < rep; ++r) {
      v = word_combine(op, v, lds[i * rep + r]);
    }
    if ((a.nword_mask >> i) & 1u) {  // NULL count -> non-null count = rows - NULLs
      int64_t rows = 0;
      for (uint32_t r = 0; r < rep; ++r) {
        rows += lds[r];
      }
      v = rows - v;
    }
    slab[i] = v;
  }
}

__global__ __launch_bounds__(kJdBlock) void hdk_join_agg_direct(JoinDirectArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  if (a.run_if && *a.run_if != 1) {
    return;  // (the sliced path did the job -- or the launch was interrupted: 2)
  }
  constexpr int VR = kJdVR;
  const int tid = threadIdx.x;
  const int wpe = a.wpe;
  const uint32_t rep = a.rep;
  const uint32_t ew = static_cast<uint32_t>(wpe);  // one entry
  for (uint32_t i = tid; i < ew * rep; i += kJdBlock) {
    lds[i] = word_identity(a.wop[(i / rep) % wpe]);
  }
  __syncthreads();
  const uint32_t my_rep = tid & (rep - 1);
  const int64_t* __restrict__ table = a.kp.join_hash_tables;  // one join: the table itself
  JdAcc acc;
  acc.rows = 0;
#pragma unroll
  for (int t = 0; t < kJdMaxTargets; ++t) {
    acc.val[t] = (t < a.ntargets && a.t[t].vword >= 0) ? word_identity(a.t[t].wop) : 0;
    acc.nulls[t] = 0;
  }
  int32_t err = 0;
  const Watch watch = watch_begin(a.kp);
  constexpr int64_t kTileRows = static_cast<int64_t>(kJdBlock) * VR;
  if (!a.clustered) {
    const uint64_t nfrag = *a.kp.num_fragments;
    const uint32_t ntab = *a.kp.num_tables;
    int64_t tile = blockIdx.x;
    int64_t frag_tile_begin = 0;
    for (uint64_t f = 0; f < nfrag; ++f) {
      const int64_t nrows = a.kp.num_rows[f * ntab];
      const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
      const int8_t* const* cols = a.kp.col_buffers[f];
      const int8_t* kcol = cols[a.key_buf_idx];
      const int8_t* xcol = a.x_buf_idx >= 0 ? cols[a.x_buf_idx] : nullptr;
      for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
        HDK_WATCH_TILE(watch, err, tile)
        const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
        if (row0 + kTileRows <= nrows) {
          // full tile: rows dealt in adjacent pairs, 16-byte non-temporal loads, all issued before the first probe
          int64_t k[VR], x[VR];
#pragma unroll
          for (int u = 0; u < VR / 2; ++u) {
            const int64_t r = row0 + (static_cast<int64_t>(u) * kJdBlock + tid) * 2;
            const bf_i64x2 kk = gload<bf_i64x2>(kcol, r >> 1, true);
            k[2 * u] = kk.x;
            k[2 * u + 1] = kk.y;
            if (xcol) {
              const bf_i64x2 xx = gload<bf_i64x2>(xcol, r >> 1, true);
              x[2 * u] = xx.x;
              x[2 * u + 1] = xx.y;
            } else {
              x[2 * u] = 0;
              x[2 * u + 1] = 0;
            }
          }
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            jd_row(a, table, k[r], x[r], acc, err);
          }
        } else {
          for (int64_t r = row0 + tid; r < nrows; r += kJdBlock) {
            jd_row(a, table, gload<int64_t>(kcol, r, true), xcol ? gload<int64_t>(xcol, r, true) : 0, acc, err);
          }
        }
      }
      frag_tile_begin += ntiles;
    }
  } else {
    // clustered tuples: the sub-slabs in order (consecutive blocks work on the same few key ranges), then the overflow area
    const uint32_t nsub = kClusterBins * kClusterXcds;
    int64_t tile = blockIdx.x;
    int64_t tile_begin = 0;
    for (uint32_t f = 0; f <= nsub; ++f) {
      const uint64_t n = f < nsub ? min(static_cast<uint64_t>(a.fill[static_cast<size_t>(f) * kClusterCursorStride]), a.sub)
                                  : min(static_cast<uint64_t>(*a.fill_ovf), a.cap_ovf);
      const int64_t ntiles = static_cast<int64_t>((n + kTileRows - 1) / kTileRows);
      const int64_t* in = a.tuples + static_cast<size_t>(f) * a.sub * a.tw;
      for (; tile < tile_begin + ntiles; tile += gridDim.x) {
        HDK_WATCH_TILE(watch, err, tile)
        const uint64_t i0 = static_cast<uint64_t>(tile - tile_begin) * kTileRows;
        if (a.tw == 2 && i0 + kTileRows <= n) {
          bf_i64x2 tp[VR];
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            tp[r] = gload<bf_i64x2>(reinterpret_cast<const int8_t*>(in), static_cast<int64_t>(i0 + static_cast<uint64_t>(r) * kJdBlock + tid), true);
          }
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            jd_row(a, table, tp[r].x, tp[r].y, acc, err);
          }
        } else {
          for (uint64_t i = i0 + tid; i < n && i < i0 + kTileRows; i += kJdBlock) {
            const int64_t key = gload<int64_t>(reinterpret_cast<const int8_t*>(in), static_cast<int64_t>(i * a.tw), true);
            const int64_t x = a.tw == 2 ? gload<int64_t>(reinterpret_cast<const int8_t*>(in), static_cast<int64_t>(i * a.tw + 1), true) : 0;
            jd_row(a, table, key, x, acc, err);
          }
        }
      }
      tile_begin += ntiles;
    }
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
  jd_flush(a, acc, lds, my_rep, tid, kJdBlock);
}

}  // namespace hdk
