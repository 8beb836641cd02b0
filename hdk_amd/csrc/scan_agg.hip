// scan_agg.hip -- the hot pass: scan -> filter -> join probe -> group -> aggregate, LDS-privatised.
//
// Replaces what HDK JITs per query (multifrag_query_hoisted_literals -> query_group_by_template ->
// row_func, reference QE/RuntimeFunctions.cpp:1692-1768, QE/QueryTemplateGenerator.cpp:488-785)
// with fixed kernels for NonGroupedAggregate and GroupByPerfectHash plans whose partial table fits
// in LDS.  Reference GPU behaviour being replaced: one global atomic per row and target
// (agg_*_shared, QE/cuda_mapd_rt.cu:424-478,886-957) onto <= entry_count contended addresses.
//
// MI355X design:
//   * persistent grid (blocks = CUs x k), static tile walk over ALL fragments (tile -> fragment by a
//     scalar walk of NUM_ROWS), so there is no per-fragment tail;
//   * every lane reads its rows with the widest naturally aligned load the column allows; the
//     specialised kernels (scan_agg_fast.hip) read 16 B per lane per column;
//   * aggregation in LDS with ds_add_u64 / ds_add_f64 / ds_min/max_i64 into a table replicated REP
//     times (lane & (REP-1) picks the replica) so that 64 lanes hitting one key do not serialise;
//   * one slab store per block, then `k_finalize` folds the slabs into the output buffer with the
//     reference's exact agg_*[_skip_val] semantics -- no global atomics, deterministic slab order.
#include "agg_common.h"
#include "watch.h"

namespace hdk {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// LDS atomics on 64-bit words
// ---------------------------------------------------------------------------------------------
HDK_DEV void lds_add_u64(int64_t* p, int64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
}
HDK_DEV void lds_add_f64(int64_t* p, double v) {
  atomicAdd(reinterpret_cast<double*>(p), v);  // ds_add_f64 (-munsafe-fp-atomics)
}
HDK_DEV void lds_min_i64(int64_t* p, int64_t v) { atomicMin(reinterpret_cast<long long*>(p), static_cast<long long>(v)); }
HDK_DEV void lds_max_i64(int64_t* p, int64_t v) { atomicMax(reinterpret_cast<long long*>(p), static_cast<long long>(v)); }
HDK_DEV void lds_min_f64(int64_t* p, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(p);
  unsigned long long old = *a;
  while (v < bits_to_double(static_cast<int64_t>(old))) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, static_cast<unsigned long long>(double_to_bits(v)));
    if (old == assumed) {
      break;
    }
  }
}
HDK_DEV void lds_max_f64(int64_t* p, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(p);
  unsigned long long old = *a;
  while (bits_to_double(static_cast<int64_t>(old)) < v) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, static_cast<unsigned long long>(double_to_bits(v)));
    if (old == assumed) {
      break;
    }
  }
}

struct ScanArgs {
  const hdk_hip_plan* plan;  // device copy (workspace head)
  KernParams kp;
  int64_t* slabs;            // [gridDim.x][entry_count * wpe]
  uint32_t entry_count;      // 1 for non-grouped
  uint32_t rep;              // power of two, <= 32
  uint32_t rows_per_tile;    // kBlock * rows per thread
};

// ---------------------------------------------------------------------------------------------
// K1 (generic): any plan the library accepts, evaluated by the plan interpreter.
// ---------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(kBlock) void hdk_scan_agg_generic(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  __shared__ WordLayout wl;
  const hdk_hip_plan* __restrict__ p = a.plan;
  const int tid = threadIdx.x;
  if (tid == 0) {
    make_word_layout(p, &wl);
  }
  __syncthreads();
  const int wpe = wl.wpe;
  const uint32_t rep = a.rep;
  const uint32_t ew = a.entry_count * wpe;
  const uint32_t total_words = ew * rep;
  for (uint32_t i = tid; i < total_words; i += kBlock) {
    lds[i] = word_identity(wl.wop[(i / rep) % wpe]);
  }
  __syncthreads();

  const uint32_t my_rep = tid & (rep - 1);
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const int64_t tile_rows = a.rows_per_tile;
  const bool grouped = p->query_kind != HDK_Q_NON_GROUPED;
  const int nt = p->num_targets;

  RowCtx c;
  c.plan = p;
  c.join_row[0] = 0;
  c.join_row[1] = 0;
  int32_t err = 0;

  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  const Watch watch = watch_begin(a.kp);
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + tile_rows - 1) / tile_rows;
    c.cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row0 = (tile - frag_tile_begin) * tile_rows;
      const int64_t row_end = min(row0 + tile_rows, nrows);
      for (int64_t row = row0 + tid; row < row_end; row += kBlock) {
        c.pos = row;
        for_each_row_match(c, a.kp.join_hash_tables, err, [&]() {
        int64_t entry = 0;
        if (grouped) {
          entry = perfect_hash_entry(c, err);
          if (static_cast<uint64_t>(entry) >= a.entry_count) {
            err = HDK_HIP_ERR_OUT_OF_SLOTS;  // key outside the range the layout was sized for
            return;
          }
        }
        int64_t* base = lds + (static_cast<uint32_t>(entry) * wpe) * rep + my_rep;
        lds_add_u64(base, 1);
        for (int t = 0; t < nt; ++t) {
          const hdk_hip_target& tg = p->targets[t];
          const int vw = wl.vword[t];
          const int nw = wl.nword[t];
          if (vw < 0 && nw < 0) {
            continue;
          }
          bool is_null;
          const int64_t v = eval_target_arg(c, tg, is_null, err);
          if (is_null) {
            if (nw >= 0) {
              lds_add_u64(base + nw * rep, 1);  // counts NULLs; the flush turns it into the non-null count
            }
            continue;
          }
          if (vw >= 0) {
            int64_t* wp = base + vw * rep;
            switch (wl.wop[vw]) {
              case WOP_ADD_U64: lds_add_u64(wp, v); break;
              case WOP_ADD_F64: lds_add_f64(wp, bits_to_double(v)); break;
              case WOP_MIN_I64: lds_min_i64(wp, v); break;
              case WOP_MAX_I64: lds_max_i64(wp, v); break;
              case WOP_MIN_F64: lds_min_f64(wp, bits_to_double(v)); break;
              default: lds_max_f64(wp, bits_to_double(v)); break;
            }
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
  __syncthreads();
  // flush: fold the REP replicas, one slab per block
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * ew;
  for (uint32_t i = tid; i < ew; i += kBlock) {
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

// ---------------------------------------------------------------------------------------------
// K2: fold the per-block slabs into the output buffer.  One wave per entry.
// ---------------------------------------------------------------------------------------------
struct FinalizeArgs {
  const hdk_hip_plan* plan;
  const int64_t* slabs;
  int64_t** groupby_buf;  // GROUPBY_BUF
  uint32_t num_slabs;
  uint32_t entry_count;
  const uint32_t* skip_if;  // nullptr, or: do nothing when *skip_if != 0 (slabs of a strategy whose statistics did not hold)
};

HDK_DEV int64_t shfl_down_i64(int64_t v, int delta) {
  int lo = static_cast<int>(v & 0xffffffff);
  int hi = static_cast<int>(v >> 32);
  lo = __shfl_down(lo, delta, kWave);
  hi = __shfl_down(hi, delta, kWave);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

HDK_DEV void apply_count(int8_t* slot, int width, int64_t cnt) {
  if (width == 4) {
    *reinterpret_cast<int32_t*>(slot) =
        static_cast<int32_t>(static_cast<uint32_t>(*reinterpret_cast<int32_t*>(slot)) + static_cast<uint32_t>(cnt));
  } else {
    *reinterpret_cast<int64_t*>(slot) =
        static_cast<int64_t>(static_cast<uint64_t>(*reinterpret_cast<int64_t*>(slot)) + static_cast<uint64_t>(cnt));
  }
}

// agg_{sum,min,max}[_double][_skip_val] applied once with the block-folded partial
HDK_DEV void apply_value(const hdk_hip_target& tg, int8_t* slot, int64_t partial, int64_t nn, int64_t rowcount) {
  const bool skip = tg.skip_null;
  if ((skip && nn == 0) || rowcount == 0) {
    return;  // nothing but NULLs (or no rows): the slot keeps its value
  }
  const int agg = tg.agg;
  if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) {  // float accumulator in the slot's low 4 bytes; the partial is a double
    int32_t* s = reinterpret_cast<int32_t*>(slot);
    const float v = static_cast<float>(bits_to_double(partial));
    const int32_t old = *s;
    if (skip && old == float_slot_null(tg)) {
      *s = __float_as_int(v);
      return;
    }
    const float o = __int_as_float(old);
    float r;
    if (agg == HDK_AGG_MIN) {
      r = (v < o) ? v : o;
    } else if (agg == HDK_AGG_MAX) {
      r = (o < v) ? v : o;
    } else {
      // the block partials were summed in double: one rounding to float here (the reference's float atomics
      // round at every row; tests compare float sums with a tolerance)
      r = static_cast<float>(static_cast<double>(o) + bits_to_double(partial));
    }
    *s = __float_as_int(r);
    return;
  }
  if (tg.slot_width == 1 || tg.slot_width == 2) {  // logical-sized MIN / MAX slots (columnar)
    small_min_max(agg, skip, tg.slot_width, slot, partial, tg.null_val);
    return;
  }
  if (tg.slot_width == 4) {  // int32 slots (integer arguments)
    int32_t* s = reinterpret_cast<int32_t*>(slot);
    const int32_t v = static_cast<int32_t>(partial);
    const int32_t nullv = static_cast<int32_t>(tg.null_val);
    const int32_t old = *s;
    if (skip && old == nullv) {
      *s = v;
    } else if (agg == HDK_AGG_MIN) {
      *s = old < v ? old : v;
    } else if (agg == HDK_AGG_MAX) {
      *s = old > v ? old : v;
    } else {
      *s = static_cast<int32_t>(static_cast<uint32_t>(old) + static_cast<uint32_t>(v));
    }
    return;
  }
  int64_t* s = reinterpret_cast<int64_t*>(slot);
  const int64_t old = *s;
  if (skip && old == tg.null_val) {  // first non-NULL value replaces the sentinel (bit compare)
    *s = partial;
    return;
  }
  if (tg.arg_is_fp) {
    const double o = bits_to_double(old);
    const double v = bits_to_double(partial);
    double r;
    if (agg == HDK_AGG_MIN) {
      r = (v < o) ? v : o;
    } else if (agg == HDK_AGG_MAX) {
      r = (o < v) ? v : o;
    } else {
      r = o + v;
    }
    *s = double_to_bits(r);
  } else if (agg == HDK_AGG_MIN) {
    *s = old < partial ? old : partial;
  } else if (agg == HDK_AGG_MAX) {
    *s = old > partial ? old : partial;
  } else {
    *s = static_cast<int64_t>(static_cast<uint64_t>(old) + static_cast<uint64_t>(partial));
  }
}

// One entry's folded partial words into the output buffer: the exact agg_*[_skip_val] semantics (QE/RuntimeFunctions.cpp:
// 387-875), so that repeated launches accumulate like repeated row-function calls.  words[w * ws]: word w of the layout.
HDK_DEV void finalize_apply(const hdk_hip_plan* __restrict__ p, const WordLayout& wl, const FinalizeArgs& a, uint32_t entry,
                            const int64_t* words, int ws) {
  const int64_t rowcount = words[0];
  if (rowcount == 0) {
    return;
  }
  int64_t* buf = a.groupby_buf[0];
  const bool grouped = p->query_kind != HDK_Q_NON_GROUPED;
  const bool columnar = p->output_columnar;
  const int nk = p->key_count;
  // reconstruct the (translated) key components from the entry index
  int64_t keys[HDK_HIP_MAX_KEYS];
  if (grouped) {
    uint64_t rem = entry;
#pragma unroll
    for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
      if (k < nk) {
        uint64_t comp = rem;
        if (nk > 1) {
          comp = rem % static_cast<uint64_t>(p->key_card[k]);
          rem /= static_cast<uint64_t>(p->key_card[k]);
        }
        const int64_t bucket = p->key_bucket[k] ? p->key_bucket[k] : 1;
        keys[k] = p->key_min[k] + static_cast<int64_t>(comp) * bucket;
      }
    }
    if (!p->keyless) {
      if (columnar) {
        // set_matching_group_value_perfect_hash_columnar / get_columnar_group_bin_offset
        if (buf[entry] == HDK_EMPTY_KEY_64) {
#pragma unroll
          for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
            if (k < nk) {
              buf[static_cast<size_t>(k) * a.entry_count + entry] = keys[k];
            }
          }
        }
      } else {
        int64_t* row = buf + static_cast<size_t>(entry) * p->row_size_quad;
        if (row[0] == HDK_EMPTY_KEY_64) {  // get_group_value_fast / get_matching_group_value_perfect_hash
#pragma unroll
          for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
            if (k < nk) {
              row[k] = keys[k];
            }
          }
        }
      }
    }
  }
  const int nt = p->num_targets;
  int slot_idx = 0;
  for (int t = 0; t < nt; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    int8_t* s1;
    int8_t* s2 = nullptr;
    if (!grouped) {
      s1 = reinterpret_cast<int8_t*>(a.groupby_buf[slot_idx]);
      if (tg.agg == HDK_AGG_AVG) {
        s2 = reinterpret_cast<int8_t*>(a.groupby_buf[slot_idx + 1]);
      }
    } else if (columnar) {
      s1 = reinterpret_cast<int8_t*>(buf) + columnar_slot_off(p, a.entry_count, slot_idx) +
           static_cast<size_t>(entry) * tg.slot_width;
      if (tg.agg == HDK_AGG_AVG) {
        s2 = reinterpret_cast<int8_t*>(buf) + columnar_slot_off(p, a.entry_count, slot_idx + 1) +
             static_cast<size_t>(entry) * tg.slot2_width;
      }
    } else {
      int8_t* row = reinterpret_cast<int8_t*>(buf + static_cast<size_t>(entry) * p->row_size_quad);
      s1 = row + tg.slot_off;
      s2 = row + tg.slot2_off;
    }
    const int vw = wl.vword[t];
    const int nw = wl.nword[t];
    const int64_t nn = nw >= 0 ? words[nw * ws] : rowcount;
    switch (tg.agg) {
      case HDK_AGG_COUNT:
        apply_count(s1, tg.slot_width, nn);
        break;
      case HDK_AGG_ID: {
        int64_t kv = keys[0];
#pragma unroll
        for (int k = 1; k < HDK_HIP_MAX_KEYS; ++k) {
          if (k == tg.key_idx) {
            kv = keys[k];
          }
        }
        const int ki = tg.key_idx;
        if (p->key_has_nulls[ki] && kv == p->key_null_translated[ki]) {
          kv = p->keys[ki].null_val;  // the target re-evaluates the key expression: untranslated NULL
        }
        if (tg.slot_width == 4) {
          *reinterpret_cast<int32_t*>(s1) = static_cast<int32_t>(kv);
        } else {
          *reinterpret_cast<int64_t*>(s1) = kv;
        }
        break;
      }
      case HDK_AGG_AVG:
        apply_count(s2, tg.slot2_width, nn);
        apply_value(tg, s1, words[vw * ws], nn, rowcount);
        break;
      default:
        apply_value(tg, s1, words[vw * ws], nn, rowcount);
        break;
    }
    slot_idx += tg.agg == HDK_AGG_AVG ? 2 : 1;
  }
}


extern "C" __global__ __launch_bounds__(kBlock) void hdk_finalize(FinalizeArgs a) {
  __shared__ WordLayout wl;
  __shared__ int64_t s_words[kBlock / kWave][kMaxWordsPerEntry];
  const hdk_hip_plan* __restrict__ p = a.plan;
  if (a.skip_if && *a.skip_if) {
    return;
  }
  if (threadIdx.x == 0) {
    make_word_layout(p, &wl);
  }
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t entry = blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave);
  if (entry >= a.entry_count) {
    return;
  }
  const int wpe = wl.wpe;
  const size_t ew = static_cast<size_t>(a.entry_count) * wpe;
  // fold slabs: lane l takes slabs l, l+64, ...; then a fixed shuffle tree (deterministic order)
  int64_t* words = s_words[threadIdx.x / kWave];  // written and read by lane 0 only
  for (int w = 0; w < wpe; ++w) {
    const int32_t op = wl.wop[w];
    int64_t acc = word_identity(op);
    for (uint32_t b = lane; b < a.num_slabs; b += kWave) {
      acc = word_combine(op, acc, a.slabs[b * ew + static_cast<size_t>(entry) * wpe + w]);
    }
    for (int d = kWave / 2; d > 0; d >>= 1) {
      acc = word_combine(op, acc, shfl_down_i64(acc, d));
    }
    if (lane == 0) {
      words[w] = acc;
    }
  }
  if (lane != 0) {
    return;
  }
  finalize_apply(p, wl, a, entry, words, 1);
}

// Few slabs, many entries (the two-pass group-bys leave 8 slabs of a 100 K-entry table: a WAVE per entry took 343 us for it):
// a THREAD per entry folds the slabs' words itself.
constexpr int kFinalizeFlatBlock = 256;
extern "C" __global__ __launch_bounds__(kFinalizeFlatBlock) void hdk_finalize_flat(FinalizeArgs a) {
  __shared__ WordLayout wl;
  __shared__ int64_t s_words[kMaxWordsPerEntry][kFinalizeFlatBlock];
  const hdk_hip_plan* __restrict__ p = a.plan;
  if (a.skip_if && *a.skip_if) {
    return;
  }
  if (threadIdx.x == 0) {
    make_word_layout(p, &wl);
  }
  __syncthreads();
  const uint32_t entry = blockIdx.x * kFinalizeFlatBlock + threadIdx.x;
  if (entry >= a.entry_count) {
    return;
  }
  const int wpe = wl.wpe;
  const size_t ew = static_cast<size_t>(a.entry_count) * wpe;
  for (int w = 0; w < wpe; ++w) {
    const int32_t op = wl.wop[w];
    int64_t acc = word_identity(op);
    for (uint32_t b = 0; b < a.num_slabs; ++b) {
      acc = word_combine(op, acc, a.slabs[b * ew + static_cast<size_t>(entry) * wpe + w]);
    }
    s_words[w][threadIdx.x] = acc;
  }
  finalize_apply(p, wl, a, entry, &s_words[0][threadIdx.x], kFinalizeFlatBlock);
}

}  // namespace hdk

// =============================================================================================
// host side: strategy choice + launch
// =============================================================================================
#include <mutex>
#include <utility>
#include <vector>

#include <stddef.h>
#include <string.h>

#include "host_match.h"
#include "scan_bh_host.h"
#include "scan_bhm_host.h"
#include "scan_agg_fast.h"
#include "scan_agg_cols.h"
#include "scan_agg_vec.h"
#include "scan_agg_keys.h"
#include "scan_cluster.h"
#include "scan_join_direct.h"
#include "scan_join_sliced.h"
#include "scan_join_sliced2.h"

using namespace hdk;

namespace hdk {

// ---- plan validation: every index, width and enumerator a kernel or a host matcher will use unchecked ------------------
static int32_t check_leaf(const hdk_hip_plan* p, const hdk_hip_leaf& l, bool may_be_none, const char* what) {
  HDK_REQUIRE(l.kind >= (may_be_none ? HDK_LEAF_NONE : HDK_LEAF_COL) && l.kind <= HDK_LEAF_FP, "%s: bad leaf kind %d", what, l.kind);
  if (l.kind == HDK_LEAF_COL) {
    HDK_REQUIRE(l.col >= 0 && l.col < p->num_cols, "%s: column %d out of range (num_cols %d)", what, l.col, p->num_cols);
  }
  return HDK_HIP_OK;
}

static int32_t check_expr(const hdk_hip_plan* p, const hdk_hip_expr& e, const char* what) {
  HDK_REQUIRE(e.nsteps >= 0 && e.nsteps <= HDK_HIP_MAX_EXPR_STEPS, "%s: nsteps %d out of range", what, e.nsteps);
  HDK_REQUIRE(e.vclass == HDK_VC_INT || e.vclass == HDK_VC_FP, "%s: bad value class", what);
  int32_t st = check_leaf(p, e.leaf0, false, what);
  if (st) return st;
  for (int i = 0; i < e.nsteps; ++i) {
    const hdk_hip_step& sp = e.steps[i];
    HDK_REQUIRE(sp.op >= HDK_OP_ADD && sp.op <= HDK_OP_CAST_FP_TO_INT, "%s: step %d: bad op %d", what, i, sp.op);
    HDK_REQUIRE(sp.out_class == HDK_VC_INT || sp.out_class == HDK_VC_FP, "%s: step %d: bad out_class", what, i);
    HDK_REQUIRE(sp.check_width == 0 || sp.check_width == 1 || sp.check_width == 2 || sp.check_width == 4 || sp.check_width == 8,
                "%s: step %d: check_width must be 0/1/2/4/8", what, i);
    const bool unary = sp.op == HDK_OP_EXTRACT_YEAR || sp.op == HDK_OP_CAST_INT_TO_FP || sp.op == HDK_OP_CAST_FP_TO_INT;
    st = check_leaf(p, sp.rhs, unary, what);
    if (st) return st;
  }
  return HDK_HIP_OK;
}

// `exprs`: also every expression, column and join the scan kernels read.  The reductions and the table helpers are
// handed plans that only describe the output layout (a QueryMemoryDescriptor's worth: hdk_amd/glue/HipPlanBuilder.h
// make_plan; the reference's ResultSetReduction needs nothing else): they check with exprs = false.
static int32_t validate_plan_impl(const hdk_hip_plan* p, bool exprs) {
  HDK_REQUIRE(p, "plan is NULL");
  HDK_REQUIRE(p->abi_version == HDK_HIP_PLAN_ABI, "plan ABI %u != library ABI %u", p->abi_version,
              HDK_HIP_PLAN_ABI);
  HDK_REQUIRE(p->num_cols >= 0 && p->num_cols <= HDK_HIP_MAX_COLS, "bad num_cols");
  HDK_REQUIRE(p->num_quals >= 0 && p->num_quals <= HDK_HIP_MAX_QUALS, "bad num_quals");
  HDK_REQUIRE(p->num_joins >= 0 && p->num_joins <= HDK_HIP_MAX_JOINS, "bad num_joins");
  HDK_REQUIRE(p->key_count >= 0 && p->key_count <= HDK_HIP_MAX_KEYS, "bad key_count");
  int32_t st;
  for (int i = 0; exprs && i < p->num_cols; ++i) {
    const hdk_hip_col& c = p->cols[i];
    HDK_REQUIRE(c.width == 1 || c.width == 2 || c.width == 4 || c.width == 8, "column %d: width %d is not 1/2/4/8", i, c.width);
    HDK_REQUIRE(c.kind >= HDK_COL_INT && c.kind <= HDK_COL_SMALL_DATE, "column %d: bad kind %d", i, c.kind);
    HDK_REQUIRE(c.kind != HDK_COL_SMALL_DATE || c.width == 2 || c.width == 4, "column %d: a DATE in days is 2 or 4 bytes wide", i);
    HDK_REQUIRE(c.kind != HDK_COL_FLOAT || c.width == 4, "column %d: a float column is 4 bytes wide", i);
    HDK_REQUIRE(c.kind != HDK_COL_DOUBLE || c.width == 8, "column %d: a double column is 8 bytes wide", i);
    HDK_REQUIRE(c.table >= -p->num_joins && c.table <= p->num_joins, "column %d: table %d out of range (%d joins)", i, c.table,
                p->num_joins);
    HDK_REQUIRE(c.buf_idx >= 0 && c.buf_idx < 4096, "column %d: buf_idx %d out of range", i, c.buf_idx);
    if (c.table < 0) {  // payload word of a fused join table
      const hdk_hip_join& jn = p->joins[-c.table - 1];
      HDK_REQUIRE(jn.kind == HDK_JOIN_ONE_TO_ONE_FUSED && c.buf_idx >= 1 && c.buf_idx < jn.fused_stride && c.width == 8,
                  "column %d: not a payload word of join %d's fused table", i, -c.table - 1);
    }
    HDK_REQUIRE(c.has_stats == 0 || c.has_stats == 1, "column %d: has_stats must be 0 or 1", i);
  }
  for (int i = 0; exprs && i < p->num_quals; ++i) {
    const hdk_hip_qual& q = p->quals[i];
    if ((st = check_expr(p, q.lhs, "filter lhs"))) return st;
    if ((st = check_leaf(p, q.rhs, false, "filter rhs"))) return st;
    HDK_REQUIRE(q.cmp >= HDK_CMP_EQ && q.cmp <= HDK_CMP_GE, "filter %d: bad comparison %d", i, q.cmp);
    HDK_REQUIRE(q.after_joins == 0 || q.after_joins == 1, "filter %d: after_joins must be 0 or 1", i);
  }
  HDK_REQUIRE(p->filter_after_joins == 0 || p->filter_after_joins == 1, "filter_after_joins must be 0 or 1");
  for (int k = 0; exprs && k < p->key_count; ++k) {
    if ((st = check_expr(p, p->keys[k], "group-by key"))) return st;
  }
  for (int j = 0; exprs && j < p->num_joins; ++j) {
    const hdk_hip_join& jn = p->joins[j];
    HDK_REQUIRE(jn.kind >= HDK_JOIN_ONE_TO_ONE && jn.kind <= HDK_JOIN_KEYED_ONE_TO_MANY, "bad join kind");
    HDK_REQUIRE(jn.type >= HDK_JOIN_INNER && jn.type <= HDK_JOIN_ANTI, "bad join type");
    if (jn.type == HDK_JOIN_SEMI || jn.type == HDK_JOIN_ANTI) {  // first-row-wins fills exist for the one-to-one tables only
      HDK_REQUIRE(jn.kind == HDK_JOIN_ONE_TO_ONE || jn.kind == HDK_JOIN_ONE_TO_ONE_FUSED || jn.kind == HDK_JOIN_KEYED_ONE_TO_ONE,
                  "join %d: a SEMI / ANTI join probes a one-to-one table", j);
    }
    HDK_REQUIRE(jn.null_mode >= HDK_JOIN_NULL_NONE && jn.null_mode <= HDK_JOIN_NULL_BITWISE, "join %d: bad null_mode", j);
    HDK_REQUIRE(jn.table_idx >= 0 && jn.table_idx < p->num_joins, "join %d: table_idx %d out of range", j, jn.table_idx);
    HDK_REQUIRE(jn.bucket >= 0, "join %d: negative bucket", j);
    if ((st = check_expr(p, jn.outer_key, "join key"))) return st;
    if (jn.kind == HDK_JOIN_ONE_TO_ONE_FUSED) {
      HDK_REQUIRE(jn.fused_stride >= 1 && jn.fused_stride <= 8, "join %d: fused_stride %d out of range", j, jn.fused_stride);
    }
    if (jn.kind <= HDK_JOIN_ONE_TO_ONE_FUSED) {
      HDK_REQUIRE(jn.max_key >= jn.min_key, "join %d: empty key range", j);
      if (jn.null_mode == HDK_JOIN_NULL_BITWISE && jn.entry_count > 0 && jn.translated_null >= jn.min_key) {
        // the slot a NULL key probes ([bucketized_]hash_join_idx_bitwise) must lie inside the table
        const int64_t b = jn.bucket > 1 ? jn.bucket : 1;
        HDK_REQUIRE((jn.translated_null - jn.min_key) / b < jn.entry_count, "join %d: translated NULL key outside the table", j);
      }
    }
    if (jn.kind >= HDK_JOIN_KEYED_ONE_TO_ONE) {
      for (int k = 0; k + 1 < jn.key_component_count && k < HDK_HIP_MAX_JOIN_KEYS - 1; ++k) {
        if ((st = check_expr(p, jn.extra_keys[k], "join key component"))) return st;
      }
    }
    if (jn.kind == HDK_JOIN_ONE_TO_MANY || jn.kind >= HDK_JOIN_KEYED_ONE_TO_ONE) {
      HDK_REQUIRE(jn.entry_count > 0 && jn.entry_count < (int64_t(1) << 31), "join table entry_count out of range");
    }
    if (jn.kind >= HDK_JOIN_KEYED_ONE_TO_ONE) {
      HDK_REQUIRE(jn.key_component_count >= 1 && jn.key_component_count <= HDK_HIP_MAX_JOIN_KEYS,
                  "bad key_component_count");
      HDK_REQUIRE(jn.key_component_width == 4 || jn.key_component_width == 8, "bad key_component_width");
    }
  }
  HDK_REQUIRE(p->num_filter_ops >= 0 && p->num_filter_ops <= HDK_HIP_MAX_FILTER_OPS, "bad num_filter_ops");
  if (p->num_filter_ops) {  // a well-formed postfix program that leaves exactly one value
    int depth = 0;
    for (int i = 0; i < p->num_filter_ops; ++i) {
      const uint32_t op = p->filter_ops[i];
      if (op < HDK_F_AND) {
        HDK_REQUIRE(static_cast<int>(op) < p->num_quals, "filter program: qual %u out of range", op);
        ++depth;
      } else if (op == HDK_F_NOT) {
        HDK_REQUIRE(depth >= 1, "filter program: NOT on an empty stack");
      } else {
        HDK_REQUIRE(op == HDK_F_AND || op == HDK_F_OR, "filter program: unknown op %u", op);
        HDK_REQUIRE(depth >= 2, "filter program: AND / OR needs two values");
        --depth;
      }
      HDK_REQUIRE(depth <= 16, "filter program: more than 16 values on the stack");
    }
    HDK_REQUIRE(depth == 1, "filter program: leaves %d values", depth);
  }
  HDK_REQUIRE(p->num_targets > 0 && p->num_targets <= HDK_HIP_MAX_TARGETS, "bad num_targets");
  HDK_REQUIRE(p->query_kind >= HDK_Q_NON_GROUPED && p->query_kind <= HDK_Q_PROJECTION,
              "bad query_kind");
  if (p->query_kind == HDK_Q_PROJECTION) {
    HDK_REQUIRE(p->entry_count > 0, "entry_count must be positive");
    HDK_REQUIRE(p->key_count == 0 && !p->keyless, "projection plans have no group-by keys");
    HDK_REQUIRE(p->output_columnar || p->row_size_quad > 0, "row_size_quad must be positive");
    for (int t = 0; t < p->num_targets; ++t) {
      const hdk_hip_target& tg = p->targets[t];
      HDK_REQUIRE(tg.agg == HDK_AGG_ID && tg.has_arg, "projection targets are plain expressions");
      HDK_REQUIRE(tg.arg_is_fp >= HDK_FP_SLOT_NONE && tg.arg_is_fp <= HDK_FP_SLOT_FLOAT, "target %d: arg_is_fp out of range", t);
      if (exprs && (st = check_expr(p, tg.arg, "projection target"))) return st;
      HDK_REQUIRE(tg.slot_off >= 0 && (p->output_columnar || static_cast<int64_t>(tg.slot_off) + tg.slot_width <=
                                                                  static_cast<int64_t>(p->row_size_quad) * 8),
                  "target %d: slot at offset %d does not fit the row", t, tg.slot_off);
      HDK_REQUIRE(tg.slot_width == 1 || tg.slot_width == 2 || tg.slot_width == 4 || tg.slot_width == 8,
                  "slot width must be 1/2/4/8");
      HDK_REQUIRE(p->output_columnar || tg.slot_width == 8, "row-wise projection slots are 8 bytes");
    }
    return HDK_HIP_OK;
  }
  if (p->query_kind != HDK_Q_NON_GROUPED) {
    HDK_REQUIRE(p->entry_count > 0, "entry_count must be positive");
    HDK_REQUIRE(p->key_count > 0, "group-by plan without keys");
    HDK_REQUIRE(p->output_columnar || p->row_size_quad > 0, "row_size_quad must be positive");
  }
  HDK_REQUIRE(p->query_kind == HDK_Q_NON_GROUPED || p->key_width == 4 || p->key_width == 8, "key_width must be 4 or 8");
  int nslots = 0;
  for (int t = 0; t < p->num_targets; ++t) {
    nslots += p->targets[t].agg == HDK_AGG_AVG ? 2 : 1;
  }
  if (p->keyless) {
    HDK_REQUIRE(p->query_kind == HDK_Q_PERFECT_HASH, "only perfect-hash plans can be keyless");
    HDK_REQUIRE(p->idx_target_as_key >= 0 && p->idx_target_as_key < nslots, "idx_target_as_key %d out of range (%d slots)",
                p->idx_target_as_key, nslots);
  }
  const int64_t row_bytes = static_cast<int64_t>(p->row_size_quad) * 8;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    HDK_REQUIRE(tg.agg >= HDK_AGG_COUNT && tg.agg <= HDK_AGG_SINGLE_VALUE, "target %d: bad aggregate kind %d", t, tg.agg);
    if (tg.agg == HDK_AGG_SINGLE_VALUE) {
      HDK_REQUIRE(p->query_kind != HDK_Q_PROJECTION && tg.has_arg, "target %d: SINGLE_VALUE is an aggregate over an argument", t);
      HDK_REQUIRE(tg.slot_width == 4 || tg.slot_width == 8, "target %d: SINGLE_VALUE slots are 4 or 8 bytes wide", t);
    }
    HDK_REQUIRE(tg.has_arg == 0 || tg.has_arg == 1, "target %d: has_arg must be 0 or 1", t);
    HDK_REQUIRE(tg.arg_is_fp >= HDK_FP_SLOT_NONE && tg.arg_is_fp <= HDK_FP_SLOT_FLOAT, "target %d: arg_is_fp out of range", t);
    HDK_REQUIRE(!exprs || tg.has_arg || tg.agg == HDK_AGG_COUNT || (tg.agg == HDK_AGG_ID && p->query_kind != HDK_Q_NON_GROUPED),
                "target %d: only COUNT(*) and projected keys have no argument", t);
    if (exprs && tg.has_arg && (st = check_expr(p, tg.arg, "target argument"))) return st;
    if (p->query_kind != HDK_Q_NON_GROUPED && tg.slot_width != 0) {
      // a slot lies inside the row (row-wise) / starts on its own alignment (columnar)
      HDK_REQUIRE(tg.slot_off >= 0 && (p->output_columnar || static_cast<int64_t>(tg.slot_off) + tg.slot_width <= row_bytes),
                  "target %d: slot at offset %d does not fit a row of %lld bytes", t, tg.slot_off, static_cast<long long>(row_bytes));
      if (tg.agg == HDK_AGG_AVG) {
        HDK_REQUIRE(tg.slot2_width == 4 || tg.slot2_width == 8, "target %d: AVG count slot width must be 4 or 8", t);
        HDK_REQUIRE(tg.slot2_off >= 0 && (p->output_columnar || static_cast<int64_t>(tg.slot2_off) + tg.slot2_width <= row_bytes),
                    "target %d: AVG count slot at offset %d does not fit the row", t, tg.slot2_off);
      }
    }
    if (tg.slot_width == 0) {
      HDK_REQUIRE(p->query_kind == HDK_Q_BASELINE_HASH && tg.agg == HDK_AGG_ID && tg.key_idx >= 0 &&
                      tg.key_idx < p->key_count,
                  "only a projected key of a baseline-hash plan may have a zero-width slot");
      continue;
    }
    if (tg.slot_width == 1 || tg.slot_width == 2) {
      // logical-sized columns (RS/ColSlotContext.cpp, QE/MemoryLayoutBuilder.cpp:559-652): MIN / MAX over TINYINT /
      // SMALLINT in a columnar buffer keep 1- / 2-byte slots (the reference's small-slot runtime exists for those two
      // aggregates only: AGGREGATE_ONE_NULLABLE_VALUE_SMALL, QE/ResultSetReduction.cpp:1136-1172)
      HDK_REQUIRE(p->output_columnar && (tg.agg == HDK_AGG_MIN || tg.agg == HDK_AGG_MAX) && !tg.arg_is_fp,
                  "target %d: 1- and 2-byte slots hold integer MIN / MAX of a columnar buffer only", t);
      continue;
    }
    HDK_REQUIRE(tg.slot_width == 4 || tg.slot_width == 8, "slot width must be 1, 2 (MIN / MAX, columnar), 4 or 8");
    if (tg.agg == HDK_AGG_ID && p->query_kind != HDK_Q_NON_GROUPED) {
      // a group-by plan writes a non-aggregate target from key #key_idx (hdk_finalize, the global kernels)
      HDK_REQUIRE(tg.key_idx >= 0 && tg.key_idx < p->key_count,
                  "target %d: HDK_AGG_ID in a group-by plan needs 0 <= key_idx < key_count (got %d)", t, tg.key_idx);
    }
    HDK_REQUIRE(tg.arg_is_fp >= HDK_FP_SLOT_NONE && tg.arg_is_fp <= HDK_FP_SLOT_FLOAT, "target %d: arg_is_fp out of range", t);
    if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) {
      HDK_REQUIRE(tg.agg == HDK_AGG_SUM || tg.agg == HDK_AGG_MIN || tg.agg == HDK_AGG_MAX || tg.agg == HDK_AGG_AVG ||
                      tg.agg == HDK_AGG_SINGLE_VALUE,
                  "target %d: a float accumulator belongs to SUM / MIN / MAX / AVG / SINGLE_VALUE", t);
      HDK_REQUIRE(!exprs || (tg.has_arg && tg.arg.vclass == HDK_VC_FP), "target %d: a float accumulator needs a floating-point argument", t);
    } else if (tg.slot_width == 4 && tg.arg_is_fp && tg.agg != HDK_AGG_COUNT) {
      set_error("a 4-byte slot cannot hold a double: float arguments use arg_is_fp = HDK_FP_SLOT_FLOAT");
      return HDK_HIP_ERR_UNSUPPORTED;
    }
  }
  if (p->query_kind == HDK_Q_PERFECT_HASH) {
    HDK_REQUIRE(p->key_width == 8, "perfect hash uses 8-byte keys");
  }
  if (p->output_columnar) {
    HDK_REQUIRE(p->key_width == 8, "columnar output uses 8-byte keys");
  }
  return HDK_HIP_OK;
}

int32_t validate_plan(const hdk_hip_plan* p) { return validate_plan_impl(p, true); }
int32_t validate_plan_layout(const hdk_hip_plan* p) { return validate_plan_impl(p, false); }

struct FastArgs;
static bool match_fast(const hdk_hip_plan* p, const LaunchShape& shape, FastArgs* fa, int* kw_out, int* vw_out);
static bool match_keys(const hdk_hip_plan* p, const LaunchShape& shape, KeysArgs* ka);
static bool match_cols(const hdk_hip_plan* p, const LaunchShape& shape, ColsArgs* ca);
static bool match_keys_values(const hdk_hip_plan* p, KeysArgs* ka);
static bool match_join_direct(const hdk_hip_plan* p, const LaunchShape& shape, JoinDirectArgs* ja);
// which of the join kernels takes a plan (one place: the launch, its description and its grid agree)
enum JoinRoute { JOIN_ROUTE_NONE = 0, JOIN_ROUTE_DIRECT = 1, JOIN_ROUTE_SLICED2 = 2 };
static JoinRoute route_join(const hdk_hip_plan* plan, const LaunchShape& shape, const hdk_hip_kernel_options* ko, JoinDirectArgs* ja,
                            Slice2Args* ga);
static bool join_direct_clusters(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko);
// the counting form keeps 32-bit counters: half the bytes of the 8-byte words the shape was sized for
static uint32_t keys_lds_bytes(const KeysArgs& ka, const LaunchShape& shape) {
  return ka.nvals ? shape.lds_bytes : shape.lds_bytes / 2;
}
// counting / value form, rows dealt 2 (some 8-byte column is read) or 4 at a time
static const void* keys_kernel(const KeysArgs& ka, uint32_t block = kKeysBlock) {
  int wmax = 1;
  for (int k = 0; k < ka.nkeys; ++k) wmax = ka.key[k].width > wmax ? ka.key[k].width : wmax;
  for (int v = 0; v < ka.nvals; ++v) wmax = ka.val[v].width > wmax ? ka.val[v].width : wmax;
  if (ka.nvals && block == kKeysWideBlock) {
    return wmax == 8 ? reinterpret_cast<const void*>(hdk_scan_agg_keys<true, 2, kKeysWideBlock>)
                     : reinterpret_cast<const void*>(hdk_scan_agg_keys<true, 4, kKeysWideBlock>);
  }
  if (ka.nvals) {
    return wmax == 8 ? reinterpret_cast<const void*>(hdk_scan_agg_keys<true, 2>)
                     : reinterpret_cast<const void*>(hdk_scan_agg_keys<true, 4>);
  }
  return wmax == 8 ? reinterpret_cast<const void*>(hdk_scan_agg_keys<false, 2>)
                   : reinterpret_cast<const void*>(hdk_scan_agg_keys<false, 4>);
}


// arms a launch's own interrupt / watchdog words (watch.h)
__global__ void k_arm_watch(LaunchWatch* w, uint32_t flags, uint32_t watchdog_ms) {
  w->flags = flags;
  w->deadline = __builtin_amdgcn_s_memrealtime() + static_cast<uint64_t>(watchdog_ms) * 100000ull;
}

LaunchShape choose_shape(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko,
                         const hdk_hip_device_properties* props) {
  LaunchShape s;
  WordLayout wl;
  make_word_layout(p, &wl);
  s.wpe = wl.wpe;
  s.entry_count = p->query_kind == HDK_Q_NON_GROUPED ? 1u : p->entry_count;
  s.grid = (ko && ko->grid_dim_x) ? ko->grid_dim_x : static_cast<uint32_t>(props->grid_size);
  s.block = kBlock;
  bool force_global = ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS);
  for (int t = 0; t < p->num_targets; ++t) {
    // SINGLE_VALUE lives on the final table only (checked_single_agg_id_shared): no per-block images to fold
    force_global = force_global || p->targets[t].agg == HDK_AGG_SINGLE_VALUE;
  }
  const uint64_t words = static_cast<uint64_t>(s.entry_count) * s.wpe;
  if (p->query_kind == HDK_Q_PROJECTION) {
    s.strategy = STRAT_PROJECT;
    s.rep = 1;
    s.lds_bytes = 0;
    s.slab_words = 0;
  } else if (p->query_kind != HDK_Q_BASELINE_HASH && words <= kLdsMaxTableWords && !force_global) {
    s.strategy = STRAT_LDS;
    uint32_t rep = 32;
    while (rep > 1 && words * rep > kLdsWordBudget) {
      rep >>= 1;
    }
    s.rep = rep;
    s.lds_bytes = static_cast<uint32_t>(words * rep * 8);
    s.slab_words = words;
  } else {
    s.strategy = STRAT_GLOBAL;
    s.rep = 1;
    s.lds_bytes = 0;
    s.slab_words = 0;
  }
  FastArgs fa;
  int kw = 0, vw = 0;
  if (!(ko && ko->grid_dim_x) && match_fast(p, s, &fa, &kw, &vw)) {
    // Blocks per CU of the streaming kernel, measured at 256 M rows (grid sweep 256..2048 blocks):
    //   16 B/row, one LDS atomic per row (C2), or no key (C1): 2 per CU is best (C2 @1e9 rows: 512
    //     blocks 2.60 ms, 1024 blocks 2.71 ms -- fewer slabs to fold, same bytes in flight);
    //   narrow rows carry more LDS atomics per byte streamed and want more waves to hide them:
    //     COUNT(*) by a 4-byte key (taxi Q1) 4 per CU (4.4 -> 5.6 TB/s), 2-byte key + AVG (taxi Q2)
    //     3 per CU.
    const uint32_t cu = static_cast<uint32_t>(props->num_cu);
    const bool static_ops = fa.nops == 1 || ((fa.nops == 2 || fa.nops == 3) && fa.op_kind[0] == FOP_ADD_ONE &&
                                             (fa.op_kind[1] == FOP_ADD_U64 || fa.op_kind[1] == FOP_ADD_F64));
    if (fa.nxq || fa.vform) {
      s.grid = 3u * cu;  // X mode (24-32 B/row over three or four streamed columns): c2x at 256 M rows 768 blocks 0.960 ms, 1024 1.054, 1280 1.079, 1536 1.035
    } else if (fa.nquals) {
      s.grid = 4u * cu;  // filtered: the filter-column gathers want more waves (C2 + WHERE: 1.09 vs 1.14 ms at 2 per CU)
    } else if (kw == 0 || kw + vw >= 16) {
      s.grid = 2u * cu;
    } else if (vw == 0) {
      s.grid = 4u * cu;
    } else if (static_ops) {
      s.grid = 3u * cu;  // taxi Q2 with the compile-time op list: 768 blocks 0.419 ms, 512 0.507, 1024 0.440
    } else {
      s.grid = 8u * cu;  // run-time op list (scalar dispatch per op): more waves hide it
    }
    const uint32_t lds_limited = static_cast<uint32_t>((160u * 1024u) / (s.lds_bytes + 1024u)) * cu;
    if (lds_limited >= cu && s.grid > lds_limited) {
      s.grid = lds_limited;  // a large unreplicated table: fewer blocks fit per CU, keep them all resident
    }
  } else if (ColsArgs cols_args; !(ko && ko->grid_dim_x) && !(ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR))) &&
             match_cols(p, s, &cols_args)) {
    // one column streamed at a time, kColsU x 16 bytes in flight per lane: two 256-thread blocks per CU carry 64 KB per CU
    const char* per_cu = hdk_sw(SW_COLS_BLOCKS_PER_CU);  // (A/B measurements)
    s.grid = (per_cu && atoi(per_cu) > 0 ? static_cast<uint32_t>(atoi(per_cu)) : kColsBlocksPerCu) * static_cast<uint32_t>(props->num_cu);
  } else if (!(ko && ko->grid_dim_x)) {
    const bool scalar = (ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_SCALAR)) || needs_join_loops(p);
    const bool generic = ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR));
    const void* k;
    int block = kBlock;
    size_t lds_for_occupancy = s.lds_bytes;
    KeysArgs ka;
    if (s.strategy == STRAT_LDS && !generic && match_keys(p, s, &ka)) {
      lds_for_occupancy = keys_lds_bytes(ka, s);
      block = kKeysBlock;
      // value form with a table that leaves room for fewer than four blocks per CU: twice the threads per table
      if (ka.nvals && (160u * 1024u) / (lds_for_occupancy + 1024u) < 4 && !hdk_sw(SW_KEYS_NO_WIDE_BLOCK)) {
        block = kKeysWideBlock;
      }
      s.block = static_cast<uint32_t>(block);
      k = keys_kernel(ka, s.block);
    } else if (s.strategy == STRAT_LDS) {
      k = scalar ? reinterpret_cast<const void*>(hdk_scan_agg_generic)
                 : (p->num_joins ? (plan_is_single_matching_set_join(p) ? reinterpret_cast<const void*>(hdk_scan_agg_vec_many)
                                    : plan_has_keyed_join(p)            ? reinterpret_cast<const void*>(hdk_scan_agg_vec_keyed)
                                                                        : reinterpret_cast<const void*>(hdk_scan_agg_vec_join))
                                 : reinterpret_cast<const void*>(hdk_scan_agg_vec));
      block = scalar ? kBlock : kVecBlock;
    } else if (s.strategy == STRAT_PROJECT) {
      s.grid = project_grid(p, ko, props);
      return s;
    } else {
      s.grid = baseline_grid(p, ko, props);
      return s;
    }
    s.grid = resident_grid(k, block, lds_for_occupancy, props);
  }
  return s;
}

size_t workspace_bytes_for(const LaunchShape& s) {
  return kPlanRegionBytes + static_cast<size_t>(s.grid) * s.slab_words * 8;
}

// ---- optional per-launch timing of the scan kernel (HDK_HIP_LAUNCH_RECORD_EVENTS) -----------------
// HIP events recorded on the launch stream right around the dominant kernel; bench.py collects
// them after its final synchronize (the roofline's `achieved` is algorithmic bytes / this).
struct ScanEventLog {
  std::mutex mu;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
static ScanEventLog g_scan_events[16];

int32_t scan_events_begin(int32_t device_id, hipStream_t s, hipEvent_t* e0, hipEvent_t* e1) {
  HDK_HIP_CHECK(hipEventCreate(e0));
  HDK_HIP_CHECK(hipEventCreate(e1));
  HDK_HIP_CHECK(hipEventRecord(*e0, s));
  std::lock_guard<std::mutex> lk(g_scan_events[device_id].mu);
  g_scan_events[device_id].pending.emplace_back(*e0, *e1);
  return HDK_HIP_OK;
}

// ---- fast-path matcher: plan -> FastArgs (scan_agg_fast.h) -----------------------------------------

static bool match_fast(const hdk_hip_plan* p, const LaunchShape& shape, FastArgs* fa, int* kw_out, int* vw_out) {
  if (shape.strategy != STRAT_LDS || p->num_joins || p->key_count > 1) return false;
  if (p->query_kind == HDK_Q_BASELINE_HASH) return false;
  int kw = 0;
  memset(fa, 0, sizeof(*fa));
  if (p->key_count == 1) {
    int kc;
    if (!plain_outer_col(p, p->keys[0], &kc)) return false;
    const hdk_hip_col& c = p->cols[kc];
    if (c.kind != HDK_COL_INT || p->key_bucket[0] > 1) return false;
    kw = c.width;
    fa->key_buf_idx = c.buf_idx;
    fa->key_min = p->key_min[0];
    fa->key_null = p->keys[0].null_val;
    fa->key_null_translated = p->key_null_translated[0];
    fa->key_translate_null = p->key_has_nulls[0] && p->keys[0].nullable;
    if (kw <= 4) {  // the kernel does a narrow key's arithmetic in 32 bits (fast_row): the rare plan that cannot goes to the interpreter
      constexpr int64_t kLim = 1ll << 30;
      if (fa->key_min < -kLim || fa->key_min > kLim) return false;
      if (fa->key_translate_null && fa->key_null_translated != static_cast<int32_t>(fa->key_null_translated)) return false;
    }
  }
  int vw = 0;
  bool need_real_rowcount = false, need_real_nn = false;
  const hdk_hip_expr* vexpr = nullptr;  // every aggregate argument is this one expression: column a, a op column b, a op literal
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_ID) {
      if (tg.key_idx != 0) return false;
      continue;
    }
    if (!tg.has_arg) {
      if (tg.agg != HDK_AGG_COUNT) return false;
      need_real_rowcount = true;
      continue;
    }
    const hdk_hip_expr& e = tg.arg;
    if (vexpr && memcmp(vexpr, &e, sizeof(e)) != 0) return false;
    vexpr = &e;
    if (e.nsteps > 1 || e.leaf0.kind != HDK_LEAF_COL || p->cols[e.leaf0.col].table != 0) return false;
    const int c = e.leaf0.col;

    const hdk_hip_col& col = p->cols[c];
    const bool fp = col.kind == HDK_COL_DOUBLE;
    if (col.kind != HDK_COL_INT && !fp) return false;
    if (tg.agg != HDK_AGG_COUNT && (tg.arg_is_fp != 0) != fp) return false;  // no int->fp promotion here
    if (col.width != 4 && col.width != 8) return false;
    const int nullable = tg.skip_null && tg.arg.nullable;
    if (vw && fa->val_nullable != nullable) return false;
    vw = col.width;
    fa->val_buf_idx = col.buf_idx;
    fa->val_null = tg.arg.null_val;
    fa->val_nullable = nullable;
    fa->val_is_fp = fp;
    if (e.nsteps == 1) {
      // a op b / a op literal over 8-byte integer columns (the Q instantiations: fast_value_expr)
      const hdk_hip_step& sp = e.steps[0];
      if (fp || col.width != 8 || tg.arg_is_fp || sp.out_class != HDK_VC_INT) return false;
      if (sp.op != HDK_OP_ADD && sp.op != HDK_OP_SUB && sp.op != HDK_OP_MUL) return false;
      // a NULL operand gives the step's NULL, which must be what the target skips (eval_target_arg: arg.null_val, and the
      // slot's own sentinel for a value that collides with it)
      if (tg.skip_null && (e.null_val != sp.null_out || tg.null_val != sp.null_out)) return false;
      if (!tg.skip_null && (e.leaf0.nullable || sp.rhs.nullable)) return false;
      fa->a_nullable = e.leaf0.nullable;
      fa->a_null = e.leaf0.null_val;
      fa->vop = sp.op;
      fa->v_check_width = sp.check_width;
      fa->v_null_out = sp.null_out;
      if (sp.rhs.kind == HDK_LEAF_INT) {
        fa->vform = 2;
        fa->v_lit = sp.rhs.ival;
      } else if (sp.rhs.kind == HDK_LEAF_COL) {
        const hdk_hip_col& bc = p->cols[sp.rhs.col];
        if (bc.table != 0 || bc.kind != HDK_COL_INT || bc.width != 8 || sp.rhs.col == c) return false;
        fa->vform = 1;
        fa->x_buf_idx[0] = bc.buf_idx;  // extra streamed column 0
        fa->nx = 1;
        fa->b_src = 1;
        fa->b_nullable = sp.rhs.nullable;
        fa->b_null = sp.rhs.null_val;
      } else {
        return false;
      }
    }
    const bool counts_rows = tg.agg == HDK_AGG_COUNT || tg.agg == HDK_AGG_AVG;
    if (counts_rows && target_has_nn_word(tg)) need_real_nn = true;
    if (counts_rows && !target_has_nn_word(tg)) need_real_rowcount = true;
  }
  if (kw == 0 && vw == 0) return false;
  WordLayout wl;
  make_word_layout(p, &wl);
  fa->wpe = wl.wpe;
  for (int w = 0; w < wl.wpe; ++w) fa->wop[w] = wl.wop[w];
  fa->entry_count = shape.entry_count;
  fa->rep = shape.rep;
  fa->mask_mode = shape.entry_count <= 64 && !need_real_rowcount && !need_real_nn;
  int nops = 0;
  auto push = [&](int kind, int word) {
    if (nops < kFastMaxOps) {
      fa->op_kind[nops] = kind;
      fa->op_word[nops] = word;
    }
    ++nops;
  };
  if (!fa->mask_mode) push(FOP_ADD_ONE, 0);
  for (int t = 0; t < p->num_targets; ++t) {
    if (wl.vword[t] >= 0) {
      int kind;
      switch (wl.wop[wl.vword[t]]) {
        case WOP_ADD_U64: kind = FOP_ADD_U64; break;
        case WOP_ADD_F64: kind = FOP_ADD_F64; break;
        case WOP_MIN_I64: kind = FOP_MIN_I64; break;
        case WOP_MAX_I64: kind = FOP_MAX_I64; break;
        case WOP_MIN_F64: kind = FOP_MIN_F64; break;
        default: kind = FOP_MAX_F64; break;
      }
      push(kind, wl.vword[t]);
    }
    if (wl.nword[t] >= 0) {
      if (fa->mask_mode) {
        fa->nn_words[fa->n_nn_words++] = wl.nword[t];
      } else {
        push(FOP_ADD_ONE_IF_NULL, wl.nword[t]);
        fa->nword_mask |= 1u << wl.nword[t];
      }
    }
  }
  if (nops > kFastMaxOps) return false;
  fa->nops = nops;
  if (p->num_quals || fa->vform) {
    // filtered plans / expression arguments: only the instantiations scan_fast.hip has (grouped, 8-byte value column)
    if (kw == 0 || vw != 8) return false;
    // X mode: every filter operand is the value column, the (8-byte) key column, one of <= 2 other 8-byte integer columns
    // -- streamed with 16-byte loads beside the value column -- or an integer literal
    FastArgs x = *fa;
    // (filters against literals only, no expression: the gathered form -- any column width, and measured faster for that
    // shape: C2 + WHERE c < 0 at 256 M rows 0.98 ms against 1.10 ms with c streamed)
    bool needs_x = fa->vform != 0;
    for (int i = 0; i < p->num_quals; ++i) needs_x = needs_x || p->quals[i].rhs.kind == HDK_LEAF_COL;
    // an AND / OR / NOT program: in X mode when every operand is a column the kernel streams anyway (the value and the key
    // column: `WHERE val < 0 OR key = 3` reads nothing extra), else over gathered `column cmp literal` leaves
    const bool program = p->num_filter_ops != 0;
    if (program && (p->num_filter_ops > kMaxPlainProg || p->num_quals > kMaxPlainQuals ||
                    filter_program_depth(p) > kPlainProgStack)) return false;
    bool xmode = (needs_x || program) && !fa->val_is_fp && !hdk_sw(SW_FAST_NO_XMODE);
    auto src_of = [&](int col) -> int {
      const hdk_hip_col& cc = p->cols[col];
      if (cc.table != 0 || cc.kind != HDK_COL_INT || cc.width != 8) return -1;
      if (cc.buf_idx == x.val_buf_idx) return 0;
      if (kw == 8 && cc.buf_idx == x.key_buf_idx) return 3;
      for (int i = 0; i < x.nx; ++i) {
        if (x.x_buf_idx[i] == cc.buf_idx) return 1 + i;
      }
      if (x.nx == 2) return -1;
      x.x_buf_idx[x.nx] = cc.buf_idx;
      return 1 + x.nx++;
    };
    for (int i = 0; xmode && i < p->num_quals; ++i) {
      const hdk_hip_qual& q = p->quals[i];
      FastArgs::XQual& xq = x.xq[i];
      if (i >= kMaxPlainQuals || q.lhs.nsteps || q.lhs.leaf0.kind != HDK_LEAF_COL) { xmode = false; break; }
      xq.lhs_src = src_of(q.lhs.leaf0.col);
      xq.lhs_nullable = q.lhs.leaf0.nullable;
      xq.lhs_null = q.lhs.leaf0.null_val;
      xq.cmp = q.cmp;
      if (q.rhs.kind == HDK_LEAF_INT) {
        xq.rhs_src = 4;
        xq.lit = q.rhs.ival;
      } else if (q.rhs.kind == HDK_LEAF_COL) {
        xq.rhs_src = src_of(q.rhs.col);
        xq.rhs_nullable = q.rhs.nullable;
        xq.rhs_null = q.rhs.null_val;
      } else {
        xq.rhs_src = -1;
      }
      if (xq.lhs_src < 0 || xq.rhs_src < 0) xmode = false;
    }
    if (xmode && program && x.nx != 0) xmode = false;  // (the program's X kernel streams no extra column; a third column just
                                                       //  for a literal compare is cheaper gathered anyway)
    if (program && needs_x && !xmode) return false;
    if (xmode) {
      x.nxq = p->num_quals;
      if (program) {
        x.nxprog = p->num_filter_ops;
        for (int i = 0; i < p->num_filter_ops; ++i) x.xprog[i] = p->filter_ops[i];
      }
      *fa = x;
    } else {
      // gathered `column cmp literal` filters (any column width, fp literals): the Q instantiations; no expression argument
      if (fa->vform || !match_plain_quals(p, fa->q, true)) return false;
      fa->nquals = p->num_quals;
    }
  }
  *kw_out = kw;
  *vw_out = vw;
  return true;
}

// ---- column-by-column matcher: plan -> ColsArgs (scan_agg_cols.h) --------------------------------------------------------
// Non-grouped, unfiltered, no join; every target COUNT(*) or an aggregate of a plain outer column (integer or double), at most
// kColsMaxCols distinct columns: NonGroupedAgg/NGA01-05.sql.  (One column with every argument alike is match_fast's.)
static bool match_cols(const hdk_hip_plan* p, const LaunchShape& shape, ColsArgs* ca) {
  if (shape.strategy != STRAT_LDS || p->query_kind != HDK_Q_NON_GROUPED || p->num_joins || p->num_quals || p->num_filter_ops) return false;
  if (hdk_sw(SW_NO_COLS_KERNEL)) return false;
  memset(ca, 0, sizeof(*ca));
  WordLayout wl;
  make_word_layout(p, &wl);
  ca->wpe = wl.wpe;
  for (int w = 0; w < kMaxWordsPerEntry; ++w) {
    ca->wcol[w] = -1;
    ca->wkind[w] = w == 0 ? CW_ROWS : CW_UNUSED;
  }
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg != HDK_AGG_COUNT && tg.agg != HDK_AGG_SUM && tg.agg != HDK_AGG_MIN && tg.agg != HDK_AGG_MAX && tg.agg != HDK_AGG_AVG) return false;
    if (!tg.has_arg) {
      if (tg.agg != HDK_AGG_COUNT) return false;
      continue;
    }
    int c;
    if (!plain_outer_col(p, tg.arg, &c)) return false;
    const hdk_hip_col& col = p->cols[c];
    const bool fp = col.kind == HDK_COL_DOUBLE;
    if (col.kind != HDK_COL_INT && !fp) return false;
    if (fp ? col.width != 8 : (col.width != 1 && col.width != 2 && col.width != 4 && col.width != 8)) return false;
    if (tg.agg != HDK_AGG_COUNT && (tg.arg_is_fp != 0) != fp) return false;  // (no int -> fp promotion, no float accumulators)
    if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) return false;
    const int nullable = tg.skip_null && tg.arg.nullable;
    int ci = -1;
    for (int i = 0; i < ca->ncols; ++i) {
      if (ca->col[i].buf_idx == col.buf_idx) ci = i;
    }
    if (ci < 0) {
      if (ca->ncols == kColsMaxCols) return false;
      ci = ca->ncols++;
      ca->col[ci].buf_idx = col.buf_idx;
      ca->col[ci].width = col.width;
      ca->col[ci].fp = fp;
      ca->col[ci].nullable = nullable;
      ca->col[ci].null_val = tg.arg.null_val;
    } else if (ca->col[ci].nullable != nullable || (nullable && ca->col[ci].null_val != tg.arg.null_val) || ca->col[ci].width != col.width ||
               ca->col[ci].fp != static_cast<int32_t>(fp)) {
      return false;
    }
    if (wl.vword[t] >= 0) {
      ca->wcol[wl.vword[t]] = ci;
      ca->wkind[wl.vword[t]] = (tg.agg == HDK_AGG_SUM || tg.agg == HDK_AGG_AVG) ? CW_SUM : (tg.agg == HDK_AGG_MIN ? CW_MIN : CW_MAX);
    }
    if (wl.nword[t] >= 0) {
      ca->wcol[wl.nword[t]] = ci;
      ca->wkind[wl.nword[t]] = CW_NN;
    }
  }
  return ca->ncols > 0;
}

// (same decisions as launch_scan_lds)
static const char* scan_kernel_name(const hdk_hip_plan* p, const LaunchShape& s, const hdk_hip_kernel_options* ko, bool force_generic,
                                    bool force_scalar) {
  FastArgs fa;
  int kw, vw;
  if (!force_generic && match_fast(p, s, &fa, &kw, &vw)) return "hdk_scan_agg_direct";
  ColsArgs ca;
  if (!force_generic && match_cols(p, s, &ca)) return "hdk_scan_agg_cols";
  JoinDirectArgs ja;
  Slice2Args ga;
  const JoinRoute route = force_generic ? JOIN_ROUTE_NONE : route_join(p, s, ko, &ja, &ga);
  if (route == JOIN_ROUTE_DIRECT) return "hdk_join_agg_direct";
  if (route == JOIN_ROUTE_SLICED2) return "hdk_scan_agg_vec_join";  // (armed behind the sliced passes)
  KeysArgs ka;
  if (!force_generic && match_keys(p, s, &ka)) return ka.nvals ? "hdk_scan_agg_keys_values" : "hdk_scan_agg_keys";
  if (needs_join_loops(p) || force_scalar) return "hdk_scan_agg_generic";
  return p->num_joins ? (plan_is_single_matching_set_join(p) ? "hdk_scan_agg_vec_many"
                         : plan_has_keyed_join(p)            ? "hdk_scan_agg_vec_keyed"
                                                             : "hdk_scan_agg_vec_join")
                      : "hdk_scan_agg_vec";
}

// A GroupByPerfectHash plan of the packed on-chip kernels' shape (scan_bh_packed.h: one plain key column, every aggregate over
// one integer column inside its statistics) that would otherwise run on the batched interpreter or -- its table beyond LDS --
// on the perfect-partitioned / global-atomics kernels: the reference's PerfectHashSingleCol benchmark queries (PHS001-005:
// count, sum, max, min, avg by x10 ... x100k) are this shape; the interpreter ran them at 7 % of the HBM roofline, and 1 000
// groups fell to global atomics (1.5 s per 256 M rows, profiles/r05_bh_configs.txt).
static bool perfect_goes_packed(const hdk_hip_plan* p, const LaunchShape& s, const hdk_hip_kernel_options* ko) {
  if (p->query_kind != HDK_Q_PERFECT_HASH || !bh_packed_kernel_name(p, ko)) return false;
  if (s.strategy == STRAT_GLOBAL) return true;
  if (s.strategy != STRAT_LDS) return false;
  return strcmp(scan_kernel_name(p, s, ko, false, false), "hdk_scan_agg_vec") == 0;
}

// ---- hdk_join_agg_direct (scan_join_direct.h): the matcher ------------------------------------------------------------
static bool jd_leaf_from(const hdk_hip_plan* p, const hdk_hip_leaf& l, int kc, int xc, JdLeaf* out) {
  out->nullable = l.nullable;
  out->null_val = l.null_val;
  out->ival = l.ival;
  if (l.kind == HDK_LEAF_INT) {
    out->kind = JD_LITERAL;
    out->nullable = 0;
    return true;
  }
  if (l.kind != HDK_LEAF_COL) return false;
  const hdk_hip_col& c = p->cols[l.col];
  if (c.table == 0 && l.col == xc && l.col != kc) {
    out->kind = JD_X;
    return true;
  }
  if (c.table == -1 && c.buf_idx == 1) {  // payload word 1 of join 0's fused entry
    out->kind = JD_PAYLOAD;
    return true;
  }
  return false;
}

static bool match_join_direct(const hdk_hip_plan* p, const LaunchShape& shape, JoinDirectArgs* ja) {
  if (shape.strategy != STRAT_LDS || p->query_kind != HDK_Q_NON_GROUPED || p->num_joins != 1 || p->num_quals ||
      p->num_filter_ops || p->num_targets > kJdMaxTargets) {
    return false;
  }
  if (shape.rep == 0 || (shape.rep & (shape.rep - 1))) return false;
  const hdk_hip_join& jn = p->joins[0];
  if (jn.kind != HDK_JOIN_ONE_TO_ONE_FUSED || jn.fused_stride != 2 || !join_type_inner_like(jn.type) || jn.bucket > 1 ||
      jn.null_mode == HDK_JOIN_NULL_BITWISE || jn.table_idx != 0) {
    return false;
  }
  int kc;
  if (!plain_outer_col(p, jn.outer_key, &kc) || p->cols[kc].width != 8 || p->cols[kc].kind != HDK_COL_INT) return false;
  int xc = -1;
  for (int i = 0; i < p->num_cols; ++i) {
    const hdk_hip_col& c = p->cols[i];
    if (c.table == 0) {
      if (i == kc) continue;
      if (xc >= 0 || c.width != 8 || c.kind != HDK_COL_INT) return false;  // one more outer column, 8-byte integer
      xc = i;
    } else if (!(c.table == -1 && c.buf_idx == 1)) {
      return false;  // an inner column read through the row id, or a second payload word
    }
  }
  memset(ja, 0, sizeof(*ja));
  WordLayout wl;
  make_word_layout(p, &wl);
  if (wl.wpe != static_cast<int>(shape.wpe)) return false;
  ja->wpe = wl.wpe;
  for (int w = 0; w < wl.wpe; ++w) ja->wop[w] = wl.wop[w];
  ja->rep = shape.rep;
  ja->key_buf_idx = p->cols[kc].buf_idx;
  ja->x_buf_idx = xc >= 0 ? p->cols[xc].buf_idx : -1;
  ja->min_key = jn.min_key;
  ja->max_key = jn.max_key;
  ja->key_null = jn.null_val;
  ja->key_nullable = jn.null_mode == HDK_JOIN_NULL_NULLABLE;
  ja->ntargets = p->num_targets;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    JdTarget& jt = ja->t[t];
    if (tg.agg == HDK_AGG_ID || tg.slot_width != 8 || tg.arg_is_fp) return false;
    if (tg.agg == HDK_AGG_AVG && tg.slot2_width != 8) return false;
    jt.has_arg = tg.has_arg;
    jt.vword = wl.vword[t];
    jt.nword = wl.nword[t];
    jt.wop = wl.vword[t] >= 0 ? wl.wop[wl.vword[t]] : WOP_ADD_U64;
    if (jt.wop != WOP_ADD_U64 && jt.wop != WOP_MIN_I64 && jt.wop != WOP_MAX_I64) return false;
    if (wl.nword[t] >= 0) ja->nword_mask |= 1u << wl.nword[t];
    if (!tg.has_arg) {
      if (tg.agg != HDK_AGG_COUNT) return false;
      continue;
    }
    const hdk_hip_expr& e = tg.arg;
    if (e.vclass != HDK_VC_INT || e.nsteps > 1 || !jd_leaf_from(p, e.leaf0, kc, xc, &jt.a) || jt.a.kind == JD_LITERAL) return false;
    jt.nsteps = e.nsteps;
    if (e.nsteps == 1) {
      const hdk_hip_step& st = e.steps[0];
      if ((st.op != HDK_OP_ADD && st.op != HDK_OP_SUB && st.op != HDK_OP_MUL) || st.out_class != HDK_VC_INT) return false;
      if (!jd_leaf_from(p, st.rhs, kc, xc, &jt.b)) return false;
      jt.op = st.op;
      jt.check_width = st.check_width;
      jt.step_null = st.null_out;
    }
    // what eval_target_arg tests: after a step the value is nullable with the step's NULL (eval_expr), else the leaf's
    jt.arg_null = e.null_val;
    jt.arg_nullable = e.nullable;
    jt.skip_null = tg.skip_null;
    jt.slot_null = tg.null_val;
  }
  return true;
}

// does the direct join kernel read key-range-clustered tuples?
static bool join_direct_clusters(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko) {
  if (!ko || ko->total_rows == 0 || (ko->flags & HDK_HIP_LAUNCH_NO_CLUSTER_PROBES)) return false;
  // Only on request.  Measured on C3 (256 M rows, 160 MB table): pre-pass 1.96 ms + probes over the clustered tuples
  // 3.19 ms (a quarter of them still miss L2: several key ranges are in flight at once) against 5.08 ms in row order --
  // no gain yet; with the table L2-resident this kernel needs 2.06 ms (the interpreter 3.09).
  (void)p;
  return (ko->flags & HDK_HIP_LAUNCH_CLUSTER_PROBES) != 0;
}

// ---- hdk_join_agg_sliced (scan_join_sliced.h): key-range slices of the join table probed out of LDS ----------------
// Taken by default for the hdk_join_agg_direct shape when the table is far larger than the caches and the column
// statistics let payload and key offset travel in 32 bits; HDK_HIP_LAUNCH_CLUSTER_PROBES takes it whatever the sizes
// (tests), HDK_HIP_LAUNCH_NO_CLUSTER_PROBES never.
static bool match_join_sliced(const hdk_hip_plan* p, const JoinDirectArgs& ja, const hdk_hip_kernel_options* ko, uint32_t grid,
                              SliceArgs* sa) {
  if (!ko || ko->total_rows == 0 || (ko->flags & HDK_HIP_LAUNCH_NO_CLUSTER_PROBES)) return false;
  const bool forced = (ko->flags & HDK_HIP_LAUNCH_CLUSTER_PROBES) != 0;
  const hdk_hip_join& jn = p->joins[0];
  if (jn.max_key < jn.min_key) return false;
  const uint64_t range = static_cast<uint64_t>(jn.max_key - jn.min_key) + 1;
  if (range >= 0xFFFFFFFFull) return false;
  // worth it when a probe in row order costs a memory line: table beyond the 4 MB L2s and the 256 MB Infinity Cache's
  // comfortable share, enough rows to pay for the scatter pass
  if (!forced && (range * 16 < (32ull << 20) || ko->total_rows < (32ull << 20))) return false;
  uint32_t slice = static_cast<uint32_t>((range + kSliceMaxBins - 1) / kSliceMaxBins);
  if (slice < 64) slice = 64;
  if (slice > kSliceMaxEntries) return false;  // > 10.2 M keys: a slice no longer fits the LDS of a CU
  memset(sa, 0, sizeof(*sa));
  // payload: in 32 bits with two values to spare (column statistics of the fused inner column)
  const hdk_hip_col* pay = nullptr;
  const hdk_hip_col* xcol = nullptr;
  for (int i = 0; i < p->num_cols; ++i) {
    if (p->cols[i].table == -1 && p->cols[i].buf_idx == 1) pay = &p->cols[i];
    if (p->cols[i].table == 0 && p->cols[i].buf_idx == ja.x_buf_idx) xcol = &p->cols[i];
  }
  if (pay && !(pay->has_stats && pay->min_val > static_cast<int64_t>(INT32_MIN) + 1 && pay->max_val <= static_cast<int64_t>(INT32_MAX))) {
    return false;
  }
  sa->key_buf_idx = ja.key_buf_idx;
  sa->x_buf_idx = ja.x_buf_idx;
  sa->key_min = jn.min_key;
  sa->key_range = range;
  sa->key_nullable = ja.key_nullable;
  sa->key_null = ja.key_null;
  sa->slice = slice;
  magic_u32(slice, &sa->slice_magic, &sa->slice_shift);
  sa->nbins = static_cast<uint32_t>((range + slice - 1) / slice);
  // every block of the probe pass flushes into slab[blockIdx.x], and the workspace holds (and hdk_finalize folds) as many
  // slabs as the launch shape's grid -- which the caller may have made small (KernelOptions::gridDimX): one block per
  // slice at least, or the slices are not used
  if (sa->nbins > grid) return false;
  // the NULL sentinels of x and of the payload, as the targets' leaves name them
  for (int t = 0; t < ja.ntargets; ++t) {
    for (const JdLeaf* l : {&ja.t[t].a, &ja.t[t].b}) {
      if (!ja.t[t].has_arg) continue;
      if (l->kind == JD_X && l->nullable) sa->x_null = l->null_val, sa->x_null32 = 1;
      if (l->kind == JD_PAYLOAD && l->nullable) sa->pay_null = l->null_val, sa->pay_nullable = 1;
    }
  }
  // x: 32 bits when the statistics allow (a nullable column gives up INT32_MIN for its NULL), else 16-byte tuples
  sa->narrow = 1;
  if (ja.x_buf_idx >= 0) {
    const bool x_has_nulls = !xcol || !xcol->has_stats || xcol->has_nulls;
    if (!(xcol && xcol->has_stats && xcol->min_val >= static_cast<int64_t>(INT32_MIN) + (x_has_nulls ? 1 : 0) &&
          xcol->max_val <= static_cast<int64_t>(INT32_MAX)) || hdk_sw(SW_SLICE_WIDE)) {
      sa->narrow = 0;
    }
    if (!x_has_nulls) {
      // no NULLs announced: INT32_MIN is an ordinary value; a NULL showing up anyway (the leaf is nullable, x_null is set)
      // is caught as stale by the scatter pass
      sa->x_null_is_stale = sa->x_null32;
      sa->x_null32 = 0;
    }
    else if (sa->narrow && !sa->x_null32) sa->narrow = 0;  // NULLs possible but no sentinel known from the leaves
  } else {
    sa->x_null32 = 0;
  }
  // the FAST form of the probe pass: ONE target, SUM(x + payload) in either order, into an ADD word
  if (ja.ntargets == 1 && ja.t[0].has_arg && ja.t[0].nsteps == 1 && ja.t[0].op == HDK_OP_ADD && ja.t[0].vword >= 0 &&
      ja.t[0].wop == WOP_ADD_U64 && !hdk_sw(SW_SLICE_GENERAL)) {
    const JdLeaf &la = ja.t[0].a, &lb = ja.t[0].b;
    const JdLeaf* lx = la.kind == JD_X ? &la : (lb.kind == JD_X ? &lb : nullptr);
    const JdLeaf* lp = la.kind == JD_PAYLOAD ? &la : (lb.kind == JD_PAYLOAD ? &lb : nullptr);
    if (lx && lp && lx != lp) {
      sa->fast = 1;
      sa->fast_x_nullable = lx->nullable;
      sa->fast_x_null = lx->null_val;
      sa->fast_p_nullable = lp->nullable;
      sa->fast_p_null = lp->null_val;
    }
  }
  const uint64_t rows = ko->total_rows;
  const uint64_t nsub = static_cast<uint64_t>(sa->nbins) * kSliceXcds;
  sa->sub = ((rows / nsub) * 17 / 16 + 4096 + 15) & ~15ull;
  sa->cap_ovf = rows / 8 + 4096;
  if (sa->sub > 0xFFFFFFF0ull || sa->cap_ovf > 0xFFFFFFF0ull) return false;  // 32-bit cursors
  return true;
}

template <bool NARROW, bool FAST>
static int32_t launch_sliced_kernels(SliceArgs& sa, const LaunchShape& shape, const hdk_hip_device_properties* props, hipStream_t s,
                                     bool* launched) {
  *launched = false;
  const void* kagg = reinterpret_cast<const void*>(hdk_join_agg_sliced<NARROW, FAST>);
  const size_t lds_agg = static_cast<size_t>(shape.wpe) * shape.rep * 8 + static_cast<size_t>(sa.slice) * 4;
  if (lds_agg > 48 * 1024 &&
      hipFuncSetAttribute(kagg, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_agg)) != hipSuccess) {
    (void)hipGetLastError();
    return HDK_HIP_OK;  // no block this large here: probe in row order
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kagg, kSliceAggBlock, lds_agg) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    return HDK_HIP_OK;
  }
  constexpr int VR = NARROW ? 8 : 4;
  constexpr int TW = NARROW ? 1 : 2;
  const size_t lds_sc = static_cast<size_t>(kSliceBlock) * VR * TW * 8 + static_cast<size_t>(kSliceBlock) * VR + 16;
  const unsigned g_sc = scatter_grid(reinterpret_cast<const void*>(hdk_join_scatter_slices<NARROW>), kSliceBlock, lds_sc, props, 2);
  uint32_t members = static_cast<uint32_t>(per_cu) * static_cast<uint32_t>(props->num_cu) / sa.nbins;
  if (members < 1) members = 1;
  if (sa.nbins * members > shape.grid) members = shape.grid / sa.nbins;  // slab[blockIdx.x] must exist (match_join_sliced: nbins <= grid)
  if (members < 1) return HDK_HIP_OK;
  hipLaunchKernelGGL(hdk_join_order_probe, dim3(256), dim3(256), 0, s, sa);
  hipLaunchKernelGGL(hdk_join_scatter_slices<NARROW>, dim3(g_sc), dim3(kSliceBlock), lds_sc, s, sa);
  hipLaunchKernelGGL((hdk_join_agg_sliced<NARROW, FAST>), dim3(sa.nbins * members), dim3(kSliceAggBlock), lds_agg, s, sa);
  HDK_HIP_CHECK(hipGetLastError());
  *launched = true;
  return HDK_HIP_OK;
}

static int32_t launch_join_direct(const hdk_hip_plan* plan, JoinDirectArgs ja, const hdk_hip_kernel_options* ko,
                                  const LaunchShape& shape, const hdk_hip_device_properties* props, hipStream_t s) {
  AsyncScratch scratch(s);
  const hdk_hip_join& jn = plan->joins[0];
  SliceArgs sa;
  if (match_join_sliced(plan, ja, ko, shape.grid, &sa)) {
    // key-range slices probed out of LDS; the row-order kernel below stays armed for what the slices cannot carry
    auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
    const size_t tw = sa.narrow ? 1 : 2;
    const size_t nsub = static_cast<size_t>(sa.nbins) * kSliceXcds;
    const size_t b_tuples = up((nsub * sa.sub + sa.cap_ovf) * tw * 8);
    const size_t b_fill = up((nsub * kSliceCursorStride + 8) * sizeof(uint32_t));
    if (hipMallocAsync(&scratch.p, b_tuples + b_fill, s) == hipSuccess) {
      int8_t* q = static_cast<int8_t*>(scratch.p);
      sa.kp = ja.kp;
      sa.tuples = reinterpret_cast<int64_t*>(q);
      sa.fill = reinterpret_cast<uint32_t*>(q + b_tuples);
      sa.fill_ovf = sa.fill + nsub * kSliceCursorStride;
      sa.mode = sa.fill_ovf + 1;
      sa.probe = sa.mode + 1;
      sa.jd = ja;
      sa.num_slabs = shape.grid;
      HDK_HIP_CHECK(hipMemsetAsync(sa.fill, 0, b_fill, s));
      bool launched = false;
      const int32_t st = sa.narrow ? (sa.fast ? launch_sliced_kernels<true, true>(sa, shape, props, s, &launched)
                                              : launch_sliced_kernels<true, false>(sa, shape, props, s, &launched))
                                   : (sa.fast ? launch_sliced_kernels<false, true>(sa, shape, props, s, &launched)
                                              : launch_sliced_kernels<false, false>(sa, shape, props, s, &launched));
      if (st) return st;
      if (launched) {
        ja.run_if = sa.mode;
      }
    } else {
      (void)hipGetLastError();  // no scratch: probe in row order
      scratch.p = nullptr;
    }
  } else if (join_direct_clusters(plan, ko) && static_cast<uint64_t>(jn.max_key - jn.min_key) < 0xFFFFFFFFull) {
    ClusterArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.kp = ja.kp;
    ca.aos = 1;
    ca.ncols = ja.x_buf_idx >= 0 ? 2 : 1;
    ca.buf_idx[0] = ja.key_buf_idx;
    ca.buf_idx[1] = ja.x_buf_idx;
    ca.key_min = jn.min_key;
    ca.key_range = static_cast<uint64_t>(jn.max_key - jn.min_key) + 1;
    ca.bin_mult = (static_cast<uint64_t>(kClusterBins) << 32) / ca.key_range;
    const uint64_t rows = ko->total_rows;
    ca.sub = ((rows / (kClusterBins * kClusterXcds)) * 17 / 16 + 4096 + 15) & ~15ull;
    ca.cap_ovf = rows;
    const uint64_t nsub = static_cast<uint64_t>(kClusterBins) * kClusterXcds;
    auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
    const size_t b_tuples = up((nsub * ca.sub + ca.cap_ovf) * ca.ncols * 8);
    const size_t b_fill = up((nsub * kClusterCursorStride + 4) * sizeof(uint32_t));
    if (hipMallocAsync(&scratch.p, b_tuples + b_fill, s) == hipSuccess) {
      int8_t* q = static_cast<int8_t*>(scratch.p);
      ca.out[0] = reinterpret_cast<int64_t*>(q);
      ca.fill = reinterpret_cast<uint32_t*>(q + b_tuples);
      ca.fill_ovf = ca.fill + nsub * kClusterCursorStride;
      HDK_HIP_CHECK(hipMemsetAsync(ca.fill, 0, b_fill, s));
      const size_t lds = static_cast<size_t>(kClusterTile) * ca.ncols * 8 + kClusterTile + 16;
      const unsigned grid = resident_grid(reinterpret_cast<const void*>(hdk_cluster_by_key), kClusterBlock, lds, props);
      hipLaunchKernelGGL(hdk_cluster_by_key, dim3(grid), dim3(kClusterBlock), lds, s, ca);
      ja.clustered = 1;
      ja.tw = ca.ncols;
      ja.tuples = ca.out[0];
      ja.fill = ca.fill;
      ja.fill_ovf = ca.fill_ovf;
      ja.sub = ca.sub;
      ja.cap_ovf = ca.cap_ovf;
    } else {
      (void)hipGetLastError();  // no scratch: probe in row order
      scratch.p = nullptr;
    }
  }
  hipLaunchKernelGGL(hdk_join_agg_direct, dim3(shape.grid), dim3(kJdBlock), shape.lds_bytes, s, ja);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;  // (`scratch` is freed here, stream-ordered after the kernels above)
}

// ---- the general sliced join (scan_join_sliced2.h): join + perfect-hash GROUP BY on the joined column / + filters / any
// list of integer aggregates over x and the payload, everything in 8-byte tuples ----------------------------------------
// leaf of an aggregate argument: 0 = x (the one other 8-byte outer column), 1 = the payload, 2 = an integer literal
// (`yc`: the outer column that rides in the tuple's spare bits, if any -- it reads like a third payload word, pidx 2)
static int s2_leaf_kind(const hdk_hip_plan* p, const hdk_hip_leaf& l, int kc, int* xc, int* pidx = nullptr, int yc = -1) {
  if (l.kind == HDK_LEAF_INT) return 2;
  if (l.kind != HDK_LEAF_COL) return -1;
  const hdk_hip_col& c = p->cols[l.col];
  if (c.table == -1 && (c.buf_idx == 1 || c.buf_idx == 2)) {
    if (pidx) *pidx = c.buf_idx - 1;
    return 1;
  }
  if (l.col == yc) {  // (match_join_sliced2 checked it: an integer column of the outer table, 8 or 4 bytes wide)
    if (pidx) *pidx = 2;
    return 1;
  }
  if (c.table != 0 || l.col == kc || c.width != 8 || c.kind != HDK_COL_INT) return -1;
  if (*xc >= 0 && *xc != l.col) return -1;  // one value column travels in the tuple
  *xc = l.col;
  return 0;
}

static bool match_join_sliced2(const hdk_hip_plan* p, const LaunchShape& shape, const hdk_hip_kernel_options* ko, Slice2Args* ga) {
  if (!ko || ko->total_rows == 0 || (ko->flags & HDK_HIP_LAUNCH_NO_CLUSTER_PROBES) || hdk_sw(SW_NO_SLICED2)) return false;
  if (shape.strategy != STRAT_LDS || p->num_joins != 1 || p->num_filter_ops || p->num_targets > HDK_HIP_MAX_TARGETS) return false;
  if (p->query_kind != HDK_Q_NON_GROUPED && p->query_kind != HDK_Q_PERFECT_HASH) return false;
  if (plan_reads_small_dates(p)) return false;
  const bool forced = (ko->flags & HDK_HIP_LAUNCH_CLUSTER_PROBES) != 0;
  const hdk_hip_join& jn = p->joins[0];
  if (jn.kind != HDK_JOIN_ONE_TO_ONE_FUSED || (jn.fused_stride != 2 && jn.fused_stride != 3) || !join_type_inner_like(jn.type) || jn.bucket > 1 ||
      jn.null_mode == HDK_JOIN_NULL_BITWISE || jn.table_idx != 0 || jn.max_key < jn.min_key) {
    return false;
  }
  int kc;
  if (!plain_outer_col(p, jn.outer_key, &kc) || p->cols[kc].width != 8 || p->cols[kc].kind != HDK_COL_INT) return false;
  const uint64_t range = static_cast<uint64_t>(jn.max_key - jn.min_key) + 1;
  if (range >= 0xFFFFFFFFull) return false;
  if (!forced && (range * 16 < (32ull << 20) || ko->total_rows < (32ull << 20))) return false;
  memset(ga, 0, sizeof(*ga));
  SliceArgs& sa = ga->s;
  const int npay = jn.fused_stride - 1;
  const hdk_hip_col* pay[2] = {nullptr, nullptr};
  for (int i = 0; i < p->num_cols; ++i) {
    const hdk_hip_col& c = p->cols[i];
    if (c.table == -1 && c.buf_idx >= 1 && c.buf_idx <= npay) pay[c.buf_idx - 1] = &c;
    else if (c.table != 0) return false;  // an inner column read through the row id, or a third payload word
  }
  ga->npay = npay;
  // ---- a second outer column: y, in the bits of the tuple's low word the key offset leaves free ------------------------------
  // Which one: the group key's column when that is an outer column (GROUP BY fact.g ... WHERE dim.d < c); else, when the
  // aggregates read two outer columns, the one with the narrower statistics.  It must fit: codes 1 .. max - min + 1.
  uint32_t key_bits = 1;
  while (key_bits < 32 && (1ull << key_bits) < range) ++key_bits;
  auto y_fits = [&](int ci) {
    const hdk_hip_col& c = p->cols[ci];
    if (c.table != 0 || ci == kc || (c.width != 8 && c.width != 4) || c.kind != HDK_COL_INT || !c.has_stats || c.max_val < c.min_val) return false;
    if (c.min_val <= static_cast<int64_t>(INT32_MIN) + 2 || c.max_val > static_cast<int64_t>(INT32_MAX)) return false;
    return static_cast<uint64_t>(c.max_val - c.min_val) + 2 <= (1ull << (32 - key_bits));
  };
  int yc = -1;
  if (!hdk_sw(SW_S2_NO_Y)) {
    if (p->query_kind == HDK_Q_PERFECT_HASH && p->key_count == 1 && p->keys[0].leaf0.kind == HDK_LEAF_COL &&
        p->cols[p->keys[0].leaf0.col].table == 0) {
      yc = p->keys[0].leaf0.col;
      if (!y_fits(yc)) return false;
    } else {
      int oc[2] = {-1, -1}, noc = 0;
      auto note = [&](const hdk_hip_leaf& l) {
        if (l.kind != HDK_LEAF_COL || p->cols[l.col].table != 0 || l.col == kc) return true;
        for (int i = 0; i < noc; ++i) {
          if (oc[i] == l.col) return true;
        }
        if (noc == 2) return false;
        oc[noc++] = l.col;
        return true;
      };
      for (int t = 0; t < p->num_targets; ++t) {
        const hdk_hip_target& tg = p->targets[t];
        if (tg.agg == HDK_AGG_ID || !tg.has_arg) continue;
        if (!note(tg.arg.leaf0) || (tg.arg.nsteps >= 1 && !note(tg.arg.steps[0].rhs))) return false;
      }
      if (noc == 2) {
        const bool f0 = y_fits(oc[0]), f1 = y_fits(oc[1]);
        if (!f0 && !f1) return false;
        if (f0 && f1) {
          yc = (p->cols[oc[0]].max_val - p->cols[oc[0]].min_val) <= (p->cols[oc[1]].max_val - p->cols[oc[1]].min_val) ? oc[0] : oc[1];
        } else {
          yc = f0 ? oc[0] : oc[1];
        }
      }
    }
  }
  // ---- targets ----------------------------------------------------------------------------------------------------------
  WordLayout wl;
  make_word_layout(p, &wl);
  if (wl.wpe != static_cast<int>(shape.wpe)) return false;
  ga->wpe = wl.wpe;
  for (int w = 0; w < wl.wpe; ++w) ga->wop[w] = wl.wop[w];
  int xc = -1;
  int nt = 0;
  int64_t x_null = 0, p_null[3] = {0, 0, 0};  // ([2]: y)
  bool x_nullable = false, p_nullable[3] = {false, false, false};
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_ID) {
      if (p->query_kind != HDK_Q_PERFECT_HASH || tg.key_idx != 0) return false;
      continue;  // hdk_finalize rebuilds a projected key from the entry index
    }
    if (tg.agg == HDK_AGG_SINGLE_VALUE || tg.arg_is_fp || tg.slot_width != 8) return false;
    if (tg.agg == HDK_AGG_AVG && tg.slot2_width != 8) return false;
    if (!tg.has_arg) {
      if (tg.agg != HDK_AGG_COUNT) return false;
      continue;  // COUNT(*): the row count word
    }
    if (wl.vword[t] < 0 && wl.nword[t] < 0) continue;  // COUNT(not-null expression): the row count too
    if (nt == kS2MaxTargets) return false;
    S2Target& st = ga->t[nt++];
    st.vword = wl.vword[t];
    st.nword = wl.nword[t];
    st.wop = wl.vword[t] >= 0 ? wl.wop[wl.vword[t]] : WOP_ADD_U64;
    if (st.wop != WOP_ADD_U64 && st.wop != WOP_MIN_I64 && st.wop != WOP_MAX_I64) return false;
    if (wl.nword[t] >= 0) ga->nword_mask |= 1u << wl.nword[t];
    const hdk_hip_expr& e = tg.arg;
    if (e.vclass != HDK_VC_INT || e.nsteps > 1) return false;
    int pi = 0;
    const int ka = s2_leaf_kind(p, e.leaf0, kc, &xc, &pi, yc);
    if (ka != 0 && ka != 1) return false;
    bool nullable = e.leaf0.nullable != 0;
    if (ka == 0 && e.leaf0.nullable) x_nullable = true, x_null = e.leaf0.null_val;
    if (ka == 1 && e.leaf0.nullable) p_nullable[pi] = true, p_null[pi] = e.leaf0.null_val;
    if (ka == 1) st.pidx = pi;
    st.null_if_x = ka == 0 && e.leaf0.nullable;
    st.null_if_p = ka == 1 && e.leaf0.nullable;
    if (e.nsteps == 0) {
      st.src = ka == 0 ? S2_X : S2_P;
    } else {
      const hdk_hip_step& sp = e.steps[0];
      if ((sp.op != HDK_OP_ADD && sp.op != HDK_OP_SUB && sp.op != HDK_OP_MUL) || sp.out_class != HDK_VC_INT) return false;
      if (sp.check_width != 0 && sp.check_width != 8) return false;  // (a 4-byte SQL type can overflow: the interpreter checks)
      int pj = 0;
      const int kb = s2_leaf_kind(p, sp.rhs, kc, &xc, &pj, yc);
      if (kb < 0) return false;
      if (kb == 2) {
        if (sp.rhs.ival <= -(int64_t(1) << 31) || sp.rhs.ival >= (int64_t(1) << 31)) return false;
        st.src = ka == 0 ? S2_X_OP_LIT : S2_P_OP_LIT;
        st.op = sp.op;
        st.lit = sp.rhs.ival;
      } else {
        if (kb == ka) return false;  // x op x, payload op payload
        nullable = nullable || sp.rhs.nullable;
        if (kb == 0 && sp.rhs.nullable) x_nullable = true, x_null = sp.rhs.null_val, st.null_if_x = 1;
        if (kb == 1) st.pidx = pj;
        if (kb == 1 && sp.rhs.nullable) p_nullable[pj] = true, p_null[pj] = sp.rhs.null_val, st.null_if_p = 1;
        if (sp.op == HDK_OP_ADD) st.src = S2_X_ADD_P;
        else if (sp.op == HDK_OP_MUL) st.src = S2_X_MUL_P;
        else st.src = ka == 0 ? S2_X_SUB_P : S2_P_SUB_X;
      }
    }
    // a NULL argument is skipped (and counted) -- the only reading of a NULL the kernel has; a plan that would AGGREGATE
    // the sentinel (nullable leaf, skip_null off) is the interpreter's
    if (nullable && !tg.skip_null) return false;
  }
  ga->ntargets = nt;
  // ---- filters: `outer column cmp literal` in pass 1, `payload cmp literal` in pass 2 ----------------------------------------
  {
    hdk_hip_plan outer_only;
    memset(&outer_only, 0, sizeof(outer_only));
    int nq = 0;
    for (int i = 0; i < p->num_quals; ++i) {
      const hdk_hip_qual& q = p->quals[i];
      if (q.lhs.nsteps == 0 && q.lhs.leaf0.kind == HDK_LEAF_COL && p->cols[q.lhs.leaf0.col].table == -1 &&
          p->cols[q.lhs.leaf0.col].buf_idx >= 1 && p->cols[q.lhs.leaf0.col].buf_idx <= npay) {
        if (q.rhs.kind != HDK_LEAF_INT || ga->npq == kS2MaxPayQuals) return false;
        const int qi = p->cols[q.lhs.leaf0.col].buf_idx - 1;
        if (q.lhs.leaf0.nullable) p_nullable[qi] = true, p_null[qi] = q.lhs.leaf0.null_val;
        ga->pq[ga->npq].pidx = qi;
        ga->pq[ga->npq].cmp = q.cmp;
        ga->pq[ga->npq].rhs = q.rhs.ival;
        ++ga->npq;
      } else {
        if (nq == kMaxPlainQuals) return false;
        outer_only.quals[nq++] = q;
      }
    }
    outer_only.num_quals = nq;
    outer_only.num_filter_ops = 0;
    outer_only.num_cols = p->num_cols;
    memcpy(outer_only.cols, p->cols, sizeof(p->cols));
    if (nq && !match_plain_quals(&outer_only, sa.q)) return false;
    sa.nquals = nq;
  }
  // ---- group key: the payload, or payload / literal; perfect hash without a bucket ----------------------------------------------
  ga->entry_count = shape.entry_count;
  if (p->query_kind == HDK_Q_PERFECT_HASH) {
    if (p->key_count != 1 || p->key_bucket[0] > 1) return false;
    const hdk_hip_expr& ke = p->keys[0];
    int dummy = -1, ki = 0;
    if (ke.nsteps > 1 || s2_leaf_kind(p, ke.leaf0, kc, &dummy, &ki, yc) != 1) return false;
    ga->key_pidx = ki;
    if (ke.leaf0.nullable) p_nullable[ki] = true, p_null[ki] = ke.leaf0.null_val;
    if (ke.nsteps == 1) {
      const hdk_hip_step& sp = ke.steps[0];
      if ((sp.op != HDK_OP_DIV && sp.op != HDK_OP_MOD) || sp.out_class != HDK_VC_INT || sp.rhs.kind != HDK_LEAF_INT || sp.rhs.ival < 1 ||
          sp.rhs.ival > INT32_MAX) {
        return false;
      }
      ga->key_div = static_cast<int32_t>(sp.rhs.ival);
      ga->key_mod = sp.op == HDK_OP_MOD;
      if (ga->key_div >= 2) magic_u32(static_cast<uint32_t>(ga->key_div), &ga->div_magic, &ga->div_shift);
    }
    ga->grouped = 1;
    ga->key_min = p->key_min[0];
    // eval_key: a NULL key takes the translated value when the layout has a slot for NULLs, else stays the NULL sentinel
    const int64_t null_key = (p->key_has_nulls[0] && ke.nullable) ? p->key_null_translated[0] : ke.null_val;
    ga->null_entry = static_cast<int64_t>(static_cast<uint64_t>(null_key) - static_cast<uint64_t>(p->key_min[0]));
  }
  // ---- the 8-byte tuple: payload and x inside 32 bits by the column statistics ------------------------------------------------------
  for (int j = 0; j < npay; ++j) {
    if (pay[j] && !(pay[j]->has_stats && pay[j]->min_val > static_cast<int64_t>(INT32_MIN) + 1 && pay[j]->max_val <= static_cast<int64_t>(INT32_MAX))) return false;
  }
  sa.key_buf_idx = p->cols[kc].buf_idx;
  sa.x_buf_idx = xc >= 0 ? p->cols[xc].buf_idx : -1;
  sa.key_min = jn.min_key;
  sa.key_range = range;
  sa.key_nullable = jn.null_mode == HDK_JOIN_NULL_NULLABLE;
  sa.key_null = jn.null_val;
  sa.pay_null = p_null[0];
  sa.pay_nullable = p_nullable[0];
  ga->pay_null1 = p_null[1];
  ga->pay_nullable1 = p_nullable[1];
  sa.narrow = 1;
  if (yc >= 0) {
    const hdk_hip_col& ycol = p->cols[yc];
    if (ycol.has_nulls && !p_nullable[2]) return false;  // NULLs possible but no sentinel known from the leaves
    sa.y_buf_idx = ycol.buf_idx;
    sa.y_width = ycol.width;
    sa.y_shift = key_bits;
    sa.y_min = ycol.min_val;
    sa.y_codes = static_cast<uint32_t>(ycol.max_val - ycol.min_val) + 2;
    sa.y_nullable = p_nullable[2];
    sa.y_null = p_null[2];
  }
  // two payload words in one LDS word when their statistics leave room (codes: word 0 two reserved, word 1 one)
  if (npay == 2 && pay[0] && pay[1] && !hdk_sw(SW_S2_NO_PACKED_PAIR)) {
    const uint64_t n0 = static_cast<uint64_t>(pay[0]->max_val - pay[0]->min_val) + 3, n1 = static_cast<uint64_t>(pay[1]->max_val - pay[1]->min_val) + 2;
    uint32_t b0 = 1, b1 = 1;
    while (b0 < 32 && (1ull << b0) < n0) ++b0;
    while (b1 < 32 && (1ull << b1) < n1) ++b1;
    if (pay[0]->max_val >= pay[0]->min_val && pay[1]->max_val >= pay[1]->min_val && b0 + b1 <= 32) {
      ga->packed = 1;
      ga->pk_bits0 = b0;
      ga->pk_codes0 = static_cast<uint32_t>(n0);  // codes in all (b0 <= 31: at most 2^31)
      ga->pk_codes1 = static_cast<uint32_t>(n1);
      ga->pk_min0 = pay[0]->min_val;
      ga->pk_min1 = pay[1]->min_val;
    }
  }
  if (xc >= 0) {
    const hdk_hip_col& xcol = p->cols[xc];
    const bool x_has_nulls = !xcol.has_stats || xcol.has_nulls;
    if (!(xcol.has_stats && xcol.min_val >= static_cast<int64_t>(INT32_MIN) + (x_has_nulls ? 1 : 0) && xcol.max_val <= static_cast<int64_t>(INT32_MAX))) return false;
    sa.x_null = x_null;
    sa.x_null32 = x_nullable ? 1 : 0;
    if (!x_has_nulls) {
      sa.x_null_is_stale = sa.x_null32;
      sa.x_null32 = 0;
    } else if (!x_nullable) {
      return false;  // NULLs possible but no sentinel known from the leaves
    }
  }
  // ---- geometry: slices whose payloads fit LDS next to the group table; <= 256 of them: one scatter level, else two ------------
  const uint64_t table_bytes1 = static_cast<uint64_t>(shape.entry_count) * wl.wpe * 8;
  if (table_bytes1 > kS2MaxTableBytes) return false;
  const uint64_t rows = ko->total_rows;
  uint32_t slice = static_cast<uint32_t>((range + kSliceMaxBins - 1) / kSliceMaxBins);
  if (slice < 64) slice = 64;
  const uint64_t kb4 = 4ull * (ga->packed ? 1 : npay);  // LDS bytes per key of a slice
  const bool one_level = static_cast<uint64_t>(slice) * kb4 + table_bytes1 <= kS2LdsBytes && !hdk_sw(SW_SLICE_TWO_LEVELS);
  auto pick_rep = [&](uint32_t keys) {
    uint32_t rep = 1;
    while (rep < 32 && static_cast<uint64_t>(keys) * kb4 + table_bytes1 * (rep * 2) <= kS2LdsBytes && table_bytes1 * (rep * 2) <= 16 * 1024) rep *= 2;
    return rep;
  };
  if (one_level) {
    ga->rep = pick_rep(slice);
    sa.slice = slice;
    magic_u32(slice, &sa.slice_magic, &sa.slice_shift);
    sa.nbins = static_cast<uint32_t>((range + slice - 1) / slice);
    ga->fslice = slice;
    ga->nslices = sa.nbins;
    ga->nsl_par = sa.nbins;
    if (sa.nbins > shape.grid) return false;  // one block (and one slab) per slice at least
  } else {
    // fine slices of <= 32 K keys (128 KB of LDS; less when the group table is big), `fpc` of them per coarse bin
    uint64_t fs = (kS2LdsBytes - table_bytes1) / kb4;
    if (fs > 32768) fs = 32768;
    fs &= ~static_cast<uint64_t>(63);
    if (fs < 1024) return false;
    if (const char* e = hdk_sw(SW_SLICE_FINE_KEYS)) fs = static_cast<uint64_t>(atoi(e)) & ~63ull;  // (tests: small tables, two levels)
    if (fs < 64) fs = 64;
    const uint64_t nslices = (range + fs - 1) / fs;
    const uint64_t fpc = (nslices + kSliceMaxBins - 1) / kSliceMaxBins;
    if (fpc > static_cast<uint64_t>(kSliceMaxBins) || fpc * fs > 0xFFFFFFFFull) return false;
    const uint64_t coarse = fpc * fs;
    ga->two_level = 1;
    ga->fpc = static_cast<uint32_t>(fpc);
    ga->fslice = static_cast<uint32_t>(fs);
    magic_u32(ga->fslice, &ga->fmagic, &ga->fshift);
    sa.slice = static_cast<uint32_t>(coarse);
    magic_u32(sa.slice, &sa.slice_magic, &sa.slice_shift);
    sa.nbins = static_cast<uint32_t>((range + coarse - 1) / coarse);
    ga->nslices = sa.nbins * ga->fpc;  // (the last coarse bin's slices past the key range stay empty)
    ga->rep = pick_rep(ga->fslice);
    ga->cap2 = ((rows / ((range + fs - 1) / fs)) * 17 / 16 + 2048 + 15) & ~15ull;
    if (ga->cap2 > 0xFFFFFFF0ull) return false;
    ga->nsl_par = ga->nslices < shape.grid ? ga->nslices : shape.grid;  // (launch_join_sliced2 lowers it to what is resident)
  }
  const uint64_t nsub = static_cast<uint64_t>(sa.nbins) * kSliceXcds;
  sa.sub = ((rows / nsub) * 17 / 16 + 4096 + 15) & ~15ull;
  sa.cap_ovf = rows / 8 + 4096;
  if (sa.sub > 0xFFFFFFF0ull || sa.cap_ovf > 0xFFFFFFF0ull) return false;  // 32-bit cursors
  return true;
}

// `bh`: the launch belongs to a GroupByBaselineHash plan run through an internal dense table (launch_baseline_sliced_join):
// armed behind the passes is the open-addressing interpreter of that plan, and the slabs are folded into the open-addressing
// table by hdk_bh_fold_dense instead of hdk_finalize.
typedef void (*S2Kernel)(Slice2Args);
template <bool GROUPED, bool HASY>
static S2Kernel s2_agg_kernel_npay(int npay_form) {
  return npay_form == 3 ? hdk_join_agg_sliced2<GROUPED, 3, HASY> : (npay_form == 2 ? hdk_join_agg_sliced2<GROUPED, 2, HASY> : hdk_join_agg_sliced2<GROUPED, 1, HASY>);
}
static S2Kernel s2_agg_kernel(const Slice2Args& ga) {
  const int form = ga.packed ? 3 : (ga.npay == 2 ? 2 : 1);
  const bool y = ga.s.y_shift != 0;
  return ga.grouped ? (y ? s2_agg_kernel_npay<true, true>(form) : s2_agg_kernel_npay<true, false>(form))
                    : (y ? s2_agg_kernel_npay<false, true>(form) : s2_agg_kernel_npay<false, false>(form));
}
typedef void (*S2Scatter)(SliceArgs);
static S2Scatter s2_scatter_kernel(const SliceArgs& sa) {
  if (sa.y_shift && sa.y_width == 4) return sa.nquals ? hdk_join_scatter_slices<true, true, 4> : hdk_join_scatter_slices<true, false, 4>;
  if (sa.y_shift) return sa.nquals ? hdk_join_scatter_slices<true, true, 8> : hdk_join_scatter_slices<true, false, 8>;
  return sa.nquals ? hdk_join_scatter_slices<true, true, 0> : hdk_join_scatter_slices<true, false, 0>;
}

struct BhBehindSliced {
  const hdk_hip_plan* plan;  // the baseline-hash plan (host)
  const hdk_hip_kernel_options* ko;
  BhDenseFold fold;
};
static int32_t launch_join_sliced2(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, Slice2Args& ga,
                                   const LaunchShape& shape, int64_t* slabs, const hdk_hip_device_properties* props, hipStream_t s,
                                   bool* launched, BhBehindSliced* bh = nullptr) {
  *launched = false;
  SliceArgs& sa = ga.s;
  const S2Kernel kagg_fn = s2_agg_kernel(ga);
  const void* kagg = reinterpret_cast<const void*>(kagg_fn);
  const size_t lds_agg = static_cast<size_t>(ga.entry_count) * ga.wpe * ga.rep * 8 + static_cast<size_t>(ga.fslice) * 4 * ((ga.npay == 2 && !ga.packed) ? 2 : 1);
  if (lds_agg > 48 * 1024 && hipFuncSetAttribute(kagg, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_agg)) != hipSuccess) {
    (void)hipGetLastError();
    return HDK_HIP_OK;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kagg, kSliceAggBlock, lds_agg) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    return HDK_HIP_OK;
  }
  auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t nsub = static_cast<size_t>(sa.nbins) * kSliceXcds;
  const size_t b_tuples = up((nsub * sa.sub + sa.cap_ovf) * 8);
  const size_t b_tuples2 = ga.two_level ? up(static_cast<size_t>(ga.nslices) * ga.cap2 * 8) : 0;
  const size_t n_fill1 = nsub * kSliceCursorStride + 8;
  const size_t n_fill2 = ga.two_level ? static_cast<size_t>(ga.nslices) * kSliceCursorStride : 0;
  const size_t b_fill = up((n_fill1 + n_fill2) * sizeof(uint32_t));
  AsyncScratch scratch(s);
  if (hipMallocAsync(&scratch.p, b_tuples + b_tuples2 + b_fill, s) != hipSuccess) {
    (void)hipGetLastError();  // no scratch: the interpreter probes in row order
    scratch.p = nullptr;
    return HDK_HIP_OK;
  }
  int8_t* q = static_cast<int8_t*>(scratch.p);
  sa.kp = kp;
  sa.tuples = reinterpret_cast<int64_t*>(q);
  ga.tuples2 = reinterpret_cast<int64_t*>(q + b_tuples);
  sa.fill = reinterpret_cast<uint32_t*>(q + b_tuples + b_tuples2);
  sa.fill_ovf = sa.fill + nsub * kSliceCursorStride;
  sa.mode = sa.fill_ovf + 1;
  sa.probe = sa.mode + 1;
  ga.fill2 = sa.fill + n_fill1;
  sa.num_slabs = shape.grid;
  ga.slabs = slabs;
  ga.error_code = kp.error_code;
  HDK_HIP_CHECK(hipMemsetAsync(sa.fill, 0, b_fill, s));
  constexpr int VR = 8;
  const size_t lds_sc = static_cast<size_t>(kSliceBlock) * VR * 8 + static_cast<size_t>(kSliceBlock) * VR + 16;
  const S2Scatter ksc_fn = s2_scatter_kernel(sa);
  const unsigned g_sc = scatter_grid(reinterpret_cast<const void*>(ksc_fn), kSliceBlock, lds_sc, props, 2);
  const uint32_t resident = static_cast<uint32_t>(per_cu) * static_cast<uint32_t>(props->num_cu);
  uint32_t members = 1;
  if (ga.two_level) {  // persistent blocks, each walking slices b, b + nsl_par, ...
    if (ga.nsl_par > resident) ga.nsl_par = resident;
    if (ga.nsl_par > shape.grid) ga.nsl_par = shape.grid;
  } else {
    members = resident / sa.nbins;
    if (members < 1) members = 1;
    if (sa.nbins * members > shape.grid) members = shape.grid / sa.nbins;
  }
  if (members < 1 || ga.nsl_par < 1) return HDK_HIP_OK;
  hipLaunchKernelGGL(hdk_join_order_probe, dim3(256), dim3(256), 0, s, sa);
  hipLaunchKernelGGL(ksc_fn, dim3(g_sc), dim3(kSliceBlock), lds_sc, s, sa);
  if (ga.two_level) {
    const size_t lds_l2 = static_cast<size_t>(kSliceBlock) * VR * 8 + static_cast<size_t>(kSliceBlock) * VR + 16;
    const uint32_t ncx = (sa.nbins + kSliceXcds - 1) / kSliceXcds;
    const uint32_t res2 = resident_grid(reinterpret_cast<const void*>(hdk_join_scatter_level2), kSliceBlock, lds_l2, props);
    uint32_t m2 = res2 / (ncx * kSliceXcds);
    if (m2 < 1) m2 = 1;
    hipLaunchKernelGGL(hdk_join_scatter_level2, dim3(ncx * kSliceXcds * m2), dim3(kSliceBlock), lds_l2, s, ga);
  }
  hipLaunchKernelGGL(kagg_fn, dim3(ga.nsl_par * members), dim3(kSliceAggBlock), lds_agg, s, ga);
  // armed behind the passes: the batched interpreter over the plan's own columns, in row order -- clustered input, stale
  // statistics, an overflow area that filled up
  if (bh) {
    int32_t st = launch_bh_vec_armed(bh->plan, d_plan, kp, bh->ko, props, s, sa.mode);
    if (st) return st;
    bh->fold.slabs = slabs;
    bh->fold.num_slabs = shape.grid;
    st = launch_bh_fold_dense(bh->fold, d_plan, kp, s);
    if (st) return st;
    *launched = true;
    return HDK_HIP_OK;
  }
  VecArgs v;
  v.plan = d_plan;
  v.kp = kp;
  v.slabs = slabs;
  v.entry_count = shape.entry_count;
  v.rep = shape.rep;
  v.run_if = sa.mode;
  v.bh_cap_log2 = 0;
  v.bh_pad_ = 0;
  hipLaunchKernelGGL(hdk_scan_agg_vec_join, dim3(shape.grid), dim3(kVecBlock), shape.lds_bytes, s, v);
  HDK_HIP_CHECK(hipGetLastError());
  *launched = true;
  return HDK_HIP_OK;  // (`scratch` is freed here, stream-ordered after the kernels above)
}

static JoinRoute route_join(const hdk_hip_plan* plan, const LaunchShape& shape, const hdk_hip_kernel_options* ko, JoinDirectArgs* ja,
                            Slice2Args* ga) {
  const bool direct = match_join_direct(plan, shape, ja);
  if (direct) {
    SliceArgs sa;
    if (match_join_sliced(plan, *ja, ko, shape.grid, &sa) && sa.fast && !hdk_sw(SW_SLICED2_ALWAYS)) {
      return JOIN_ROUTE_DIRECT;  // BASELINE config 3 itself: SUM(x + payload), its compile-time form
    }
  }
  if (match_join_sliced2(plan, shape, ko, ga)) return JOIN_ROUTE_SLICED2;
  return direct ? JOIN_ROUTE_DIRECT : JOIN_ROUTE_NONE;
}

// ---- GroupByBaselineHash behind the sliced join (`GROUP BY dim.dval % 64`) -----------------------------------------
// The reference's planner has no range for a modulo (QE/ExpressionRange.cpp:391-419), so the OUTPUT is an open-addressing
// table -- but the kernel can bound the key itself: payload % m lies in (-m, m).  The plan runs as an INTERNAL perfect-hash
// plan over that range (the same scatter and LDS passes as C3g, a dense private group table beside every slice); the slabs
// are folded into the open-addressing table through the reference's probe sequence (scan_bh.hip: hdk_bh_fold_dense).
static bool baseline_sliced_join_plan(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props,
                                      hdk_hip_plan* inner, LaunchShape* shape, Slice2Args* ga, BhDenseFold* fold) {
  if (p->query_kind != HDK_Q_BASELINE_HASH || p->num_joins != 1 || p->key_count != 1 || p->key_width != 8 || p->output_columnar) return false;
  if (!ko || (ko->flags & (HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS | HDK_HIP_LAUNCH_FORCE_PARTITIONED)) || launch_forces_generic(ko)) return false;
  if (!bh_lds_kernel_name(p, ko)) return false;  // (the armed fallback is the open-addressing interpreter)
  const hdk_hip_expr& ke = p->keys[0];
  if (ke.nsteps != 1 || ke.leaf0.kind != HDK_LEAF_COL) return false;
  const hdk_hip_step& sp = ke.steps[0];
  if (sp.op != HDK_OP_MOD || sp.out_class != HDK_VC_INT || sp.rhs.kind != HDK_LEAF_INT || sp.rhs.ival < 1 || sp.rhs.ival > 4096) return false;
  const hdk_hip_col& kc = p->cols[ke.leaf0.col];
  const int64_t m = sp.rhs.ival;
  const bool nonneg = kc.has_stats && kc.min_val >= 0;
  const int64_t lo = nonneg ? 0 : -(m - 1), hi = m - 1;
  const bool has_null = ke.nullable != 0;
  *inner = *p;
  inner->query_kind = HDK_Q_PERFECT_HASH;
  inner->keyless = 0;
  inner->idx_target_as_key = -1;
  inner->key_min[0] = lo;
  inner->key_bucket[0] = 0;
  inner->key_card[0] = static_cast<uint32_t>(hi - lo + 1 + (has_null ? 1 : 0));
  inner->key_has_nulls[0] = has_null ? 1 : 0;
  inner->key_null_translated[0] = hi + 1;
  inner->entry_count = static_cast<uint32_t>(hi - lo + 1 + (has_null ? 1 : 0));
  *shape = choose_shape(inner, ko, props);
  if (shape->strategy != STRAT_LDS || !match_join_sliced2(inner, *shape, ko, ga)) return false;
  memset(fold, 0, sizeof(*fold));
  fold->entries = inner->entry_count;
  fold->out_entry_count = p->entry_count;
  fold->wpe = ga->wpe;
  for (int w = 0; w < ga->wpe; ++w) fold->wop[w] = ga->wop[w];
  fold->nword_mask = ga->nword_mask;
  fold->key_lo = lo;
  fold->null_entry = has_null ? static_cast<uint32_t>(hi + 1 - lo) : 0xFFFFFFFFu;
  fold->null_key = ke.null_val;
  return true;
}

const char* baseline_sliced_join_names(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props) {
  hdk_hip_plan inner;
  LaunchShape shape;
  Slice2Args ga;
  BhDenseFold fold;
  if (!baseline_sliced_join_plan(p, ko, props, &inner, &shape, &ga, &fold)) return nullptr;
  return ga.two_level ? "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_scatter_level2,hdk_join_agg_sliced2,hdk_scan_agg_bh_vec_join,hdk_bh_fold_dense"
                      : "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2,hdk_scan_agg_bh_vec_join,hdk_bh_fold_dense";
}

int32_t launch_baseline_sliced_join(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                                    const hdk_hip_device_properties* props, hipStream_t s, bool* launched) {
  *launched = false;
  hdk_hip_plan inner;
  LaunchShape shape;
  Slice2Args ga;
  BhBehindSliced bh;
  if (!baseline_sliced_join_plan(plan, ko, props, &inner, &shape, &ga, &bh.fold)) return HDK_HIP_OK;
  bh.plan = plan;
  bh.ko = ko;
  AsyncScratch slabs(s);
  if (hipMallocAsync(&slabs.p, static_cast<size_t>(shape.grid) * shape.slab_words * 8, s) != hipSuccess) {
    (void)hipGetLastError();
    slabs.p = nullptr;
    return HDK_HIP_OK;
  }
  return launch_join_sliced2(&inner, d_plan, kp, ga, shape, static_cast<int64_t*>(slabs.p), props, s, launched, &bh);
}

static int32_t launch_scan_lds(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                               const LaunchShape& shape, int64_t* slabs, hipStream_t s, bool force_generic,
                               bool force_scalar, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props) {
  FastArgs fa;
  int kw, vw;
  if (!force_generic && match_fast(plan, shape, &fa, &kw, &vw)) {
    fa.kp = kp;
    fa.slabs = slabs;
    return launch_fast_direct(kw, vw, fa, shape, s);  // (scan_fast.hip: the instantiations live in their own translation unit)
  }
  ColsArgs ca;
  if (!force_generic && match_cols(plan, shape, &ca)) {
    ca.kp = kp;
    ca.slabs = slabs;
    return launch_cols(ca, shape, s);  // (scan_cols.hip)
  }
  JoinDirectArgs ja;
  Slice2Args ga;
  const JoinRoute route = force_generic ? JOIN_ROUTE_NONE : route_join(plan, shape, ko, &ja, &ga);
  if (route == JOIN_ROUTE_SLICED2) {
    bool launched = false;
    const int32_t st = launch_join_sliced2(plan, d_plan, kp, ga, shape, slabs, props, s, &launched);
    if (st || launched) return st;
    // (no scratch / no block this large: fall through to the kernels that probe in row order)
    if (match_join_direct(plan, shape, &ja)) {
      ja.kp = kp;
      ja.slabs = slabs;
      return launch_join_direct(plan, ja, ko, shape, props, s);
    }
  } else if (route == JOIN_ROUTE_DIRECT) {
    ja.kp = kp;
    ja.slabs = slabs;
    return launch_join_direct(plan, ja, ko, shape, props, s);
  }
  KeysArgs ka;
  if (!force_generic && match_keys(plan, shape, &ka)) {
    ka.kp = kp;
    ka.slabs = slabs;
    void* kargs[] = {&ka};
    const uint32_t kblock = shape.block == kKeysWideBlock ? kKeysWideBlock : kKeysBlock;
    HDK_HIP_CHECK(hipLaunchKernel(keys_kernel(ka, kblock), dim3(shape.grid), dim3(kblock), kargs, keys_lds_bytes(ka, shape), s));
    HDK_HIP_CHECK(hipGetLastError());
    return HDK_HIP_OK;
  }
  if (!force_scalar && !needs_join_loops(plan)) {
    VecArgs v;
    v.plan = d_plan;
    v.kp = kp;
    v.slabs = slabs;
    v.entry_count = shape.entry_count;
    v.rep = shape.rep;
    v.run_if = nullptr;
    if (plan_is_single_matching_set_join(plan)) {
      hipLaunchKernelGGL(hdk_scan_agg_vec_many, dim3(shape.grid), dim3(kVecBlock), shape.lds_bytes, s, v);
    } else if (plan->num_joins && plan_has_keyed_join(plan)) {
      hipLaunchKernelGGL(hdk_scan_agg_vec_keyed, dim3(shape.grid), dim3(kVecBlock), shape.lds_bytes, s, v);
    } else if (plan->num_joins) {
      hipLaunchKernelGGL(hdk_scan_agg_vec_join, dim3(shape.grid), dim3(kVecBlock), shape.lds_bytes, s, v);
    } else {
      hipLaunchKernelGGL(hdk_scan_agg_vec, dim3(shape.grid), dim3(kVecBlock), shape.lds_bytes, s, v);
    }
    HDK_HIP_CHECK(hipGetLastError());
    return HDK_HIP_OK;
  }
  ScanArgs a;
  a.plan = d_plan;
  a.kp = kp;
  a.slabs = slabs;
  a.entry_count = shape.entry_count;
  a.rep = shape.rep;
  a.rows_per_tile = kBlock * 4;
  hipLaunchKernelGGL(hdk_scan_agg_generic, dim3(shape.grid), dim3(kBlock), shape.lds_bytes, s, a);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// value form of hdk_scan_agg_keys: every aggregate argument is a plain column of the outer table (at most two of
// them), the slot type follows the column (no int -> fp promotion); the per-row update list comes from the word layout
static bool match_keys_values(const hdk_hip_plan* p, KeysArgs* ka) {
  WordLayout wl;
  make_word_layout(p, &wl);
  if (wl.wpe != ka->wpe) return false;
  for (int w = 0; w < wl.wpe; ++w) ka->wop[w] = wl.wop[w];
  int nops = 0;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (wl.vword[t] < 0 && wl.nword[t] < 0) continue;  // projected key, COUNT(*), COUNT(not-null column)
    int c;
    if (!tg.has_arg || !plain_outer_col(p, tg.arg, &c)) return false;
    const hdk_hip_col& col = p->cols[c];
    const bool fp = col.kind == HDK_COL_DOUBLE || col.kind == HDK_COL_FLOAT;
    if (col.width != 1 && col.width != 2 && col.width != 4 && col.width != 8) return false;
    if (col.kind == HDK_COL_UNSIGNED && col.width == 8) return false;
    if (target_has_value_word(tg) && (tg.arg_is_fp != 0) != fp) return false;
    const int nullable = tg.skip_null && tg.arg.nullable;
    int vi = -1;
    for (int k = 0; k < ka->nvals; ++k) {
      if (ka->val[k].buf_idx == col.buf_idx) vi = k;
    }
    if (vi < 0) {
      if (ka->nvals == kKeysMaxVals) return false;
      vi = ka->nvals++;
      KeysVal& kv = ka->val[vi];
      kv.buf_idx = col.buf_idx;
      kv.width = col.width;
      kv.kind = col.kind;
      kv.nullable = nullable;
      kv.null_val = tg.arg.null_val;
    } else if (ka->val[vi].nullable != nullable) {
      return false;  // one column, two NULL conventions: the interpreter sorts that out
    }
    auto push = [&](int kind, int word) {
      if (nops < kKeysMaxOps) {
        ka->op_kind[nops] = kind;
        ka->op_word[nops] = word;
        ka->op_val[nops] = vi;
      }
      ++nops;
    };
    if (wl.vword[t] >= 0) {
      int kind;
      switch (wl.wop[wl.vword[t]]) {
        case WOP_ADD_U64: kind = FOP_ADD_U64; break;
        case WOP_ADD_F64: kind = FOP_ADD_F64; break;
        case WOP_MIN_I64: kind = FOP_MIN_I64; break;
        case WOP_MAX_I64: kind = FOP_MAX_I64; break;
        case WOP_MIN_F64: kind = FOP_MIN_F64; break;
        default: kind = FOP_MAX_F64; break;
      }
      push(kind, wl.vword[t]);
    }
    if (wl.nword[t] >= 0) {
      push(FOP_ADD_ONE_IF_NULL, wl.nword[t]);
      ka->nword_mask |= 1u << wl.nword[t];
    }
  }
  if (nops > kKeysMaxOps || ka->nvals == 0) return false;
  ka->nops = nops;
  return true;
}

// the shape hdk_scan_agg_keys takes (scan_agg_keys.h): perfect hash on 1-3 integer outer columns, each
// plain or under ONE of extract-year / decimal scale-down, plain filters; row counts only (the counting form) or
// aggregates over at most two plain outer columns (the value form, match_keys_values)
static bool match_keys(const hdk_hip_plan* p, const LaunchShape& shape, KeysArgs* ka) {
  if (shape.strategy != STRAT_LDS || p->query_kind != HDK_Q_PERFECT_HASH || p->num_joins) return false;
  if (p->key_count < 1 || p->key_count > kKeysMax) return false;
  if (shape.rep == 0 || (shape.rep & (shape.rep - 1))) return false;
  if (!match_plain_quals(p, ka->q)) return false;  // (no filter programs here: scan_agg_keys.h, HDK_KEYS_PROG)
  ka->wpe = shape.wpe;
  ka->nvals = 0;
  ka->nops = 0;
  ka->nword_mask = 0;
  if (shape.wpe != 1 && !match_keys_values(p, ka)) return false;
  ka->nquals = p->num_quals;
  ka->nkeys = p->key_count;
  ka->entry_count = shape.entry_count;
  ka->rep = shape.rep;
  ka->rep_shift = 0;
  while ((1u << ka->rep_shift) < shape.rep) ++ka->rep_shift;
  // 32-bit key terms wrap modulo 2^32: with |kmin| this far from the int32 limits no out-of-range value of a
  // 32-bit key can wrap into [0, card)
  constexpr int64_t kMinGuard = static_cast<int64_t>(INT32_MAX) - (1 << 25);
  uint64_t stride = 1;
  for (int k = 0; k < p->key_count; ++k) {
    const hdk_hip_expr& e = p->keys[k];
    if (e.leaf0.kind != HDK_LEAF_COL || e.nsteps > 1 || p->key_bucket[k]) return false;
    const hdk_hip_col& c = p->cols[e.leaf0.col];
    if (c.table != 0 || (c.kind != HDK_COL_INT && c.kind != HDK_COL_UNSIGNED)) return false;
    if (c.width != 1 && c.width != 2 && c.width != 4 && c.width != 8) return false;
    KeysKey& kk = ka->key[k];
    kk.buf_idx = c.buf_idx;
    kk.width = c.width;
    kk.is_unsigned = c.kind == HDK_COL_UNSIGNED;
    kk.xf = KXF_NONE;
    kk.param = 1;
    kk.col_null = e.leaf0.null_val;
    kk.col_nullable = e.leaf0.nullable;
    kk.div_magic = 0;
    kk.div_shift = 0;
    int64_t step_null = 0;
    if (e.nsteps == 1) {
      const hdk_hip_step& st = e.steps[0];
      if (st.out_class != HDK_VC_INT) return false;
      if (st.op == HDK_OP_EXTRACT_YEAR) {
        kk.xf = KXF_YEAR;
      } else if (st.op == HDK_OP_SCALE_DOWN && st.rhs.ival >= 2 && st.rhs.ival <= INT32_MAX) {
        kk.xf = KXF_SCALE_DOWN;
        kk.param = st.rhs.ival;
        // unsigned 32-bit division by an invariant divisor, round-up method in its branch-free form
        const uint32_t d = static_cast<uint32_t>(kk.param);
        uint32_t log2d = 31;
        while (!(d >> log2d)) --log2d;
        if ((d & (d - 1)) == 0) {
          kk.div_magic = 0;
          kk.div_shift = static_cast<int32_t>(log2d) - 1;
        } else {
          const uint64_t two_k = 1ull << (32 + log2d);
          uint64_t m = two_k / d;
          const uint32_t rem = static_cast<uint32_t>(two_k - m * d);
          uint32_t m32 = static_cast<uint32_t>(m) * 2u;
          const uint32_t twice_rem = rem * 2u;
          if (twice_rem >= d || twice_rem < rem) m32 += 1;
          kk.div_magic = m32 + 1u;
          kk.div_shift = static_cast<int32_t>(log2d);
        }
      } else {
        return false;
      }
      step_null = st.null_out;
    }
    kk.translate = p->key_has_nulls[k] && e.nullable;
    kk.key_null = e.null_val;
    kk.kmin = p->key_min[k];
    if (p->key_card[k] < 1 || static_cast<uint64_t>(p->key_card[k]) > shape.entry_count) return false;
    kk.card = static_cast<uint32_t>(p->key_card[k]);
    kk.stride = static_cast<uint32_t>(stride);
    stride *= kk.card;
    if (stride > shape.entry_count) return false;  // the table must hold every combination
    const bool kmin_small = kk.kmin >= -kMinGuard && kk.kmin <= kMinGuard;
    kk.kmin32 = kmin_small ? static_cast<int32_t>(kk.kmin) : 0;
    kk.year_base = 1900u - static_cast<uint32_t>(kk.kmin32);
    // the value a NULL key takes (eval_key_v, vec_eval.h) and what it contributes
    int64_t null_value;
    if (kk.xf == KXF_NONE) {
      null_value = p->key_null_translated[k];  // only reached when translate is on
      const bool null32 = kk.key_null >= INT32_MIN && kk.key_null <= INT32_MAX;
      kk.narrow = kmin_small && (c.width < 4 || (c.width == 4 && !kk.is_unsigned)) && (!kk.translate || null32);
    } else {
      if (!kmin_small) return false;  // the 32-bit forms of the transforms subtract kmin in 32 bits
      null_value = (kk.translate && step_null == kk.key_null) ? p->key_null_translated[k] : step_null;
      kk.narrow = 0;
    }
    const uint64_t nt = static_cast<uint64_t>(null_value) - static_cast<uint64_t>(kk.kmin);
    kk.null_ok = nt < kk.card;
    kk.null_term = static_cast<uint32_t>(nt);
    kk.null_term_p = kk.null_ok ? kk.null_term : kKeysPoison;
    if (kk.xf != KXF_NONE && kk.col_nullable) {
      // the tile body recognises a NULL of a transformed column only among the rows outside the transform's 32-bit
      // fast range (scan_agg_keys.h): true of every signed 4- and 8-byte column, checked here with the kernel's tests
      bool null_is_fast;
      if (kk.xf == KXF_YEAR) {
        null_is_fast = static_cast<uint64_t>(kk.col_null) <= static_cast<uint64_t>(UINT32_MAX - 2208988800u);
      } else {
        const uint64_t half = static_cast<uint64_t>(kk.param >> 1);
        const int64_t tmp = static_cast<int64_t>(kk.col_null >= 0 ? static_cast<uint64_t>(kk.col_null) + half
                                                                  : static_cast<uint64_t>(kk.col_null) - half);
        const int32_t t32 = static_cast<int32_t>(tmp);
        null_is_fast = t32 == tmp && t32 != INT32_MIN;
      }
      if (null_is_fast) return false;
    }
    if (shape.entry_count >= kKeysPoison) return false;  // (cannot happen: kLdsMaxTableWords < kKeysPoison)
  }
  return true;
}





// Head of every launch: the plan copy and the launch's interrupt / watchdog words go to the front of `workspace`
// (kPlanRegionBytes), the 12 launch pointers become a KernParams.
int32_t launch_head(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT], const hdk_hip_kernel_options* ko,
                    int32_t device_id, void* workspace, hipStream_t s, hdk_hip_plan** d_plan_out, KernParams* kp_out) {
  // the plan is read by the kernels from device memory (wave-uniform scalar loads); behind it sit the launch's own
  // interrupt / watchdog words (watch.h): zeros travel with the plan upload, anything else is written by k_arm_watch
  // on the launch stream.  A PLAN_RESIDENT launch always runs k_arm_watch: its workspace may hold the words of an
  // earlier launch, and as a captured kernel node it re-arms every replay of the graph.
  hdk_hip_plan* d_plan = static_cast<hdk_hip_plan*>(workspace);
  *d_plan_out = d_plan;
  LaunchWatch* d_watch = reinterpret_cast<LaunchWatch*>(static_cast<int8_t*>(workspace) + kWatchOffset);
  const bool plan_resident = ko && (ko->flags & HDK_HIP_LAUNCH_PLAN_RESIDENT);
  if (!plan_resident) {
    struct alignas(16) Head {
      int8_t bytes[kWatchOffset + sizeof(LaunchWatch)];
    } head;
    static_assert(sizeof(LaunchWatch) == 16, "LaunchWatch is 16 bytes");
    memcpy(head.bytes, plan, sizeof(hdk_hip_plan));
    memset(head.bytes + sizeof(hdk_hip_plan), 0, sizeof(head.bytes) - sizeof(hdk_hip_plan));
    HDK_HIP_CHECK(hipMemcpyAsync(d_plan, head.bytes, sizeof(head.bytes), hipMemcpyHostToDevice, s));
  }
  {
    const uint32_t wf = ((ko && (ko->flags & HDK_HIP_LAUNCH_CHECK_INTERRUPT)) ? 1u : 0u) | ((ko && ko->watchdog_ms) ? 2u : 0u);
    if (wf || plan_resident) {
      hipLaunchKernelGGL(k_arm_watch, dim3(1), dim3(1), 0, s, d_watch, wf, ko ? ko->watchdog_ms : 0u);
      HDK_HIP_CHECK(hipGetLastError());
    }
  }

  KernParams& kp = *kp_out;
  kp.col_buffers = reinterpret_cast<const int8_t* const* const*>(params[HDK_KP_COL_BUFFERS]);
  kp.num_fragments = reinterpret_cast<const uint64_t*>(params[HDK_KP_NUM_FRAGMENTS]);
  kp.literals = params[HDK_KP_LITERALS];
  kp.num_rows = reinterpret_cast<const int64_t*>(params[HDK_KP_NUM_ROWS]);
  kp.frag_row_offsets = reinterpret_cast<const uint64_t*>(params[HDK_KP_FRAG_ROW_OFFSETS]);
  kp.max_matched = reinterpret_cast<const int32_t*>(params[HDK_KP_MAX_MATCHED]);
  kp.total_matched = reinterpret_cast<int32_t*>(params[HDK_KP_TOTAL_MATCHED]);
  kp.init_agg_vals = reinterpret_cast<const int64_t*>(params[HDK_KP_INIT_AGG_VALS]);
  kp.groupby_buf = reinterpret_cast<int64_t**>(params[HDK_KP_GROUPBY_BUF]);
  kp.error_code = reinterpret_cast<int32_t*>(params[HDK_KP_ERROR_CODE]);
  kp.num_tables = reinterpret_cast<const uint32_t*>(params[HDK_KP_NUM_TABLES]);
  kp.join_hash_tables = reinterpret_cast<const int64_t*>(params[HDK_KP_JOIN_HASH_TABLES]);
  kp.watch = d_watch;
  kp.interrupt = device_interrupt_word(device_id);

  return HDK_HIP_OK;
}

}  // namespace hdk

extern "C" int32_t hdk_hip_validate_plan(const hdk_hip_plan* plan, int32_t layout_only) {
  return layout_only ? validate_plan_layout(plan) : validate_plan(plan);
}

extern "C" int32_t hdk_hip_workspace_size(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko,
                                          int32_t device_id, size_t* bytes) {
  HDK_REQUIRE(bytes, "bytes is NULL");
  const int32_t st = validate_plan(plan);
  if (st) return st;
  const hdk_hip_device_properties* props = device_props(device_id);
  if (!props) return HDK_HIP_ERR_RUNTIME;
  *bytes = workspace_bytes_for(choose_shape(plan, ko, props));
  return HDK_HIP_OK;
}

static bool match_cluster_join(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, ClusterArgs* ca);

extern "C" int32_t hdk_hip_describe_launch(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko,
                                           int32_t device_id, char* out, size_t out_len) {
  HDK_REQUIRE(out && out_len, "out is NULL");
  const int32_t st = validate_plan(plan);
  if (st) return st;
  // (HDK_HIP_DEVICE_ASSUMED_MI355X: the answer for an MI355X without touching a device -- kernel routing is host arithmetic
  // over the plan, the options and these few numbers, and the CPU test-suite pins it: tests/test_kernel_routing_cpu.py)
  static const hdk_hip_device_properties assumed = [] {
    hdk_hip_device_properties p;
    memset(&p, 0, sizeof(p));
    p.global_mem = 288ull << 30;
    p.num_cu = 256;
    p.max_threads_per_block = 1024;
    p.wavefront_size = 64;
    p.grid_size = 4 * 256;
    p.shared_mem_per_block = 160 << 10;  // (what hipDeviceProp_t::sharedMemPerBlock reports on the MI355X)
    p.has_shared_memory_atomics = p.can_load_async = p.has_fp64 = 1;
    strncpy(p.arch_name, "gfx950 (assumed)", sizeof(p.arch_name) - 1);
    return p;
  }();
  const hdk_hip_device_properties* props = device_id == HDK_HIP_DEVICE_ASSUMED_MI355X ? &assumed : device_props(device_id);
  if (!props) return HDK_HIP_ERR_RUNTIME;
  const LaunchShape s = choose_shape(plan, ko, props);
  {
    ClusterArgs ca;
    JoinDirectArgs jd;
    Slice2Args g2;
    const bool generic = ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR));
    const JoinRoute route = generic ? JOIN_ROUTE_NONE : route_join(plan, s, ko, &jd, &g2);
    const bool direct = route == JOIN_ROUTE_DIRECT;
    SliceArgs sl;
    if (route == JOIN_ROUTE_SLICED2) {
      const int n = snprintf(out, out_len, g2.two_level ? "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_scatter_level2,hdk_join_agg_sliced2,"
                                                        : "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2,");
      if (n > 0 && static_cast<size_t>(n) < out_len) {
        out += n;
        out_len -= static_cast<size_t>(n);
      }
    } else if (direct && match_join_sliced(plan, jd, ko, s.grid, &sl)) {
      const int n = snprintf(out, out_len, "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced,");
      if (n > 0 && static_cast<size_t>(n) < out_len) {
        out += n;
        out_len -= static_cast<size_t>(n);
      }
    } else if (direct && join_direct_clusters(plan, ko)) {
      const int n = snprintf(out, out_len, "hdk_cluster_by_key,");
      if (n > 0 && static_cast<size_t>(n) < out_len) {
        out += n;
        out_len -= static_cast<size_t>(n);
      }
    } else if (route == JOIN_ROUTE_NONE && !baseline_sliced_join_names(plan, ko, props) &&
               match_cluster_join(plan, ko, &ca)) {  // the pre-pass that clusters the outer rows by join-key range
      const int n = snprintf(out, out_len, "hdk_cluster_by_key,hdk_cluster_params,");
      if (n > 0 && static_cast<size_t>(n) < out_len) {
        out += n;
        out_len -= static_cast<size_t>(n);
      }
    }
  }
  if (perfect_goes_packed(plan, s, ko)) {
    snprintf(out, out_len, "%s", bh_packed_kernel_name(plan, ko));
  } else if (s.strategy == STRAT_LDS) {
    snprintf(out, out_len, "%s,hdk_finalize",
             scan_kernel_name(plan, s, ko, ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR)),
                              ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_SCALAR)));
  } else if (s.strategy == STRAT_PROJECT) {
    project_describe(plan, ko, out, out_len);
  } else {
    if (const char* sj = baseline_sliced_join_names(plan, ko, props)) {  // a bounded key behind a large join table
      snprintf(out, out_len, "%s", sj);
    } else {
      baseline_describe(plan, ko, out, out_len);
    }
  }
  return HDK_HIP_OK;
}

// ---- clustering pre-pass for join probes (scan_cluster.h) -----------------------------------------------------------
static bool match_cluster_join(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, ClusterArgs* ca) {
  if (!ko || ko->total_rows == 0 || (ko->flags & HDK_HIP_LAUNCH_NO_CLUSTER_PROBES)) return false;
  if (ko->flags & (HDK_HIP_LAUNCH_FORCE_SCALAR | HDK_HIP_LAUNCH_FORCE_GENERIC)) return false;
  if (p->num_joins != 1 || p->query_kind == HDK_Q_PROJECTION || needs_join_loops(p) || p->num_filter_ops) return false;
  const hdk_hip_join& jn = p->joins[0];
  if ((jn.kind != HDK_JOIN_ONE_TO_ONE && jn.kind != HDK_JOIN_ONE_TO_ONE_FUSED) || !join_type_inner_like(jn.type)) return false;
  if (jn.null_mode == HDK_JOIN_NULL_BITWISE || jn.bucket > 1) return false;  // (a NULL key that matches: not dropped)
  int kc;
  if (!plain_outer_col(p, jn.outer_key, &kc)) return false;
  if (jn.max_key < jn.min_key || static_cast<uint64_t>(jn.max_key - jn.min_key) >= 0xFFFFFFFFull) return false;
  const uint64_t range = static_cast<uint64_t>(jn.max_key - jn.min_key) + 1;
  const uint64_t entry_bytes = jn.kind == HDK_JOIN_ONE_TO_ONE_FUSED ? 8ull * static_cast<uint64_t>(jn.fused_stride) : 4ull;
  // Only on request.  Measured on C3 (256 M rows, 10 M-key dimension, 160 MB fused table): pre-pass 2.2 ms + probes over
  // the clustered rows 4.2 ms against 5.0 ms for probes in row order -- the batched interpreter that consumes the rows
  // needs 3.1 ms even when the whole table sits in L2, so the locality cannot pay for the pre-pass until a specialised
  // consumer exists.
  (void)entry_bytes;
  if (!(ko->flags & HDK_HIP_LAUNCH_CLUSTER_PROBES)) return false;
  memset(ca, 0, sizeof(*ca));
  for (int i = 0; i < HDK_HIP_MAX_COLS; ++i) ca->outer_slot[i] = -1;
  // the join key first, then every other outer column of the plan: all of them 8 bytes wide
  ca->buf_idx[0] = p->cols[kc].buf_idx;
  ca->outer_slot[kc] = 0;
  ca->ncols = 1;
  for (int i = 0; i < p->num_cols; ++i) {
    const hdk_hip_col& c = p->cols[i];
    if (c.table < 0) continue;  // payload word of a fused table: no buffer
    if (c.table != 0) continue;
    if (c.width != 8 || (c.kind != HDK_COL_INT && c.kind != HDK_COL_DOUBLE)) return false;
    if (c.buf_idx != i) return false;  // (plan.py numbers them alike; the parameter builder relies on it)
    if (i == kc) continue;
    if (ca->ncols == kClusterMaxCols) return false;
    ca->buf_idx[ca->ncols] = c.buf_idx;
    ca->outer_slot[i] = ca->ncols++;
  }
  if (p->cols[kc].width != 8 || p->cols[kc].kind != HDK_COL_INT || p->cols[kc].buf_idx != kc) return false;
  ca->ncols_total = p->num_cols;
  ca->key_min = jn.min_key;
  ca->key_range = range;
  ca->bin_mult = (static_cast<uint64_t>(kClusterBins) << 32) / range;  // floor: bin <= kClusterBins - 1 for every key in range
  const uint64_t rows = ko->total_rows;
  ca->sub = ((rows / (kClusterBins * kClusterXcds)) * 17 / 16 + 4096 + 15) & ~15ull;
  ca->cap_ovf = rows;
  return true;
}

// runs the pre-pass and points `kp` at the permuted data; *scratch is freed (stream-ordered) by the caller after the scan.
// Out of scratch memory is not an error: the launch then simply probes in row order.
static int32_t launch_cluster_join(ClusterArgs ca, KernParams* kp, const hdk_hip_device_properties* props, hipStream_t s,
                                   void** scratch) {
  const uint64_t nsub = static_cast<uint64_t>(kClusterBins) * kClusterXcds;
  const uint64_t col_rows = nsub * ca.sub + ca.cap_ovf;
  const uint64_t nfr = nsub + 1;
  const uint32_t ntab_max = 1 + HDK_HIP_MAX_JOINS;
  auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t b_cols = up(col_rows * 8) * ca.ncols;
  const size_t b_fill = up((nsub * kClusterCursorStride + 4) * sizeof(uint32_t));
  const size_t b_colptrs = up(nfr * ca.ncols_total * sizeof(void*));
  const size_t b_fragptrs = up(nfr * sizeof(void*));
  const size_t b_rows = up(nfr * ntab_max * sizeof(int64_t));
  int8_t* q = nullptr;
  if (hipMallocAsync(reinterpret_cast<void**>(&q), b_cols + b_fill + b_colptrs + b_fragptrs + 2 * b_rows + 256, s) != hipSuccess) {
    (void)hipGetLastError();
    return HDK_HIP_OK;
  }
  *scratch = q;
  for (int c = 0; c < ca.ncols; ++c) {
    ca.out[c] = reinterpret_cast<int64_t*>(q);
    q += up(col_rows * 8);
  }
  ca.fill = reinterpret_cast<uint32_t*>(q);
  ca.fill_ovf = ca.fill + nsub * kClusterCursorStride;
  HDK_HIP_CHECK(hipMemsetAsync(ca.fill, 0, b_fill, s));
  q += b_fill;
  ca.col_ptrs = reinterpret_cast<const int8_t**>(q); q += b_colptrs;
  ca.frag_ptrs = reinterpret_cast<const int8_t* const**>(q); q += b_fragptrs;
  ca.num_rows = reinterpret_cast<int64_t*>(q); q += b_rows;
  ca.frag_offs = reinterpret_cast<uint64_t*>(q); q += b_rows;
  ca.num_fragments = reinterpret_cast<uint64_t*>(q);
  ca.kp = *kp;
  const size_t lds = static_cast<size_t>(kClusterTile) * ca.ncols * 8 + kClusterTile + 16;
  unsigned grid = resident_grid(reinterpret_cast<const void*>(hdk_cluster_by_key), kClusterBlock, lds, props);
  hipLaunchKernelGGL(hdk_cluster_by_key, dim3(grid), dim3(kClusterBlock), lds, s, ca);
  hipLaunchKernelGGL(hdk_cluster_params, dim3(8), dim3(256), 0, s, ca);
  HDK_HIP_CHECK(hipGetLastError());
  kp->col_buffers = reinterpret_cast<const int8_t* const* const*>(ca.frag_ptrs);
  kp->num_fragments = ca.num_fragments;
  kp->num_rows = ca.num_rows;
  kp->frag_row_offsets = ca.frag_offs;
  return HDK_HIP_OK;
}

int32_t hdk::launch_finalize_slabs(const hdk_hip_plan* d_plan, const int64_t* slabs, int64_t** groupby_buf, uint32_t num_slabs,
                                   uint32_t entry_count, const uint32_t* skip_if, hipStream_t s) {
  FinalizeArgs fa;
  fa.plan = d_plan;
  fa.slabs = slabs;
  fa.groupby_buf = groupby_buf;
  fa.num_slabs = num_slabs;
  fa.entry_count = entry_count;
  fa.skip_if = skip_if;
  if (num_slabs <= 16 && entry_count >= 2048) {  // few slabs of a large table: a thread per entry
    hipLaunchKernelGGL(hdk_finalize_flat, dim3((entry_count + kFinalizeFlatBlock - 1) / kFinalizeFlatBlock), dim3(kFinalizeFlatBlock), 0, s, fa);
    HDK_HIP_CHECK(hipGetLastError());
    return HDK_HIP_OK;
  }
  const unsigned fblocks = (entry_count + (kBlock / kWave) - 1) / (kBlock / kWave);
  hipLaunchKernelGGL(hdk_finalize, dim3(fblocks), dim3(kBlock), 0, s, fa);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// HDK_HIP_LAUNCH_INIT_OUTPUT for the strategies that do not fuse it: the init kernel, on the launch stream
static int32_t init_row_wise_output(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT], int32_t device_id,
                                    hipStream_t s) {
  const uint32_t key_count = plan->keyless ? 0u : static_cast<uint32_t>(plan->key_count);
  return launch_init_row_wise_indirect(reinterpret_cast<int64_t* const*>(params[HDK_KP_GROUPBY_BUF]),
                                       reinterpret_cast<const int64_t*>(params[HDK_KP_INIT_AGG_VALS]), plan->entry_count,
                                       key_count, static_cast<uint32_t>(plan->key_width), plan->row_size_quad,
                                       plan->keyless, device_props(device_id), s);
}

extern "C" int32_t hdk_hip_launch(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT],
                                  const hdk_hip_kernel_options* ko, int32_t device_id, void* stream,
                                  void* workspace, size_t workspace_bytes) {
  int32_t st = validate_plan(plan);
  if (st) return st;
  HDK_REQUIRE(params, "params is NULL");
  HDK_REQUIRE(params[HDK_KP_NUM_FRAGMENTS] && params[HDK_KP_NUM_ROWS] && params[HDK_KP_NUM_TABLES] &&
                  params[HDK_KP_GROUPBY_BUF] && params[HDK_KP_ERROR_CODE],
              "a required kernel parameter is NULL");
  HDK_REQUIRE(plan->num_joins == 0 || params[HDK_KP_JOIN_HASH_TABLES], "JOIN_HASH_TABLES is NULL");
  // aggregate kernels start their private tables from the init values (the reference always passes them,
  // QE/QueryExecutionContext.cpp:788-964); a projection has no slots
  HDK_REQUIRE(plan->query_kind == HDK_Q_PROJECTION || params[HDK_KP_INIT_AGG_VALS], "INIT_AGG_VALS is NULL");
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  const hdk_hip_device_properties* props = device_props(device_id);
  const LaunchShape shape = choose_shape(plan, ko, props);
  for (int j = 0; j < plan->num_joins; ++j) {
    if (plan->joins[j].kind == HDK_JOIN_ONE_TO_ONE_FUSED) {
      HDK_REQUIRE(plan->joins[j].fused_stride >= 1, "fused join table needs fused_stride >= 1");
      if ((shape.strategy == STRAT_GLOBAL && !bh_lds_kernel_name(plan, ko)) || (ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_SCALAR)) ||
          needs_join_loops(plan)) {
        set_error("fused join tables are only read by the batched interpreter kernels");
        return HDK_HIP_ERR_UNSUPPORTED;
      }
    }
  }
  if (shape.strategy == STRAT_GLOBAL) {
    for (int t = 0; t < plan->num_targets; ++t) {
      if (plan->targets[t].slot_width == 1 || plan->targets[t].slot_width == 2) {
        // (global atomics on 1- and 2-byte slots would need CAS loops on the enclosing word: not in the library;
        // tables that fit LDS are flushed by hdk_finalize, which has one writer per slot)
        set_error("1- and 2-byte slots are only written by the LDS strategy");
        return HDK_HIP_ERR_UNSUPPORTED;
      }
    }
  }
  HDK_REQUIRE(workspace && workspace_bytes >= workspace_bytes_for(shape),
              "workspace too small: %zu < %zu", workspace_bytes, workspace_bytes_for(shape));
  HDK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "workspace must be 16-byte aligned");
  bool init_output = ko && (ko->flags & HDK_HIP_LAUNCH_INIT_OUTPUT);
  if (init_output) {  // (checked before anything is enqueued or allocated)
    HDK_REQUIRE((plan->query_kind == HDK_Q_PERFECT_HASH || plan->query_kind == HDK_Q_BASELINE_HASH) && !plan->output_columnar,
                "HDK_HIP_LAUNCH_INIT_OUTPUT is for row-wise group-by buffers");
    HDK_REQUIRE(params[HDK_KP_INIT_AGG_VALS], "HDK_HIP_LAUNCH_INIT_OUTPUT needs INIT_AGG_VALS");
  }

  hdk_hip_plan* d_plan = nullptr;
  KernParams kp;
  st = launch_head(plan, params, ko, device_id, workspace, s, &d_plan, &kp);
  if (st) return st;

  const bool timed = ko && (ko->flags & HDK_HIP_LAUNCH_RECORD_EVENTS);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {  // everything the launch enqueues from here on is inside the recorded interval
    st = scan_events_begin(device_id, s, &e0, &e1);
    if (st) return st;
  }
  // join probes over a table far larger than L2: permute the outer columns by key range first (scan_cluster.h)
  AsyncScratch cluster_scratch(s);  // (freed on every way out of the launch)
  {
    ClusterArgs ca;
    JoinDirectArgs jd;
    Slice2Args g2;
    const bool generic = ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR));
    // (the direct and the sliced join kernels cluster their own input, as tuples)
    const JoinRoute route = generic ? JOIN_ROUTE_NONE : route_join(plan, shape, ko, &jd, &g2);
    // (a baseline-hash plan behind the sliced join clusters its own input, as tuples: launch_baseline_sliced_join)
    if (route == JOIN_ROUTE_NONE && !baseline_sliced_join_names(plan, ko, props) && match_cluster_join(plan, ko, &ca)) {
      st = launch_cluster_join(ca, &kp, props, s, &cluster_scratch.p);
      if (st) return st;
    }
  }
  if (perfect_goes_packed(plan, shape, ko)) {
    if (init_output) {
      st = init_row_wise_output(plan, params, device_id, s);
      if (st) return st;
    }
    bool launched = false;
    st = launch_bh_packed(plan, d_plan, kp, ko, props, s, &launched);
    if (st) return st;
    if (launched) {
      if (timed) {
        HDK_HIP_CHECK(hipEventRecord(e1, s));
      }
      return HDK_HIP_OK;
    }
  }
  if (shape.strategy == STRAT_LDS) {
    int64_t* slabs = reinterpret_cast<int64_t*>(static_cast<int8_t*>(workspace) + kPlanRegionBytes);
    if (init_output) {
      st = init_row_wise_output(plan, params, device_id, s);
      if (st) return st;
    }
    st = launch_scan_lds(plan, d_plan, kp, shape, slabs, s, ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR)),
                         ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_SCALAR), ko, props);
    if (st) return st;
    if (timed) {
      HDK_HIP_CHECK(hipEventRecord(e1, s));
    }
    FinalizeArgs fa;
    fa.skip_if = nullptr;
    fa.plan = d_plan;
    fa.slabs = slabs;
    fa.groupby_buf = kp.groupby_buf;
    fa.num_slabs = shape.grid;
    fa.entry_count = shape.entry_count;
    const unsigned fblocks = (shape.entry_count + (kBlock / kWave) - 1) / (kBlock / kWave);
    hipLaunchKernelGGL(hdk_finalize, dim3(fblocks), dim3(kBlock), 0, s, fa);
    HDK_HIP_CHECK(hipGetLastError());
    return HDK_HIP_OK;
  }
  if (shape.strategy == STRAT_PROJECT) {
    HDK_REQUIRE(params[HDK_KP_MAX_MATCHED] && params[HDK_KP_TOTAL_MATCHED], "MAX_MATCHED / TOTAL_MATCHED is NULL");
    st = launch_project(plan, d_plan, kp, ko, shape, props, s);
  } else {
    st = launch_baseline(plan, d_plan, kp, ko, shape, init_output, props, s);
  }
  if (st) return st;
  if (timed) {
    HDK_HIP_CHECK(hipEventRecord(e1, s));
  }
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_collect_scan_times(int32_t device_id, float* ms_out, int32_t capacity, int32_t* count) {
  HDK_REQUIRE(count, "count is NULL");
  HDK_REQUIRE(device_id >= 0 && device_id < 16, "bad device");
  std::lock_guard<std::mutex> lk(g_scan_events[device_id].mu);
  auto& v = g_scan_events[device_id].pending;
  int32_t n = 0;
  for (auto& pr : v) {
    HDK_HIP_CHECK(hipEventSynchronize(pr.second));
    float ms = 0.f;
    HDK_HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
    if (ms_out && n < capacity) {
      ms_out[n] = ms;
    }
    ++n;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  v.clear();
  *count = n;
  return HDK_HIP_OK;
}
