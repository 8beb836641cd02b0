// scan_agg_perfect_part.h -- perfect-hash GROUP BY whose table does not fit LDS, by ENTRY-RANGE PARTITIONS.
//
// (One to three plain key columns: perfect_key_hash.)  A GroupByPerfectHash layout with more than kLdsMaxTableWords words (some ten thousand groups up to the planner's
// switch to open addressing at 2^30 / ((keys + targets) x 8) entries, QE/MemoryLayoutBuilder.cpp:176-179) used to go to
// hdk_scan_agg_global: one or two memory-side atomics per row, 1.1e10 rows/s whatever the table size -- seven times
// slower than the OPEN-ADDRESSING group-by of the same keys (scan_agg_partitioned.h), although the perfect layout is the
// easier one: entry = key - min, no hash, no probe sequence.  Same remedy, simpler passes:
//   pass A  k_pp_scatter     rows -> filters -> tuples [entry | argument words], scattered by entry range (<= 256 bins)
//   pass A2 k_pp_scatter2    (more than 256 slices) one block per (bin, XCD) sub-slab, by slice
//   pass B  k_pp_aggregate   one block per slice of 2^k entries: the slice's rows of the (initialised) table into LDS, the
//                            tuples applied there (the agg_* of part_apply_targets on LDS rows), the rows written back
// The reference's row function for this layout: get_group_value_fast[_keyless] + agg_* (QE/RowFuncBuilder.cpp:597-745,
// QE/GroupByRuntime.cpp:202-246).  A key distribution that overflows a slab (capacities come from the row count: a hot key)
// raises a flag; pass B then leaves the table alone and hdk_scan_agg_global, armed behind it, does the launch.
#pragma once
#include "part_scatter_batch.h"  // pb_scatter_batch: the run-staging scatter
#include "scan_agg_partitioned.h"  // PartTarget, part_apply_targets
#include "plain_quals.h"

namespace hdk {

constexpr int kPpMaxArgs = 2;
constexpr int kPpMaxKeys = 3;
constexpr int kPpMaxKeySlots = 8;  // projected-key slots of a row (HDK_AGG_ID targets)
constexpr uint32_t kPpLdsBytes = 60 * 1024;

struct PpArgs {
  KernParams kp;
  // the keys: one to three plain integer columns of the outer table; entry = sum((key_k - min_k) * stride_k), perfect_key_hash
  // (QE/RowFuncBuilder.cpp:748-801)
  int32_t nkeys;
  int32_t pad0_;
  ProjFastCol key[kPpMaxKeys];
  int32_t key_nullable[kPpMaxKeys];
  int32_t null_has_entry[kPpMaxKeys];  // a NULL key has a slot (translated) -- else it is out of range like any other stranger
  int64_t key_null[kPpMaxKeys];
  int64_t key_min[kPpMaxKeys];
  int64_t key_translated[kPpMaxKeys];  // what a NULL key is stored / hashed as
  uint32_t key_card[kPpMaxKeys];
  uint32_t key_stride[kPpMaxKeys];
  uint32_t entry_count;
  uint32_t row_bytes;
  int32_t nargs;
  int32_t nquals;
  ProjFastCol arg[kPpMaxArgs];
  // an argument may be ONE integer step over plain columns: a op b / a op literal (eval_expr's checked + - *), computed by
  // the scatter pass; the tuple carries the value (NULL operands give the step's NULL, which the target skips)
  struct ArgExpr {
    int32_t form;            // 0: the column itself; 1: a op column b; 2: a op literal
    int32_t op;              // HDK_OP_ADD / SUB / MUL
    int32_t check_width;
    int32_t a_nullable, b_nullable, pad_;
    ProjFastCol b;
    int64_t a_null, b_null, lit, null_out;
  } ax[kPpMaxArgs];
  ProjFastQual q[kMaxPlainQuals];
  int32_t ntargets;          // aggregate targets (PartTarget); projected keys apart
  int32_t nkeyslots;
  PartTarget tg[HDK_HIP_MAX_TARGETS];
  int32_t keyslot_off[kPpMaxKeySlots];
  int32_t keyslot_width[kPpMaxKeySlots];
  int32_t keyslot_key[kPpMaxKeySlots];         // which key the slot holds
  int32_t keyslot_translated[kPpMaxKeySlots];  // 1: the layout's own key slot (holds the TRANSLATED key, as get_group_value_fast stores it); 0: a projected key (agg_id of the key expression: a NULL stays NULL)
  // geometry
  uint32_t slice_log2;       // entries per slice = 1 << slice_log2 (their rows fit kPpLdsBytes)
  uint32_t nslices;
  uint32_t fpc_log2;         // slices per level-1 bin = 1 << fpc_log2 (0: one level)
  uint32_t nb1;
  uint32_t two_level;
  uint32_t members2;         // level 2: blocks per level-1 bin
  uint32_t pad1_;
  // packed: ONE integer argument inside 32 bits by the column statistics travels in the entry's word -- [argument : entry],
  // 8 bytes a tuple instead of 16 (its NULL as INT32_MIN; a value the statistics did not announce raises the flag)
  uint32_t packed;
  int64_t packed_null;       // the argument column's in-band NULL
  int32_t packed_nullable;
  int32_t pad_;
  uint64_t cap1, cap2;
  int64_t* tuples1;          // [nb1][kPbXcds][cap1][TW]
  int64_t* tuples2;          // [nslices][cap2][TW]
  uint32_t* fill1;
  uint32_t* fill2;
  uint32_t* flag;            // 0: partitions; != 0: the armed global-atomics kernel takes the launch
};

// ---- pass A -------------------------------------------------------------------------------------------------------------
// NK: key columns the code is compiled for (1: the single-key form -- entry = key - min, no strides; kPpMaxKeys: any number)
template <int TW, int VR, int NK>
__global__ __launch_bounds__(kPbBlock) void k_pp_scatter(PpArgs a) {
  constexpr int kTile = kPbBlock * VR;
  __shared__ uint32_t s_cnt[kPbMaxBins];
  __shared__ uint4 s_run[kPbMaxBins];
  __shared__ uint32_t s_total;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + static_cast<size_t>(kTile) * TW);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kPbXcds - 1);
  for (int i = tid; i < kPbMaxBins; i += kPbBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const uint32_t bin_shift = a.slice_log2 + a.fpc_log2;
  bool stranger = false, stale = false, overflow = false;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTile - 1) / kTile;
    const int8_t* const* cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      const int64_t row0 = (tile - frag_tile_begin) * kTile;
      int64_t row[VR];
      bool live[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        row[r] = row0 + static_cast<int64_t>(r) * kPbBlock + tid;
        live[r] = row[r] < nrows;
        row[r] = live[r] ? row[r] : 0;
      }
      if (a.nquals) {
        plain_quals_pass<VR>(a.q, a.nquals, cols, row, live, true);
      }
      int64_t tup[VR][TW];
      uint32_t bin[VR];
      // all loads of the batch first (the decoder's width switch is wave-uniform: one branch per column, the row loop inside --
      // with the switch inside the row loop the loads went out one at a time and the pass ran at half the copy rate)
      int64_t kv[NK][VR], xv[VR];
#pragma unroll
      for (int ki = 0; ki < NK; ++ki) {
        if (ki < a.nkeys) {
          const int8_t* kb = cols[a.key[ki].buf_idx];
          if (a.key[ki].width == 8) {
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              kv[ki][r] = gload<int64_t>(kb, row[r], true);
            }
          } else {
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              kv[ki][r] = decode_col_g(kb, a.key[ki].width, a.key[ki].kind, row[r], true);
            }
          }
        }
      }
      if (TW == 1 && a.packed) {
        const int8_t* xb = cols[a.arg[0].buf_idx];
        if (a.arg[0].width == 8) {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            xv[r] = gload<int64_t>(xb, row[r], true);
          }
        } else {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            xv[r] = decode_col_g(xb, a.arg[0].width, a.arg[0].kind, row[r], true);
          }
        }
      }
      // argument columns (and the second operand of an expression argument): the same, loads first
      int64_t av[TW > 1 ? TW - 1 : 1][VR], bv[TW > 1 ? TW - 1 : 1][VR];
#pragma unroll
      for (int w = 1; w < TW; ++w) {
        const int8_t* ab = cols[a.arg[w - 1].buf_idx];
        if (a.arg[w - 1].width == 8 && a.arg[w - 1].kind != HDK_COL_FLOAT) {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            av[w - 1][r] = gload<int64_t>(ab, row[r], true);
          }
        } else {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            av[w - 1][r] = decode_col_g(ab, a.arg[w - 1].width, a.arg[w - 1].kind, row[r], true);
          }
        }
        if (a.ax[w - 1].form == 1) {
          const int8_t* bb = cols[a.ax[w - 1].b.buf_idx];
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            bv[w - 1][r] = decode_col_g(bb, a.ax[w - 1].b.width, a.ax[w - 1].b.kind, row[r], true);
          }
        } else {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            bv[w - 1][r] = 0;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        uint64_t entry = 0;
        bool inside = true;
#pragma unroll
        for (int ki = 0; ki < NK; ++ki) {
          if (ki < a.nkeys) {
            int64_t k = kv[ki][r];
            if (a.key_nullable[ki] && k == a.key_null[ki]) {
              inside = inside && a.null_has_entry[ki];
              k = a.key_translated[ki];
            }
            const uint64_t term = static_cast<uint64_t>(k) - static_cast<uint64_t>(a.key_min[ki]);
            inside = inside && term < a.key_card[ki];
            entry += NK == 1 ? term : term * a.key_stride[ki];
          }
        }
        if (live[r] && !(inside && entry < a.entry_count)) {
          stranger = true;  // a key outside the range the layout was sized for (get_group_value_fast would write past the buffer)
          live[r] = false;
        }
        const uint32_t e32 = live[r] ? static_cast<uint32_t>(entry) : 0u;
        bin[r] = e32 >> bin_shift;
        tup[r][0] = static_cast<int64_t>(e32);
#pragma unroll
        for (int w = 1; w < TW; ++w) {
          int64_t va = av[w - 1][r];
          const PpArgs::ArgExpr& ex = a.ax[w - 1];
          if (ex.form) {
            const int64_t vb = ex.form == 1 ? bv[w - 1][r] : ex.lit;
            if ((ex.a_nullable && va == ex.a_null) || (ex.form == 1 && ex.b_nullable && vb == ex.b_null)) {
              va = ex.null_out;
            } else {
              int64_t res;
              const bool ovf = ex.op == HDK_OP_ADD ? checked_arith(HDK_OP_ADD, va, vb, ex.check_width, &res)
                                                   : (ex.op == HDK_OP_SUB ? checked_arith(HDK_OP_SUB, va, vb, ex.check_width, &res)
                                                                          : checked_arith(HDK_OP_MUL, va, vb, ex.check_width, &res));
              overflow |= live[r] && ovf;
              va = res;
            }
          }
          tup[r][w] = va;
        }
        if (TW == 1 && a.packed) {
          const int64_t x = xv[r];
          const bool is_null = a.packed_nullable && x == a.packed_null;
          const int32_t x32 = is_null ? INT32_MIN : static_cast<int32_t>(x);
          stale |= live[r] && !is_null && (static_cast<int64_t>(x32) != x || x32 == INT32_MIN);
          tup[r][0] = static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(x32)) << 32) | e32);
        }
      }
      pb_scatter_batch<TW, VR>(
          tup, bin, live, s_cnt, s_run, &s_total, s_stage, s_binof, a.tuples1,
          [&](uint32_t b, uint32_t n, uint32_t* base, uint32_t* nfit) {
            *base = atomicAdd(a.fill1 + (static_cast<size_t>(b) * kPbXcds + xcd) * kPbCursorStride, n);
            *nfit = static_cast<uint64_t>(*base) >= a.cap1 ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(n), a.cap1 - *base));
            if (*nfit < n) {
              atomicMax(a.flag, 1u);  // a hot entry range: the global-atomics kernel takes the launch
            }
          },
          [&](uint32_t b, uint64_t pos) { return (static_cast<uint64_t>(b) * kPbXcds + xcd) * a.cap1 + pos; });
    }
    frag_tile_begin += ntiles;
  }
  if (__any(stranger) && (threadIdx.x & (kWave - 1)) == 0) {
    record_error(a.kp.error_code, HDK_HIP_ERR_OUT_OF_SLOTS);
  }
  if (__any(stale) && (threadIdx.x & (kWave - 1)) == 0) {
    atomicMax(a.flag, 1u);  // a value outside what the statistics announced: the global-atomics kernel reads the columns
  }
  if (__any(overflow) && (threadIdx.x & (kWave - 1)) == 0) {
    record_error(a.kp.error_code, HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW);
  }
}

// ---- pass A2 ------------------------------------------------------------------------------------------------------------
template <int TW, int VR>
__global__ __launch_bounds__(kPbBlock) void k_pp_scatter2(PpArgs a) {
  constexpr int kTile = kPbBlock * VR;
  __shared__ uint32_t s_cnt[kPbMaxBins];
  __shared__ uint4 s_run[kPbMaxBins];
  __shared__ uint32_t s_total, s_stop;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + static_cast<size_t>(kTile) * TW);
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_stop = __hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int i = tid; i < kPbMaxBins; i += kPbBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  if (s_stop) {
    return;
  }
  const uint32_t fmask = (1u << a.fpc_log2) - 1u;
  // Blocks b with b % 8 == c % 8 work on level-1 bin c, `members` of them per bin (block ids are dealt to the XCDs round
  // robin: all writers of a slice's slab then sit behind ONE L2 and the partial lines at the ends of their runs merge there;
  // an affinity for speed, any placement is correct).  The bin's eight sub-slabs are one sequence of tiles, dealt to the
  // members in turn.
  const uint32_t members = a.members2;
  const uint32_t lane8 = blockIdx.x % kPbXcds, idx8 = blockIdx.x / kPbXcds;
  const uint32_t member = idx8 % members;
  for (uint32_t b1 = lane8 + kPbXcds * (idx8 / members); b1 < a.nb1; b1 += kPbXcds * (gridDim.x / (kPbXcds * members))) {
   uint32_t turn = 0;
   for (uint32_t x8 = 0; x8 < kPbXcds; ++x8) {
    const uint32_t sub = b1 * kPbXcds + x8;
    const uint64_t n = min(static_cast<uint64_t>(a.fill1[static_cast<size_t>(sub) * kPbCursorStride]), a.cap1);
    const int64_t* src = a.tuples1 + static_cast<uint64_t>(sub) * a.cap1 * TW;
    for (uint64_t t0 = 0; t0 < n; t0 += kTile, ++turn) {
      if (turn % members != member) {
        continue;
      }
      int64_t tup[VR][TW];
      uint32_t bin[VR];
      bool live[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint64_t i = t0 + static_cast<uint64_t>(r) * kPbBlock + tid;
        live[r] = i < n;
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          tup[r][w] = live[r] ? __builtin_nontemporal_load(src + i * TW + w) : 0;
        }
        bin[r] = (static_cast<uint32_t>(tup[r][0]) >> a.slice_log2) & fmask;
      }
      pb_scatter_batch<TW, VR>(
          tup, bin, live, s_cnt, s_run, &s_total, s_stage, s_binof, a.tuples2,
          [&](uint32_t f, uint32_t cnt, uint32_t* base, uint32_t* nfit) {
            const uint32_t slice = (b1 << a.fpc_log2) + f;
            *base = atomicAdd(a.fill2 + static_cast<size_t>(slice) * kPbCursor2Stride, cnt);
            *nfit = static_cast<uint64_t>(*base) >= a.cap2 ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(cnt), a.cap2 - *base));
            if (*nfit < cnt) {
              atomicMax(a.flag, 1u);
            }
          },
          [&](uint32_t f, uint64_t pos) { return static_cast<uint64_t>((b1 << a.fpc_log2) + f) * a.cap2 + pos; });
    }
   }
  }
}

// ---- pass B -------------------------------------------------------------------------------------------------------------
// dynamic LDS: the slice's rows
template <int TW>
__global__ __launch_bounds__(kPbBlock) void k_pp_aggregate(PpArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t s_rows[];
  __shared__ PartTarget s_tg[HDK_HIP_MAX_TARGETS];
  __shared__ uint32_t s_stop;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) {
    s_stop = __hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < static_cast<uint32_t>(a.ntargets)) {
    s_tg[tid] = a.tg[tid];
  }
  __syncthreads();
  if (s_stop) {
    return;  // (hdk_scan_agg_global, armed behind this kernel, aggregates the launch)
  }
  const uint32_t slots = 1u << a.slice_log2;
  const uint32_t rq = a.row_bytes / 8;
  int64_t* table = a.kp.groupby_buf[0];
  for (uint32_t s = blockIdx.x; s < a.nslices; s += gridDim.x) {
    const uint32_t e0 = s << a.slice_log2;
    const uint32_t n_e = min(slots, a.entry_count - e0);
    const uint32_t nq = n_e * rq;
    int64_t* rows = table + static_cast<size_t>(e0) * rq;
    for (uint32_t i = tid; i < nq; i += kPbBlock) {
      s_rows[i] = rows[i];
    }
    __syncthreads();
    const int nsrc = a.two_level ? 1 : kPbXcds;
    for (int x = 0; x < nsrc; ++x) {
      const int64_t* src;
      uint64_t n;
      if (a.two_level) {
        src = a.tuples2 + static_cast<uint64_t>(s) * a.cap2 * TW;
        n = min(static_cast<uint64_t>(a.fill2[static_cast<size_t>(s) * kPbCursor2Stride]), a.cap2);
      } else {
        const uint64_t sub = static_cast<uint64_t>(s) * kPbXcds + x;
        src = a.tuples1 + sub * a.cap1 * TW;
        n = min(static_cast<uint64_t>(a.fill1[sub * kPbCursorStride]), a.cap1);
      }
      constexpr int U = 4;  // tuples in flight per lane
      for (uint64_t i0 = tid; i0 < n; i0 += static_cast<uint64_t>(U) * kPbBlock) {
        int64_t tt[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint64_t i = i0 + static_cast<uint64_t>(u) * kPbBlock;
#pragma unroll
          for (int w = 0; w < 3; ++w) {
            tt[u][w] = (w < TW && i < n) ? __builtin_nontemporal_load(src + i * TW + w) : 0;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (i0 + static_cast<uint64_t>(u) * kPbBlock >= n) {
            break;
          }
          int64_t* t = tt[u];
          const uint32_t entry = static_cast<uint32_t>(t[0]);
          if (TW == 1 && a.packed) {
            const int32_t x32 = static_cast<int32_t>(static_cast<uint64_t>(t[0]) >> 32);
            t[1] = (a.packed_nullable && x32 == INT32_MIN) ? a.packed_null : static_cast<int64_t>(x32);
          }
          const uint32_t local = entry - e0;
          if (local < n_e) {
            int8_t* rowb = reinterpret_cast<int8_t*>(s_rows + static_cast<size_t>(local) * rq);
            if (a.nkeyslots) {  // agg_id / the layout's key slots: every row of a group stores the same keys
              for (int ks = 0; ks < a.nkeyslots; ++ks) {
                const int ki = a.keyslot_key[ks];
                const uint32_t term = a.nkeys == 1 ? entry : (entry / a.key_stride[ki]) % a.key_card[ki];
                const int64_t stored = static_cast<int64_t>(static_cast<uint64_t>(a.key_min[ki]) + term);
                const int64_t key = (!a.keyslot_translated[ks] && a.null_has_entry[ki] && stored == a.key_translated[ki]) ? a.key_null[ki] : stored;
                if (a.keyslot_width[ks] == 4) {
                  *reinterpret_cast<int32_t*>(rowb + a.keyslot_off[ks]) = static_cast<int32_t>(key);
                } else {
                  *reinterpret_cast<int64_t*>(rowb + a.keyslot_off[ks]) = key;
                }
              }
            }
            part_apply_targets(s_tg, a.ntargets, rowb, t);
          }
        }
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nq; i += kPbBlock) {
      rows[i] = s_rows[i];
    }
    __syncthreads();
  }
}

}  // namespace hdk
