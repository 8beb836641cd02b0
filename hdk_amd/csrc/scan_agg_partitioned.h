// scan_agg_partitioned.h -- open-addressing group-by through radix partitioning + LDS aggregation.
//
// Why: random global atomics run at ~2.4e10/s chip-wide whatever the table size
// (scripts/microbench/atomics.hip), so hdk_scan_agg_baseline_direct cannot pass ~2e10 rows/s.  The only
// way around the memory-side atomic units is locality: bring all rows of a slot range together and
// aggregate them in LDS.  Same result contract as the other baseline kernels (reference get_group_value +
// agg_*: QE/GroupByRuntime.cpp:31-200, QE/RuntimeFunctions.cpp:387-875): every group sits in exactly one
// entry of the GroupByBaselineHash buffer with the reference's row layout; which entry is unobservable
// (consumers iterate entries; the reference's own placement depends on thread timing).
//
//   fine partition f of a key = mulhi32(key_hash(key), PF)  ->  table entries [f*S, (f+1)*S)
//   pass 1  filter rows, scatter (keys + argument columns -> tuples of <= 3 words) into 128 coarse slabs (c = f / P2)
//   pass 2  scatter each coarse slab into its P2 fine slabs
//   pass 3  one block per fine slab: load its (initialised) table region into LDS, insert/aggregate the
//           slab's tuples there with the ordinary claim protocol and agg_* functions, store the region
//   pass 4  tuples that did not fit their slab (slabs are sized for a uniform hash with slack, not counted)
//           are applied to their region with global atomics
//   skew    when even the overflow area fills up (a heavy-hitter key: its rows all land in one slab) the
//           scatter passes raise a device-side flag, the later passes return at once, and the launch's last
//           kernel -- hdk_scan_agg_baseline_direct, armed by that flag -- redoes the job with global atomics on
//           the still untouched table.  No host round trip.
// Scatter = per 2048-row batch (512 threads x 4 rows): LDS histogram by partition, ONE global cursor atomic per partition and
// batch, LDS staging ordered by partition, coalesced copy-out of the runs (~16 tuples = 256 B each).
#pragma once
#include "scan_agg_baseline_fast.h"

namespace hdk {

constexpr int kPartBlock = 512;               // scatter passes (sweep at C5: 256x8 3.7 ms, 512x4 2.9 ms, 1024x2 3.9 ms for pass 1)
constexpr int kPartAggBlock = 1024;              // aggregation pass: 2 blocks x 60 KiB LDS per CU, all 32 wave slots busy
constexpr int kPartVR = 4;
constexpr int kPartTile = kPartBlock * kPartVR;  // tuples per scatter batch
constexpr int kPartP1 = 128;                     // coarse partitions
constexpr int kPartMaxBins = 256;                // bins a scatter pass distinguishes (P1, or P2 <= 256)
constexpr int kPartMaxArgs = 2;                  // argument columns carried in a tuple (at most)
constexpr int kPartMaxTW = 1 + kPartMaxArgs;     // tuple words: 1-2 keys + argument columns, 3 in all (LDS staging)
constexpr uint32_t kPartLdsWords = 7680;         // 60 KiB LDS table per fine partition

struct PartArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  uint32_t entry_count;
  uint32_t slots;        // S: entries per fine partition
  uint32_t fine_count;   // PF = P2 * kPartP1
  uint32_t p2;           // fine partitions per coarse partition
  int32_t tw;            // tuple words
  int32_t key_buf_idx, key_width, key_kind;
  int32_t nkeys;         // 1 or 2 key columns: tuple = [key0, (key1), arguments...]
  int32_t key2_buf_idx, key2_width, key2_kind;
  int32_t nargs;
  BaseFastTarget arg[kPartMaxArgs];  // argument columns (buf_idx / width / kind); .target unused
  int32_t ntargets;
  int32_t tgt_index[HDK_HIP_MAX_TARGETS];  // plan target index
  int32_t tgt_arg[HDK_HIP_MAX_TARGETS];    // tuple word of its argument (>= nkeys), or 0 for none
  uint64_t cap1, cap2, cap_ovf;            // slab capacities in tuples
  int64_t* slab1;        // [kPartP1][cap1][tw]
  int64_t* slab2;        // [fine_count][cap2][tw]
  int64_t* ovf;          // [cap_ovf][tw]
  uint32_t* fill1;       // [kPartP1]
  uint32_t* fill2;       // [fine_count]
  uint32_t* fill_ovf;    // [1]
  uint32_t* fallback;    // [1]: set when the overflow area is exhausted -> the atomics kernel takes over
  int32_t nquals;        // plain filters, applied in pass 1
  ProjFastQual q[kMaxPlainQuals];
};

template <typename K>
HDK_DEV uint32_t part_fine_id(const int64_t* tup, int nkeys, uint32_t fine_count) {
  const K k[2] = {static_cast<K>(tup[0]), static_cast<K>(tup[1])};
  const uint32_t h = nkeys == 2 ? key_hash_dev<K>(k, 2) : key_hash_dev<K>(k, 1);  // constant trip counts unroll
  return static_cast<uint32_t>((static_cast<uint64_t>(h) * fine_count) >> 32);
}

// ---- scatter: LEVEL 1 reads the columns, LEVEL 2 reads coarse slab blockIdx.y -------------------------
template <int LEVEL, typename K>
__global__ __launch_bounds__(kPartBlock) void hdk_part_scatter(PartArgs a) {
  __shared__ uint32_t s_cnt[kPartMaxBins];
  __shared__ uint32_t s_lpos[kPartMaxBins];   // start of the bin's run in the staging area
  __shared__ uint32_t s_base[kPartMaxBins];   // first slab position claimed for the bin
  extern __shared__ __attribute__((aligned(16))) int64_t s_stage[];  // [kPartTile][tw] (dynamic: 16 KiB per word)
  __shared__ uint32_t s_pos[kPartTile];       // destination: position inside the bin's slab / overflow area
  __shared__ uint16_t s_bin[kPartTile];       // destination: bin, 0xFFFF = overflow area, 0xFFFE = dropped
  constexpr int VR = kPartVR;
  const int tid = threadIdx.x;
  const int tw = a.tw;
  const uint32_t nbins = LEVEL == 1 ? kPartP1 : a.p2;
  const uint64_t cap = LEVEL == 1 ? a.cap1 : a.cap2;
  uint32_t* fill = LEVEL == 1 ? a.fill1 : a.fill2 + static_cast<size_t>(blockIdx.y) * a.p2;
  int64_t* out = LEVEL == 1 ? a.slab1 : a.slab2 + static_cast<size_t>(blockIdx.y) * a.p2 * cap * tw;
  for (int i = tid; i < kPartMaxBins; i += kPartBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();

  auto do_batch = [&](const bool (&live)[VR], int64_t (&tup)[VR][kPartMaxTW]) {
    // 1. histogram + rank inside the bin
    uint32_t bin[VR], rank[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      bin[r] = 0;
      rank[r] = 0;
      if (live[r]) {
        const uint32_t f = part_fine_id<K>(tup[r], a.nkeys, a.fine_count);
        bin[r] = LEVEL == 1 ? f / a.p2 : f % a.p2;
        rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
      }
    }
    __syncthreads();
    // 2. one slab claim per bin; run starts in the staging area (exclusive scan by wave 0)
    if (tid < kPartMaxBins) {
      const uint32_t n = tid < static_cast<int>(nbins) ? s_cnt[tid] : 0;
      s_base[tid] = n ? atomicAdd(fill + tid, n) : 0u;
    }
    if (tid < kWave) {
      uint32_t carry = 0;
      for (int c0 = 0; c0 < kPartMaxBins; c0 += kWave) {
        const uint32_t n = s_cnt[c0 + tid];
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint32_t v = __shfl_up(incl, d, kWave);
          if (tid >= d) {
            incl += v;
          }
        }
        s_lpos[c0 + tid] = carry + incl - n;
        carry += __shfl(incl, kWave - 1, kWave);
      }
    }
    __syncthreads();
    // 3. stage the tuples ordered by bin, remember where each goes
    uint32_t staged = 0;
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      if (live[r]) {
        const uint32_t si = s_lpos[bin[r]] + rank[r];
        const uint64_t pos = static_cast<uint64_t>(s_base[bin[r]]) + rank[r];
        if (pos < cap) {
          s_bin[si] = static_cast<uint16_t>(bin[r]);
          s_pos[si] = static_cast<uint32_t>(pos);
        } else {  // the slab is full (heavy hitter): overflow area, applied with atomics at the end
          const uint32_t o = atomicAdd(a.fill_ovf, 1u);
          s_bin[si] = o < a.cap_ovf ? 0xFFFFu : 0xFFFEu;
          s_pos[si] = o;
          if (o >= a.cap_ovf) {
            atomicExch(a.fallback, 1u);  // too skewed for slabs: hand the launch to the atomics kernel
          }
        }
#pragma unroll
        for (int w = 0; w < kPartMaxTW; ++w) {
          if (w < tw) {
            s_stage[static_cast<size_t>(si) * tw + w] = tup[r][w];
          }
        }
        ++staged;
      }
    }
    (void)staged;
    __syncthreads();
    // 4. copy out: consecutive staging slots of a bin go to consecutive slab positions
    const uint32_t total = s_lpos[kPartMaxBins - 1] + s_cnt[kPartMaxBins - 1];
    for (uint32_t i = tid; i < total; i += kPartBlock) {
      const uint32_t b = s_bin[i];
      if (b == 0xFFFEu) {
        continue;
      }
      int64_t* q = b == 0xFFFFu ? a.ovf + static_cast<size_t>(s_pos[i]) * tw
                                : out + (static_cast<size_t>(b) * cap + s_pos[i]) * tw;
#pragma unroll
      for (int w = 0; w < kPartMaxTW; ++w) {
        if (w < tw) {
          q[w] = s_stage[static_cast<size_t>(i) * tw + w];
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < kPartMaxBins; i += kPartBlock) {
      s_cnt[i] = 0;
    }
    __syncthreads();
  };

  if (LEVEL == 2 && __hip_atomic_load(a.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    return;
  }
  if (LEVEL == 1) {
    const uint64_t nfrag = *a.kp.num_fragments;
    const uint32_t ntab = *a.kp.num_tables;
    int64_t tile = blockIdx.x;
    int64_t frag_tile_begin = 0;
    for (uint64_t f = 0; f < nfrag; ++f) {
      const int64_t nrows = a.kp.num_rows[f * ntab];
      const int64_t ntiles = (nrows + kPartTile - 1) / kPartTile;
      const int8_t* const* cols = a.kp.col_buffers[f];
      for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
        // block-uniform exit: thread 0 samples the flag, everyone agrees before the batch's barriers
        if (tid == 0) {
          s_base[0] = __hip_atomic_load(a.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const uint32_t give_up = s_base[0];
        __syncthreads();
        if (give_up) {
          return;
        }
        const int64_t row0 = (tile - frag_tile_begin) * kPartTile + tid;
        bool live[VR];
        int64_t tup[VR][kPartMaxTW];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          live[r] = row0 + static_cast<int64_t>(r) * kPartBlock < nrows;
        }
        if (a.nquals) {
          int64_t rows[VR];
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            rows[r] = row0 + static_cast<int64_t>(r) * kPartBlock;
          }
          plain_quals_pass<VR>(a.q, a.nquals, cols, rows, live, true);
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
          tup[r][0] = live[r] ? decode_col_g(cols[a.key_buf_idx], a.key_width, a.key_kind, row, true) : 0;
        }
        const int nk = a.nkeys;
#pragma unroll
        for (int w = 1; w < kPartMaxTW; ++w) {
          if (w < nk) {  // second key column
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
              tup[r][w] = live[r] ? decode_col_g(cols[a.key2_buf_idx], a.key2_width, a.key2_kind, row, true) : 0;
            }
          } else if (w - nk < a.nargs) {
            const BaseFastTarget c = a.arg[w - nk > 0 ? 1 : 0];
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
              tup[r][w] = live[r] ? decode_col_g(cols[c.buf_idx], c.width, c.kind, row, true) : 0;
            }
          }
        }
        do_batch(live, tup);
      }
      frag_tile_begin += ntiles;
    }
  } else {
    const uint32_t c = blockIdx.y;
    const uint64_t n = min(static_cast<uint64_t>(a.fill1[c]), a.cap1);
    const int64_t* in = a.slab1 + static_cast<size_t>(c) * a.cap1 * tw;
    for (uint64_t t0 = static_cast<uint64_t>(blockIdx.x) * kPartTile; t0 < n; t0 += static_cast<uint64_t>(gridDim.x) * kPartTile) {
      bool live[VR];
      int64_t tup[VR][kPartMaxTW];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint64_t i = t0 + static_cast<uint64_t>(r) * kPartBlock + tid;
        live[r] = i < n;
#pragma unroll
        for (int w = 0; w < kPartMaxTW; ++w) {
          tup[r][w] = (live[r] && w < tw) ? __builtin_nontemporal_load(in + i * tw + w) : 0;
        }
      }
      do_batch(live, tup);
    }
  }
}

// What a tuple needs to know about a target, gathered once per block into LDS: reading the plan (global
// memory) per tuple put ~25 dependent global loads on every tuple's path (10.7 ms for the C5 shape).
struct PartTarget {
  int32_t agg, has_arg, skip_null, arg_is_fp;
  int32_t slot_width, slot2_width, slot_off, slot2_off;
  int32_t arg_word, arg_fp, arg_nullable, pad_;
  int64_t null_val;      // skip value of the slot
  int64_t arg_null_val;  // in-band NULL of the argument
};

HDK_DEV void part_load_targets(const PartArgs& a, const hdk_hip_plan* p, PartTarget* s_tg) {
  if (static_cast<int>(threadIdx.x) < a.ntargets) {
    const int t = threadIdx.x;
    const hdk_hip_target& tg = p->targets[a.tgt_index[t]];
    PartTarget d;
    d.agg = tg.agg;
    d.has_arg = tg.has_arg;
    d.skip_null = tg.skip_null;
    d.arg_is_fp = tg.arg_is_fp;
    d.slot_width = tg.slot_width;
    d.slot2_width = tg.slot2_width;
    d.slot_off = tg.slot_off;
    d.slot2_off = tg.slot2_off;
    d.arg_word = a.tgt_arg[t];
    const int32_t kind = d.arg_word ? a.arg[d.arg_word - a.nkeys].kind : HDK_COL_INT;
    d.arg_fp = kind == HDK_COL_FLOAT || kind == HDK_COL_DOUBLE;
    d.arg_nullable = tg.arg.nullable;
    d.pad_ = 0;
    d.null_val = tg.null_val;
    d.arg_null_val = tg.arg.null_val;
    s_tg[t] = d;
  }
}

// one tuple -> its group in `table` (LDS image of a region, or the region itself): claim + aggregates
template <typename K>
HDK_DEV void part_apply_tuple(const TableShape shape, const PartTarget* s_tg, int ntargets, uint32_t rq, int64_t* table,
                              uint32_t slots, const int64_t* tup, int32_t& err) {
  const K key[2] = {static_cast<K>(tup[0]), static_cast<K>(tup[1])};  // (word 1 is only read as a key when key_count == 2)
  bool fresh;
  const int64_t e = find_or_claim<K>(shape, table, slots, key, &fresh);
  if (e < 0) {
    err = HDK_HIP_ERR_OUT_OF_SLOTS;  // more groups in this slot range than it has entries
    return;
  }
  int8_t* rowb = reinterpret_cast<int8_t*>(table + static_cast<size_t>(e) * rq);
  for (int t = 0; t < ntargets; ++t) {
    const PartTarget tg = s_tg[t];
    int8_t* s1 = rowb + tg.slot_off;
    int8_t* s2 = rowb + tg.slot2_off;
    int64_t v = tg.arg_word == 1 ? tup[1] : (tg.arg_word == 2 ? tup[2] : 0);
    bool is_null = false;
    if (tg.has_arg) {  // eval_target_arg for a plain column argument (device_common.h)
      const bool arg_fp = tg.arg_fp != 0;
      if (tg.skip_null && is_null_val(v, tg.arg_null_val, tg.arg_nullable, arg_fp)) {
        is_null = true;
      } else {
        if (tg.arg_is_fp && !arg_fp) {
          v = double_to_bits(static_cast<double>(v));
        }
        if (tg.skip_null) {
          is_null = tg.arg_is_fp ? (bits_to_double(v) == bits_to_double(tg.null_val)) : (v == tg.null_val);
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
      g_agg32(tg.agg, tg.skip_null, static_cast<int32_t>(tg.null_val), reinterpret_cast<int32_t*>(s1), static_cast<int32_t>(v));
    } else {
      g_agg64(tg.agg, tg.arg_is_fp, tg.skip_null, tg.null_val, reinterpret_cast<int64_t*>(s1), v);
    }
  }
}

// ---- pass 3: one block per fine partition ----------------------------------------------------------------
template <typename K>
__global__ __launch_bounds__(kPartAggBlock) void hdk_part_aggregate(PartArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds_table[];
  __shared__ PartTarget s_tg[HDK_HIP_MAX_TARGETS];
  const hdk_hip_plan* __restrict__ p = a.plan;
  const int tid = threadIdx.x;
  const uint32_t f = blockIdx.x;
  const TableShape shape = table_shape(p);
  const uint32_t rq = shape.row_quads;
  const int ntargets = a.ntargets;
  const uint32_t slots = a.slots;
  part_load_targets(a, p, s_tg);
  const uint32_t words = a.slots * rq;
  int64_t* region = a.kp.groupby_buf[0] + static_cast<size_t>(f) * words;
  const uint64_t n = min(static_cast<uint64_t>(a.fill2[f]), a.cap2);
  if (n == 0 || *a.fallback) {
    return;  // the region keeps its initialised (empty) image
  }
  for (uint32_t i = tid; i < words; i += kPartAggBlock) {
    lds_table[i] = region[i];  // the init kernel's image: empty keys + init values, whatever the layout
  }
  __syncthreads();
  const int tw = a.tw;
  const int64_t* in = a.slab2 + static_cast<size_t>(f) * a.cap2 * tw;
  int32_t err = 0;
  // Two tuples per trip, the next pair's loads issued before the current pair is applied: the LDS
  // claim/aggregate chain (ds_* ops, lgkmcnt) of one pair hides the HBM latency (vmcnt) of the next.
  // Plain scalars on purpose -- no per-thread tuple arrays that could end up in scratch.
  int64_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
  auto fetch = [&](uint64_t i, int64_t& t0, int64_t& t1, int64_t& t2) {
    if (i < n) {
      const int64_t* q = in + i * tw;
      t0 = __builtin_nontemporal_load(q);
      t1 = tw > 1 ? __builtin_nontemporal_load(q + 1) : 0;
      t2 = tw > 2 ? __builtin_nontemporal_load(q + 2) : 0;
    }
  };
  fetch(tid, a0, a1, a2);
  fetch(static_cast<uint64_t>(tid) + kPartAggBlock, b0, b1, b2);
  for (uint64_t i = tid; i < n; i += 2 * kPartAggBlock) {
    const int64_t ta[kPartMaxTW] = {a0, a1, a2};
    const int64_t tb[kPartMaxTW] = {b0, b1, b2};
    const bool has_b = i + kPartAggBlock < n;
    fetch(i + 2 * kPartAggBlock, a0, a1, a2);
    fetch(i + 3 * kPartAggBlock, b0, b1, b2);
    part_apply_tuple<K>(shape, s_tg, ntargets, rq, lds_table, slots, ta, err);
    if (has_b) {
      part_apply_tuple<K>(shape, s_tg, ntargets, rq, lds_table, slots, tb, err);
    }
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
  __syncthreads();
  for (uint32_t i = tid; i < words; i += kPartAggBlock) {
    region[i] = lds_table[i];
  }
}

// ---- pass 4: overflow tuples, straight onto their region with global atomics -----------------------------
template <typename K>
__global__ __launch_bounds__(kPartBlock) void hdk_part_overflow(PartArgs a) {
  __shared__ PartTarget s_tg[HDK_HIP_MAX_TARGETS];
  const hdk_hip_plan* __restrict__ p = a.plan;
  if (*a.fallback) {
    return;
  }
  part_load_targets(a, p, s_tg);
  __syncthreads();
  const uint64_t n = min(static_cast<uint64_t>(*a.fill_ovf), a.cap_ovf);
  const TableShape shape = table_shape(p);
  const uint32_t rq = shape.row_quads;
  const int tw = a.tw;
  int32_t err = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kPartBlock + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * kPartBlock) {
    int64_t tup[kPartMaxTW];
#pragma unroll
    for (int w = 0; w < kPartMaxTW; ++w) {
      tup[w] = w < tw ? a.ovf[i * tw + w] : 0;
    }
    const uint32_t f = part_fine_id<K>(tup, a.nkeys, a.fine_count);
    int64_t* region = a.kp.groupby_buf[0] + static_cast<size_t>(f) * a.slots * rq;
    part_apply_tuple<K>(shape, s_tg, a.ntargets, rq, region, a.slots, tup, err);
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
