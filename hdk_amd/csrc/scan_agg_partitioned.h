// scan_agg_partitioned.h -- open-addressing group-by through radix partitioning + LDS aggregation.
//
// Why: random global atomics run at ~2.4e10/s chip-wide whatever the table size
// (scripts/microbench/atomics.hip), so hdk_scan_agg_baseline_direct cannot pass ~2e10 rows/s.  The only
// way around the memory-side atomic units is locality: bring all rows of an entry range together and
// aggregate them in LDS.  Same result contract as the other baseline kernels (reference get_group_value +
// agg_*: QE/GroupByRuntime.cpp:31-200, QE/RuntimeFunctions.cpp:387-875), INCLUDING the placement: a group
// sits where the reference's probe sequence (key_hash % entry_count, then linearly on) finds it, so the
// table can be handed to anything that looks groups up -- a second launch into the same buffer,
// hdk_hip_reduce_buffers (ResultSetReduction::reduceOneEntryBaseline, QE/ResultSetReduction.cpp:694-731).
//
//   home(key)   = key_hash(key) % entry_count                      (the reference's first probe)
//   region f    = home >> slots_log2: table entries [f * S, min((f + 1) * S, entry_count)),  S = 2^slots_log2
//   pass 1  filter rows, scatter (keys + argument columns -> tuples of <= 3 words) into P1 coarse slabs
//           (c = f >> p2_log2)
//   pass 2  scatter each coarse slab into its P2 = 2^p2_log2 fine slabs (one per region)
//   pass 3  one block per region: load its (initialised) image into LDS, insert/aggregate the slab's tuples
//           there, probing from home - f * S to the END of the region (no wrap), store the image back.  A
//           tuple whose probe runs off the region's end is set aside ("spilled")
//   pass 4  spilled tuples, and tuples that did not fit their slab (slabs are sized for a uniform hash with
//           slack, not counted), go through the ordinary whole-table find_or_claim with global atomics: the
//           probe walks over the full tail of the tuple's region into the next one, exactly as the reference's
//   skew    when even the overflow area fills up (a heavy-hitter key: its rows all land in one slab) the
//           scatter passes raise a device-side flag, the later passes return at once, and the launch's last
//           kernel -- hdk_scan_agg_baseline_direct, armed by that flag -- redoes the job with global atomics on
//           the still untouched table.  No host round trip.
//
// Scatter, per batch of kPartTile tuples: LDS histogram by bin -> one slab claim per bin -> LDS staging ordered
// by bin -> copy-out.  What the passes wait for is the copy-out (scripts/microbench/partition.hip): runs that
// start and end inside 128-byte lines took 2.3-2.7 ms per 256 M tuples, the same bytes as whole aligned lines
// 1.7 ms (4.7 TB/s of 32 B/tuple), with or without the cursor atomics.  So a block only ever writes WHOLE
// LINES: per bin it flushes a multiple of G tuples (G * tuple bytes = a multiple of 128) at a slab position that
// is a multiple of G, and carries the remaining < G tuples into its next batch (they are re-read from the staging
// area and re-ranked like new tuples).  After its last batch a block puts what is left (< G per bin) into a
// small per-bin TAIL slab with an exact count, so readers never meet a hole.  Slab cursors sit one per 128-byte
// line (128 cursors in 512 bytes serialise on four lines: 2.68 -> 2.34 ms).
#pragma once
#include "scan_agg_baseline_fast.h"

namespace hdk {

constexpr int kPartBlock = 512;                  // scatter passes (sweep at C5: 256x8 3.7 ms, 512x4 2.9 ms, 1024x2 3.9 ms for pass 1)
constexpr int kPartAggBlock = 1024;              // aggregation pass: 2 blocks x 64 KiB LDS per CU, all 32 wave slots busy
constexpr int kPartVR = 4;
constexpr int kPartTile = kPartBlock * kPartVR;  // new tuples per scatter batch
constexpr int kPartLV = 4;                       // carried tuples a thread can take along (kPartBlock * kPartLV >= bins * (G - 1))
constexpr int kPartMaxBins = 256;                // bins a scatter pass distinguishes (P1 <= 256, P2 <= 256)
constexpr int kPartMaxArgs = 2;                  // argument columns carried in a tuple (at most)
constexpr int kPartMaxTW = 1 + kPartMaxArgs;     // tuple words: 1-2 keys + argument columns, 3 in all (LDS staging)
constexpr uint32_t kPartLdsBytes = 64 * 1024;    // LDS image of a region
constexpr uint32_t kPartCursorStride = 32;       // uint32 cursors one per 128-byte line
constexpr int kPartG2X = 16;                     // pass-2 blocks per coarse slab

struct PartArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  uint32_t entry_count;
  uint64_t total_rows;   // upper bound on the rows of the launch (hdk_hip_kernel_options::total_rows)
  uint32_t slots_log2;   // S = 1 << slots_log2 entries per region
  uint32_t fine_count;   // PF = ceil(entry_count / S) regions
  uint32_t p2_log2;      // regions per coarse partition = 1 << p2_log2
  uint32_t p1;           // coarse partitions = ceil(PF / P2)
  uint32_t mod_magic, mod_shift;  // h % entry_count without a division (fastmod_u32)
  int32_t tw;            // tuple words
  int32_t g_log2;        // flush granule G = 1 << g_log2 tuples (whole 128-byte lines)
  int32_t key_buf_idx, key_width, key_kind;
  int32_t nkeys;         // 1 or 2 key columns: tuple = [key0, (key1), arguments...]
  int32_t key2_buf_idx, key2_width, key2_kind;
  int32_t nargs;
  BaseFastTarget arg[kPartMaxArgs];  // argument columns (buf_idx / width / kind); .target unused
  int32_t ntargets;
  int32_t tgt_index[HDK_HIP_MAX_TARGETS];  // plan target index
  int32_t tgt_arg[HDK_HIP_MAX_TARGETS];    // tuple word of its argument (>= nkeys), or 0 for none
  uint64_t cap1, cap2, cap_ovf, cap_spill; // capacities in tuples (cap1, cap2: multiples of G)
  uint32_t tcap1, tcap2; // tail slab capacities: blocks writing the bin x (G - 1)
  int64_t* slab1;        // [p1][cap1][tw]            (pass 3 reuses this memory as the spill area)
  int64_t* tail1;        // [p1][tcap1][tw]
  int64_t* slab2;        // [fine_count][cap2][tw]
  int64_t* tail2;        // [fine_count][tcap2][tw]
  int64_t* ovf;          // [cap_ovf][tw]
  uint32_t* fill1;       // [p1] x kPartCursorStride
  uint32_t* tfill1;      // [p1]
  uint32_t* fill2;       // [fine_count]
  uint32_t* tfill2;      // [fine_count]
  uint32_t* fill_ovf;    // [1]
  uint32_t* fill_spill;  // [1]
  uint32_t* fallback;    // [1]: set when the overflow area is exhausted -> the atomics kernel takes over
  int32_t nquals;        // plain filters, applied in pass 1
  ProjFastQual q[kMaxPlainQuals];
};

// h % d for any 32-bit h: unsigned division by an invariant divisor, round-up method in its branch-free form
// (33-bit magic number, low 32 bits in `magic`): one v_mul_hi_u32 + one v_mul_lo_u32 instead of a division
HDK_DEV uint32_t fastmod_u32(uint32_t h, uint32_t magic, uint32_t shift, uint32_t d) {
  const uint32_t t = __umulhi(magic, h);
  const uint32_t q = (((h - t) >> 1) + t) >> shift;
  return h - q * d;
}

// the reference's first probe position of a tuple's key
template <typename K, int TW = kPartMaxTW>
HDK_DEV uint32_t part_home(const PartArgs& a, const int64_t* tup) {
  const K k[2] = {static_cast<K>(tup[0]), TW > 1 ? static_cast<K>(tup[TW > 1 ? 1 : 0]) : K(0)};
  const uint32_t h = (TW > 1 && a.nkeys == 2) ? key_hash_dev<K>(k, 2) : key_hash_dev<K>(k, 1);  // constant trip counts unroll
  return fastmod_u32(h, a.mod_magic, a.mod_shift, a.entry_count);
}

// staging capacity of a scatter batch: the new tuples plus at most G - 1 carried ones per bin (multiple of 8)
__host__ __device__ inline uint32_t part_stage_tuples(uint32_t nbins, uint32_t gmask) {
  return (kPartTile + nbins * gmask + 7u) & ~7u;
}
__host__ inline size_t part_scatter_lds_bytes(uint32_t nbins, uint32_t gmask, int tw) {
  const size_t cs = part_stage_tuples(nbins, gmask);
  return cs * tw * 8 + cs * 2 + static_cast<size_t>(nbins) * gmask * 2 + 16;
}

// ---- scatter: LEVEL 1 reads the columns, LEVEL 2 reads coarse slab blockIdx.y -------------------------
// dynamic LDS: [cap_stage][tw] staging | uint16 bin of every staged tuple [cap_stage] | uint16 staging index of
// every carried tuple [bins x (G - 1)]
template <int LEVEL, typename K, int TW>
__global__ __launch_bounds__(kPartBlock) void hdk_part_scatter(PartArgs a) {
  __shared__ uint32_t s_cnt[kPartMaxBins];     // tuples of the bin in this batch (carried + new); rank source
  __shared__ uint32_t s_lpos[kPartMaxBins];    // start of the bin's run in the staging area
  // per bin and batch, read as one 16-byte word by the copy-out: .x start of the run in the staging area, .y tuples
  // of the run that leave in this batch, .z first position claimed for them, .w 0 slab, 1 tail slab, 2 slab +
  // overflow area, 3 slab + dropped (fallback armed)
  __shared__ uint4 s_run[kPartMaxBins];
  __shared__ uint32_t s_nfit[kPartMaxBins];    // kinds 2, 3: how many of them still fit the slab
  __shared__ uint32_t s_obase[kPartMaxBins];   // kind 2: overflow-area position of the rest
  __shared__ uint32_t s_total, s_nleft, s_stop;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  constexpr int VR = kPartVR, LV = kPartLV;
  constexpr int tw = TW;  // tuple words, compile time: the tuples of a batch live in registers
  const int tid = threadIdx.x;
  const uint32_t nbins = LEVEL == 1 ? a.p1 : (1u << a.p2_log2);
  const uint32_t gmask = (1u << a.g_log2) - 1;
  const uint32_t cap_stage = part_stage_tuples(nbins, gmask);
  int64_t* s_stage = s_dyn;
  uint16_t* s_binof = reinterpret_cast<uint16_t*>(s_dyn + static_cast<size_t>(cap_stage) * tw);
  uint16_t* s_left = s_binof + cap_stage;
  const uint64_t cap = LEVEL == 1 ? a.cap1 : a.cap2;
  const uint32_t tcap = LEVEL == 1 ? a.tcap1 : a.tcap2;
  const uint32_t cstride = LEVEL == 1 ? kPartCursorStride : 1;
  const size_t bin0 = LEVEL == 1 ? 0 : (static_cast<size_t>(blockIdx.y) << a.p2_log2);  // first region of the coarse slab
  uint32_t* fill = (LEVEL == 1 ? a.fill1 : a.fill2) + bin0 * cstride;
  uint32_t* tfill = (LEVEL == 1 ? a.tfill1 : a.tfill2) + bin0;
  int64_t* out = (LEVEL == 1 ? a.slab1 : a.slab2) + bin0 * cap * tw;
  int64_t* tout = (LEVEL == 1 ? a.tail1 : a.tail2) + bin0 * tcap * tw;
  for (int i = tid; i < kPartMaxBins; i += kPartBlock) {
    s_cnt[i] = 0;
  }
  if (tid == 0) {
    s_nleft = 0;
    s_stop = 0;
  }
  __syncthreads();

  // One batch: `live` new tuples in registers plus the tuples the previous batch carried over.  `last`: nothing
  // new, everything that is left goes to the tail slabs.
  auto do_batch = [&](const bool (&live)[VR], int64_t (&tup)[VR][TW], const bool last) {
    // 0. carried tuples: back from the staging area into registers (with the bin they were staged under)
    const uint32_t nleft = s_nleft;
    bool clive[LV];
    uint32_t cbin[LV];
    int64_t ctup[LV][TW];
#pragma unroll
    for (int l = 0; l < LV; ++l) {
      const uint32_t k = tid + l * kPartBlock;
      clive[l] = k < nleft;
      cbin[l] = 0;
      if (clive[l]) {
        const uint32_t si = s_left[k];
        cbin[l] = s_binof[si];
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          ctup[l][w] = s_stage[static_cast<size_t>(si) * tw + w];
        }
      }
    }
    // 1. histogram + rank inside the bin
    uint32_t bin[VR], rank[VR], crank[LV];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      bin[r] = 0;
      rank[r] = 0;
      if (live[r]) {
        const uint32_t f = part_home<K, TW>(a, tup[r]) >> a.slots_log2;
        bin[r] = LEVEL == 1 ? f >> a.p2_log2 : f & (nbins - 1);
        rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
      }
    }
#pragma unroll
    for (int l = 0; l < LV; ++l) {
      crank[l] = clive[l] ? atomicAdd(&s_cnt[cbin[l]], 1u) : 0u;
    }
    __syncthreads();
    // 2. per bin: how much leaves (whole lines), where to; run starts in the staging area (exclusive scan by wave 0)
    if (tid < kPartMaxBins) {
      const uint32_t n = tid < static_cast<int>(nbins) ? s_cnt[tid] : 0;
      const uint32_t nf = last ? n : n & ~gmask;
      uint32_t kind = 0, base = 0, nfit = nf, obase = 0;
      if (nf) {
        if (last) {  // < G tuples: the bin's tail slab, exact count (tcap = writers x (G - 1): cannot overflow)
          kind = 1;
          base = atomicAdd(tfill + tid, nf);
        } else {
          base = atomicAdd(fill + static_cast<size_t>(tid) * cstride, nf);
          // what does not fit the slab any more (heavy hitter) goes to the overflow area -- split at the slab's end, a
          // multiple of G like `base`: readers take the slab as [0, min(cursor, cap)), every position of it is written
          nfit = static_cast<uint64_t>(base) >= cap ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(nf), cap - base));
          if (nfit < nf) {
            obase = atomicAdd(a.fill_ovf, nf - nfit);
            kind = 2;
            if (static_cast<uint64_t>(obase) + (nf - nfit) > a.cap_ovf) {
              kind = 3;
              atomicExch(a.fallback, 1u);  // too skewed for slabs: hand the launch to the atomics kernel
            }
          }
        }
      }
      s_run[tid].y = nf;
      s_run[tid].z = base;
      s_run[tid].w = kind;
      s_nfit[tid] = nfit;
      s_obase[tid] = obase;
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
        s_run[c0 + tid].x = carry + incl - n;
        carry += __shfl(incl, kWave - 1, kWave);
      }
      if (tid == 0) {
        s_total = carry;
        s_nleft = 0;
      }
    }
    __syncthreads();
    // 3. stage the tuples ordered by bin
    auto stage_one = [&](uint32_t b, uint32_t rk, const int64_t* t) {
      const uint32_t si = s_lpos[b] + rk;
      s_binof[si] = static_cast<uint16_t>(b);
#pragma unroll
      for (int w = 0; w < TW; ++w) {
        s_stage[static_cast<size_t>(si) * tw + w] = t[w];
      }
    };
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      if (live[r]) {
        stage_one(bin[r], rank[r], tup[r]);
      }
    }
#pragma unroll
    for (int l = 0; l < LV; ++l) {
      if (clive[l]) {
        stage_one(cbin[l], crank[l], ctup[l]);
      }
    }
    if (tid < kPartMaxBins) {
      s_cnt[tid] = 0;  // (read for the last time in step 2)
    }
    __syncthreads();
    // 4. copy out whole lines: consecutive staging slots of a bin go to consecutive slab positions; what stays
    //    behind (< G per bin) is listed for the next batch (one list claim per wave)
    const uint32_t total = s_total;
    for (uint32_t i0 = 0; i0 < total; i0 += kPartBlock) {
      const uint32_t i = i0 + tid;
      const bool in = i < total;
      uint4 run = make_uint4(0, 0, 0, 0);
      uint32_t b = 0, r = 0;
      if (in) {
        b = s_binof[i];
        run = s_run[b];
        r = i - run.x;
      }
      const bool stays = in && r >= run.y;
      const uint64_t stay_mask = __ballot(stays);
      if (stay_mask) {
        const int lane = tid & (kWave - 1);
        uint32_t lbase = 0;
        if (lane == __ffsll(static_cast<long long>(stay_mask)) - 1) {
          lbase = atomicAdd(&s_nleft, static_cast<uint32_t>(__popcll(stay_mask)));
        }
        lbase = __shfl(lbase, __ffsll(static_cast<long long>(stay_mask)) - 1, kWave);
        if (stays) {
          s_left[lbase + __popcll(stay_mask & ((1ull << lane) - 1))] = static_cast<uint16_t>(i);
        }
      }
      if (!in || stays) {
        continue;
      }
      const uint32_t kind = run.w;
      uint32_t nfit = run.y;
      if (kind >= 2) {
        nfit = s_nfit[b];
        if (r >= nfit && kind == 3) {
          continue;
        }
      }
      const size_t pos = static_cast<size_t>(run.z) + r;
      int64_t* q = r >= nfit   ? a.ovf + (static_cast<size_t>(s_obase[b]) + (r - nfit)) * tw
                   : kind == 1 ? tout + (static_cast<size_t>(b) * tcap + pos) * tw
                               : out + (static_cast<size_t>(b) * cap + pos) * tw;
      if (TW == 2) {
        *reinterpret_cast<bf_i64x2*>(q) = *reinterpret_cast<const bf_i64x2*>(s_stage + static_cast<size_t>(i) * 2);
      } else {
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          q[w] = s_stage[static_cast<size_t>(i) * tw + w];
        }
      }
    }
    __syncthreads();
  };

  if (LEVEL == 2 && __hip_atomic_load(a.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    return;
  }
  bool none[VR];
  int64_t zero[VR][TW];
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    none[r] = false;
#pragma unroll
    for (int w = 0; w < TW; ++w) {
      zero[r][w] = 0;
    }
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
          s_stop = __hip_atomic_load(a.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_stop) {
          return;
        }
        const int64_t row0 = (tile - frag_tile_begin) * kPartTile + tid;
        bool live[VR];
        int64_t tup[VR][TW];
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
        for (int w = 1; w < TW; ++w) {
          if (w < nk) {  // second key column
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
              tup[r][w] = live[r] ? decode_col_g(cols[a.key2_buf_idx], a.key2_width, a.key2_kind, row, true) : 0;
            }
          } else {
            const BaseFastTarget c = a.arg[w - nk > 0 ? 1 : 0];
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
              tup[r][w] = live[r] ? decode_col_g(cols[c.buf_idx], c.width, c.kind, row, true) : 0;
            }
          }
        }
        do_batch(live, tup, false);
      }
      frag_tile_begin += ntiles;
    }
  } else {
    // coarse slab c = its main part [0, min(fill, cap)) followed by its tail slab [0, tfill)
    const uint32_t c = blockIdx.y;
    const uint64_t n_main = min(static_cast<uint64_t>(a.fill1[static_cast<size_t>(c) * kPartCursorStride]), a.cap1);
    const uint64_t n = n_main + a.tfill1[c];
    const int64_t* in_main = a.slab1 + static_cast<size_t>(c) * a.cap1 * tw;
    const int64_t* in_tail = a.tail1 + static_cast<size_t>(c) * a.tcap1 * tw;
    for (uint64_t t0 = static_cast<uint64_t>(blockIdx.x) * kPartTile; t0 < n; t0 += static_cast<uint64_t>(gridDim.x) * kPartTile) {
      bool live[VR];
      int64_t tup[VR][TW];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint64_t i = t0 + static_cast<uint64_t>(r) * kPartBlock + tid;
        live[r] = i < n;
        const int64_t* q = i < n_main ? in_main + i * tw : in_tail + (i - n_main) * tw;
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          tup[r][w] = live[r] ? __builtin_nontemporal_load(q + w) : 0;
        }
      }
      do_batch(live, tup, false);
    }
  }
  do_batch(none, zero, true);  // what is still carried: into the tail slabs
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

// the aggregates of one tuple onto its group's row (`rowb`: in an LDS image or in the table itself)
HDK_DEV void part_apply_targets(const PartTarget* s_tg, int ntargets, int8_t* rowb, const int64_t* tup) {
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

// ---- pass 3: one block per region -------------------------------------------------------------------------
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
  const uint32_t first = f << a.slots_log2;                           // first table entry of the region
  const uint32_t slots = min(1u << a.slots_log2, a.entry_count - first);  // (the last region may be short)
  part_load_targets(a, p, s_tg);
  const uint32_t words = slots * rq;
  int64_t* region = a.kp.groupby_buf[0] + static_cast<size_t>(first) * rq;
  const uint64_t n_main = min(static_cast<uint64_t>(a.fill2[f]), a.cap2);
  const uint64_t n = n_main + a.tfill2[f];
  if (n == 0 || *a.fallback) {
    return;  // the region keeps its initialised (empty) image
  }
  for (uint32_t i = tid; i < words; i += kPartAggBlock) {
    lds_table[i] = region[i];  // the init kernel's image (or an earlier launch's groups), whatever the layout
  }
  __syncthreads();
  const int tw = a.tw;
  const int64_t* in_main = a.slab2 + static_cast<size_t>(f) * a.cap2 * tw;
  const int64_t* in_tail = a.tail2 + static_cast<size_t>(f) * a.tcap2 * tw;
  int64_t* spill = a.slab1;  // dead since pass 2 and large enough for every tuple of the launch
  // Two tuples per trip, the next pair's loads issued before the current pair is applied: the LDS
  // claim/aggregate chain (ds_* ops, lgkmcnt) of one pair hides the HBM latency (vmcnt) of the next.
  // Plain scalars on purpose -- no per-thread tuple arrays that could end up in scratch.
  int64_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
  auto fetch = [&](uint64_t i, int64_t& t0, int64_t& t1, int64_t& t2) {
    if (i < n) {
      const int64_t* q = i < n_main ? in_main + i * tw : in_tail + (i - n_main) * tw;
      t0 = __builtin_nontemporal_load(q);
      t1 = tw > 1 ? __builtin_nontemporal_load(q + 1) : 0;
      t2 = tw > 2 ? __builtin_nontemporal_load(q + 2) : 0;
    }
  };
  auto apply = [&](const int64_t (&tup)[kPartMaxTW]) {
    const K key[2] = {static_cast<K>(tup[0]), static_cast<K>(tup[1])};  // (word 1 is only read as a key when key_count == 2)
    const uint32_t local = part_home<K>(a, tup) - first;
    bool fresh;
    const int64_t e = find_or_claim_from<K, false>(shape, lds_table, slots, local, key, &fresh);
    if (e < 0) {  // the group lives past the end of this region: pass 4 places it with the whole-table probe
      const uint32_t o = atomicAdd(a.fill_spill, 1u);
      if (o < a.cap_spill) {
        for (int w = 0; w < tw; ++w) {
          spill[static_cast<size_t>(o) * tw + w] = tup[w];
        }
      }
      return;
    }
    part_apply_targets(s_tg, ntargets, reinterpret_cast<int8_t*>(lds_table + static_cast<size_t>(e) * rq), tup);
  };
  fetch(tid, a0, a1, a2);
  fetch(static_cast<uint64_t>(tid) + kPartAggBlock, b0, b1, b2);
  for (uint64_t i = tid; i < n; i += 2 * kPartAggBlock) {
    const int64_t ta[kPartMaxTW] = {a0, a1, a2};
    const int64_t tb[kPartMaxTW] = {b0, b1, b2};
    const bool has_b = i + kPartAggBlock < n;
    fetch(i + 2 * kPartAggBlock, a0, a1, a2);
    fetch(i + 3 * kPartAggBlock, b0, b1, b2);
    apply(ta);
    if (has_b) {
      apply(tb);
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < words; i += kPartAggBlock) {
    region[i] = lds_table[i];
  }
}

// ---- pass 4: overflow and spilled tuples, straight onto the table with the reference's probe sequence --------
template <typename K>
__global__ __launch_bounds__(kPartBlock) void hdk_part_overflow(PartArgs a) {
  __shared__ PartTarget s_tg[HDK_HIP_MAX_TARGETS];
  const hdk_hip_plan* __restrict__ p = a.plan;
  if (*a.fallback) {
    return;
  }
  part_load_targets(a, p, s_tg);
  __syncthreads();
  const uint64_t n_ovf = min(static_cast<uint64_t>(*a.fill_ovf), a.cap_ovf);
  const uint64_t n_spill = min(static_cast<uint64_t>(*a.fill_spill), a.cap_spill);
  const TableShape shape = table_shape(p);
  const uint32_t rq = shape.row_quads;
  const int tw = a.tw;
  int64_t* table = a.kp.groupby_buf[0];
  int32_t err = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kPartBlock + threadIdx.x; i < n_ovf + n_spill;
       i += static_cast<uint64_t>(gridDim.x) * kPartBlock) {
    const int64_t* q = i < n_ovf ? a.ovf + i * tw : a.slab1 + (i - n_ovf) * tw;
    int64_t tup[kPartMaxTW];
#pragma unroll
    for (int w = 0; w < kPartMaxTW; ++w) {
      tup[w] = w < tw ? q[w] : 0;
    }
    const K key[2] = {static_cast<K>(tup[0]), static_cast<K>(tup[1])};
    bool fresh;
    const int64_t e = find_or_claim_from<K, true>(shape, table, a.entry_count, part_home<K>(a, tup), key, &fresh);
    if (e < 0) {
      err = HDK_HIP_ERR_OUT_OF_SLOTS;  // the table is full (the reference's get_group_value returns NULL)
      continue;
    }
    part_apply_targets(s_tg, a.ntargets, reinterpret_cast<int8_t*>(table + static_cast<size_t>(e) * rq), tup);
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
