// scan_join_sliced.h -- probe-and-aggregate for BASELINE config 3's shape when the join table is far larger than the
// caches: the table is cut into key-range SLICES that fit the LDS of a CU, the outer rows are scattered by slice, and
// every slice is probed out of LDS.
//
// Why: probing a 160 MB table in row order costs one 128-byte memory line per probe whatever the entry size
// (profiles/r02_c3_pmc.json: 141 bytes of HBM traffic per row for 16 algorithmic ones, 20.3 ms per 1 B rows), and the
// chip moves ~7 TB/s of such lines however many are in flight.  Locality is the only lever, and column statistics make
// it cheap: key - min_key fits 32 bits, so does (per ChunkStats) the one other outer column x and the inner payload.
//   pass 0 (hdk_join_order_probe)   samples 256 tiles of the key column: when nearly all of them span less than 1/128
//                                   of the key range the input is already clustered (a fact table sorted by the
//                                   foreign key) and probing in row order already hits L2 -- the launch's mode word
//                                   is set to 1 and passes 1-2 return at once; hdk_join_agg_direct, armed by that
//                                   word, does the job in row order.
//   pass 1 (hdk_join_scatter_slices) rows -> tuples [x as int32 : key - min as uint32] (8 bytes; 16-byte tuples
//                                   [key - min | x] when x does not fit), scattered into one slab per slice: per batch
//                                   an LDS histogram, one cursor claim per slice and XCD, LDS staging ordered by slice,
//                                   copy-out (the machinery of scan_agg_partitioned.h, level 1).  Rows whose key is NULL
//                                   or outside [min, max] have no partner in an INNER join and are dropped.  Rows that
//                                   do not fit their sub-slab (skewed keys) go to a shared overflow area.
//   pass 2 (hdk_join_agg_sliced)    one block per slice (1 024 threads, one per CU when the slice is ~150 KB): the
//                                   slice's payloads -- int32, with two reserved values for "no partner" and NULL --
//                                   are loaded from the fused table into LDS, the slab's tuples streamed with 16-byte
//                                   loads and probed there: no memory line per probe, the pass reads 8 bytes per row.
//                                   Aggregates live in registers (jd_eval, the row function of scan_join_direct.h); the
//                                   overflow area is probed against the table in memory, by all blocks together.
// Anything the narrow forms cannot carry -- a payload or an x outside 32 bits although the statistics said otherwise,
// an overflow area that fills up -- sets the mode word to 1 as well: the armed row-order kernel then redoes the whole
// launch and overwrites every slab, so the answer never depends on the metadata being right.
// Reference behaviour replaced: the probe of the row function, hash_join_idx (QE/GroupByRuntime.cpp:298-366), in row
// order (QE/IRCodegen.cpp:497-667); aggregates do not depend on row order, so the permutation is invisible.
#pragma once
#include "plain_quals.h"
#include "scan_join_direct.h"

namespace hdk {

constexpr int kSliceBlock = 512;        // scatter pass
constexpr int kSliceAggBlock = 1024;    // probe pass
constexpr int kSliceMaxBins = 256;
constexpr int kSliceXcds = 8;
constexpr uint32_t kSliceCursorStride = 32;
constexpr uint32_t kSliceMaxEntries = 39936;  // 156 KB of int32 payloads: one 1 024-thread block per CU
constexpr int32_t kSliceNoMatch = INT32_MIN;      // reserved payload values
constexpr int32_t kSliceNull = INT32_MIN + 1;

struct SliceArgs {
  KernParams kp;
  int32_t key_buf_idx, x_buf_idx;     // outer columns (8-byte integers); x_buf_idx < 0: none
  int64_t key_min;
  uint64_t key_range;                 // max - min + 1 (< 2^32)
  int32_t key_nullable;
  int64_t key_null;
  uint32_t slice;                     // keys per slice
  uint32_t slice_magic, slice_shift;  // (key - min) / slice
  uint32_t nbins;                     // slices in use (<= 256)
  int32_t narrow;                     // 8-byte tuples [x as int32 : key - min]; else 16-byte [key - min | x]
  int32_t x_null32;                   // narrow: INT32_MIN stands for x's in-band NULL
  int32_t x_null_is_stale;            // narrow: the statistics announce no NULLs in x but its leaf is nullable -- a value equal
                                      // to x_null contradicts them (and fits 32 bits when x is a narrow column's value)
  int64_t x_null;
  int64_t pay_null;                   // the payload column's in-band NULL
  int32_t pay_nullable;
  uint64_t sub;                       // tuples of a (slice, XCD) sub-slab (multiple of 16)
  uint64_t cap_ovf;
  int64_t* tuples;                    // [nbins][kSliceXcds][sub] tuples, then the overflow area [cap_ovf]
  uint32_t* fill;                     // [nbins][kSliceXcds] x kSliceCursorStride
  uint32_t* fill_ovf;
  uint32_t* mode;                     // 0: slices; != 0: the armed row-order kernel takes over
  uint32_t* probe;                    // [2]: narrow tiles, finished blocks (hdk_join_order_probe)
  JoinDirectArgs jd;                  // targets, word layout, slabs (scan_join_direct.h)
  uint32_t num_slabs;                 // slabs hdk_finalize folds (the launch shape's grid)
  // FAST form of pass 2: ONE target SUM(x + payload) (BASELINE config 3); the NULL rules of its two leaves
  int32_t fast;
  int32_t fast_x_nullable, fast_p_nullable;
  int64_t fast_x_null, fast_p_null;
  // filters `outer column cmp literal` (plain_quals.h), applied by pass 1 before the scatter: a row that fails is never a
  // tuple (hdk_join_scatter_slices<NARROW, true>; the general sliced plans of scan_join_sliced2.h)
  int32_t nquals;
  int32_t pad_q_;
  ProjFastQual q[kMaxPlainQuals];
  // a SECOND outer column y riding in the tuple's low word above the key offset (scan_join_sliced2.h: a group key or a
  // measure of the outer table): [x : code(y) << y_shift | key - min], code = y - y_min + 1, 0 = NULL; the key range fits
  // y_shift bits and the column statistics put every code below 2^(32 - y_shift).  y_shift == 0: no such column
  int32_t y_buf_idx;
  int32_t y_nullable;
  uint32_t y_shift;
  int32_t y_width;                    // 8 or 4 bytes
  int32_t pad_y_;
  uint32_t y_codes;                   // codes in use: 1 .. y_codes - 1 (a value outside them contradicts the statistics)
  int64_t y_min, y_null;
};

// the key offset of a narrow tuple's low word (all of it when no second outer column rides along)
HDK_DEV uint32_t slice_key_of(const SliceArgs& a, uint32_t lo) {
  return a.y_shift ? lo & ((1u << a.y_shift) - 1u) : lo;
}

// ---- pass 0: is the key column already clustered? ------------------------------------------------------------------
__global__ __launch_bounds__(256) void hdk_join_order_probe(SliceArgs a) {
  __shared__ int64_t s_lo[4], s_hi[4];
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kRows = 256 * 8;
  // tile `blockIdx.x` of gridDim.x, spread evenly over the rows of the launch
  int64_t total_tiles = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    total_tiles += a.kp.num_rows[f * ntab] / kRows;
  }
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  bool sampled = false;
  if (total_tiles > 0) {
    int64_t t = static_cast<int64_t>((static_cast<unsigned __int128>(blockIdx.x) * static_cast<uint64_t>(total_tiles)) / gridDim.x);
    for (uint64_t f = 0; f < nfrag; ++f) {
      const int64_t nt = a.kp.num_rows[f * ntab] / kRows;
      if (t < nt) {
        const int8_t* kcol = a.kp.col_buffers[f][a.key_buf_idx];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bf_i64x2 kk = gload<bf_i64x2>(kcol, (t * kRows >> 1) + u * 256 + threadIdx.x, false);
          const int64_t k2[2] = {kk.x, kk.y};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint64_t d = static_cast<uint64_t>(k2[j]) - static_cast<uint64_t>(a.key_min);
            if (d < a.key_range) {  // (keys without a partner do not probe)
              lo = min(lo, k2[j]);
              hi = max(hi, k2[j]);
            }
          }
        }
        sampled = true;
        break;
      }
      t -= nt;
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    lo = min(lo, __shfl_xor(lo, d, 64));
    hi = max(hi, __shfl_xor(hi, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      lo = min(lo, s_lo[w]);
      hi = max(hi, s_hi[w]);
    }
    const bool narrow = sampled && (hi < lo || static_cast<uint64_t>(hi - lo) < a.key_range / 128);
    if (narrow) {
      atomicAdd(&a.probe[0], 1u);
    }
    __threadfence();
    if (atomicAdd(&a.probe[1], 1u) == gridDim.x - 1) {  // the last block decides
      const uint32_t n = __hip_atomic_load(&a.probe[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (total_tiles >= static_cast<int64_t>(gridDim.x) && n * 10 >= gridDim.x * 9) {
        atomicMax(a.mode, 1u);
      }
    }
  }
}

// ---- pass 1: rows -> tuples, scattered by slice -----------------------------------------------------------------------
// dynamic LDS: [kTile][TW] staging | uint8 slice of every staging slot [kTile]
// YW: byte width of the second outer column (8 or 4; 0: none)
template <bool NARROW, bool Q = false, int YW = 0>
__global__ __launch_bounds__(kSliceBlock) void hdk_join_scatter_slices(SliceArgs a) {
  static_assert(NARROW || YW == 0, "the second outer column rides in 8-byte tuples");
  constexpr bool Y = YW != 0;
  typedef int __attribute__((ext_vector_type(2))) i32x2;
  constexpr int VR = NARROW ? 8 : 4;
  constexpr int TW = NARROW ? 1 : 2;
  constexpr int kTile = kSliceBlock * VR;
  __shared__ uint32_t s_cnt[kSliceMaxBins];
  __shared__ uint4 s_run[kSliceMaxBins];  // .x start in the staging area, .y tuples that fit the sub-slab, .z slab position, .w overflow position
  __shared__ uint32_t s_total, s_stop;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + static_cast<size_t>(kTile) * TW);
  const int tid = threadIdx.x;
  // (block-uniform, through LDS: another block of this kernel may raise the word between the reads of two waves)
  if (tid == 0) {
    s_stop = __hip_atomic_load(a.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_stop) {
    return;  // clustered input: probed in row order
  }
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kSliceXcds - 1);
  for (int i = tid; i < kSliceMaxBins; i += kSliceBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const Watch watch = watch_begin(a.kp);
  bool stale = false;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTile - 1) / kTile;
    const int8_t* const* cols = a.kp.col_buffers[f];
    const int8_t* kcol = cols[a.key_buf_idx];
    const int8_t* xcol = a.x_buf_idx >= 0 ? cols[a.x_buf_idx] : nullptr;
    const int8_t* ycol = Y ? cols[a.y_buf_idx] : nullptr;
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      if (tid == 0) {  // block-uniform exit (the batch below has barriers)
        uint32_t stop = __hip_atomic_load(a.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (watch.flags) {
          if (const int32_t w_ = watch_poll(watch)) {
            record_error(a.kp.error_code, w_);
            atomicMax(a.mode, 2u);  // interrupted: nobody redoes the launch
            stop = 2;
          }
        }
        s_stop = stop;
      }
      __syncthreads();
      if (s_stop) {
        return;
      }
      const int64_t tile_row0 = (tile - frag_tile_begin) * kTile;
      int64_t k[VR], x[VR];
      uint32_t ycode[VR];  // (Y) the second outer column, already as its code shifted above the key offset
      bool live[VR];
      auto y_code = [&](int64_t y) -> uint32_t {
        if (a.y_nullable && y == a.y_null) {
          return 0u;
        }
        const uint64_t c = static_cast<uint64_t>(y) - static_cast<uint64_t>(a.y_min) + 1u;
        stale |= c - 1u >= static_cast<uint64_t>(a.y_codes - 1u);  // (a row dropped later may raise it too: only slower, never wrong)
        return static_cast<uint32_t>(c) << a.y_shift;
      };
      if (tile_row0 + kTile <= nrows) {
        // full tile: rows dealt in adjacent pairs, one 16-byte non-temporal load per lane, pair and column
#pragma unroll
        for (int u = 0; u < VR / 2; ++u) {
          const int64_t p = (tile_row0 >> 1) + static_cast<int64_t>(u) * kSliceBlock + tid;
          const bf_i64x2 kk = gload<bf_i64x2>(kcol, p, true);
          k[2 * u] = kk.x;
          k[2 * u + 1] = kk.y;
          bf_i64x2 xx;
          xx.x = 0;
          xx.y = 0;
          if (xcol) {
            xx = gload<bf_i64x2>(xcol, p, true);
          }
          x[2 * u] = xx.x;
          x[2 * u + 1] = xx.y;
          if (YW == 8) {
            const bf_i64x2 yy = gload<bf_i64x2>(ycol, p, true);
            ycode[2 * u] = y_code(yy.x);
            ycode[2 * u + 1] = y_code(yy.y);
          } else if (YW == 4) {
            const i32x2 yy = gload<i32x2>(ycol, p, true);
            ycode[2 * u] = y_code(yy.x);
            ycode[2 * u + 1] = y_code(yy.y);
          }
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          live[r] = true;
        }
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const int64_t row = tile_row0 + static_cast<int64_t>(r) * kSliceBlock + tid;
          live[r] = row < nrows;
          k[r] = live[r] ? gload<int64_t>(kcol, row, true) : 0;
          x[r] = (live[r] && xcol) ? gload<int64_t>(xcol, row, true) : 0;
          if (Y) {
            ycode[r] = !live[r] ? 0u : y_code(YW == 4 ? static_cast<int64_t>(gload<int32_t>(ycol, row, true)) : gload<int64_t>(ycol, row, true));
          }
        }
      }
      if (Q) {  // the plan's filters on outer columns: three-valued, NULL fails (plain_quals_pass)
        int64_t rown[VR];
        const bool full = tile_row0 + kTile <= nrows;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          rown[r] = full ? tile_row0 + (static_cast<int64_t>(r >> 1) * kSliceBlock + tid) * 2 + (r & 1)
                         : tile_row0 + static_cast<int64_t>(r) * kSliceBlock + tid;
        }
        plain_quals_pass<VR>(a.q, a.nquals, cols, rown, live, true);
      }
      // 1. slice + rank inside the slice.  A key outside [min, max] (a NULL among them) has no partner: the row is dropped
      uint32_t bin[VR], rank[VR];
      int64_t tup[VR][TW];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint64_t d = static_cast<uint64_t>(k[r]) - static_cast<uint64_t>(a.key_min);
        live[r] = live[r] && d < a.key_range && !(a.key_nullable && k[r] == a.key_null);
        const uint32_t d32 = static_cast<uint32_t>(d);
        const uint32_t t = __umulhi(a.slice_magic, d32);
        bin[r] = live[r] ? ((((d32 - t) >> 1) + t) >> a.slice_shift) : 0u;
        if (NARROW) {
          const bool is_null = a.x_null32 && x[r] == a.x_null;
          const int32_t x32 = is_null ? INT32_MIN : static_cast<int32_t>(x[r]);
          stale |= live[r] && ((!is_null && (static_cast<int64_t>(x32) != x[r] || (a.x_null32 && x32 == INT32_MIN))) ||
                               (a.x_null_is_stale && x[r] == a.x_null));
          tup[r][0] = static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(x32)) << 32) | (Y ? d32 | ycode[r] : d32));
        } else {
          tup[r][0] = static_cast<int64_t>(d);
          tup[r][TW - 1] = x[r];
        }
        rank[r] = 0;
        if (live[r]) {
          rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
        }
      }
      __syncthreads();
      // 2. one claim per slice on this XCD's sub-slab; what does not fit goes to the overflow area
      if (tid < kSliceMaxBins) {
        const uint32_t n = s_cnt[tid];
        uint32_t base = 0, nfit = 0, obase = 0;
        if (n) {
          base = atomicAdd(a.fill + (static_cast<size_t>(tid) * kSliceXcds + xcd) * kSliceCursorStride, n);
          nfit = static_cast<uint64_t>(base) >= a.sub ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(n), a.sub - base));
          if (nfit < n) {
            obase = atomicAdd(a.fill_ovf, n - nfit);
            if (static_cast<uint64_t>(obase) + (n - nfit) > a.cap_ovf) {
              atomicMax(a.mode, 1u);  // too skewed for slabs: the row-order kernel takes over
            }
          }
        }
        s_run[tid].y = nfit;
        s_run[tid].z = base;
        s_run[tid].w = obase;
      }
      if (tid < kWave) {  // exclusive scan of the counts: where each slice's run starts in the staging area
        uint32_t carry = 0;
        for (int c0 = 0; c0 < kSliceMaxBins; c0 += kWave) {
          const uint32_t n = s_cnt[c0 + tid];
          uint32_t incl = n;
#pragma unroll
          for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d, kWave);
            if (tid >= d) {
              incl += v;
            }
          }
          s_run[c0 + tid].x = carry + incl - n;
          carry += __shfl(incl, kWave - 1, kWave);
        }
        if (tid == 0) {
          s_total = carry;
        }
      }
      __syncthreads();
      // 3. stage the tuples ordered by slice
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        if (live[r]) {
          const uint32_t si = s_run[bin[r]].x + rank[r];
          s_binof[si] = static_cast<uint8_t>(bin[r]);
          if (TW == 2) {
            bf_i64x2 v;
            v.x = tup[r][0];
            v.y = tup[r][TW - 1];
            reinterpret_cast<bf_i64x2*>(s_stage)[si] = v;
          } else {
            s_stage[si] = tup[r][0];
          }
        }
      }
      if (tid < kSliceMaxBins) {
        s_cnt[tid] = 0;
      }
      __syncthreads();
      // 4. copy out: consecutive staging slots of a slice go to consecutive positions of its sub-slab
      const uint32_t total = s_total;
      for (uint32_t i = tid; i < total; i += kSliceBlock) {
        const uint32_t b = s_binof[i];
        const uint4 run = s_run[b];
        const uint32_t r = i - run.x;
        uint64_t dest;
        if (r < run.y) {
          dest = (static_cast<uint64_t>(b) * kSliceXcds + xcd) * a.sub + run.z + r;
        } else {
          const uint64_t o = static_cast<uint64_t>(run.w) + (r - run.y);
          if (o >= a.cap_ovf) {
            continue;  // (mode is 1: the launch is redone in row order)
          }
          dest = static_cast<uint64_t>(a.nbins) * kSliceXcds * a.sub + o;
        }
        if (TW == 2) {
          reinterpret_cast<bf_i64x2*>(a.tuples)[dest] = reinterpret_cast<const bf_i64x2*>(s_stage)[i];
        } else {
          a.tuples[dest] = s_stage[i];
        }
      }
      __syncthreads();
    }
    frag_tile_begin += ntiles;
  }
  if (NARROW && __any(stale) && (threadIdx.x & (kWave - 1)) == 0) {
    atomicMax(a.mode, 1u);  // a value the statistics did not announce
  }
}

// ---- pass 2: one block per slice, the slice's payloads in LDS ---------------------------------------------------------
// dynamic LDS: [wpe x rep] aggregate words | int32 payload[slice]
// FAST: the plan has ONE target, SUM(x + payload), fixed at compile time -- the general row function (jd_eval: up to
// four targets, leaves, operator, width and combine op all looked up per row) costs 124 vector + 80 scalar
// instructions per tuple, and with one 1 024-thread block per CU (16 waves) the pass is bound by instruction issue
// (profiles/r03_c3_pmc.json), not by its 8 bytes per row
template <bool NARROW, bool FAST>
__global__ __launch_bounds__(kSliceAggBlock) void hdk_join_agg_sliced(SliceArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  const JoinDirectArgs& j = a.jd;
  const int tid = threadIdx.x;
  const int wpe = j.wpe;
  const uint32_t rep = j.rep;
  const uint32_t ew = static_cast<uint32_t>(wpe);
  const uint32_t my_rep = tid & (rep - 1);
  int32_t* s_pay = reinterpret_cast<int32_t*>(lds + ew * rep);
  // (block-uniform, through LDS: another block of this kernel may raise the word between the reads of two waves)
  __shared__ uint32_t s_off;
  if (tid == 0) {
    s_off = __hip_atomic_load(a.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const bool off = s_off != 0;
  // slabs beyond this grid (hdk_finalize folds the launch shape's count) hold identities
  for (uint32_t sidx = blockIdx.x + (off ? 0u : gridDim.x); sidx < a.num_slabs; sidx += gridDim.x) {
    for (uint32_t i = tid; i < ew; i += kSliceAggBlock) {
      j.slabs[static_cast<size_t>(sidx) * ew + i] = (j.nword_mask >> i) & 1u ? 0 : word_identity(j.wop[i]);
    }
  }
  if (off) {
    return;  // the armed row-order kernel writes the real slabs
  }
  for (uint32_t i = tid; i < ew * rep; i += kSliceAggBlock) {
    lds[i] = word_identity(j.wop[(i / rep) % wpe]);
  }
  const int64_t* __restrict__ table = j.kp.join_hash_tables;  // fused: [row id | payload] per key
  const uint32_t bin = blockIdx.x % a.nbins, member = blockIdx.x / a.nbins, members = gridDim.x / a.nbins;
  const uint32_t first = bin * a.slice;
  const uint32_t nkeys = static_cast<uint32_t>(min(static_cast<uint64_t>(a.slice), a.key_range - first));
  bool bad = false;
  for (uint32_t i = tid; i < nkeys; i += kSliceAggBlock) {
    const bf_i64x2 e = *reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(
        reinterpret_cast<uintptr_t>(table) + static_cast<uint64_t>(first + i) * 16);
    int32_t p32 = kSliceNoMatch;
    if (e.x >= 0) {
      if (a.pay_nullable && e.y == a.pay_null) {
        p32 = kSliceNull;
      } else {
        p32 = static_cast<int32_t>(e.y);
        bad |= static_cast<int64_t>(p32) != e.y || p32 == kSliceNoMatch || p32 == kSliceNull;
      }
    }
    s_pay[i] = p32;
  }
  if (__any(bad) && (tid & (kWave - 1)) == 0) {
    atomicMax(a.mode, 1u);  // a payload outside what the statistics announced: redone in row order
  }
  __syncthreads();
  JdAcc acc;
  acc.rows = 0;
#pragma unroll
  for (int t = 0; t < kJdMaxTargets; ++t) {
    acc.val[t] = (t < j.ntargets && j.t[t].vword >= 0) ? word_identity(j.t[t].wop) : 0;
    acc.nulls[t] = 0;
  }
  int32_t err = 0;
  // FAST: SUM(x + payload) with agg_sum[_skip_val]'s rules, everything about the target in scalar registers
  const bool f_xn = a.fast_x_nullable != 0, f_pn = a.fast_p_nullable != 0;
  const int64_t f_xnull = a.fast_x_null, f_pnull = a.fast_p_null;
  const JdTarget& ft = j.t[0];
  const int64_t f_step_null = ft.step_null, f_arg_null = ft.arg_null, f_slot_null = ft.slot_null;
  const bool f_skip = ft.skip_null != 0, f_argn = ft.arg_nullable != 0;
  const int f_width = ft.check_width;
  uint64_t f_rows = 0, f_nulls = 0;
  int64_t f_sum = 0;
  auto joined = [&](int64_t x, int64_t pay) {
    if (!FAST) {
      jd_eval(j, x, pay, acc, err);
      return;
    }
    // (plain scalars: as members of `acc` the two counters were merged into one load-add-store on a scratch slot
    // selected per row -- with a full vmcnt wait in front of the prefetched tuple loads)
    f_rows += 1;
    int64_t v;
    if ((f_xn && x == f_xnull) || (f_pn && pay == f_pnull)) {
      v = f_step_null;
    } else {
      if (checked_arith(HDK_OP_ADD, x, pay, f_width, &v)) {
        err = HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW;
      }
    }
    const bool is_null = f_skip && ((f_argn && v == f_arg_null) || v == f_slot_null);
    f_nulls += is_null ? 1u : 0u;
    f_sum += is_null ? 0 : v;
  };
  auto row = [&](uint32_t krel, int64_t x) {
    const uint32_t local = krel - first;
    const int32_t p32 = local < nkeys ? s_pay[local] : kSliceNoMatch;
    if (p32 != kSliceNoMatch) {
      joined(x, p32 == kSliceNull ? a.pay_null : static_cast<int64_t>(p32));
    }
  };
  auto narrow_row = [&](int64_t w) {
    const int32_t x32 = static_cast<int32_t>(static_cast<uint64_t>(w) >> 32);
    row(static_cast<uint32_t>(w), (a.x_null32 && x32 == INT32_MIN) ? a.x_null : static_cast<int64_t>(x32));
  };
  // the slice's eight sub-slabs: every lane takes 16-byte words w, w + stride, ... (members of a slice interleave), two
  // loads in flight ahead of the word being applied.  One applied word per trip keeps the loop body small -- with
  // eight tuples per trip the unrolled row function outgrew the instruction cache (16 K lines of ISA, 4.9 ms per 1 B rows)
#pragma unroll 1
  for (int xq = 0; xq < kSliceXcds; ++xq) {
    const size_t sidx = static_cast<size_t>(bin) * kSliceXcds + xq;
    const uint64_t n = min(static_cast<uint64_t>(a.fill[sidx * kSliceCursorStride]), a.sub);
    const uint64_t nwords = n * (NARROW ? 1 : 2);       // int64 words of the sub-slab
    const uint64_t npairs = (nwords + 1) / 2;           // 16-byte words (sub is a multiple of 16: the last one is inside the slab)
    const int8_t* in = reinterpret_cast<const int8_t*>(a.tuples + sidx * a.sub * (NARROW ? 1 : 2));
    const uint64_t stride = static_cast<uint64_t>(members) * kSliceAggBlock;
    bf_i64x2 n0, n1;
    n0.x = n0.y = n1.x = n1.y = 0;
    uint64_t pw = static_cast<uint64_t>(member) * kSliceAggBlock + tid;
    if (pw < npairs) {
      n0 = gload<bf_i64x2>(in, static_cast<int64_t>(pw), true);
    }
    if (pw + stride < npairs) {
      n1 = gload<bf_i64x2>(in, static_cast<int64_t>(pw + stride), true);
    }
#pragma unroll 1
    for (; pw < npairs; pw += stride) {
      const bf_i64x2 cur = n0;
      n0 = n1;
      if (pw + 2 * stride < npairs) {
        n1 = gload<bf_i64x2>(in, static_cast<int64_t>(pw + 2 * stride), true);
      }
      if (NARROW) {
        narrow_row(cur.x);
        if (2 * pw + 1 < nwords) {
          narrow_row(cur.y);
        }
      } else {
        row(static_cast<uint32_t>(cur.x), cur.y);
      }
    }
  }
  // the overflow area (tuples of any slice): probed against the table in memory, all blocks together
  {
    const uint64_t n = min(static_cast<uint64_t>(*a.fill_ovf), a.cap_ovf);
    const int64_t* in = a.tuples + static_cast<size_t>(a.nbins) * kSliceXcds * a.sub * (NARROW ? 1 : 2);
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kSliceAggBlock + tid; i < n; i += static_cast<uint64_t>(gridDim.x) * kSliceAggBlock) {
      uint32_t krel;
      int64_t x;
      if (NARROW) {
        const int64_t w = in[i];
        const int32_t x32 = static_cast<int32_t>(static_cast<uint64_t>(w) >> 32);
        krel = static_cast<uint32_t>(w);
        x = (a.x_null32 && x32 == INT32_MIN) ? a.x_null : static_cast<int64_t>(x32);
      } else {
        krel = static_cast<uint32_t>(in[2 * i]);
        x = in[2 * i + 1];
      }
      const bf_i64x2 e = *reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(
          reinterpret_cast<uintptr_t>(table) + static_cast<uint64_t>(krel) * 16);
      if (e.x >= 0) {
        joined(x, e.y);
      }
    }
  }
  if (err) {
    record_error(j.kp.error_code, err);
  }
  if (FAST) {
    acc.rows = f_rows;
    acc.val[0] = f_sum;
    acc.nulls[0] = f_nulls;
  }
  jd_flush(j, acc, lds, my_rep, tid, kSliceAggBlock);
}

}  // namespace hdk
