// scan_cluster.h -- clustering pre-pass for equi-join probes: permute the outer table's columns by join-key RANGE so
// that the probes of consecutive rows fall into a slice of the join table that stays in L2.
//
// Why: a probe into a table much larger than L2 costs one 128-byte line through the fabric whatever the entry size
// (profiles/r02_c3_pmc.json: 1.10 memory-side requests per probing row, 8 TB/s of line fetches at 5.0e10 probes/s);
// the chip answers ~5e10 such misses per second however many are in flight.  From an L2-resident table the same gather
// runs at 1.85e11/s (scripts/microbench/gather.hip).  The reference probes in row order (QE/IRCodegen.cpp:497-667) and
// has no such step; aggregates do not depend on row order, so the step is invisible in the result.
//
//   pass 1 (hdk_cluster_by_key)   rows whose key lies in [min_key, max_key] are scattered into nbins = 256 key-range
//                                 bins, the permuted columns written column-wise: per 2 048-row batch an LDS histogram,
//                                 one cursor claim per bin, LDS staging ordered by bin, copy-out.  A bin is eight
//                                 sub-slabs, one per XCD (scan_agg_partitioned.h explains why); rows that do not fit
//                                 their sub-slab (skewed keys) go to an overflow fragment that has room for every row.
//                                 Rows with a NULL or out-of-range key cannot match an INNER join and are dropped.
//   pass 2 (hdk_cluster_params)   builds the kernel parameters of the permuted data: every (bin, XCD) sub-slab and the
//                                 overflow area become a FRAGMENT (COL_BUFFERS / NUM_ROWS / FRAG_ROW_OFFSETS /
//                                 NUM_FRAGMENTS in scratch memory); inner-table columns keep their buffers.
//   then                          the plan's ordinary kernel runs over those fragments, unchanged.
#pragma once
#include "device_common.h"

namespace hdk {

constexpr int kClusterBlock = 512;
constexpr int kClusterVR = 4;
constexpr int kClusterTile = kClusterBlock * kClusterVR;
constexpr int kClusterMaxCols = 4;
constexpr int kClusterBins = 256;
constexpr int kClusterXcds = 8;
constexpr uint32_t kClusterCursorStride = 32;  // one cursor per 128-byte line

struct ClusterArgs {
  KernParams kp;                      // the launch's own parameters (the unpermuted data)
  int32_t ncols;                      // outer columns (all 8 bytes wide); [0] is the join key
  int32_t buf_idx[kClusterMaxCols];   // their COL_BUFFERS indices
  int32_t ncols_total;                // plan->num_cols: entries of a COL_BUFFERS row
  int32_t outer_slot[HDK_HIP_MAX_COLS];  // per plan column: index into out[] (outer column) or -1 (inner: keeps its buffer)
  int64_t key_min;
  uint64_t key_range;                 // max_key - min_key + 1
  uint64_t bin_mult;                  // bin = ((key - min) * bin_mult) >> 32
  uint64_t sub;                       // rows of a (bin, XCD) sub-slab (multiple of 16)
  uint64_t cap_ovf;                   // rows of the overflow fragment
  int64_t* out[kClusterMaxCols];      // permuted columns: [kClusterBins * kClusterXcds * sub | cap_ovf] rows each
  int32_t aos;                        // 1: out[0] holds ROWS of ncols words instead (tuples for hdk_join_agg_direct)
  uint32_t* fill;                     // [kClusterBins][kClusterXcds] x kClusterCursorStride
  uint32_t* fill_ovf;
  // pass 2 outputs
  const int8_t** col_ptrs;            // [nfrag][ncols_total]
  const int8_t* const** frag_ptrs;    // [nfrag] -> rows of col_ptrs
  int64_t* num_rows;                  // [nfrag * ntab]
  uint64_t* frag_offs;                // [nfrag * ntab]
  uint64_t* num_fragments;            // [1]
};

__global__ __launch_bounds__(kClusterBlock) void hdk_cluster_by_key(ClusterArgs a) {
  constexpr int VR = kClusterVR;
  __shared__ uint32_t s_cnt[kClusterBins];
  __shared__ uint4 s_run[kClusterBins];  // .x start in the staging area, .y slots that fit the sub-slab, .z slab position, .w overflow position
  __shared__ uint32_t s_total;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  const int nc = a.ncols;
  int64_t* s_stage = s_dyn;                                                     // [kClusterTile][nc]
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + static_cast<size_t>(kClusterTile) * nc);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kClusterXcds - 1);
  for (int i = tid; i < kClusterBins; i += kClusterBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kClusterTile - 1) / kClusterTile;
    const int8_t* const* cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      const int64_t tile_row0 = (tile - frag_tile_begin) * kClusterTile;
      int64_t v[VR][kClusterMaxCols];
      bool live[VR];
      if (tile_row0 + kClusterTile <= nrows) {
        // full tile: rows dealt in adjacent pairs, one 16-byte non-temporal load per lane, pair and column
#pragma unroll
        for (int c = 0; c < kClusterMaxCols; ++c) {
          if (c < nc) {
            const uint64_t b = reinterpret_cast<uintptr_t>(cols[a.buf_idx[c]]) + static_cast<uint64_t>(tile_row0) * 8;
            const uint32_t b_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b));
            const uint32_t b_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b >> 32));
            const __attribute__((address_space(1))) int8_t* base =
                reinterpret_cast<const __attribute__((address_space(1))) int8_t*>((static_cast<uint64_t>(b_hi) << 32) | b_lo);
#pragma unroll
            for (int u = 0; u < VR / 2; ++u) {
              const bf_i64x2 x = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(
                  base + static_cast<uint32_t>(u * kClusterBlock + tid) * 16u));
              v[2 * u][c] = x.x;
              v[2 * u + 1][c] = x.y;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          live[r] = true;
        }
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const int64_t row = tile_row0 + static_cast<int64_t>(r) * kClusterBlock + tid;
          live[r] = row < nrows;
#pragma unroll
          for (int c = 0; c < kClusterMaxCols; ++c) {
            v[r][c] = (c < nc && live[r]) ? gload<int64_t>(cols[a.buf_idx[c]], row, true) : 0;
          }
        }
      }
      // 1. bin + rank inside the bin.  A key outside [min, max] (a NULL among them) has no partner: the row is dropped
      uint32_t bin[VR], rank[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint64_t d = static_cast<uint64_t>(v[r][0]) - static_cast<uint64_t>(a.key_min);
        live[r] = live[r] && d < a.key_range;
        bin[r] = static_cast<uint32_t>((static_cast<uint64_t>(static_cast<uint32_t>(d)) * a.bin_mult) >> 32);
        rank[r] = 0;
        if (live[r]) {
          rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
        }
      }
      __syncthreads();
      // 2. one claim per bin on this XCD's sub-slab; what does not fit goes to the overflow fragment
      if (tid < kClusterBins) {
        const uint32_t n = s_cnt[tid];
        uint32_t base = 0, nfit = 0, obase = 0;
        if (n) {
          base = atomicAdd(a.fill + (static_cast<size_t>(tid) * kClusterXcds + xcd) * kClusterCursorStride, n);
          nfit = static_cast<uint64_t>(base) >= a.sub ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(n), a.sub - base));
          if (nfit < n) {
            obase = atomicAdd(a.fill_ovf, n - nfit);
          }
        }
        s_run[tid].y = nfit;
        s_run[tid].z = base;
        s_run[tid].w = obase;
      }
      if (tid < kWave) {  // exclusive scan of the bin counts: where each bin's run starts in the staging area
        uint32_t carry = 0;
        for (int c0 = 0; c0 < kClusterBins; c0 += kWave) {
          const uint32_t n = s_cnt[c0 + tid];
          uint32_t incl = n;
#pragma unroll
          for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t x = __shfl_up(incl, d, kWave);
            if (tid >= d) {
              incl += x;
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
      // 3. stage the rows ordered by bin
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        if (live[r]) {
          const uint32_t si = s_run[bin[r]].x + rank[r];
          s_binof[si] = static_cast<uint8_t>(bin[r]);
#pragma unroll
          for (int c = 0; c < kClusterMaxCols; ++c) {
            if (c < nc) {
              s_stage[static_cast<size_t>(si) * nc + c] = v[r][c];
            }
          }
        }
      }
      if (tid < kClusterBins) {
        s_cnt[tid] = 0;
      }
      __syncthreads();
      // 4. copy out, column by column: consecutive staging slots of a bin go to consecutive rows of its sub-slab
      const uint32_t total = s_total;
      for (uint32_t i = tid; i < total; i += kClusterBlock) {
        const uint32_t b = s_binof[i];
        const uint4 run = s_run[b];
        const uint32_t r = i - run.x;
        if (r >= run.y && static_cast<uint64_t>(run.w) + (r - run.y) >= a.cap_ovf) {
          // cannot happen while total_rows really bounds the rows of the launch (the overflow area holds that many);
          // a caller that understated it gets an error instead of a write past the scratch buffer
          record_error(a.kp.error_code, HDK_HIP_ERR_OUT_OF_SLOTS);
          continue;
        }
        const uint64_t dest = r < run.y ? (static_cast<uint64_t>(b) * kClusterXcds + xcd) * a.sub + run.z + r
                                        : static_cast<uint64_t>(kClusterBins) * kClusterXcds * a.sub + run.w + (r - run.y);
        if (a.aos) {
          if (nc == 2) {
            reinterpret_cast<bf_i64x2*>(a.out[0])[dest] = reinterpret_cast<const bf_i64x2*>(s_stage)[i];
          } else {
            for (int c = 0; c < nc; ++c) {
              a.out[0][dest * nc + c] = s_stage[static_cast<size_t>(i) * nc + c];
            }
          }
        } else {
          for (int c = 0; c < nc; ++c) {
            a.out[c][dest] = s_stage[static_cast<size_t>(i) * nc + c];
          }
        }
      }
      __syncthreads();
    }
    frag_tile_begin += ntiles;
  }
}

// the permuted data as kernel parameters: fragment f < bins * xcds = sub-slab f, the last fragment = the overflow area
__global__ __launch_bounds__(256) void hdk_cluster_params(ClusterArgs a) {
  const uint32_t ntab = *a.kp.num_tables;
  const uint32_t nsub = kClusterBins * kClusterXcds;
  const uint32_t nfr = nsub + 1;
  const int8_t* const* cols0 = a.kp.col_buffers[0];  // inner-table columns: the same (linearised) buffer in every fragment
  for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < nfr; f += gridDim.x * blockDim.x) {
    const uint64_t rows = f < nsub ? min(static_cast<uint64_t>(a.fill[static_cast<size_t>(f) * kClusterCursorStride]), a.sub)
                                   : min(static_cast<uint64_t>(*a.fill_ovf), a.cap_ovf);
    const uint64_t first = f < nsub ? static_cast<uint64_t>(f) * a.sub : static_cast<uint64_t>(nsub) * a.sub;
    for (int ci = 0; ci < a.ncols_total; ++ci) {
      const int slot = a.outer_slot[ci];
      a.col_ptrs[static_cast<size_t>(f) * a.ncols_total + ci] =
          slot >= 0 ? reinterpret_cast<const int8_t*>(a.out[slot] + first) : cols0[ci];
    }
    a.frag_ptrs[f] = a.col_ptrs + static_cast<size_t>(f) * a.ncols_total;
    a.num_rows[static_cast<size_t>(f) * ntab] = static_cast<int64_t>(rows);
    a.frag_offs[static_cast<size_t>(f) * ntab] = 0;
    for (uint32_t t = 1; t < ntab; ++t) {
      a.num_rows[static_cast<size_t>(f) * ntab + t] = a.kp.num_rows[t];  // (fragment 0's entry: the inner tables' row counts)
      a.frag_offs[static_cast<size_t>(f) * ntab + t] = 0;
    }
    if (f == 0) {
      *a.num_fragments = nfr;
    }
  }
}

}  // namespace hdk
