// scan_agg_cols.h -- non-grouped aggregates over SEVERAL plain columns, streamed column by column.
//
// Shape: NonGroupedAggregate, no filter, no join; every target is COUNT(*) or COUNT / SUM / MIN / MAX / AVG of a plain
// outer column (integer of 1 / 2 / 4 / 8 bytes, or double) -- the reference's NonGroupedAgg benchmark queries
// (omniscidb/Benchmarks/synthetic_benchmark/queries/NonGroupedAgg/NGA01-05.sql: six aggregates over six int columns), which
// the one-argument streaming kernel (scan_agg_fast.h) does not take and the batched interpreter ran at a quarter of the
// HBM roofline.  Reference being replaced: agg_sum / agg_count / agg_min / agg_max[_skip_val] called once per row and target
// (QE/RuntimeFunctions.cpp:456-476,612-660) and the fold of the per-thread partials (Executor::reduceResults,
// QE/Execute.cpp:889-1110).
//
// Without a filter or a key the columns are INDEPENDENT: a row of one column never has to meet the same row of another.  So
// there are no "rows" in this kernel at all.  Every block walks the columns one after another; a column of a fragment is a
// flat array of 16-byte chunks, dealt to the blocks in tiles of BLOCK x U chunks (U loads of 16 bytes per lane in flight,
// fully coalesced, non-temporal), and a lane keeps ONE accumulator set -- sum, min, max, non-NULL count -- in registers
// whatever the number of targets.  At the end of a column: wave shuffle tree, one LDS step across the waves, and the block
// writes the words of the targets that read this column into its slab (agg_common.h: the slab format of the LDS-privatised
// kernels, one entry), which hdk_finalize folds into the output with the exact agg_*[_skip_val] semantics.
// Algorithmic bytes: the sum of the distinct columns' widths per row; nothing else is read or written.
#pragma once
#include "watch.h"
#include "agg_common.h"
#include "scan_agg_fast.h"

namespace hdk {

constexpr int kColsBlock = 256;
constexpr int kColsU = 8;          // 16-byte loads in flight per lane
constexpr int kColsMaxCols = 8;    // distinct argument columns
constexpr uint32_t kColsBlocksPerCu = 2;

enum ColsWordKind : int32_t { CW_ROWS = 0, CW_SUM = 1, CW_MIN = 2, CW_MAX = 3, CW_NN = 4, CW_UNUSED = 5 };

struct ColsArgs {
  KernParams kp;
  int64_t* slabs;
  int32_t ncols;
  int32_t wpe;
  struct Col {
    int32_t buf_idx;
    int32_t width;      // 1 / 2 / 4 / 8
    int32_t fp;         // double
    int32_t nullable;   // skip the in-band NULL
    int64_t null_val;   // widened (integers) / double bits
  } col[kColsMaxCols];
  int32_t wcol[kMaxWordsPerEntry];   // column of word w (-1: the row count)
  int32_t wkind[kMaxWordsPerEntry];  // ColsWordKind
};

// a lane's accumulators for one column
template <bool FP>
struct ColsAcc {
  using V = std::conditional_t<FP, double, int64_t>;
  V sum, mn, mx;
  uint64_t nn;
};

template <bool FP>
HDK_DEV void cols_acc_init(ColsAcc<FP>& a) {
  a.nn = 0;
  if constexpr (FP) {
    a.sum = 0.0;
    a.mn = bits_to_double(word_identity(WOP_MIN_F64));
    a.mx = bits_to_double(word_identity(WOP_MAX_F64));
  } else {
    a.sum = 0;
    a.mn = INT64_MAX;
    a.mx = INT64_MIN;
  }
}

// one element.  Integers of <= 4 bytes keep min / max in 32 bits (E = int32_t): half the vector instructions of the 64-bit
// compare-and-select; the sum is 64-bit always (u64 wrap, like agg_sum on an int64 slot).
template <int W, bool FP, bool NULLABLE, typename E>
HDK_DEV void cols_elem(E v, E null_e, uint64_t& sum_bits, E& mn, E& mx, uint32_t& nn) {
  if constexpr (FP) {
    const bool live = !NULLABLE || !(v == null_e);
    double s = bits_to_double(static_cast<int64_t>(sum_bits));
    s = live ? s + v : s;
    sum_bits = static_cast<uint64_t>(double_to_bits(s));
    mn = (live && v < mn) ? v : mn;
    mx = (live && mx < v) ? v : mx;
    nn += live ? 1u : 0u;
  } else {
    const bool live = !NULLABLE || v != null_e;
    sum_bits += live ? static_cast<uint64_t>(static_cast<int64_t>(v)) : 0ull;
    mn = (live && v < mn) ? v : mn;
    mx = (live && v > mx) ? v : mx;
    nn += live ? 1u : 0u;
  }
}

// the pass of one column over every fragment of the launch; returns the lane's partials through `out`
template <int W, bool FP, bool NULLABLE>
HDK_DEV void cols_pass(const ColsArgs& a, const ColsArgs::Col& c, ColsAcc<FP>& out, const Watch& watch, int32_t& err) {
  using E = std::conditional_t<FP, double, std::conditional_t<(W <= 4), int32_t, int64_t>>;
  constexpr int R = 16 / W;          // elements per 16-byte chunk
  constexpr int REGS = 4;
  const int tid = threadIdx.x;
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  E null_e;
  if constexpr (FP) {
    null_e = bits_to_double(c.null_val);
  } else {
    null_e = static_cast<E>(c.null_val);
  }
  uint64_t sum_bits = FP ? static_cast<uint64_t>(double_to_bits(0.0)) : 0ull;
  E mn, mx;
  if constexpr (FP) {
    mn = bits_to_double(word_identity(WOP_MIN_F64));
    mx = bits_to_double(word_identity(WOP_MAX_F64));
  } else {
    mn = W <= 4 ? static_cast<E>(INT32_MAX) : static_cast<E>(INT64_MAX);
    mx = W <= 4 ? static_cast<E>(INT32_MIN) : static_cast<E>(INT64_MIN);
  }
  uint64_t nn64 = 0;
  constexpr int64_t kTileChunks = static_cast<int64_t>(kColsBlock) * kColsU;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t nchunks = nrows / R;                  // whole 16-byte chunks of the column
    const int64_t ntiles = (nchunks + kTileChunks - 1) / kTileChunks + 1;  // (+ 1: the tile of the last < R elements)
    const gcol_t col = (gcol_t)a.kp.col_buffers[f][c.buf_idx];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t t = tile - frag_tile_begin;
      uint32_t nn = 0;
      if (t == ntiles - 1) {
        // the last elements of the fragment that do not fill a chunk: a lane each
        const int64_t r = nchunks * R + tid;
        if (r < nrows) {
          E v;
          if constexpr (FP) {
            v = bits_to_double(load_elem<8>(col, r));
          } else {
            v = static_cast<E>(load_elem<W>(col, r));
          }
          cols_elem<W, FP, NULLABLE, E>(v, null_e, sum_bits, mn, mx, nn);
        }
        nn64 += nn;
        continue;
      }
      const int64_t c0 = t * kTileChunks;
      if (c0 + kTileChunks <= nchunks) {
        uint32_t regs[kColsU][REGS];
#pragma unroll
        for (int u = 0; u < kColsU; ++u) {
          load_bytes<16, true>(col + (c0 + static_cast<int64_t>(u) * kColsBlock + tid) * 16, regs[u]);
        }
#pragma unroll
        for (int u = 0; u < kColsU; ++u) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            E v;
            if constexpr (FP) {
              v = bits_to_double(extract_elem<8>(regs[u], i));
            } else {
              v = static_cast<E>(extract_elem<W>(regs[u], i));
            }
            cols_elem<W, FP, NULLABLE, E>(v, null_e, sum_bits, mn, mx, nn);
          }
        }
      } else {
        // the fragment's last (partial) tile: a chunk per lane and trip
        for (int64_t ch = c0 + tid; ch < nchunks; ch += kColsBlock) {
          uint32_t regs[REGS];
          load_bytes<16, true>(col + ch * 16, regs);
#pragma unroll
          for (int i = 0; i < R; ++i) {
            E v;
            if constexpr (FP) {
              v = bits_to_double(extract_elem<8>(regs, i));
            } else {
              v = static_cast<E>(extract_elem<W>(regs, i));
            }
            cols_elem<W, FP, NULLABLE, E>(v, null_e, sum_bits, mn, mx, nn);
          }
        }
      }
      nn64 += nn;  // (a tile holds < 2^32 elements per lane)
    }
    frag_tile_begin += ntiles;
  }
  out.nn = nn64;
  if constexpr (FP) {
    out.sum = bits_to_double(static_cast<int64_t>(sum_bits));
    out.mn = mn;
    out.mx = mx;
  } else {
    out.sum = static_cast<int64_t>(sum_bits);
    // (a lane that saw no live element keeps the identities, widened)
    out.mn = (W <= 4 && mn == static_cast<E>(INT32_MAX) && out.nn == 0) ? INT64_MAX : static_cast<int64_t>(mn);
    out.mx = (W <= 4 && mx == static_cast<E>(INT32_MIN) && out.nn == 0) ? INT64_MIN : static_cast<int64_t>(mx);
  }
}

HDK_DEV int64_t cols_shfl_down(int64_t v, int delta) {
  int lo = static_cast<int>(v & 0xffffffff);
  int hi = static_cast<int>(v >> 32);
  lo = __shfl_down(lo, delta, kWave);
  hi = __shfl_down(hi, delta, kWave);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

// block-wide fold of the lanes' partials: shuffle tree per wave (fixed order), then wave 0 over the waves' results in LDS
template <bool FP>
HDK_DEV void cols_block_fold(ColsAcc<FP>& acc, int64_t (*s_part)[4]) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  constexpr int32_t op_sum = FP ? WOP_ADD_F64 : WOP_ADD_U64, op_min = FP ? WOP_MIN_F64 : WOP_MIN_I64,
                    op_max = FP ? WOP_MAX_F64 : WOP_MAX_I64;
  int64_t w[4];
  if constexpr (FP) {
    w[0] = double_to_bits(acc.sum);
    w[1] = double_to_bits(acc.mn);
    w[2] = double_to_bits(acc.mx);
  } else {
    w[0] = acc.sum;
    w[1] = acc.mn;
    w[2] = acc.mx;
  }
  w[3] = static_cast<int64_t>(acc.nn);
  const int32_t ops[4] = {op_sum, op_min, op_max, WOP_ADD_U64};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    for (int d = kWave / 2; d > 0; d >>= 1) {
      w[k] = word_combine(ops[k], w[k], cols_shfl_down(w[k], d));
    }
  }
  __syncthreads();  // (s_part is reused column after column)
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_part[wave][k] = w[k];
    }
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t r = s_part[0][k];
      for (int wv = 1; wv < kColsBlock / kWave; ++wv) {
        r = word_combine(ops[k], r, s_part[wv][k]);
      }
      s_part[0][k] = r;
    }
  }
  __syncthreads();
}

// (a template so that the matcher's translation unit can include this header without emitting the kernel)
template <int U = kColsU>
__global__ __launch_bounds__(kColsBlock) void hdk_scan_agg_cols(ColsArgs a) {
  static_assert(U == kColsU, "one instantiation");
  __shared__ int64_t s_part[kColsBlock / kWave][4];
  const int tid = threadIdx.x;
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * a.wpe;
  int32_t err = 0;
  const Watch watch = watch_begin(a.kp);
  // word 0, the row count (COUNT(*), and what tells hdk_finalize that the launch saw rows at all): block 0 states it
  if (tid == 0) {
    int64_t rows = 0;
    if (blockIdx.x == 0) {
      const uint64_t nfrag = *a.kp.num_fragments;
      const uint32_t ntab = *a.kp.num_tables;
      for (uint64_t f = 0; f < nfrag; ++f) {
        rows += a.kp.num_rows[f * ntab];
      }
    }
    slab[0] = rows;
  }
  for (int ci = 0; ci < a.ncols; ++ci) {
    const ColsArgs::Col c = a.col[ci];
    int64_t res[4];  // sum, min, max, non-NULL count of the block (thread 0)
    if (c.fp) {
      ColsAcc<true> acc;
      if (c.nullable) {
        cols_pass<8, true, true>(a, c, acc, watch, err);
      } else {
        cols_pass<8, true, false>(a, c, acc, watch, err);
      }
      cols_block_fold<true>(acc, s_part);
    } else {
      ColsAcc<false> acc;
#define HDK_COLS_PASS(W)                                   \
  if (c.nullable) {                                        \
    cols_pass<W, false, true>(a, c, acc, watch, err);      \
  } else {                                                 \
    cols_pass<W, false, false>(a, c, acc, watch, err);     \
  }
      switch (c.width) {
        case 1: HDK_COLS_PASS(1) break;
        case 2: HDK_COLS_PASS(2) break;
        case 4: HDK_COLS_PASS(4) break;
        default: HDK_COLS_PASS(8) break;
      }
#undef HDK_COLS_PASS
      cols_block_fold<false>(acc, s_part);
    }
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        res[k] = s_part[0][k];
      }
      for (int w = 1; w < a.wpe; ++w) {
        if (a.wcol[w] == ci) {
          const int32_t kind = a.wkind[w];
          slab[w] = kind == CW_SUM ? res[0] : (kind == CW_MIN ? res[1] : (kind == CW_MAX ? res[2] : res[3]));
        }
      }
    }
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
