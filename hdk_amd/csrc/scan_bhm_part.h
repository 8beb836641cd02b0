// scan_bhm_part.h -- the multi-argument / multi-key group-by (scan_bhm.h) for tables BEYOND a CU's LDS: two passes.
//
// A dense table of 10 K - 500 K groups (MultiStep/MSBS002-003, MSPHS002-003, PerfectHashMultiCol/PHM003-005, BaselineHash/
// BH005) is cut into bins of 2^w consecutive entries, each small enough for one block's LDS:
//   pass A  hdk_bhm_scatter     every row becomes ONE 4-byte tuple (2-byte when entry and codes fit 16 bits: BH005's 9 + 4)
//                                   [entry inside the bin : w | code of argument column 0 | code of column 1 | ...]
//                               (code = value - min + 1 by the column's statistics, 0 = NULL: x100 takes 7 bits, x10 four) and
//                               goes behind its bin's cursor -- batches of 8 192 rows per block, staged in LDS ordered by bin, so
//                               that a bin receives a RUN of consecutive tuples per batch (128 bytes on average at 256 bins),
//                               one cursor claim per bin, XCD and batch;
//   pass B  hdk_bhm_aggregate   one block per (bin, XCD sub-slab): the bin's 2^w entries in LDS, the row body of scan_bhm.h
//                               (bhm_update: packed [rows : sum] words, MIN / MAX fields) over the decoded tuples, then the
//                               block's part of slab `XCD` of the whole dense table.
// The eight slabs are folded exactly like the one-pass kernel's (hdk_finalize for a perfect-hash plan, hdk_bhm_fold for open
// addressing).  Bytes per row: the columns once, 4 written, 4 read -- against 8 + 8 of the hash-bin passes and one table
// access per row and target of the global-atomics kernel the suite's shapes ran on before.
// A key or argument outside the statistics, or a sub-slab that overflows (a hot key: more than twice a bin's share of the
// rows), raises the launch's flag: the folds skip and the armed global-atomics kernel redoes the launch.
// Reference being replaced: get_group_value[_fast] + agg_* on the final table for every row (QE/GroupByRuntime.cpp:31-55,198-246).
#pragma once
#include "scan_bhm.h"
#include "part_scatter_batch.h"

namespace hdk {

#ifndef HDK_BHM_PART_BLOCK
#define HDK_BHM_PART_BLOCK 512
#endif
constexpr int kBhmPartBlock = HDK_BHM_PART_BLOCK;  // pass A's block (A/B builds: -DHDK_BHM_PART_BLOCK=256)
// rows per lane and batch of pass A: 16 over 4-byte columns (8 192 rows a batch), 8 over 8-byte ones (their loads take twice the registers)
#ifndef HDK_BHM_PART_VR
#define HDK_BHM_PART_VR 16
#endif
#ifndef HDK_BHM_PART_WAVES
#define HDK_BHM_PART_WAVES 4  // waves per SIMD the register budget is held to (A/B builds: -DHDK_BHM_PART_VR=8 -DHDK_BHM_PART_WAVES=6)
#endif
constexpr int kBhmPartWaves = HDK_BHM_PART_WAVES;
constexpr int bhm_part_vr(int w) { return w == 8 ? 8 : HDK_BHM_PART_VR; }
constexpr size_t bhm_scatter_lds(int w) { return static_cast<size_t>(kBhmPartBlock) * bhm_part_vr(w) * 8; }  // {tuple, place} of every row of a batch
constexpr int kBhmAggBlock = 256;

// words of BhmPartArgs::layout
constexpr uint32_t kBlHist = 0;       // [kPbMaxBins] sampled rows per bin
constexpr uint32_t kBlSampled = 256;  // sampled rows
constexpr uint32_t kBlCap = 512;      // [kPbMaxBins] tuples of a sub-slab of the bin (a multiple of 8)
constexpr uint32_t kBlOff = 768;      // [kPbMaxBins] where it starts inside an XCD's region
constexpr uint32_t kBlRegion = 1024;  // tuples of one XCD's region
constexpr uint32_t kBlWords = 1032;

struct BhmPartArgs {
  BhmArgs b;               // columns, statistics, argument descriptors; the LDS geometry of pass B (entries = 2^w)
  uint32_t w;              // bits of a tuple's entry inside its bin
  uint32_t nbins;          // <= 256
  uint32_t total_entries;  // the dense table
  uint32_t tw;             // bytes of a tuple: 2 when entry and codes fit 16 bits (BH005 / PHS005: 9 + 4), else 4
  uint32_t cshift[kBhmMaxSrc];  // position of argument column s's code in the tuple
  uint32_t cmask[kBhmMaxSrc];
  uint64_t cap_limit;      // most tuples a (bin, XCD) sub-slab may hold (its size is a 32-bit word of the layout)
  uint64_t region_max;     // tuples of one XCD's region the allocation has room for
  uint64_t total_rows;     // the launch's row bound
  uint32_t* tuples;        // [kPbXcds] regions of `region` tuples of `tw` bytes (+ one batch of slack behind the last): sub-slab
                           // (x, bin) starts at layout[kBlOff + bin] of region x and holds layout[kBlCap + bin] tuples
  uint32_t* fill;          // [kPbXcds][nbins] x kPbCursorStride
  uint32_t* layout;        // kBl* words: what the sample saw and the sub-slabs made from it (hdk_bhm_part_sample / _layout)
  uint32_t sample_stride;  // every sample_stride-th tile of 16-byte steps is looked at
  uint32_t parts;          // pass B: blocks a (bin, XCD) sub-slab may be cut into (grid.y) ...
  uint32_t part_tuples;    // ... when it holds more than this many tuples per part (a multiple of 4 096): a hot key's bin
  uint32_t generation;     // tuples one LDS table of pass B may take (its packed fields: rows in 24 bits, sums in 40): a block
                           // with more flushes its table into the slab and starts the next generation (a multiple of 4 096)
  int32_t wop[kMaxWordsPerEntry];  // how a slab word of a later generation joins the earlier ones (agg_common.h's word_combine)
};

// ---- the sub-slabs are sized from a SAMPLE of the keys ----------------------------------------------------------------------------
// Sub-slabs of one size for every bin (twice the even share) made any uneven bin a fallback: 2 % of NULL keys -- one entry, one
// bin -- sent MSPHS003 from 1.3 ms to 1 270 ms per 256 M rows through the armed global-atomics launch.  Now every
// sample_stride-th tile's keys are binned first (a few per cent of one column's bytes), and a bin's sub-slabs hold twice what
// the sample promises (+ 4 096).  What the sample did not see still overflows into the flag and the fallback.
template <int NK, int W>
HDK_DEV uint32_t bhm_part_sample_bin(const BhmPartArgs& g, const uint32_t (&kr)[NK][4], int i, bool nulls) {
  const BhmArgs& a = g.b;
  uint32_t idx = 0;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const BhmKey& key = a.key[kk];
    bool wide;
    const int32_t kv = bhm_narrow<W>(kr[kk], i, wide);
    uint32_t d = min(static_cast<uint32_t>(kv) - static_cast<uint32_t>(key.min), key.n - 1u);
    if (nulls && key.nullable != 0 && kv == key.null32) {
      d = key.null_d;
    }
    idx += NK == 1 ? d : __umul24(d, key.stride);
  }
  return min(idx >> g.w, g.nbins - 1u);
}

template <int NK, int W>
__global__ __launch_bounds__(256) void hdk_bhm_part_sample(BhmPartArgs g) {
  constexpr int R = 16 / W;
  constexpr int64_t kTile = 256 * R;
  const BhmArgs& a = g.b;
  __shared__ uint32_t s_hist[kPbMaxBins];
  const int tid = threadIdx.x;
  s_hist[tid] = 0;
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const bool nulls = a.any_nullable != 0;
  const int64_t stride = g.sample_stride;
  uint32_t seen = 0;
  int64_t tile = static_cast<int64_t>(blockIdx.x) * stride;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t ntiles = a.kp.num_rows[f * ntab] / kTile;  // (full tiles only)
    const int8_t* const* cols = a.kp.col_buffers[f];
    // four sampled tiles at a time: their loads go out together (a tile at a time the kernel was a chain of load -> LDS adds)
    while (tile < frag_tile_begin + ntiles) {
      uint32_t kr[4][NK][4];
      bool have[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t t = tile + j * static_cast<int64_t>(gridDim.x) * stride;
        have[j] = t < frag_tile_begin + ntiles;
        if (have[j]) {
          const int64_t r = (t - frag_tile_begin) * kTile + static_cast<int64_t>(tid) * R;
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            load_bytes<16, false>((gcol_t)cols[a.key[kk].buf_idx] + r * W, kr[j][kk]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (have[j]) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            atomicAdd(&s_hist[bhm_part_sample_bin<NK, W>(g, kr[j], i, nulls)], 1u);
          }
          seen += R;
          tile += static_cast<int64_t>(gridDim.x) * stride;
        }
      }
    }
    frag_tile_begin += ntiles;
  }
  __syncthreads();
  if (s_hist[tid]) {
    atomicAdd(g.layout + kBlHist + tid, s_hist[tid]);
  }
  if (seen) {
    atomicAdd(g.layout + kBlSampled, seen);
  }
}

// one block: the bins' sub-slab sizes and starts from the sample
template <int DUMMY = 0>
__global__ __launch_bounds__(kPbMaxBins) void hdk_bhm_part_layout(BhmPartArgs g) {
  __shared__ uint64_t s_scan[kPbMaxBins];
  __shared__ uint32_t s_bad;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_bad = 0;
  }
  __syncthreads();
  const uint32_t sampled = g.layout[kBlSampled];
  const uint64_t even = g.total_rows / g.nbins;
  uint64_t est = 0;
  if (static_cast<uint32_t>(tid) < g.nbins) {
    // (too small a sample -- short fragments -- says nothing: the even share then)
    est = sampled >= 64u * g.nbins ? static_cast<uint64_t>(static_cast<double>(g.layout[kBlHist + tid]) * static_cast<double>(g.total_rows) /
                                                           static_cast<double>(sampled))
                                   : even;
  }
  uint64_t cap = static_cast<uint32_t>(tid) < g.nbins ? ((est / kPbXcds) * 2 + 4096 + 7) & ~7ull : 0;
  if (cap > g.cap_limit) {
    s_bad = 1;  // (a sub-slab's size must fit 32 bits)
    cap = g.cap_limit & ~7ull;
  }
  s_scan[tid] = cap;
  __syncthreads();
  for (int d = 1; d < kPbMaxBins; d <<= 1) {
    const uint64_t v = tid >= d ? s_scan[tid - d] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  uint64_t region = s_scan[kPbMaxBins - 1];
  uint64_t off = s_scan[tid] - cap;
  if (region > g.region_max) {  // (a sample that promises more rows than the launch's bound by a quarter: not with honest inputs)
    s_bad = 1;
    cap = (g.region_max / g.nbins) & ~7ull;
    off = static_cast<uint64_t>(tid) * cap;
    region = static_cast<uint64_t>(g.nbins) * cap;
  }
  g.layout[kBlCap + tid] = static_cast<uint32_t>(tid) < g.nbins ? static_cast<uint32_t>(cap) : 0u;
  g.layout[kBlOff + tid] = static_cast<uint32_t>(off);
  __syncthreads();
  if (tid == 0) {
    g.layout[kBlRegion] = static_cast<uint32_t>(region);
    g.layout[kBlRegion + 1] = static_cast<uint32_t>(region >> 32);
    if (s_bad) {
      atomicOr(g.b.flag, 1u);
    }
  }
}

// ---- pass A ---------------------------------------------------------------------------------------------------------------------
// R = 16 / W consecutive rows of one lane (one 16-byte step of every streamed column) -> their tuples and bins.  FULL: the
// tile lies inside the fragment (every row exists).
// No row is tested against its statistics here: distances from the minimum are CLAMPED for everything that addresses memory
// (the bin, the entry) and their per-lane MAXIMA are carried through the kernel (kover, sover) and compared once at its end --
// a row outside the statistics is scattered somewhere harmless and the launch is redone by the armed fallback.  With a compare
// and a select per row and column the compiler held sixteen rows' lane masks in scalar registers and spilled them (258
// v_readlane / v_writelane, 24 instructions per row for the tuple alone; pass A is bound by instruction issue).
template <int NK, int NS, int W, bool Q, bool FULL, bool NULLS>
HDK_DEV void bhm_part_rows(const BhmPartArgs& g, const int8_t* const* cols, const uint32_t (&kr)[NK][4], const uint32_t (&xr)[NS][4],
                           const uint32_t (&qr)[kMaxPlainQuals][4], bool qvec, int64_t first_row, int64_t nrows, uint32_t* tup,
                           uint32_t* bin4, uint32_t (&kover)[NK], uint32_t (&sover)[NS], uint32_t& stale) {
  constexpr int R = 16 / W;
  const BhmArgs& a = g.b;
  const uint32_t wmask = (1u << g.w) - 1u;
  int32_t kv[NK][R], xv[NS][R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      bool wide;
      kv[kk][i] = bhm_narrow<W>(kr[kk], i, wide);
      if (W == 8) stale |= wide ? 1u : 0u;  // (of a row that does not take part, too: redone needlessly, never wrongly)
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bool wide;
      xv[s][i] = bhm_narrow<W>(xr[s], i, wide);
      if (W == 8) stale |= wide ? 1u : 0u;
    }
  }
  bool ok[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    ok[i] = FULL || first_row + i < nrows;
  }
  if (Q) {  // plain filters: rows that fail are not scattered
    if (qvec && a.qvec == 2) {
      const uint32_t lean = bhm_quals_lean<R, W, NK, NS>(a, kr, xr, qr);
#pragma unroll
      for (int i = 0; i < R; ++i) {
        ok[i] = ok[i] && ((lean >> i) & 1u) != 0;
      }
    } else if (qvec) {
      plain_quals_pass_with<R, true>(
          a.q, a.nquals,
          [&](int qi, const ProjFastQual&, const bool (&)[R], int64_t (&v)[R]) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
              // (qi is wave-uniform: a select over the three register sets' values, not an indexed array -- that would live in scratch)
              const int64_t v0 = extract_elem<W>(qr[0], i), v1 = extract_elem<W>(qr[1], i), v2 = extract_elem<W>(qr[2], i);
              v[i] = qi == 0 ? v0 : (qi == 1 ? v1 : v2);
            }
          },
          ok);
    } else {
      int64_t rows[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        rows[i] = ok[i] ? first_row + i : 0;  // (only rows that are `ok` are read)
      }
      plain_quals_pass<R, true>(a.q, a.nquals, cols, rows, ok, true);
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    uint32_t idx = 0;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const BhmKey& key = a.key[kk];
      uint32_t d = static_cast<uint32_t>(kv[kk][i]) - static_cast<uint32_t>(key.min);
      const bool isnull = NULLS && (key.nullable != 0) & (kv[kk][i] == key.null32);
      uint32_t seen = d;
      if (NULLS) {
        seen = isnull ? 0u : seen;
      }
      if (!FULL || Q) {
        seen = ok[i] ? seen : 0u;
      }
      kover[kk] = max(kover[kk], seen);
      d = min(d, key.n - 1u);
      if (NULLS) {
        d = isnull ? key.null_d : d;  // (the NULL key's term: a translated value inside the range, or the entry behind it)
      }
      idx += NK == 1 ? d : __umul24(d, key.stride);
    }
    uint32_t t = idx & wmask;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const BhmSrc& src = a.src[s];
      uint32_t d = static_cast<uint32_t>(xv[s][i]) - static_cast<uint32_t>(src.raw_min);
      uint32_t code = d + 1u;
      if (NULLS) {
        const bool isnull = (src.nullable != 0) & (xv[s][i] == src.null32);
        d = isnull ? 0u : d;
        code = isnull ? 0u : code;
      }
      if (!FULL || Q) {
        d = ok[i] ? d : 0u;
      }
      sover[s] = max(sover[s], d);
      t |= code << g.cshift[s];
    }
    bin4[i] = (idx >> g.w) << 2;
    if (!FULL || Q) {
      bin4[i] = ok[i] ? bin4[i] : 4u * kPbMaxBins;
    }
    tup[i] = t;
  }
}

template <int NK, int NS, int W = 4, bool Q = false>
__global__ __launch_bounds__(kBhmPartBlock) __attribute__((amdgpu_waves_per_eu(kBhmPartWaves, kBhmPartWaves))) void hdk_bhm_scatter(BhmPartArgs g) {
  constexpr int VR = bhm_part_vr(W);
  constexpr int kBhmPartTile = kBhmPartBlock * VR;
  constexpr int R = 16 / W, U = VR / R;
  const BhmArgs& a = g.b;
  // a batch's bins: rows per bin, where a bin's run starts in the staging area, and what turns a staging position into the
  // tuple's place behind the bin's cursor.  Entry kPbMaxBins: the DEAD bin (rows that are not scattered: a ragged tile's end,
  // rows outside the statistics, rows a filter dropped) -- staged behind all live runs and never copied out, so that the row
  // loops carry no `if (live)` (the compiler kept 16 lane masks in scalar registers and spilled them: 258 v_readlane /
  // v_writelane and 111 branches in a kernel that is bound by instruction issue)
  __shared__ uint32_t s_cnt[kPbMaxBins + 1];
  __shared__ uint2 s_pair[kPbMaxBins + 1];  // {start of the bin's run in the staging area, place of the run's first tuple}
  __shared__ int32_t s_watch;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn32[];  // the staging area: {tuple, place} per row of the batch
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kPbXcds - 1);
  for (int i = tid; i <= kPbMaxBins; i += kBhmPartBlock) {
    s_cnt[i] = 0;
  }
  // this block's tuples go to its XCD's region of the slab -- positions inside it fit 32 bits
  const uint64_t region = (static_cast<uint64_t>(g.layout[kBlRegion + 1]) << 32) | g.layout[kBlRegion];
  uint8_t* const xcd_tuples = reinterpret_cast<uint8_t*>(g.tuples) + static_cast<size_t>(xcd) * region * g.tw;
  uint32_t my_cap = 0, my_off = 0;  // of the bin this thread claims for
  if (tid < kPbMaxBins) {
    my_cap = g.layout[kBlCap + tid];
    my_off = g.layout[kBlOff + tid];
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  int32_t err = 0;
  uint32_t stale = 0;
  uint32_t kover[NK], sover[NS];  // per lane: the largest distance from a column's minimum seen so far (bhm_part_rows)
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    kover[kk] = 0;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    sover[s] = 0;
  }
  const Watch watch = watch_begin(a.kp);
  const bool nulls = a.any_nullable != 0;
  const bool narrow = g.tw == 2;  // (wave-uniform: 2-byte tuples)
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  bool stop = false;
  for (uint64_t f = 0; f < nfrag && !stop; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kBhmPartTile - 1) / kBhmPartTile;
    const int8_t* const* cols = a.kp.col_buffers[f];
    gcol_t kcol[NK], xcol[NS];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      kcol[kk] = (gcol_t)cols[a.key[kk].buf_idx];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      xcol[s] = (gcol_t)cols[a.src[s].buf_idx];
    }
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      if (watch.flags) {
        if (const int32_t wv = watch_poll_block(watch, &s_watch)) {
          err = wv;
          stop = true;
          break;
        }
      }
      const int64_t row0 = (tile - frag_tile_begin) * kBhmPartTile;
      const bool full = row0 + kBhmPartTile <= nrows;
      uint32_t tup[VR], bin4[VR];  // bin4: byte offset of the row's bin in s_cnt (the dead bin: 4 * kPbMaxBins)
      // (issuing the NEXT tile's loads here, before the batch's LDS work, measured worse: 2.43 -> 2.49 ms per 1 B rows for BH005,
      // 1.31 -> 1.47 ms per 256 M for MSBS002 -- the second register set spills)
      uint32_t kr[U][NK][4], xr[U][NS][4], qr[Q ? U : 1][kMaxPlainQuals][4];
      const bool qvec = Q && full && a.qvec != 0;  // the filter's columns with the batch (scan_bhm.h); a ragged tile: by row number
      // a full tile's columns: 16 bytes per lane, step and column
#define HDK_BHM_PART_LOAD(ROW0)                                                                       \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                    \
    const int64_t r = (ROW0) + (static_cast<int64_t>(u) * kBhmPartBlock + tid) * R;                       \
    _Pragma("unroll") for (int kk = 0; kk < NK; ++kk) {                                              \
      load_bytes<16, true>(kcol[kk] + r * W, kr[u][kk]);                                             \
    }                                                                                                \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                 \
      load_bytes<16, true>(xcol[s] + r * W, xr[u][s]);                                               \
    }                                                                                                \
    _Pragma("unroll") for (int qi = 0; qi < kMaxPlainQuals; ++qi) {                                  \
      if (Q && a.qvec != 0 && qi < a.nquals && (a.qvec != 2 || a.qsrc[qi] == kBhmQualOwnLoad)) {     \
        load_bytes<16, true>((gcol_t)cols[a.q[qi].col.buf_idx] + r * W, qr[Q ? u : 0][qi]);          \
      }                                                                                              \
    }                                                                                                \
  }
      if (full) {
        HDK_BHM_PART_LOAD(row0)
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const int64_t r = row0 + (static_cast<int64_t>(u) * kBhmPartBlock + tid) * R + i;
            const bool in = r < nrows;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
              const int64_t v = in ? load_elem<W>(kcol[kk], r) : 0;
              if (W == 8) {
                kr[u][kk][(2 * i) % 4] = static_cast<uint32_t>(v);
                kr[u][kk][(2 * i + 1) % 4] = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
              } else {
                kr[u][kk][i % 4] = static_cast<uint32_t>(v);
              }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              const int64_t v = in ? load_elem<W>(xcol[s], r) : 0;
              if (W == 8) {
                xr[u][s][(2 * i) % 4] = static_cast<uint32_t>(v);
                xr[u][s][(2 * i + 1) % 4] = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
              } else {
                xr[u][s][i % 4] = static_cast<uint32_t>(v);
              }
            }
          }
        }
      }
      // rows -> tuples (bhm_part_rows); the full tile's instantiations know every row is there, the ones for plans without an
      // announced NULL carry no NULL code (wave-uniform branches around whole instantiations, nothing per row)
#define HDK_BHM_PART_ROWS(FULL_, NULLS_)                                                                                          \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                                                \
    bhm_part_rows<NK, NS, W, Q, FULL_, NULLS_>(g, cols, kr[u], xr[u], qr[Q ? u : 0], qvec,                                       \
                                               row0 + (static_cast<int64_t>(u) * kBhmPartBlock + tid) * (16 / W), nrows, &tup[u * (16 / W)], \
                                               &bin4[u * (16 / W)], kover, sover, stale);                                        \
  }
      if (full) {
        if (nulls) {
          HDK_BHM_PART_ROWS(true, true)
        } else {
          HDK_BHM_PART_ROWS(true, false)
        }
      } else if (nulls) {
        HDK_BHM_PART_ROWS(false, true)
      } else {
        HDK_BHM_PART_ROWS(false, false)
      }
#undef HDK_BHM_PART_ROWS
#undef HDK_BHM_PART_LOAD
      // the batch: LDS histogram by bin, one cursor claim per bin and XCD, staging ordered by bin, copy-out in runs.
      // LDS operations per row: the histogram's add, ONE 8-byte read of its bin's {start of the run in the staging area, place of
      // the run's first tuple behind the cursor}, ONE 8-byte staging write {tuple, place}, one 8-byte read at copy-out -- four,
      // where separate arrays (run starts, bin of every slot, staging words, deltas) took seven
      uint32_t rank[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        // (behind a filter many rows are dead: their adds would meet on one address)
        rank[r] = (!Q || bin4[r] != 4u * kPbMaxBins) ? atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(s_cnt) + bin4[r]), 1u) : 0u;
      }
      __syncthreads();
      if (tid < kPbMaxBins) {
        // a thread per bin: the claim behind the bin's cursor, and -- every wave for itself, no barrier -- the exclusive scan of
        // the counts (its own 64 bins by shuffles, the bins of the waves before it summed from LDS)
        const uint32_t n = s_cnt[tid];
        uint32_t base = 0;
        if (n) {
          base = atomicAdd(g.fill + (static_cast<size_t>(xcd) * g.nbins + tid) * kPbCursorStride, n);
        }
        const int lane = tid & (kWave - 1), wv = tid / kWave;
        uint32_t incl = n;
#pragma unroll
        for (int dd = 1; dd < kWave; dd <<= 1) {
          const uint32_t v = __shfl_up(incl, dd, kWave);
          if (lane >= dd) {
            incl += v;
          }
        }
        uint32_t before = 0;
        for (int c = 0; c < wv; ++c) {  // (wave-uniform trip count)
          uint32_t v = s_cnt[c * kWave + lane];
#pragma unroll
          for (int dd = kWave / 2; dd >= 1; dd >>= 1) {
            v += __shfl_xor(v, dd, kWave);
          }
          before += v;
        }
        const uint32_t x = before + incl - n;
        // a claim that does not fit its sub-slab (a hot key) raises the flag -- the launch is redone by the armed fallback -- and
        // is clamped to the sub-slab's end: its tuples spill over the next sub-slab's first ones (or the slack behind the last)
        if (n && static_cast<uint64_t>(base) + n > my_cap) {
          stale = 1;
          base = min(base, my_cap);
        }
        s_pair[tid] = make_uint2(x, my_off + base);
        if (tid == kPbMaxBins - 1) {
          s_pair[kPbMaxBins] = make_uint2(x + n, 0u);  // the dead run starts where the live ones end
        }
      }
      __syncthreads();
      uint2* const stage = reinterpret_cast<uint2*>(s_dyn32);
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        if (!Q || bin4[r] != 4u * kPbMaxBins) {
          const uint2 pr = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(s_pair) + 2u * bin4[r]);
          stage[pr.x + rank[r]] = make_uint2(tup[r], pr.y + rank[r]);  // (the dead run's are never read)
        }
      }
      if (tid < kPbMaxBins) {
        s_cnt[tid] = 0;
      }
      if (tid == 0) {
        s_cnt[kPbMaxBins] = 0;
      }
      __syncthreads();
      const uint32_t total = s_pair[kPbMaxBins].x;
      if (narrow) {  // (a uniform branch around the loops, not inside them)
        for (uint32_t i = tid; i < total; i += kBhmPartBlock) {
          const uint2 v = stage[i];
          reinterpret_cast<uint16_t*>(xcd_tuples)[v.y] = static_cast<uint16_t>(v.x);
        }
      } else {
        for (uint32_t i = tid; i < total; i += kBhmPartBlock) {
          const uint2 v = stage[i];
          reinterpret_cast<uint32_t*>(xcd_tuples)[v.y] = v.x;
        }
      }
      // (no barrier here: the next batch touches only the counters -- zeroed before the last barrier -- until ITS first barrier,
      // which a wave reaches after its copy-out; waves that are done start on the next tile's loads)
    }
    frag_tile_begin += ntiles;
  }
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    stale |= kover[kk] >= a.key[kk].n ? 1u : 0u;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    stale |= sover[s] > a.src[s].raw_span ? 1u : 0u;
  }
  if (stale) {
    atomicOr(a.flag, 1u);
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

// the slabs hold every word's identity before pass B (parts of a hot sub-slab join their tables with atomics)
struct BhmSlabInitArgs {
  int64_t* slabs;
  uint64_t words;
  int32_t wpe;
  int32_t wop[kMaxWordsPerEntry];
};
template <int DUMMY = 0>
__global__ __launch_bounds__(256) void hdk_bhm_slab_init(BhmSlabInitArgs a) {
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < a.words; i += static_cast<uint64_t>(gridDim.x) * 256) {
    a.slabs[i] = word_identity(a.wop[i % static_cast<uint32_t>(a.wpe)]);
  }
}

// ---- pass B: one block per (bin, XCD sub-slab) -----------------------------------------------------------------------------------
template <class C, int NS, int TW = 4>
__global__ __launch_bounds__(kBhmAggBlock) void hdk_bhm_aggregate(BhmPartArgs g) {
  const BhmArgs& a = g.b;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
  const int tid = threadIdx.x;
  if (*a.flag) {
    return;  // (pass A already knows the launch is to be redone)
  }
  const uint32_t bin = blockIdx.x / kPbXcds, x = blockIdx.x % kPbXcds;
  const size_t sub = static_cast<size_t>(x) * g.nbins + bin;  // ([xcd][bin][cap])
  const uint32_t n_sub = min(g.fill[sub * kPbCursorStride], g.layout[kBlCap + bin]);
  // A sub-slab far above the average (a hot key, a NULL key: one LDS address for a large share of the rows -- 64 lanes of a wave
  // serialise on it) is cut into parts, one block each, which JOIN their tables into the slab with global atomics (the slab
  // holds the words' identities then: hdk_bhm_slab_init); every other sub-slab is block 0's alone and plainly written.  One
  // block per sub-slab took 27.5 ms per 256 M rows with half the rows in one key, 7.8 ms with 10 % NULL keys.
  const uint32_t nparts = n_sub > 2 * static_cast<uint64_t>(g.part_tuples) ? min(g.parts, (n_sub + g.part_tuples - 1) / g.part_tuples) : 1u;
  if (blockIdx.y >= max(nparts, 1u)) {
    return;
  }
  const bool shared = nparts > 1;
  uint32_t part_begin = 0, n = n_sub;
  if (shared) {
    const uint32_t chunk = (((n_sub + nparts - 1) / nparts) + 4095u) & ~4095u;
    part_begin = min(blockIdx.y * chunk, n_sub);
    n = min(part_begin + chunk, n_sub) - part_begin;
  }
  {
    uint4* z = reinterpret_cast<uint4*>(lds8);
    const uint32_t n16 = a.lds_bytes / 16;
    for (uint32_t i = tid; i < n16; i += kBhmAggBlock) {
      z[i] = make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  uint8_t* rp = lds8 + static_cast<size_t>(tid & (a.rep - 1)) * a.rep_bytes;
  const uint64_t region = (static_cast<uint64_t>(g.layout[kBlRegion + 1]) << 32) | g.layout[kBlRegion];
  const uint8_t* t = reinterpret_cast<const uint8_t*>(g.tuples) + (static_cast<size_t>(x) * region + g.layout[kBlOff + bin] + part_begin) * TW;
  const uint32_t wmask = (1u << g.w) - 1u;
  const uint32_t dummy = a.entries;  // (= 2^w: the entry behind the bin's table)
  const bool nulls = C::nulls(a);
  constexpr int TPV = 16 / TW;  // tuples of a 16-byte load: 4, or 8 two-byte ones
  constexpr int U = 16 / TPV;   // loads per lane and step: 16 tuples either way
  constexpr int R = 4;          // rows of one bhm_update
  constexpr uint32_t kStep = kBhmAggBlock * TPV * U;
  typedef uint32_t __attribute__((ext_vector_type(4))) u32x4_t;
  // FULL: every tuple of the step exists (no per-tuple bound)
#define HDK_BHM_AGG_ROWS(FULL_)                                                                                  \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                               \
    const uint32_t i = base + (static_cast<uint32_t>(u) * kBhmAggBlock + tid) * TPV;                            \
    const uint32_t wd[4] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};                                                \
    _Pragma("unroll") for (int h = 0; h < TPV / R; ++h) {                                                       \
      uint32_t e[R];                                                                                            \
      int32_t xv[NS][R];                                                                                        \
      bool lv[NS][R];                                                                                           \
      _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                           \
        const int jj = h * R + j;                                                                               \
        const uint32_t tup = TW == 2 ? (wd[jj / 2] >> (16 * (jj & 1))) & 0xFFFFu : wd[jj % 4];                  \
        const bool in = FULL_ || (i + jj < n);                                                                  \
        e[j] = in ? tup & wmask : dummy;                                                                        \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                        \
          const uint32_t code = (tup >> g.cshift[s]) & g.cmask[s];                                              \
          xv[s][j] = static_cast<int32_t>(code);                                                                \
          lv[s][j] = nulls ? (in & (code != 0)) : true;                                                         \
        }                                                                                                       \
      }                                                                                                         \
      bhm_update<C, NS, R, false>(a, rp, dummy, e, xv, lv);                                               \
    }                                                                                                           \
  }
  // the block's part of slab x of the whole dense table: written when the tuples are through -- and before that whenever the
  // table has taken a generation's worth of them (a hot key: one bin with a large share of the rows), joined word by word
  const uint32_t e0 = bin << g.w;
  int64_t* slab = a.slabs + static_cast<size_t>(x) * g.total_entries * a.wpe;
  bool flushed = false;
  auto flush = [&]() {
    for (uint32_t ei = tid; ei < a.entries && e0 + ei < g.total_entries; ei += kBhmAggBlock) {
      int64_t* out = slab + static_cast<size_t>(e0 + ei) * a.wpe;
      if (shared) {
        int64_t w[kMaxWordsPerEntry];
        bhm_slab_entry(a, lds8, ei, w);
        for (int k = 0; k < a.wpe; ++k) {
          const int32_t op = g.wop[k];
          if (w[k] == word_identity(op)) {
            continue;  // (most entries of a hot bin's other keys in most parts: nothing to add)
          }
          if (op == WOP_MIN_I64) {
            atomicMin(reinterpret_cast<long long*>(out + k), static_cast<long long>(w[k]));
          } else if (op == WOP_MAX_I64) {
            atomicMax(reinterpret_cast<long long*>(out + k), static_cast<long long>(w[k]));
          } else {
            atomicAdd(reinterpret_cast<unsigned long long*>(out + k), static_cast<unsigned long long>(w[k]));
          }
        }
      } else if (!flushed) {
        bhm_slab_entry(a, lds8, ei, out);
      } else {
        int64_t w[kMaxWordsPerEntry];
        bhm_slab_entry(a, lds8, ei, w);
        for (int k = 0; k < a.wpe; ++k) {
          out[k] = word_combine(g.wop[k], out[k], w[k]);
        }
      }
    }
  };
  uint32_t next_flush = g.generation;
  for (uint32_t base = 0; base < n; base += kStep) {
    if (base >= next_flush) {  // (block-uniform)
      __syncthreads();
      flush();
      flushed = true;
      __syncthreads();
      uint4* z = reinterpret_cast<uint4*>(lds8);
      for (uint32_t i = tid; i < a.lds_bytes / 16; i += kBhmAggBlock) {
        z[i] = make_uint4(0, 0, 0, 0);
      }
      __syncthreads();
      next_flush += g.generation;
    }
    const bool full = base + kStep <= n;
    u32x4_t tv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + (static_cast<uint32_t>(u) * kBhmAggBlock + tid) * TPV;
      if (full || i + TPV <= n) {  // (sub-slabs start 16-byte aligned: cap is a multiple of 8)
        tv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(t + static_cast<size_t>(i) * TW));
      } else {
        uint32_t wd[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < TPV; ++j) {
          if (i + j < n) {
            if (TW == 2) {
              wd[j / 2] |= static_cast<uint32_t>(reinterpret_cast<const uint16_t*>(t)[i + j]) << (16 * (j & 1));
            } else {
              wd[j % 4] = reinterpret_cast<const uint32_t*>(t)[i + j];
            }
          }
        }
        tv[u].x = wd[0];
        tv[u].y = wd[1];
        tv[u].z = wd[2];
        tv[u].w = wd[3];
      }
    }
    if (full) {
      HDK_BHM_AGG_ROWS(true)
    } else {
      HDK_BHM_AGG_ROWS(false)
    }
  }
#undef HDK_BHM_AGG_ROWS
  __syncthreads();
  flush();
}

}  // namespace hdk
