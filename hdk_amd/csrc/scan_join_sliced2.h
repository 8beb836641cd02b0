// scan_join_sliced2.h -- the sliced join (scan_join_sliced.h) for the star-schema shapes AROUND BASELINE config 3:
//   SELECT [g(payload),] agg(f(x, payload)), ...  FROM outer JOIN inner ON outer.key = inner.key
//   [WHERE outer_col cmp literal AND ... AND payload cmp literal ...]  [GROUP BY g(payload)]
// i.e. the join followed by a perfect-hash GROUP BY on the joined column ("GROUP BY dim.dval % 64" --
// here every key the reference's planner gives a perfect-hash layout: the payload itself or payload / literal), by
// filters on either side, and by any list of up to four integer aggregates over x, the payload, x +-* payload or one of
// them +-* a literal.  The reference runs all of this in ONE row function: join loop, filters, group lookup, agg_*
// calls (QE/IRCodegen.cpp:497-667 inside QE/RowFuncBuilder.cpp:597-745); here it is the three passes of the sliced join
//   pass 0  hdk_join_order_probe                 (scan_join_sliced.h, unchanged)
//   pass 1  hdk_join_scatter_slices<true, Q>     the outer-column filters run BEFORE the scatter: fewer tuples
//   pass 1b hdk_join_scatter_level2              (key ranges beyond 256 slices: see below)
//   pass 2  hdk_join_agg_sliced2<GROUPED>        below
// with the batched interpreter (hdk_scan_agg_vec_join) armed behind them for whatever the 8-byte tuples cannot carry.
//
// Round 5 -- still 8 bytes per tuple:
//   * a SECOND OUTER COLUMN y (GROUP BY fact.g ... WHERE dim.d < c; SUM(fact.a), SUM(fact.b)): the key offset needs
//     ceil(log2(key range)) bits of the tuple's low word, y's code (y - min + 1 by its statistics, 0 = NULL) takes the
//     rest -- 8 bits beside a 10 M-key dimension, enough for any group key whose table fits LDS.  y reads like a third
//     payload word (pidx 2) in filters-free positions: the group key (y, y / lit, y % lit) or an aggregate argument (y,
//     x op y, y op lit).  An 8- or 4-byte integer column.  A value outside the statistics arms the interpreter.
//   * TWO PAYLOAD WORDS PACKED into one LDS word when their statistics fit 32 bits together (NPAY = 3): a slice keeps
//     39 936 keys, so the star shape "filter on dim.a, group by dim.b" keeps ONE scatter level up to 10.2 M keys
//     (c3f at 256 M rows: 3.27 -> 2.36 ms).
//   * FILTERS ON THE JOINED COLUMNS once per key: an INNER join followed by `WHERE dim.d < c` is the join with the filtered
//     dimension, so pass 2 applies them while it loads a slice's payloads into LDS -- a key whose payloads fail reads as a key
//     without a partner (the overflow area, probed in memory, still filters per tuple): c3f 2.36 -> 2.27 ms, c3x 2.65 -> 2.52.
//
// Pass 2, per 1 024-thread block: the slice's int32 payloads in LDS next to a PRIVATE group table
// (entry x word x replica, the layout of agg_common.h); per tuple one LDS probe, the payload filters, the group entry, and
// one LDS atomic per aggregate word.  Everything about a tuple is 32 bits wide -- x and the payload fit by the column
// statistics the 8-byte tuples rely on -- so NULL tests are 32-bit compares and no result can overflow or collide with a
// 64-bit sentinel: the row function is ~40 vector instructions where the 64-bit general one of scan_join_direct.h (jd_eval)
// needs 124.  What a target computes is looked up per tuple from wave-uniform (scalar) state: no per-query code.
//
// TWO SCATTER LEVELS (round 4: no more cliff at 256 x 39 936 = 10.2 M keys).  A block can stage 256 write targets in LDS, so
// a key range of more than 256 LDS-sized slices is cut twice, like the open-addressing group-by (scan_agg_partitioned.h):
// pass 1 scatters into <= 256 COARSE bins of `fpc` slices each (the same kernel: a coarse bin is its "slice"), pass 1b
// reads one coarse bin and scatters it into its <= 256 fine slices (all blocks of a coarse bin on one XCD, so that partial
// lines meet in one L2), pass 2 walks the fine slices, one block at a time per slice, reloading its LDS payloads per
// slice and keeping its group table across slices.  256 x 256 slices x 32 K keys cover every 32-bit key range; the
// second level costs 16 more bytes per row (read + write of the 8-byte tuples).
#pragma once
#include "scan_join_sliced.h"

namespace hdk {

constexpr int kS2MaxTargets = 4;
constexpr int kS2MaxPayQuals = 2;
constexpr uint32_t kS2LdsBytes = 160 * 1024 - 1024;  // dynamic LDS of a pass-2 block (static: a few words)
constexpr uint32_t kS2MaxTableBytes = 48 * 1024;

enum S2Src : int32_t {
  S2_NONE = 0,   // COUNT(*): the row count word
  S2_X = 1,
  S2_P = 2,
  S2_X_ADD_P = 3,
  S2_X_SUB_P = 4,
  S2_P_SUB_X = 5,
  S2_X_MUL_P = 6,
  S2_X_OP_LIT = 7,  // x (+ - *) literal, `op`
  S2_P_OP_LIT = 8
};

struct S2Target {
  int32_t src;        // S2Src
  int32_t op;         // HDK_OP_ADD / SUB / MUL for the *_OP_LIT forms
  int64_t lit;
  int32_t null_if_x;  // the argument is NULL when x is (a nullable x leaf is part of it)
  int32_t null_if_p;
  int32_t vword, nword;  // words of the entry, or -1
  int32_t wop;        // WOP_ADD_U64 / WOP_MIN_I64 / WOP_MAX_I64
  int32_t pidx;       // which payload word "P" is (0 or 1)
};

struct S2PayQual {
  int32_t cmp;        // hdk_hip_cmp
  int32_t pidx;       // payload word compared
  int64_t rhs;
};

struct Slice2Args {
  SliceArgs s;               // geometry, tuples, cursors, mode word, pass-1 filters
  // group key: none (one entry), the payload, or payload / divisor; perfect hash: entry = key - key_min
  int32_t grouped;
  int32_t key_div;           // 0: the payload itself; else the literal divisor (1 .. 2^31 - 1)
  int32_t key_mod;           // 1: the key is payload % key_div (C's truncating remainder: the sign of the payload), not the quotient
  int32_t pad_mod_;
  uint32_t div_magic, div_shift;
  int64_t key_min;
  int64_t null_entry;        // entry of a NULL key (translated, or the NULL itself: then usually out of range)
  uint32_t entry_count;
  uint32_t rep;              // replicas of the LDS table (power of two)
  int32_t wpe;
  int32_t ntargets;
  S2Target t[kS2MaxTargets];
  int32_t wop[kMaxWordsPerEntry];
  uint32_t nword_mask;
  int32_t npq;
  S2PayQual pq[kS2MaxPayQuals];
  int64_t* slabs;
  int32_t* error_code;
  // two scatter levels: s.slice / s.nbins describe the COARSE bins; fine slice f of coarse bin c is slice c * fpc + f
  int32_t two_level;
  uint32_t fpc;              // fine slices per coarse bin (<= 256)
  uint32_t fslice;           // keys per fine slice (one level: = s.slice)
  uint32_t fmagic, fshift;   // (key offset inside the coarse bin) / fslice
  uint32_t nslices;          // fine slices in all (one level: = s.nbins)
  uint64_t cap2;             // tuples a fine slab holds (multiple of 16)
  int64_t* tuples2;          // [nslices][cap2]
  uint32_t* fill2;           // [nslices] x kSliceCursorStride
  uint32_t nsl_par;          // pass 2: slices worked on at a time (blocks = nsl_par x members)
  // payload words of the fused entry ([row id | p0 | p1]): one or two int32 arrays per slice in LDS
  int32_t npay;              // 1 or 2 (0 reads as 1: a join without payload columns keeps one dummy array)
  int32_t key_pidx;          // what the group key is computed from: payload word 0 / 1, or 2: the outer column y of the tuple
  int32_t pay_nullable1;     // NULL rule of payload word 1 (word 0: s.pay_nullable / s.pay_null)
  // two payload words PACKED into one LDS word when their statistics leave room (a slice holds twice the keys: star joins
  // that filter on one dimension column and group by another keep ONE scatter level up to 10 M keys):
  //   word = c0 | c1 << pk_bits0;  c0: 0 no partner, 1 NULL, else p0 - pk_min0 + 2;  c1: 0 NULL, else p1 - pk_min1 + 1
  int32_t packed;
  int64_t pay_null1;
  uint32_t pk_bits0, pk_codes0, pk_codes1, pad_pk_;
  int64_t pk_min0, pk_min1;
};

// signed 32-bit / positive invariant divisor, truncating like C (eval_expr's `a / b`)
HDK_DEV int32_t s2_div(int32_t v, uint32_t d, uint32_t magic, uint32_t shift) {
  if (d == 1u) {
    return v;
  }
  const uint32_t n = static_cast<uint32_t>(v < 0 ? -v : v);
  const uint32_t t = __umulhi(magic, n);
  const uint32_t q = (((n - t) >> 1) + t) >> shift;
  return v < 0 ? -static_cast<int32_t>(q) : static_cast<int32_t>(q);
}

// the argument of one target; 32-bit inputs, 64-bit result (|result| < 2^62: no overflow, no sentinel collision)
HDK_DEV int64_t s2_value(const S2Target& tg, int32_t x, int32_t p) {
  const int64_t X = x, P = p;
  switch (tg.src) {
    case S2_X: return X;
    case S2_P: return P;
    case S2_X_ADD_P: return X + P;
    case S2_X_SUB_P: return X - P;
    case S2_P_SUB_X: return P - X;
    case S2_X_MUL_P: return X * P;
    default: {
      const int64_t a = tg.src == S2_X_OP_LIT ? X : P;
      return tg.op == HDK_OP_ADD ? a + tg.lit : (tg.op == HDK_OP_SUB ? a - tg.lit : a * tg.lit);
    }
  }
}

// ---- pass 1b: one coarse bin -> its fine slices ------------------------------------------------------------------------
// Blocks b with b % 8 == c % 8 work on coarse bin c (all of a bin's writers behind ONE L2: the partial 128-byte lines at
// the ends of their runs merge there before they reach memory; affinity for speed only, any placement is correct); the
// bin's eight per-XCD sub-slabs are walked as one sequence of tiles.  Per tile: LDS histogram by fine slice, one cursor
// claim per slice, staging ordered by slice, copy-out -- the machinery of pass 1.  A tuple that does not fit its fine slab
// goes to the shared overflow area (probed against the table in memory by pass 2).
// dynamic LDS: int64 staging[kTile] | uint8 fine slice of every staging slot [kTile]
__global__ __launch_bounds__(kSliceBlock) void hdk_join_scatter_level2(Slice2Args g) {
  constexpr int VR = 8;
  constexpr int kTile = kSliceBlock * VR;
  const SliceArgs& a = g.s;
  __shared__ uint32_t s_cnt[kSliceMaxBins];
  __shared__ uint4 s_run[kSliceMaxBins];  // .x start in the staging area, .y tuples that fit the fine slab, .z slab position, .w overflow position
  __shared__ uint32_t s_total, s_stop;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + kTile);
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_stop = __hip_atomic_load(a.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int i = tid; i < kSliceMaxBins; i += kSliceBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  if (s_stop) {
    return;  // clustered input / stale statistics / interrupted: nothing to refine
  }
  // block -> (coarse bin, member): bins c = x + 8 j live on XCD x
  const uint32_t ncx = (a.nbins + kSliceXcds - 1) / kSliceXcds;  // coarse bins per XCD (padded)
  const uint32_t xcd = blockIdx.x % kSliceXcds, qq = blockIdx.x / kSliceXcds;
  const uint32_t c = xcd + kSliceXcds * (qq % ncx), member = qq / ncx, members = gridDim.x / (kSliceXcds * ncx);
  if (c >= a.nbins || member >= members) {
    return;
  }
  const uint32_t cfirst = c * a.slice;  // key offset of the coarse bin's first key
  const Watch watch = watch_begin(a.kp);
  int64_t tile = member;
  int64_t part_tile_begin = 0;
#pragma unroll 1
  for (int xq = 0; xq < kSliceXcds; ++xq) {
    const size_t sidx = static_cast<size_t>(c) * kSliceXcds + xq;
    const int64_t n = static_cast<int64_t>(min(static_cast<uint64_t>(a.fill[sidx * kSliceCursorStride]), a.sub));
    const int64_t ntiles = (n + kTile - 1) / kTile;
    const int8_t* in = reinterpret_cast<const int8_t*>(a.tuples + sidx * a.sub);
#pragma unroll 1
    for (; tile < part_tile_begin + ntiles; tile += members) {
      if (tid == 0) {  // block-uniform exit (the batch below has barriers)
        uint32_t stop = __hip_atomic_load(a.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (watch.flags) {
          if (const int32_t w_ = watch_poll(watch)) {
            record_error(a.kp.error_code, w_);
            atomicMax(a.mode, 2u);
            stop = 2;
          }
        }
        s_stop = stop;
      }
      __syncthreads();
      if (s_stop) {
        return;
      }
      const int64_t t0 = (tile - part_tile_begin) * kTile;
      int64_t w[VR];
      bool live[VR];
      if (t0 + kTile <= n) {  // full tile: tuples dealt in adjacent pairs, 16-byte non-temporal loads (sub-slabs are 16-byte aligned)
#pragma unroll
        for (int u = 0; u < VR / 2; ++u) {
          const bf_i64x2 ww = gload<bf_i64x2>(in, (t0 >> 1) + static_cast<int64_t>(u) * kSliceBlock + tid, true);
          w[2 * u] = ww.x;
          w[2 * u + 1] = ww.y;
          live[2 * u] = true;
          live[2 * u + 1] = true;
        }
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const int64_t i = t0 + static_cast<int64_t>(r) * kSliceBlock + tid;
          live[r] = i < n;
          w[r] = live[r] ? gload<int64_t>(in, i, true) : 0;
        }
      }
      uint32_t bin[VR], rank[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint32_t d = slice_key_of(a, static_cast<uint32_t>(w[r])) - cfirst;  // offset inside the coarse bin (< s.slice)
        const uint32_t t = __umulhi(g.fmagic, d);
        bin[r] = live[r] ? ((((d - t) >> 1) + t) >> g.fshift) : 0u;
        if (bin[r] >= g.fpc) {
          bin[r] = g.fpc - 1;  // (cannot happen for a tuple pass 1 put into this bin)
        }
        rank[r] = 0;
        if (live[r]) {
          rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
        }
      }
      __syncthreads();
      if (tid < kSliceMaxBins) {
        const uint32_t nn = s_cnt[tid];
        uint32_t base = 0, nfit = 0, obase = 0;
        if (nn) {
          base = atomicAdd(g.fill2 + (static_cast<size_t>(c) * g.fpc + tid) * kSliceCursorStride, nn);
          nfit = static_cast<uint64_t>(base) >= g.cap2 ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(nn), g.cap2 - base));
          if (nfit < nn) {
            obase = atomicAdd(a.fill_ovf, nn - nfit);
            if (static_cast<uint64_t>(obase) + (nn - nfit) > a.cap_ovf) {
              atomicMax(a.mode, 1u);  // too skewed for slabs: the interpreter takes over
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
          const uint32_t nn = s_cnt[c0 + tid];
          uint32_t incl = nn;
#pragma unroll
          for (int dd = 1; dd < kWave; dd <<= 1) {
            const uint32_t v = __shfl_up(incl, dd, kWave);
            if (tid >= dd) {
              incl += v;
            }
          }
          s_run[c0 + tid].x = carry + incl - nn;
          carry += __shfl(incl, kWave - 1, kWave);
        }
        if (tid == 0) {
          s_total = carry;
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        if (live[r]) {
          const uint32_t si = s_run[bin[r]].x + rank[r];
          s_binof[si] = static_cast<uint8_t>(bin[r]);
          s_stage[si] = w[r];
        }
      }
      if (tid < kSliceMaxBins) {
        s_cnt[tid] = 0;
      }
      __syncthreads();
      const uint32_t total = s_total;
      for (uint32_t i = tid; i < total; i += kSliceBlock) {
        const uint32_t b = s_binof[i];
        const uint4 run = s_run[b];
        const uint32_t r = i - run.x;
        if (r < run.y) {
          g.tuples2[(static_cast<uint64_t>(c) * g.fpc + b) * g.cap2 + run.z + r] = s_stage[i];
        } else {
          const uint64_t o = static_cast<uint64_t>(run.w) + (r - run.y);
          if (o < a.cap_ovf) {  // (else mode is 1: the launch is redone in row order)
            a.tuples[static_cast<uint64_t>(a.nbins) * kSliceXcds * a.sub + o] = s_stage[i];
          }
        }
      }
      __syncthreads();
    }
    part_tile_begin += ntiles;
  }
}

// dynamic LDS: [entry_count x wpe x rep] aggregate words | int32 payload[slice]
#ifndef HDK_S2_K_GROUPED
#define HDK_S2_K_GROUPED 6  // joined rows per batch of the grouped / the non-grouped form: the scalar dispatch amortised over six rows; eight spill at the 128 registers a 1 024-thread block leaves (two payload words: 164 bytes per lane, c3f 3.47 ms against 3.04)
#endif
#ifndef HDK_S2_K_PLAIN
#define HDK_S2_K_PLAIN 6  // (8 spills 52-112 bytes per lane at the 128 registers a 1 024-thread block leaves; 6 does not and measures the same or better)
#endif

// NPAY: payload words per key in LDS -- 1, 2, or 3: two words packed into one (Slice2Args::packed); HASY: the tuples carry a
// second outer column above the key offset (SliceArgs::y_shift)
template <bool GROUPED, int NPAY, bool HASY>
__global__ __launch_bounds__(kSliceAggBlock) void hdk_join_agg_sliced2(Slice2Args g) {
  constexpr bool PACKED = NPAY == 3;
  constexpr bool TWO = NPAY >= 2;      // the row function sees two payload words
  constexpr int NW = TWO ? 2 : 1;      // payload quads of a fused entry
  extern __shared__ __attribute__((aligned(16))) int64_t lds[];
  const SliceArgs& a = g.s;
  const int tid = threadIdx.x;
  const int wpe = g.wpe;
  const uint32_t rep = g.rep;
  const uint32_t ew = g.entry_count * static_cast<uint32_t>(wpe);
  const uint32_t my_rep = tid & (rep - 1);
  int32_t* s_pay = reinterpret_cast<int32_t*>(lds + static_cast<size_t>(ew) * rep);
  int32_t* s_pay1 = s_pay + g.fslice;  // (NPAY == 2)
  const uint32_t fstride = 1u + static_cast<uint32_t>(NW);  // quads of a fused entry
  const uint32_t kmask = HASY ? (1u << a.y_shift) - 1u : ~0u;
  const int32_t y_bias = static_cast<int32_t>(a.y_min - 1);  // y = code + y_bias (code 0: NULL)
  const uint32_t pk_mask0 = PACKED ? (1u << g.pk_bits0) - 1u : 0u;
  const int32_t pk_bias0 = static_cast<int32_t>(g.pk_min0 - 2), pk_bias1 = static_cast<int32_t>(g.pk_min1 - 1);
  __shared__ uint32_t s_off;
  if (tid == 0) {  // (block-uniform, through LDS: another block of this kernel may raise the word meanwhile)
    s_off = __hip_atomic_load(a.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const bool off = s_off != 0;
  // slabs beyond this grid (hdk_finalize folds the launch shape's count) hold identities
  for (uint32_t sidx = blockIdx.x + (off ? 0u : gridDim.x); sidx < a.num_slabs; sidx += gridDim.x) {
    for (uint32_t i = tid; i < ew; i += kSliceAggBlock) {
      const uint32_t w = i % wpe;
      g.slabs[static_cast<size_t>(sidx) * ew + i] = (g.nword_mask >> w) & 1u ? 0 : word_identity(g.wop[w]);
    }
  }
  if (off) {
    return;  // the armed interpreter writes the real slabs
  }
  for (uint32_t i = tid; i < ew * rep; i += kSliceAggBlock) {
    lds[i] = word_identity(g.wop[(i / rep) % wpe]);
  }
  const int64_t* __restrict__ table = a.kp.join_hash_tables;  // fused: [row id | payload] per key
  const uint32_t member = blockIdx.x / g.nsl_par, members = gridDim.x / g.nsl_par;
  uint32_t first = 0, nkeys = 0;  // the slice at hand: key offsets [first, first + nkeys)
  // non-grouped: per-lane accumulators, one LDS update per lane at the end
  uint64_t r_rows = 0;
  int64_t r_val[kS2MaxTargets];
  uint64_t r_nulls[kS2MaxTargets];
#pragma unroll
  for (int t = 0; t < kS2MaxTargets; ++t) {
    r_val[t] = (t < g.ntargets && g.t[t].vword >= 0) ? word_identity(g.t[t].wop) : 0;
    r_nulls[t] = 0;
  }
  int32_t err = 0;
  const int nt = g.ntargets;
  const int npq = g.npq;
  const bool x_null32 = a.x_null32 != 0;
  // K joined rows at a time: everything that is the same for every row of the launch -- which comparison a filter makes,
  // what a target computes, how its word combines -- is decided ONCE per batch by scalar code, the K rows then run through
  // straight vector code.  (One row at a time the scalar dispatch was the bottleneck: ~70 scalar instructions per tuple.)
  constexpr int K = GROUPED ? HDK_S2_K_GROUPED : HDK_S2_K_PLAIN;
  // ok[k]: slot k holds a tuple with a partner; x32 / p32 its outer value and payload (kSliceNull: the column's NULL)
  // (with_quals: the filters on the joined columns are still to be applied -- tuples of the overflow area; a slice's tuples find
  // them applied to the payloads in LDS: a key whose payloads fail reads as a key without a partner)
  auto batch = [&](const int32_t (&x32)[K], const int32_t (&p0)[K], const int32_t (&p1)[K], const int32_t (&y32)[K], bool (&ok)[K],
                   const bool with_quals) {
    bool pn0[K], pn1[K], yn[K], xnull[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pn0[k] = p0[k] == kSliceNull;
      pn1[k] = TWO && p1[k] == kSliceNull;
      yn[k] = HASY && y32[k] == kSliceNull;
      xnull[k] = x_null32 && x32[k] == INT32_MIN;
    }
    // the word a filter / the key / a target reads: payload word 0 / 1, or 2: the tuple's y (wave-uniform choice)
    auto pick = [&](int pidx, int32_t (&pp)[K], bool (&pn)[K]) {
      const bool is_y = HASY && pidx == 2, is_1 = TWO && pidx == 1;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        pp[k] = is_y ? y32[k] : (is_1 ? p1[k] : p0[k]);
        pn[k] = is_y ? yn[k] : (is_1 ? pn1[k] : pn0[k]);
      }
    };
    // filters on the joined column: a NULL fails every comparison (DEF_CMP_NULLABLE)
    for (int q = 0; q < (with_quals ? npq : 0); ++q) {
      const int64_t rhs = g.pq[q].rhs;
      int32_t p32[K];
      bool pnull[K];
      pick(g.pq[q].pidx, p32, pnull);
#define HDK_S2_CMP(OP)                                                                \
  _Pragma("unroll") for (int k = 0; k < K; ++k) {                                     \
    ok[k] = ok[k] && !pnull[k] && (static_cast<int64_t>(p32[k]) OP rhs);              \
  }
      switch (g.pq[q].cmp) {
        case HDK_CMP_EQ: HDK_S2_CMP(==) break;
        case HDK_CMP_NE: HDK_S2_CMP(!=) break;
        case HDK_CMP_LT: HDK_S2_CMP(<) break;
        case HDK_CMP_GT: HDK_S2_CMP(>) break;
        case HDK_CMP_LE: HDK_S2_CMP(<=) break;
        default: HDK_S2_CMP(>=) break;
      }
#undef HDK_S2_CMP
    }
    uint32_t base[K];
    if (GROUPED) {
      const uint32_t kd = static_cast<uint32_t>(g.key_div);
      const uint32_t stride = static_cast<uint32_t>(wpe) * rep;
      int32_t p32[K];
      bool pnull[K];
      pick(g.key_pidx, p32, pnull);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int32_t quot = kd ? s2_div(p32[k], kd, g.div_magic, g.div_shift) : p32[k];
        const int32_t keyv = g.key_mod ? p32[k] - quot * static_cast<int32_t>(kd) : quot;
        const int64_t entry = pnull[k] ? g.null_entry : static_cast<int64_t>(keyv) - g.key_min;
        if (ok[k] && static_cast<uint64_t>(entry) >= g.entry_count) {
          err = HDK_HIP_ERR_OUT_OF_SLOTS;  // key outside the range the layout was sized for
          ok[k] = false;
        }
        base[k] = (ok[k] ? static_cast<uint32_t>(entry) : 0u) * stride + my_rep;
        if (ok[k]) {
          atomicAdd(reinterpret_cast<unsigned long long*>(lds + base[k]), 1ull);  // row count
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        base[k] = my_rep;
        r_rows += ok[k] ? 1u : 0u;
      }
    }
#pragma unroll
    for (int t = 0; t < kS2MaxTargets; ++t) {
      if (t < nt && g.t[t].src != S2_NONE) {
        const S2Target& tg = g.t[t];
        const bool nx = tg.null_if_x != 0, np = tg.null_if_p != 0;
        const int64_t lit = tg.lit;
        int64_t v[K];
        int32_t p32[K];
        bool pnull[K];
        pick(tg.pidx, p32, pnull);
#define HDK_S2_VAL(EXPR)                                  \
  _Pragma("unroll") for (int k = 0; k < K; ++k) {         \
    const int64_t X = x32[k], P = p32[k];                 \
    (void)X;                                              \
    (void)P;                                              \
    v[k] = (EXPR);                                        \
  }
        switch (tg.src) {
          case S2_X: HDK_S2_VAL(X) break;
          case S2_P: HDK_S2_VAL(P) break;
          case S2_X_ADD_P: HDK_S2_VAL(X + P) break;
          case S2_X_SUB_P: HDK_S2_VAL(X - P) break;
          case S2_P_SUB_X: HDK_S2_VAL(P - X) break;
          case S2_X_MUL_P: HDK_S2_VAL(X * P) break;
          case S2_X_OP_LIT:
            if (tg.op == HDK_OP_ADD) { HDK_S2_VAL(X + lit) } else if (tg.op == HDK_OP_SUB) { HDK_S2_VAL(X - lit) } else { HDK_S2_VAL(X * lit) }
            break;
          default:
            if (tg.op == HDK_OP_ADD) { HDK_S2_VAL(P + lit) } else if (tg.op == HDK_OP_SUB) { HDK_S2_VAL(P - lit) } else { HDK_S2_VAL(P * lit) }
            break;
        }
#undef HDK_S2_VAL
        bool isn[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          isn[k] = (nx && xnull[k]) || (np && pnull[k]);
        }
        const int32_t wop = tg.wop;
        if (GROUPED) {
          const uint32_t voff = static_cast<uint32_t>(tg.vword) * rep, noff = static_cast<uint32_t>(tg.nword) * rep;
          if (tg.nword >= 0) {  // counts NULLs; the flush turns it into the non-null count
#pragma unroll
            for (int k = 0; k < K; ++k) {
              if (ok[k] && isn[k]) {
                atomicAdd(reinterpret_cast<unsigned long long*>(lds + base[k] + noff), 1ull);
              }
            }
          }
          if (tg.vword >= 0) {
            if (wop == WOP_ADD_U64) {
#pragma unroll
              for (int k = 0; k < K; ++k) {
                if (ok[k] && !isn[k]) atomicAdd(reinterpret_cast<unsigned long long*>(lds + base[k] + voff), static_cast<unsigned long long>(v[k]));
              }
            } else if (wop == WOP_MIN_I64) {
#pragma unroll
              for (int k = 0; k < K; ++k) {
                if (ok[k] && !isn[k]) atomicMin(reinterpret_cast<long long*>(lds + base[k] + voff), static_cast<long long>(v[k]));
              }
            } else {
#pragma unroll
              for (int k = 0; k < K; ++k) {
                if (ok[k] && !isn[k]) atomicMax(reinterpret_cast<long long*>(lds + base[k] + voff), static_cast<long long>(v[k]));
              }
            }
          }
        } else {
          int64_t acc = r_val[t];
          uint64_t nulls = r_nulls[t];
          const bool has_v = tg.vword >= 0;
          if (wop == WOP_ADD_U64) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const bool use = ok[k] && !isn[k] && has_v;
              acc = static_cast<int64_t>(static_cast<uint64_t>(acc) + (use ? static_cast<uint64_t>(v[k]) : 0ull));
              nulls += (ok[k] && isn[k]) ? 1u : 0u;
            }
          } else if (wop == WOP_MIN_I64) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const bool use = ok[k] && !isn[k] && has_v;
              acc = (use && v[k] < acc) ? v[k] : acc;
              nulls += (ok[k] && isn[k]) ? 1u : 0u;
            }
          } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const bool use = ok[k] && !isn[k] && has_v;
              acc = (use && v[k] > acc) ? v[k] : acc;
              nulls += (ok[k] && isn[k]) ? 1u : 0u;
            }
          }
          r_val[t] = acc;
          r_nulls[t] = nulls;
        }
      }
    }
  };
  // K tuples of this slice: probe the payloads in LDS, then the batch
  auto tuples = [&](const int64_t (&w)[K], const bool (&live)[K]) {
    int32_t x32[K], p0[K], p1[K], y32[K];
    bool ok[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t lo = static_cast<uint32_t>(w[k]);
      const uint32_t local = (lo & kmask) - first;
      const bool in = live[k] && local < nkeys;
      if (PACKED) {
        const uint32_t pw = in ? static_cast<uint32_t>(s_pay[local]) : 0u;
        const uint32_t c0 = pw & pk_mask0, c1 = pw >> g.pk_bits0;
        p0[k] = c0 == 0 ? kSliceNoMatch : (c0 == 1 ? kSliceNull : static_cast<int32_t>(c0) + pk_bias0);
        p1[k] = c1 == 0 ? kSliceNull : static_cast<int32_t>(c1) + pk_bias1;
      } else {
        p0[k] = in ? s_pay[local] : kSliceNoMatch;
        p1[k] = (TWO && in) ? s_pay1[local] : 0;
      }
      const uint32_t yc = HASY ? lo >> a.y_shift : 0u;
      y32[k] = yc == 0 ? kSliceNull : static_cast<int32_t>(yc) + y_bias;
      x32[k] = static_cast<int32_t>(static_cast<uint64_t>(w[k]) >> 32);
      ok[k] = p0[k] != kSliceNoMatch;  // (INNER join: no partner, no row)
    }
    batch(x32, p0, p1, y32, ok, false);
  };
  // the block's slices (one level: exactly one; two levels: slice, slice + nsl_par, ...), each: payloads into LDS, then its
  // tuples -- 16-byte words w, w + stride, ...; two words (four tuples) per trip, the next two in flight
#pragma unroll 1
  for (uint32_t sl = blockIdx.x % g.nsl_par; sl < g.nslices; sl += g.nsl_par) {
    uint32_t coarse = sl, fine = 0;
    if (g.two_level) {
      coarse = sl / g.fpc;
      fine = sl - coarse * g.fpc;
    }
    first = coarse * a.slice + fine * g.fslice;
    const uint64_t coarse_end = min(static_cast<uint64_t>(coarse + 1) * a.slice, a.key_range);
    nkeys = first < coarse_end ? static_cast<uint32_t>(min(static_cast<uint64_t>(g.fslice), coarse_end - first)) : 0u;
    __syncthreads();  // (the previous slice's probes are done)
    bool bad = false;
    // the filters on the joined columns, once per KEY (10 M) instead of once per tuple (1 B): INNER join + WHERE on the inner
    // side = filter the inner side first; a NULL fails every comparison (DEF_CMP_NULLABLE)
    auto pay_pass = [&](int64_t y, bool ynull, int64_t z, bool znull) {
      bool pass = true;
      for (int q = 0; q < npq; ++q) {
        const bool second = TWO && g.pq[q].pidx == 1;
        const int64_t v = second ? z : y, rhs = g.pq[q].rhs;
        const bool vnull = second ? znull : ynull;
        bool c;
        switch (g.pq[q].cmp) {
          case HDK_CMP_EQ: c = v == rhs; break;
          case HDK_CMP_NE: c = v != rhs; break;
          case HDK_CMP_LT: c = v < rhs; break;
          case HDK_CMP_GT: c = v > rhs; break;
          case HDK_CMP_LE: c = v <= rhs; break;
          default: c = v >= rhs; break;
        }
        pass = pass && !vnull && c;
      }
      return pass;
    };
    for (uint32_t i = tid; i < nkeys; i += kSliceAggBlock) {
      const int64_t* e = table + static_cast<uint64_t>(first + i) * fstride;
      const int64_t rid = __builtin_nontemporal_load(e);
      int32_t p32 = kSliceNoMatch, q32 = 0;
      if (PACKED) {
        uint32_t c0 = 0, c1 = 0;
        if (rid >= 0) {
          const int64_t y = __builtin_nontemporal_load(e + 1);
          const int64_t z = __builtin_nontemporal_load(e + 2);
          c0 = 1;
          const bool yn = a.pay_nullable && y == a.pay_null, zn = g.pay_nullable1 && z == g.pay_null1;
          if (!(a.pay_nullable && y == a.pay_null)) {
            const uint64_t d0 = static_cast<uint64_t>(y) - static_cast<uint64_t>(g.pk_min0);
            bad |= d0 >= g.pk_codes0 - 2u;  // a payload outside its statistics
            c0 = static_cast<uint32_t>(d0) + 2u;
          }
          if (!(g.pay_nullable1 && z == g.pay_null1)) {
            const uint64_t d1 = static_cast<uint64_t>(z) - static_cast<uint64_t>(g.pk_min1);
            bad |= d1 >= g.pk_codes1 - 1u;
            c1 = static_cast<uint32_t>(d1) + 1u;
          }
          if (npq && !pay_pass(y, yn, z, zn)) {
            c0 = 0;
            c1 = 0;
          }
        }
        p32 = static_cast<int32_t>(c0 | (c1 << g.pk_bits0));
      } else if (rid >= 0) {
        const int64_t y = __builtin_nontemporal_load(e + 1);
        const bool yn = a.pay_nullable && y == a.pay_null;
        if (yn) {
          p32 = kSliceNull;
        } else {
          p32 = static_cast<int32_t>(y);
          bad |= static_cast<int64_t>(p32) != y || p32 == kSliceNoMatch || p32 == kSliceNull;
        }
        int64_t z = 0;
        bool zn = false;
        if (NPAY == 2) {
          z = __builtin_nontemporal_load(e + 2);
          zn = g.pay_nullable1 && z == g.pay_null1;
          if (zn) {
            q32 = kSliceNull;
          } else {
            q32 = static_cast<int32_t>(z);
            bad |= static_cast<int64_t>(q32) != z || q32 == kSliceNull;
          }
        }
        if (npq && !pay_pass(y, yn, z, zn)) {
          p32 = kSliceNoMatch;
        }
      }
      s_pay[i] = p32;
      if (NPAY == 2) {
        s_pay1[i] = q32;
      }
    }
    if (__any(bad) && (tid & (kWave - 1)) == 0) {
      atomicMax(a.mode, 1u);  // a payload outside what the statistics announced: redone in row order
    }
    __syncthreads();
    const int nparts = g.two_level ? 1 : kSliceXcds;  // one level: the slice's eight per-XCD sub-slabs
#pragma unroll 1
    for (int xq = 0; xq < nparts; ++xq) {
      uint64_t n;
      const int8_t* in;
      if (g.two_level) {
        n = min(static_cast<uint64_t>(g.fill2[static_cast<size_t>(sl) * kSliceCursorStride]), g.cap2);
        in = reinterpret_cast<const int8_t*>(g.tuples2 + static_cast<size_t>(sl) * g.cap2);
      } else {
        const size_t sidx = static_cast<size_t>(sl) * kSliceXcds + xq;
        n = min(static_cast<uint64_t>(a.fill[sidx * kSliceCursorStride]), a.sub);
        in = reinterpret_cast<const int8_t*>(a.tuples + sidx * a.sub);
      }
      const uint64_t npairs = (n + 1) / 2;
      const uint64_t stride = static_cast<uint64_t>(members) * kSliceAggBlock;
      constexpr int W = K / 2;  // 16-byte words per trip
      bf_i64x2 nx[W];
      uint64_t pw = static_cast<uint64_t>(member) * kSliceAggBlock + tid;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        nx[j].x = nx[j].y = 0;
        if (pw + j * stride < npairs) {
          nx[j] = gload<bf_i64x2>(in, static_cast<int64_t>(pw + j * stride), true);
        }
      }
#pragma unroll 1
      for (; pw < npairs; pw += W * stride) {
        int64_t w[K];
        bool live[K];
#pragma unroll
        for (int j = 0; j < W; ++j) {
          w[2 * j] = nx[j].x;
          w[2 * j + 1] = nx[j].y;
          live[2 * j] = pw + j * stride < npairs;
          live[2 * j + 1] = live[2 * j] && 2 * (pw + j * stride) + 1 < n;
          if (pw + (W + j) * stride < npairs) {
            nx[j] = gload<bf_i64x2>(in, static_cast<int64_t>(pw + (W + j) * stride), true);
          }
        }
        tuples(w, live);
      }
    }
  }
  // the overflow area (tuples of any slice, skewed keys): probed against the table in memory by all blocks together --
  // a block's group table is private, so any block can take any tuple
  {
    const uint64_t n = min(static_cast<uint64_t>(*a.fill_ovf), a.cap_ovf);
    const int64_t* in = a.tuples + static_cast<size_t>(a.nbins) * kSliceXcds * a.sub;
    bool bad2 = false;
    const uint64_t step = static_cast<uint64_t>(gridDim.x) * kSliceAggBlock;
#pragma unroll 1
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kSliceAggBlock + tid; i < n; i += K * step) {
      int32_t x32[K], p0[K], p1[K], y32[K];
      bool ok[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint64_t ik = i + static_cast<uint64_t>(k) * step;
        const bool live = ik < n;
        const int64_t w = live ? in[ik] : 0;
        int64_t rid = -1, y = 0, z = 0;
        if (live) {
          const int64_t* e = table + static_cast<uint64_t>(static_cast<uint32_t>(w) & kmask) * fstride;
          rid = e[0];
          y = e[1];
          if (TWO) {
            z = e[2];
          }
        }
        const uint32_t yc = HASY ? static_cast<uint32_t>(w) >> a.y_shift : 0u;
        y32[k] = yc == 0 ? kSliceNull : static_cast<int32_t>(yc) + y_bias;
        ok[k] = rid >= 0;
        x32[k] = static_cast<int32_t>(static_cast<uint64_t>(w) >> 32);
        p0[k] = static_cast<int32_t>(y);
        p1[k] = static_cast<int32_t>(z);
        if (ok[k]) {
          if (a.pay_nullable && y == a.pay_null) {
            p0[k] = kSliceNull;
          } else {
            bad2 |= static_cast<int64_t>(p0[k]) != y || p0[k] == kSliceNoMatch || p0[k] == kSliceNull;
          }
          if (TWO) {
            if (g.pay_nullable1 && z == g.pay_null1) {
              p1[k] = kSliceNull;
            } else {
              bad2 |= static_cast<int64_t>(p1[k]) != z || p1[k] == kSliceNull;
            }
          }
        }
      }
      batch(x32, p0, p1, y32, ok, true);
    }
    if (__any(bad2) && (tid & (kWave - 1)) == 0) {
      atomicMax(a.mode, 1u);
    }
  }
  if (err) {
    record_error(g.error_code, err);
  }
  if (!GROUPED && r_rows) {
    atomicAdd(reinterpret_cast<unsigned long long*>(lds + my_rep), static_cast<unsigned long long>(r_rows));
#pragma unroll
    for (int t = 0; t < kS2MaxTargets; ++t) {
      if (t < nt && g.t[t].src != S2_NONE) {
        if (g.t[t].vword >= 0) {
          vec_lds_op(g.t[t].wop, lds + static_cast<uint32_t>(g.t[t].vword) * rep + my_rep, r_val[t]);
        }
        if (g.t[t].nword >= 0 && r_nulls[t]) {
          atomicAdd(reinterpret_cast<unsigned long long*>(lds + static_cast<uint32_t>(g.t[t].nword) * rep + my_rep),
                    static_cast<unsigned long long>(r_nulls[t]));
        }
      }
    }
  }
  __syncthreads();
  int64_t* slab = g.slabs + static_cast<size_t>(blockIdx.x) * ew;
  for (uint32_t i = tid; i < ew; i += kSliceAggBlock) {
    const uint32_t w = i % wpe;
    const int32_t op = g.wop[w];
    int64_t v = lds[i * rep];
    for (uint32_t r = 1; r < rep; ++r) {
      v = word_combine(op, v, lds[i * rep + r]);
    }
    if ((g.nword_mask >> w) & 1u) {  // NULL count -> non-null count = rows of the entry - NULLs
      const uint32_t w0 = (i / wpe) * wpe;
      int64_t rows = 0;
      for (uint32_t r = 0; r < rep; ++r) {
        rows += lds[w0 * rep + r];
      }
      v = rows - v;
    }
    slab[i] = v;
  }
}

}  // namespace hdk
