// scan_bhm_part.h -- the multi-argument / multi-key group-by (scan_bhm.h) for tables BEYOND a CU's LDS: two passes.
//
// A dense table of 10 K - 500 K groups (MultiStep/MSBS002-003, MSPHS002-003, PerfectHashMultiCol/PHM003-005, BaselineHash/
// BH005) is cut into bins of 2^w consecutive entries, each small enough for one block's LDS:
//   pass A  hdk_bhm_scatter     every row becomes ONE 4-byte tuple
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

// rows per lane and batch of pass A: 16 over 4-byte columns (8 192 rows a batch), 8 over 8-byte ones (their loads take twice the registers)
constexpr int bhm_part_vr(int w) { return w == 8 ? 8 : 16; }
constexpr size_t bhm_scatter_lds(int w) { return static_cast<size_t>(kPbBlock) * bhm_part_vr(w) * 5 + 16; }  // uint32 staging | uint8 bin of every slot
constexpr int kBhmAggBlock = 256;

struct BhmPartArgs {
  BhmArgs b;               // columns, statistics, argument descriptors; the LDS geometry of pass B (entries = 2^w)
  uint32_t w;              // bits of a tuple's entry inside its bin
  uint32_t nbins;          // <= 256
  uint32_t total_entries;  // the dense table
  uint32_t pad_;
  uint32_t cshift[kBhmMaxSrc];  // position of argument column s's code in the tuple
  uint32_t cmask[kBhmMaxSrc];
  uint64_t cap;            // tuples of a (bin, XCD) sub-slab (a multiple of 4)
  uint32_t* tuples;        // [nbins][kPbXcds][cap]
  uint32_t* fill;          // [nbins][kPbXcds] x kPbCursorStride
};

// ---- pass A ---------------------------------------------------------------------------------------------------------------------
template <int NK, int NS, int W = 4, bool Q = false>
__global__ __launch_bounds__(kPbBlock) void hdk_bhm_scatter(BhmPartArgs g) {
  constexpr int VR = bhm_part_vr(W);
  constexpr int kBhmPartTile = kPbBlock * VR;
  constexpr int R = 16 / W, U = VR / R;
  const BhmArgs& a = g.b;
  __shared__ uint32_t s_cnt[kPbMaxBins];
  __shared__ uint4 s_run[kPbMaxBins];
  __shared__ uint32_t s_total;
  __shared__ int32_t s_watch;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn32[];
  uint32_t* s_stage = s_dyn32;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn32 + kBhmPartTile);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kPbXcds - 1);
  for (int i = tid; i < kPbMaxBins; i += kPbBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  int32_t err = 0;
  uint32_t stale = 0;
  const Watch watch = watch_begin(a.kp);
  const bool nulls = a.any_nullable != 0;
  const uint32_t wmask = (1u << g.w) - 1u;
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
      uint32_t tup[VR], bin[VR];
      bool live[VR];
      uint32_t kr[U][NK][4], xr[U][NS][4], qr[Q ? U : 1][kMaxPlainQuals][4];
      const bool qvec = Q && full && a.qvec != 0;  // the filter's columns with the batch (scan_bhm.h); a ragged tile: by row number
      if (full) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = row0 + (static_cast<int64_t>(u) * kPbBlock + tid) * R;
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            load_bytes<16, true>(kcol[kk] + r * W, kr[u][kk]);
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            load_bytes<16, true>(xcol[s] + r * W, xr[u][s]);
          }
#pragma unroll
          for (int qi = 0; qi < kMaxPlainQuals; ++qi) {
            if (Q && qvec && qi < a.nquals) {
              load_bytes<16, true>((gcol_t)cols[a.q[qi].col.buf_idx] + r * W, qr[Q ? u : 0][qi]);
            }
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const int64_t r = row0 + (static_cast<int64_t>(u) * kPbBlock + tid) * R + i;
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
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int32_t kv[NK][R], xv[NS][R];
        uint32_t widem = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            bool wide;
            kv[kk][i] = bhm_narrow<W>(kr[u][kk], i, wide);
            widem |= wide ? 1u << i : 0u;
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            bool wide;
            xv[s][i] = bhm_narrow<W>(xr[u][s], i, wide);
            widem |= wide ? 1u << i : 0u;
          }
        }
        bool ok[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          ok[i] = full || row0 + (static_cast<int64_t>(u) * kPbBlock + tid) * R + i < nrows;
        }
        if (Q) {  // plain filters: rows that fail are not scattered
          int64_t rows[R];
#pragma unroll
          for (int i = 0; i < R; ++i) {
            rows[i] = ok[i] ? row0 + (static_cast<int64_t>(u) * kPbBlock + tid) * R + i : row0;
          }
          if (qvec) {
            plain_quals_pass_with<R, true>(
                a.q, a.nquals,
                [&](int qi, const ProjFastQual&, const bool (&)[R], int64_t (&v)[R]) {
#pragma unroll
                  for (int i = 0; i < R; ++i) {
                    const int64_t v0 = extract_elem<W>(qr[Q ? u : 0][0], i), v1 = extract_elem<W>(qr[Q ? u : 0][1], i), v2 = extract_elem<W>(qr[Q ? u : 0][2], i);
                    v[i] = qi == 0 ? v0 : (qi == 1 ? v1 : v2);
                  }
                },
                ok);
          } else {
            plain_quals_pass<R, true>(a.q, a.nquals, cols, rows, ok, true);
          }
        }
        uint32_t e[R];
        const uint32_t kbad = bhm_key_entries<NK, R>(a, nulls, 0xFFFFFFFFu, kv, e) | widem;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const int r = u * R + i;
          bool bad = ((kbad >> i) & 1u) != 0;
          uint32_t t = e[i] & wmask;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const BhmSrc& src = a.src[s];
            const int32_t raw = xv[s][i];
            const uint32_t d = static_cast<uint32_t>(raw) - static_cast<uint32_t>(src.raw_min);
            const bool isnull = nulls & (src.nullable != 0) & (raw == src.null32);
            bad = bad | (!isnull & (d > src.raw_span));
            t |= (isnull ? 0u : d + 1u) << g.cshift[s];
          }
          stale |= (ok[i] & bad) ? 1u : 0u;
          live[r] = ok[i] & !bad;
          bin[r] = live[r] ? e[i] >> g.w : 0u;
          tup[r] = t;
        }
      }
      // the batch: LDS histogram by bin, one cursor claim per bin and XCD, staging ordered by bin, copy-out in runs
      uint32_t rank[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        rank[r] = live[r] ? atomicAdd(&s_cnt[bin[r]], 1u) : 0u;
      }
      __syncthreads();
      if (tid < kPbMaxBins) {
        const uint32_t n = s_cnt[tid];
        uint32_t base = 0, nfit = 0;
        if (n) {
          base = atomicAdd(g.fill + (static_cast<size_t>(tid) * kPbXcds + xcd) * kPbCursorStride, n);
          const uint64_t room = base < g.cap ? g.cap - base : 0;
          nfit = n < room ? n : static_cast<uint32_t>(room);
          if (nfit < n) {
            stale = 1;  // the sub-slab is full (a hot key): the launch is redone by the armed fallback
          }
        }
        const uint64_t at = (static_cast<uint64_t>(tid) * kPbXcds + xcd) * g.cap + base;
        s_run[tid].y = nfit;
        s_run[tid].z = static_cast<uint32_t>(at);
        s_run[tid].w = static_cast<uint32_t>(at >> 32);
      }
      if (tid < kWave) {  // exclusive scan of the counts
        uint32_t carry = 0;
        for (int c0 = 0; c0 < kPbMaxBins; c0 += kWave) {
          const uint32_t n = s_cnt[c0 + tid];
          uint32_t incl = n;
#pragma unroll
          for (int dd = 1; dd < kWave; dd <<= 1) {
            const uint32_t v = __shfl_up(incl, dd, kWave);
            if (tid >= dd) {
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
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        if (live[r]) {
          const uint32_t si = s_run[bin[r]].x + rank[r];
          s_binof[si] = static_cast<uint8_t>(bin[r]);
          s_stage[si] = tup[r];
        }
      }
      if (tid < kPbMaxBins) {
        s_cnt[tid] = 0;
      }
      __syncthreads();
      const uint32_t total = s_total;
      for (uint32_t i = tid; i < total; i += kPbBlock) {
        const uint32_t b = s_binof[i];
        const uint4 run = s_run[b];
        const uint32_t r = i - run.x;
        if (r < run.y) {
          g.tuples[((static_cast<uint64_t>(run.w) << 32) | run.z) + r] = s_stage[i];
        }
      }
      __syncthreads();
    }
    frag_tile_begin += ntiles;
  }
  if (stale) {
    atomicOr(a.flag, 1u);
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

// ---- pass B: one block per (bin, XCD sub-slab) -----------------------------------------------------------------------------------
template <class C, int NS>
__global__ __launch_bounds__(kBhmAggBlock) void hdk_bhm_aggregate(BhmPartArgs g) {
  const BhmArgs& a = g.b;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
  const int tid = threadIdx.x;
  if (*a.flag) {
    return;  // (pass A already knows the launch is to be redone)
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
  const uint32_t bin = blockIdx.x / kPbXcds, x = blockIdx.x % kPbXcds;
  const size_t sub = static_cast<size_t>(bin) * kPbXcds + x;
  const uint32_t n = static_cast<uint32_t>(min(static_cast<uint64_t>(g.fill[sub * kPbCursorStride]), g.cap));
  const uint32_t* t = g.tuples + sub * g.cap;
  const uint32_t wmask = (1u << g.w) - 1u;
  const uint32_t dummy = a.entries;  // (= 2^w: the entry behind the bin's table)
  const bool nulls = C::nulls(a);
  constexpr int R = 4, U = 4;
  constexpr uint32_t kStep = kBhmAggBlock * R * U;
  typedef uint32_t __attribute__((ext_vector_type(4))) u32x4_t;
  for (uint32_t base = 0; base < n; base += kStep) {
    const bool full = base + kStep <= n;
    u32x4_t tv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + (static_cast<uint32_t>(u) * kBhmAggBlock + tid) * R;
      if (full || i + R <= n) {  // (sub-slabs start 16-byte aligned: cap is a multiple of 4)
        tv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(t + i));
      } else {
        tv[u].x = i < n ? t[i] : 0;
        tv[u].y = i + 1 < n ? t[i + 1] : 0;
        tv[u].z = i + 2 < n ? t[i + 2] : 0;
        tv[u].w = 0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + (static_cast<uint32_t>(u) * kBhmAggBlock + tid) * R;
      const uint32_t tw[R] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
      uint32_t e[R];
      int32_t xv[NS][R];
      bool lv[NS][R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const bool in = full | (i + j < n);
        e[j] = in ? tw[j] & wmask : dummy;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const uint32_t code = (tw[j] >> g.cshift[s]) & g.cmask[s];
          xv[s][j] = static_cast<int32_t>(code) + (a.src[s].raw_min - 1);
          lv[s][j] = nulls ? (in & (code != 0)) : true;
        }
      }
      bhm_update<C, NS, R>(a, rp, dummy, e, xv, lv);
    }
  }
  __syncthreads();
  // the block's part of slab x of the whole dense table
  const uint32_t e0 = bin << g.w;
  int64_t* slab = a.slabs + static_cast<size_t>(x) * g.total_entries * a.wpe;
  for (uint32_t ei = tid; ei < a.entries && e0 + ei < g.total_entries; ei += kBhmAggBlock) {
    bhm_slab_entry(a, lds8, ei, slab + static_cast<size_t>(e0 + ei) * a.wpe);
  }
}

}  // namespace hdk
