// scan_bh_dense_part.h -- group-by tables beyond LDS whose KEYS ARE DENSE: 256 bins by key RANGE, 4-byte tuples.
//
// The hash-bin passes of scan_bh_packed.h (hdk_bh_scatter / hdk_bh_aggregate) move 8 bytes per row out and back in:
// [argument : key], because a bin chosen by the key's hash says nothing about the key.  When the key column's statistics span
// no more values than 256 LDS tables hold -- the reference's BaselineHash benchmark at 10 K and 100 K groups
// (Benchmarks/synthetic_benchmark/queries/BaselineHash/BH004-005.sql), any dictionary or surrogate key -- the bin can be the
// key's range instead: bin = (key - min) / width (width = entries / 256, rounded up: all 256 bins in use), and the tuple only
// needs the offset inside the bin next to the argument's code
//     tuple = code(argument) << w | ((key - min) - bin * width),   code = argument - val_min + 1, 0 = NULL,  2^w >= width
// 4 bytes when w + bits(code) <= 32 (w <= 11: a bin's dense table is at most 2 048 entries of 20 bytes in LDS, replicated
// while it is small).  Pass A writes 4 bytes
// a row instead of 8, pass B reads 4, finds a row's entry with one AND (no tags, no probe: bh_rows_update of
// scan_bh_packed.h is the row body), and every group is folded into the output table exactly once, by the one block that
// owns its bin.  The NULL key is entry `dense_n`, one past the largest key.  Rows whose key or argument lies outside the
// statistics, and rows that find their bin's sub-slab full (a hot key), go through the reference's own scheme
// (bh_exact_row) where they are met.
// Reference being replaced: get_group_value + agg_* on the final table for every row (QE/GroupByRuntime.cpp:31-55).
#pragma once
#include "scan_bh_packed.h"

namespace hdk {

#ifndef HDK_BH_DP_VR
#define HDK_BH_DP_VR 8  // (16: 128-byte runs per bin, but 182 registers -- one block on a CU, pass A 1.57 ms against 1.15 at 256 M rows;
                        //  loading the next tile ahead of the batch's barriers: a second register set that spills at four waves, 1.87 ms)
#endif
constexpr int kBhDpVR = HDK_BH_DP_VR;                // rows per lane and batch of pass A
constexpr int kBhDpTile = kPbBlock * kBhDpVR;
constexpr size_t kBhDpScatterLds = static_cast<size_t>(kBhDpTile) * 4 + kBhDpTile + 16;  // uint32 staging | uint8 bin of every slot
constexpr int kBhDpAggThreads = 1024;
#ifndef HDK_BH_DP_WAVES
#define HDK_BH_DP_WAVES 4  // waves of pass A on a SIMD (two 512-thread blocks per CU): the register budget, 128
#endif

struct BhDensePartArgs {
  BhPackedArgs p;          // columns, statistics, word kinds, the LDS geometry of pass B (cap_log2 = w; replicas while the table is small)
  uint32_t w;              // bits of a tuple's offset inside its bin (2^w >= width)
  uint32_t width;          // entries of a bin
  uint32_t wmagic, wshift; // entry / width
  uint32_t nbins;          // bins in use (<= 256)
  uint32_t n_entries;      // dense_n (+ 1: the NULL key's entry)
  uint64_t cap4;           // tuples of a (bin, XCD) sub-slab (multiple of 4)
  uint32_t* tuples4;       // [nbins][kPbXcds][cap4]
  uint32_t* fill;          // [nbins][kPbXcds] x kPbCursorStride
};

// the key a (bin, low bits) pair stands for, as the 64-bit key word of the output table
HDK_DEV int64_t bh_dp_key_word(const BhDensePartArgs& g, uint32_t entry) {
  const BhPackedArgs& a = g.p;
  if (entry >= a.dense_n) {  // the NULL key's entry
    return a.key_form == 1 ? a.key_null_out : a.key_null;
  }
  const int64_t key = static_cast<int64_t>(a.dense_min) + static_cast<int64_t>(entry);
  return a.key_form == 1 ? double_to_bits(static_cast<double>(key)) : key;
}

// ---- pass A ---------------------------------------------------------------------------------------------------------------
template <int KW, int VW>
__global__ __launch_bounds__(kPbBlock, HDK_BH_DP_WAVES) void hdk_bh_dscatter(BhDensePartArgs g) {
  constexpr int VR = kBhDpVR;
  const BhPackedArgs& a = g.p;
  __shared__ uint32_t s_cnt[kPbMaxBins];
  __shared__ uint4 s_run[kPbMaxBins];
  __shared__ uint32_t s_total;
  __shared__ int32_t s_watch;
  __shared__ BhExactCtx s_cx;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn32[];
  uint32_t* s_stage = s_dyn32;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn32 + kBhDpTile);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kPbXcds - 1);
  for (int i = tid; i < kPbMaxBins; i += kPbBlock) {
    s_cnt[i] = 0;
  }
  bh_exact_ctx_init(&s_cx, a, tid);
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  int64_t* const out_buf = a.kp.groupby_buf[0];
  int32_t err = 0;
  const Watch watch = watch_begin(a.kp);
  const bool key_nullable = a.key_nullable != 0;
  const uint32_t lmask = (1u << g.w) - 1u;
  auto bin_of = [&](uint32_t e) {
    const uint32_t t = __umulhi(g.wmagic, e);
    return (((e - t) >> 1) + t) >> g.wshift;
  };
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  bool stop = false;
  // one row through the reference's scheme (statistics that do not hold for it; a full sub-slab)
  auto exact = [&](int64_t k, int64_t v) {
    const bool knull = key_nullable && k == a.key_null;
    const bool vnull = a.has_val && a.val_nullable && v == a.val_null;
    const int64_t kword = a.key_form == 1 ? (knull ? a.key_null_out : double_to_bits(static_cast<double>(k))) : k;
    const int32_t xe = bh_exact_row(a.plan, out_buf, a.out_entry_count, &s_cx, kword, v, vnull);
    err = xe ? xe : err;
  };
  for (uint64_t f = 0; f < nfrag && !stop; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kBhDpTile - 1) / kBhDpTile;
    const int8_t* const* cols = a.kp.col_buffers[f];
    const gcol_t kcol = (gcol_t)cols[a.key_buf_idx];
    const gcol_t vcol = VW ? (gcol_t)cols[a.val_buf_idx] : nullptr;
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      if (watch.flags) {
        if (const int32_t wv = watch_poll_block(watch, &s_watch)) {
          err = wv;
          stop = true;
          break;
        }
      }
      const int64_t row0 = (tile - frag_tile_begin) * kBhDpTile;
      const bool full = row0 + kBhDpTile <= nrows;
      // rows of a lane: 16-byte pieces of the wider column (R rows each), U = VR / R pieces.  What is kept per row is 32 bits
      // of key and argument and three flag bits (8-byte values that do not fit go through the exact path as they are met):
      // sixteen rows of 64-bit values and row numbers took 185 - 236 registers, one 512-thread block per CU
      constexpr int WMAX = KW > VW ? KW : VW;
      constexpr int R = 16 / WMAX;
      constexpr int U = VR / R;
      int32_t k32[VR], v32[VR];
      uint32_t livem = 0, vnullm = 0, knullm = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rbase = row0 + (static_cast<int64_t>(u) * kPbBlock + tid) * R;
        int64_t k64[R], v64[R];
        bool ok[R];
        if (full) {
          uint32_t kr[KW * R / 4], vr[VW ? VW * R / 4 : 1];
          load_bytes<KW * R, true>(kcol + rbase * KW, kr);
          if (VW) load_bytes<(VW ? VW * R : 4), true>(vcol + rbase * VW, vr);
#pragma unroll
          for (int i = 0; i < R; ++i) {
            k64[i] = extract_elem<KW>(kr, i);
            v64[i] = VW ? extract_elem<(VW ? VW : 8)>(vr, i) : 0;
            ok[i] = true;
          }
        } else {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            ok[i] = rbase + i < nrows;
            k64[i] = ok[i] ? load_elem<KW>(kcol, rbase + i) : 0;
            v64[i] = (VW && ok[i]) ? load_elem<(VW ? VW : 8)>(vcol, rbase + i) : 0;
          }
        }
        if (a.nquals) {
          int64_t rows[R];
#pragma unroll
          for (int i = 0; i < R; ++i) {
            rows[i] = ok[i] ? rbase + i : row0;
          }
          plain_quals_pass<R, true>(a.q, a.nquals, cols, rows, ok, true);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const int r = u * R + i;
          const bool vnull = (VW != 0) & (a.val_nullable != 0) & (v64[i] == a.val_null);
          const bool knull = key_nullable & (k64[i] == a.key_null);
          k32[r] = static_cast<int32_t>(k64[i]);
          v32[r] = static_cast<int32_t>(v64[i]);
          bool wide = false;
          if (KW == 8) {
            wide = knull | (k64[i] != static_cast<int64_t>(k32[r]));  // (an 8-byte column's NULL has no entry in the bins)
          }
          if (VW == 8) {
            wide = wide | (!vnull & (v64[i] != static_cast<int64_t>(v32[r])));
          }
          bool pending = ok[i] & wide;
          while (__builtin_amdgcn_ballot_w64(pending)) {
            if (pending) {
              pending = false;
              exact(k64[i], v64[i]);
            }
          }
          livem |= (ok[i] & !wide) ? 1u << r : 0u;
          vnullm |= vnull ? 1u << r : 0u;
          knullm |= (KW == 4 && knull) ? 1u << r : 0u;
        }
      }
      uint32_t tup[VR], bin[VR];
      bool live[VR];
      uint32_t slow = 0;
      const uint32_t vspan = static_cast<uint32_t>(a.val_max) - static_cast<uint32_t>(a.val_min);
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const bool vnull = (vnullm >> r) & 1u, knull = (knullm >> r) & 1u, in = (livem >> r) & 1u;
        const uint32_t d = static_cast<uint32_t>(k32[r]) - static_cast<uint32_t>(a.dense_min);
        const uint32_t dv = static_cast<uint32_t>(v32[r]) - static_cast<uint32_t>(a.val_min);
        const bool kin = (d < a.dense_n) | (knull & (g.n_entries > a.dense_n));
        const bool vin = (VW == 0) | vnull | (dv <= vspan);
        const uint32_t e = knull ? a.dense_n : d;
        const uint32_t code = (VW == 0 || vnull) ? 0u : dv + 1u;
        slow |= (in & !(kin & vin)) ? 1u << r : 0u;
        live[r] = in & kin & vin;
        const uint32_t b = live[r] ? bin_of(e) : 0u;
        tup[r] = (code << g.w) | ((e - b * g.width) & lmask);
        bin[r] = b;
      }
      while (__builtin_amdgcn_ballot_w64(slow != 0)) {  // (statistics that do not hold: the lane's pending rows one after another)
        if (slow) {
          const int j = __ffs(slow) - 1;
          slow &= slow - 1;
          int32_t kj = k32[0], vj = v32[0];
#pragma unroll
          for (int i = 1; i < VR; ++i) {
            kj = i == j ? k32[i] : kj;
            vj = i == j ? v32[i] : vj;
          }
          // (the 32-bit forms are the values: what did not fit went through the exact path above; NULLs by their flags)
          exact(((knullm >> j) & 1u) ? a.key_null : static_cast<int64_t>(kj), ((vnullm >> j) & 1u) ? a.val_null : static_cast<int64_t>(vj));
        }
      }
      // the batch: LDS histogram by bin, one cursor claim per bin and XCD, staging ordered by bin, copy-out (the step of
      // part_scatter_batch.h on 4-byte tuples)
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
          const uint64_t room = base < g.cap4 ? g.cap4 - base : 0;
          nfit = n < room ? n : static_cast<uint32_t>(room);
        }
        // (.z / .w: where the run starts in the tuple array, 64 bits -- the copy-out adds its index and stores)
        const uint64_t at = (static_cast<uint64_t>(tid) * kPbXcds + xcd) * g.cap4 + base;
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
        const uint32_t t = s_stage[i];
        if (r < run.y) {
          g.tuples4[((static_cast<uint64_t>(run.w) << 32) | run.z) + r] = t;
        } else {  // the sub-slab is full (a hot key): this row through the reference's scheme, now
          const uint32_t e = b * g.width + (t & lmask);
          const uint32_t code = t >> g.w;
          const int64_t k = e >= a.dense_n ? a.key_null : static_cast<int64_t>(a.dense_min) + e;
          const int64_t v = (VW == 0) ? 0 : (code == 0 ? a.val_null : static_cast<int64_t>(a.val_min) + (code - 1));
          exact(k, v);
        }
      }
      __syncthreads();
    }
    frag_tile_begin += ntiles;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

// ---- pass B: one block per bin, the bin's 2^w entries in LDS, entry = the tuple's low bits ------------------------------------
__global__ __launch_bounds__(kBhDpAggThreads) void hdk_bh_daggregate(BhDensePartArgs g) {
  const BhPackedArgs& a = g.p;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
  __shared__ BhExactCtx s_cx;
  const int tid = threadIdx.x;
  bh_exact_ctx_init(&s_cx, a, tid);
  bh_packed_lds_init(lds32, a, tid, kBhDpAggThreads);
  __syncthreads();
  const BhHot hot = bh_hot(a);
  const uint32_t bin = blockIdx.x;
  const uint32_t cap = 1u << g.w;  // (the dummy entry's index; the bin's entries are 0 .. width - 1)
  const uint32_t lmask = cap - 1u;
  uint32_t* rp = lds32 + (tid & (a.rep - 1)) * a.rep_words;
  const int32_t vbias = a.val_min - 1;
  int32_t err = 0;
  uint32_t rows_since_flush = 0;
  // the bin's groups into the output table: this block is the only one that holds them
  auto flush = [&]() {
    __syncthreads();
    const TableShape shape = table_shape(a.plan);
    int64_t* buf = a.kp.groupby_buf[0];
    for (uint32_t ei = tid; ei < g.width; ei += kBhDpAggThreads) {
      // the entry over the replicas: rows, sums and NULL counts add up in 64 bits (a replica's packed fields hold what ITS
      // lanes saw: decoded one by one)
      BhPartial b = {0, 0, 0, INT32_MAX, INT32_MIN};
      for (uint32_t r = 0; r < a.rep; ++r) {
        const uint32_t* rb = lds32 + r * a.rep_words;
        const uint64_t pk = reinterpret_cast<const uint64_t*>(rb + a.off_packed)[ei];
        const uint32_t nl = rb[a.off_nulls + ei];
        if ((pk | nl) == 0) {
          continue;  // (rows >= 1 makes the packed word positive: |sum| < 2^39)
        }
        const BhPartial one = bh_decode(pk, nl, reinterpret_cast<const uint64_t*>(rb + a.off_mm)[ei]);
        b.rows += one.rows;
        b.nulls += one.nulls;
        b.sum += one.sum;
        b.mn = min(b.mn, one.mn);
        b.mx = max(b.mx, one.mx);
      }
      if (b.rows == 0) {
        continue;
      }
      bh_fold_group_fn(a.plan, shape, s_cx.wl, buf, a.out_entry_count, s_cx.col_off, bh_dp_key_word(g, bin * g.width + ei),
                       [&](int wd) -> int64_t { return bh_partial_word(b, s_cx.wkind[wd]); }, err);
    }
  };
  constexpr int R = 4;  // tuples per 16-byte load
  constexpr int U = 4;
  constexpr uint32_t kStep = kBhDpAggThreads * R * U;
  typedef uint32_t __attribute__((ext_vector_type(4))) u32x4_t;
  for (uint32_t x = 0; x < kPbXcds; ++x) {
    const size_t sub = static_cast<size_t>(bin) * kPbXcds + x;
    const uint32_t n = static_cast<uint32_t>(min(static_cast<uint64_t>(g.fill[sub * kPbCursorStride]), g.cap4));
    const uint32_t* t = g.tuples4 + sub * g.cap4;
    for (uint32_t base = 0; base < n; base += kStep) {
      if (rows_since_flush + kStep > a.flush_rows) {  // (block-uniform)
        flush();
        __syncthreads();
        bh_packed_lds_init(lds32, a, tid, kBhDpAggThreads);
        __syncthreads();
        rows_since_flush = 0;
      }
      rows_since_flush += kStep;
      const bool full = base + kStep <= n;
      u32x4_t tv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = base + (static_cast<uint32_t>(u) * kBhDpAggThreads + tid) * R;
        if (full || i + R <= n) {  // (sub-slabs start 16-byte aligned: cap4 is a multiple of 4)
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
        const uint32_t i = base + (static_cast<uint32_t>(u) * kBhDpAggThreads + tid) * R;
        const uint32_t tw[R] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
        uint32_t e[R];
        int32_t val[R];
        uint32_t nulls = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const uint32_t code = tw[j] >> g.w;
          const bool in = full | (i + j < n);
          const bool isnull = (a.has_val != 0) & (code == 0);
          val[j] = static_cast<int32_t>(code) + vbias;
          e[j] = in ? tw[j] & lmask : cap;  // (the dummy entry behind the table takes the rows past the end)
          nulls |= (in & isnull) ? 1u << j : 0u;
        }
        bh_rows_update<R>(hot, rp, e, nulls, val);
      }
    }
  }
  flush();
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
