// scan_bhm.h -- group-by on chip for aggregates over SEVERAL argument columns / expressions and for several key columns.
//
// The packed kernels of scan_bh_packed.h take ONE argument column; one shape further the reference's own benchmark suite fell
// off the chip: MultiStep/MSBS001 / MSPHS001 (count(*), max(x100), max(x10), max(x10 + 1), sum(x100), sum(x10 + 1) by 1 000
// groups) ran on global atomics, PerfectHashMultiCol/PHM001-002 (five aggregates by two key columns) on the interpreter or the
// perfect-partitioned passes.  Reference being replaced: get_group_value[_fast] + agg_*_shared on the final table for every row
// (QE/GroupByRuntime.cpp:31-55,198-246, QE/cuda_mapd_rt.cu:167-203,424-478); the multi-column perfect-hash index is
// perfect_key_hash (QE/RowFuncBuilder.cpp:647-689,748-801).
//
// Same pricing as scan_bh_packed.h -- an LDS atomic is the expensive thing, a read is cheap:
//   * a block keeps a DENSE table in LDS: entry = sum_k (key_k - min_k) * stride_k by the key columns' statistics (a
//     GroupByPerfectHash plan: exactly its own entry index, NULLs under their translated value), no tags, no probe;
//   * every argument that is counted, summed or averaged has ONE 64-bit word [non-NULL rows : 24 | sum : 40] per entry -- one
//     ds_add_u64 a row gives COUNT, SUM and AVG; the row count comes from an argument that cannot be NULL (else a 32-bit
//     counter);
//   * every MIN / MAX of the row lives in a FIELD of one shared 64-bit word per entry, sized from the argument's statistics
//     (x100: 7 bits + a guard bit), all of them "bigger is better" codes (MAX: v - min + 1, MIN: max - v + 1, 0 = none): the
//     row reads the word (ds_read_b64), compares all fields at once ((cur | guards) - candidates keeps every guard bit iff no
//     field improves) and only then -- after a group's first rows, never -- runs a compare-and-swap loop;
//   * arguments may be `column op literal` (+ - *), several arguments may read the same column: it is streamed once.
// MSBS001's six aggregates are two ds_add_u64 and one ds_read_b64 a row, 24 bytes of LDS per group.
// At the end a block decodes its table into a slab of the generic partial-aggregate words (agg_common.h); a GroupByPerfectHash
// plan's slabs are folded by hdk_finalize (the internal index IS the plan's), an open-addressing plan's by hdk_bhm_fold
// (find_or_claim on the reference's probe sequence, once per group).
// Statistics that do not hold (a key or an argument outside them, a NULL where none was announced, more rows in a block
// than the packed fields were sized for) raise a flag instead of an exact per-row path: the folds then skip, and the
// global-atomics kernel armed behind them redoes the launch -- never a wrong result from stale metadata, only a slower one.
#pragma once
#include "watch.h"
#include "agg_common.h"
#include "scan_agg_fast.h"
#include "scan_bh.h"

namespace hdk {

constexpr int kBhmMaxKeys = 3;
constexpr int kBhmMaxSrc = 3;     // streamed argument columns
constexpr int kBhmMaxDer = 2;     // arguments derived from one column (the column itself, column op literal)
constexpr int kBhmMaxPacked = kBhmMaxSrc * kBhmMaxDer;
constexpr int kBhmMaxFields = 2 * kBhmMaxSrc * kBhmMaxDer;
constexpr int kBhmMaxMm = 2;      // 64-bit MIN / MAX words per entry
constexpr int kBhmSumBits = 40;

enum BhmWordKind : int32_t { BMW_ROWS = 0, BMW_SUM = 1, BMW_NN = 2, BMW_MIN = 3, BMW_MAX = 4 };

struct BhmKey {
  int32_t buf_idx;
  int32_t min;        // entry term = key - min ...
  uint32_t n;         // ... for key - min < n (else the statistics do not hold)
  uint32_t stride;
  int32_t nullable;   // the NULL key has a term of its own
  int32_t null32;
  uint32_t null_d;
  uint32_t pad_;
};

struct BhmDer {
  int32_t op;         // 0: the column's value; HDK_OP_ADD / HDK_OP_SUB / HDK_OP_MUL with `lit`
  int32_t lit;
  int32_t packed;     // index of its [rows : sum] word, or -1
  int32_t mx_word, mn_word;  // which MIN / MAX word holds its MAX / MIN field, or -1
  uint32_t mx_shift, mn_shift;
  int32_t mx_bias;    // MAX code = v - mx_bias + 1 (mx_bias = the smallest value the statistics allow)
  int32_t mn_bias;    // MIN code = mn_bias - v + 1 (mn_bias = the largest)
  uint32_t mx_mask, mn_mask;  // the field without its guard bit, unshifted
  uint32_t pad_;
};

struct BhmSrc {
  int32_t buf_idx;
  int32_t nullable;   // the column's in-band NULL is announced by the statistics and skipped by every target
  int32_t null32;
  int32_t raw_min;    // statistics of the column: raw - raw_min <= raw_span
  uint32_t raw_span;
  int32_t nder;
  BhmDer der[kBhmMaxDer];
};

// one word of the slab a block writes at the end (agg_common.h's layout)
struct BhmSlabWord {
  int32_t kind;       // BhmWordKind
  int32_t packed;     // BMW_SUM / BMW_NN / BMW_ROWS: the packed word (ROWS: -1 = the 32-bit row counter)
  int32_t mm_word;    // BMW_MIN / BMW_MAX (and BMW_NN of an argument without a packed word: non-NULL iff its field is not 0)
  uint32_t shift, mask;
  int32_t bias;
};

struct BhmArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  int64_t* slabs;          // [grid][entries][wpe]
  uint32_t* flag;          // != 0: the statistics did not hold somewhere -- the folds skip, the armed fallback runs
  uint32_t entries;        // internal (dense) entries = slab rows
  uint32_t e1;             // entries + the dummy, rounded up to even
  uint32_t rep;            // replicas (power of two)
  uint32_t rep_bytes;      // bytes between replicas
  uint32_t off_mm;         // byte offsets inside a replica: packed words at 0, then the MIN / MAX words, then the row counters
  uint32_t off_rows;
  uint32_t lds_bytes;
  uint32_t max_rows_per_block;  // what the packed fields were sized for
  int32_t nkeys, nsrc;
  int32_t npacked, nmm;
  int32_t rows_packed;     // packed word whose count is the row count, or -1: the 32-bit counters
  int32_t wpe;
  BhmKey key[kBhmMaxKeys];
  BhmSrc src[kBhmMaxSrc];
  uint64_t guards[kBhmMaxMm];    // guard bits of the fields of each MIN / MAX word
  int32_t nfields[kBhmMaxMm];
  uint32_t fshift[kBhmMaxMm][kBhmMaxFields];
  uint32_t fmask[kBhmMaxMm][kBhmMaxFields];
  BhmSlabWord sw[kMaxWordsPerEntry];
  // open-addressing plans: the key word of internal entry i (hdk_bhm_fold)
  int32_t key_form;        // 0: the key column's value; 1: cast(integer AS double)
  int32_t pad_;
  int64_t key_null_word;   // key word of the NULL key's entry
};

HDK_DEV int32_t bhm_apply(int32_t op, int32_t raw, int32_t lit) {
  return op == 0 ? raw : (op == HDK_OP_ADD ? raw + lit : (op == HDK_OP_SUB ? raw - lit : raw * lit));
}

// per-field maximum of two MIN / MAX words (the rare path: a row improves some field)
HDK_DEV uint64_t bhm_merge_fields(const BhmArgs& a, int w, uint64_t cur, uint64_t cand) {
  uint64_t out = 0;
  for (int f = 0; f < a.nfields[w]; ++f) {
    const uint64_t m = static_cast<uint64_t>(a.fmask[w][f]) << a.fshift[w][f];
    const uint64_t x = cur & m, y = cand & m;
    out |= x > y ? x : y;
  }
  return out;
}

// NR rows of one lane.  k[kk][j]: key column kk of row j; x[s][j]: argument column s.  `stale` collects "the statistics do
// not hold" (the caller raises the launch's flag).
template <int NK, int NS, int NR>
HDK_DEV void bhm_rows(const BhmArgs& a, uint8_t* rp, const int32_t (&k)[NK][NR], const int32_t (&x)[NS][NR], uint32_t& stale) {
  const uint32_t dummy = a.entries;
  uint32_t e[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    uint32_t idx = 0;
    bool bad = false;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const BhmKey& key = a.key[kk];
      const int32_t kv = k[kk][j];
      const bool isnull = (key.nullable != 0) & (kv == key.null32);
      uint32_t d = static_cast<uint32_t>(kv) - static_cast<uint32_t>(key.min);
      bad = bad | (!isnull & (d >= key.n));
      d = isnull ? key.null_d : d;
      idx += __umul24(d, key.stride);
    }
    stale |= bad ? 1u : 0u;
    e[j] = bad ? dummy : idx;
  }
  uint64_t cand0[NR], cand1[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    cand0[j] = 0;
    cand1[j] = 0;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const BhmSrc& src = a.src[s];
    bool live[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int32_t raw = x[s][j];
      const bool isnull = (src.nullable != 0) & (raw == src.null32);
      const bool out = !isnull & ((static_cast<uint32_t>(raw) - static_cast<uint32_t>(src.raw_min)) > src.raw_span);
      stale |= out ? 1u : 0u;
      live[j] = !isnull & !out;
    }
#pragma unroll
    for (int i = 0; i < kBhmMaxDer; ++i) {
      if (i < src.nder) {
        const BhmDer& der = src.der[i];
        if (der.packed >= 0) {
          unsigned long long* pk = reinterpret_cast<unsigned long long*>(rp) + static_cast<uint32_t>(der.packed) * a.e1;
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            const int32_t v = bhm_apply(der.op, x[s][j], der.lit);
            // (a NULL argument is not counted and not added: its increment goes to the dummy entry)
            atomicAdd(pk + (live[j] ? e[j] : dummy),
                      (1ull << kBhmSumBits) + static_cast<unsigned long long>(static_cast<long long>(v)));
          }
        }
        if (der.mx_word >= 0) {
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            const int32_t v = bhm_apply(der.op, x[s][j], der.lit);
            const uint64_t c = live[j] ? static_cast<uint64_t>(static_cast<uint32_t>(v - der.mx_bias + 1) & der.mx_mask) << der.mx_shift : 0ull;
            cand0[j] |= der.mx_word == 0 ? c : 0ull;
            cand1[j] |= der.mx_word == 1 ? c : 0ull;
          }
        }
        if (der.mn_word >= 0) {
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            const int32_t v = bhm_apply(der.op, x[s][j], der.lit);
            const uint64_t c = live[j] ? static_cast<uint64_t>(static_cast<uint32_t>(der.mn_bias - v + 1) & der.mn_mask) << der.mn_shift : 0ull;
            cand0[j] |= der.mn_word == 0 ? c : 0ull;
            cand1[j] |= der.mn_word == 1 ? c : 0ull;
          }
        }
      }
    }
  }
  // MIN / MAX: look before touching
#pragma unroll
  for (int w = 0; w < kBhmMaxMm; ++w) {
    if (w < a.nmm) {
      unsigned long long* mm = reinterpret_cast<unsigned long long*>(rp + a.off_mm) + static_cast<uint32_t>(w) * a.e1;
      const uint64_t H = a.guards[w];
      uint64_t cur[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        cur[j] = mm[e[j]];
      }
      uint32_t better = 0;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const uint64_t cd = w == 0 ? cand0[j] : cand1[j];
        better |= ((((cur[j] | H) - cd) & H) != H) ? 1u << j : 0u;
      }
      if (__builtin_amdgcn_ballot_w64(better != 0)) {  // (after a group's first rows: never)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          if (better & (1u << j)) {
            const uint64_t cd = w == 0 ? cand0[j] : cand1[j];
            unsigned long long old = cur[j];
            while (true) {
              const unsigned long long merged = bhm_merge_fields(a, w, old, cd);
              if (merged == old) {
                break;
              }
              const unsigned long long seen = atomicCAS(mm + e[j], old, merged);
              if (seen == old) {
                break;
              }
              old = seen;
            }
          }
        }
      }
    }
  }
  if (a.rows_packed < 0) {
    uint32_t* rows = reinterpret_cast<uint32_t*>(rp + a.off_rows);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      atomicAdd(rows + e[j], 1u);
    }
  }
}

// word w of the slab for entry `ei`: the replicas' partials decoded and combined
HDK_DEV int64_t bhm_slab_word(const BhmArgs& a, const uint8_t* lds8, const BhmSlabWord& sw, uint32_t ei) {
  int64_t acc = (sw.kind == BMW_MIN) ? INT64_MAX : ((sw.kind == BMW_MAX) ? INT64_MIN : 0);
  for (uint32_t r = 0; r < a.rep; ++r) {
    const uint8_t* rp = lds8 + static_cast<size_t>(r) * a.rep_bytes;
    if (sw.kind == BMW_ROWS && sw.packed < 0) {
      acc += reinterpret_cast<const uint32_t*>(rp + a.off_rows)[ei];
    } else if (sw.kind == BMW_ROWS || sw.kind == BMW_SUM || (sw.kind == BMW_NN && sw.packed >= 0)) {
      const uint64_t pk = reinterpret_cast<const uint64_t*>(rp)[static_cast<uint32_t>(sw.packed) * a.e1 + ei];
      const int64_t sum = static_cast<int64_t>(pk << (64 - kBhmSumBits)) >> (64 - kBhmSumBits);
      const int64_t cnt = static_cast<int64_t>((pk - static_cast<uint64_t>(sum)) >> kBhmSumBits);
      acc += sw.kind == BMW_SUM ? sum : cnt;
    } else {
      const uint64_t mm = reinterpret_cast<const uint64_t*>(rp + a.off_mm)[static_cast<uint32_t>(sw.mm_word) * a.e1 + ei];
      const uint32_t code = static_cast<uint32_t>(mm >> sw.shift) & sw.mask;
      if (sw.kind == BMW_NN) {
        acc += code ? 1 : 0;  // (only "none / some" is read from the count of a MIN / MAX-only argument)
      } else if (code) {
        const int64_t v = sw.kind == BMW_MAX ? static_cast<int64_t>(sw.bias) + (code - 1) : static_cast<int64_t>(sw.bias) - (code - 1);
        acc = sw.kind == BMW_MAX ? (v > acc ? v : acc) : (v < acc ? v : acc);
      }
    }
  }
  return acc;
}

// NK key columns, NS argument columns, all 4 bytes wide; U steps of 16 bytes per lane, column and tile
template <int NK, int NS, int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void hdk_scan_agg_bhm(BhmArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
  constexpr int R = 4;
  const int tid = threadIdx.x;
  {
    uint4* z = reinterpret_cast<uint4*>(lds8);
    const uint32_t n16 = a.lds_bytes / 16;
    for (uint32_t i = tid; i < n16; i += BLOCK) {
      z[i] = make_uint4(0, 0, 0, 0);  // (every code is "0 = nothing yet")
    }
  }
  __syncthreads();
  uint8_t* rp = lds8 + static_cast<size_t>(tid & (a.rep - 1)) * a.rep_bytes;
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(BLOCK) * R * U;
  int32_t err = 0;
  uint32_t stale = 0;
  uint32_t rows_seen = 0;
  const Watch watch = watch_begin(a.kp);
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
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
      HDK_WATCH_TILE(watch, err, tile)
      rows_seen += static_cast<uint32_t>(kTileRows);
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      if (row0 + kTileRows <= nrows) {
        uint32_t kr[U][NK][4], xr[U][NS][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R;
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            load_bytes<16, true>(kcol[kk] + r * 4, kr[u][kk]);
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            load_bytes<16, true>(xcol[s] + r * 4, xr[u][s]);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int32_t kv[NK][R], xv[NS][R];
#pragma unroll
          for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
              kv[kk][i] = static_cast<int32_t>(kr[u][kk][i]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              xv[s][i] = static_cast<int32_t>(xr[u][s][i]);
            }
          }
          bhm_rows<NK, NS, R>(a, rp, kv, xv, stale);
        }
      } else {
        // the ragged tail of a fragment, a row per lane and trip
        for (int64_t r = row0 + tid; r < nrows; r += BLOCK) {
          int32_t kv[NK][1], xv[NS][1];
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            kv[kk][0] = static_cast<int32_t>(load_elem<4>(kcol[kk], r));
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            xv[s][0] = static_cast<int32_t>(load_elem<4>(xcol[s], r));
          }
          bhm_rows<NK, NS, 1>(a, rp, kv, xv, stale);
        }
      }
    }
    frag_tile_begin += ntiles;
  }
  // (more rows than the packed fields were sized for can only come from a row bound that did not hold)
  if (stale || rows_seen > a.max_rows_per_block) {
    atomicOr(a.flag, 1u);
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
  __syncthreads();
  int64_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * a.entries * a.wpe;
  const uint32_t total = a.entries * static_cast<uint32_t>(a.wpe);
  for (uint32_t i = tid; i < total; i += BLOCK) {
    const uint32_t ei = i / static_cast<uint32_t>(a.wpe), w = i % static_cast<uint32_t>(a.wpe);
    slab[i] = bhm_slab_word(a, lds8, a.sw[w], ei);
  }
}

// ---- the fold of an open-addressing plan's slabs: one wave per internal entry ---------------------------------------------------
// lanes reduce the entry's words over the slabs (a fixed shuffle tree: deterministic), lane 0 finds or claims the group's
// entry on the reference's probe sequence (bh_fold_group_fn, scan_bh.h) and adds the partial in -- every group is folded into
// the output table ONCE.
struct BhmFoldArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  const int64_t* slabs;
  const uint32_t* flag;
  uint32_t num_slabs, entries, out_entry_count;
  int32_t wpe;
  int32_t wop[kMaxWordsPerEntry];
  uint32_t nword_mask;     // words that hold non-NULL counts (the fold wants NULL counts: rows - word)
  int32_t key_form;
  int64_t key_lo;          // internal entry i is the group of key key_lo + i ...
  uint32_t null_entry;     // ... but for this one (0xFFFFFFFF: none), the NULL key's
  uint32_t pad_;
  int64_t key_null_word;
};

template <int DUMMY = 0>
__global__ __launch_bounds__(256) void hdk_bhm_fold(BhmFoldArgs a) {
  __shared__ WordLayout wl;
  __shared__ uint64_t s_col_off[2 * HDK_HIP_MAX_TARGETS];
  __shared__ int64_t s_words[4][kMaxWordsPerEntry];
  if (*a.flag) {
    return;  // the statistics did not hold: the slabs are not to be trusted (the armed fallback redoes the launch)
  }
  if (threadIdx.x == 0) {
    make_word_layout(a.plan, &wl);
  }
  if (threadIdx.x < 2 * HDK_HIP_MAX_TARGETS) {
    s_col_off[threadIdx.x] = a.plan->output_columnar ? columnar_slot_off(a.plan, a.out_entry_count, threadIdx.x) : 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t entry = blockIdx.x * 4 + wave;
  if (entry >= a.entries) {
    return;
  }
  const int wpe = a.wpe;
  const size_t ew = static_cast<size_t>(a.entries) * wpe;
  int64_t* words = s_words[wave];
  for (int w = 0; w < wpe; ++w) {
    const int32_t op = a.wop[w];
    int64_t acc = word_identity(op);
    for (uint32_t b = lane; b < a.num_slabs; b += 64) {
      acc = word_combine(op, acc, a.slabs[b * ew + static_cast<size_t>(entry) * wpe + w]);
    }
    for (int d = 32; d > 0; d >>= 1) {
      const int lo = __shfl_down(static_cast<int>(static_cast<uint32_t>(acc)), d, 64);
      const int hi = __shfl_down(static_cast<int>(static_cast<uint64_t>(acc) >> 32), d, 64);
      acc = word_combine(op, acc, static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo)));
    }
    if (lane == 0) {
      words[w] = acc;
    }
  }
  if (lane != 0 || words[0] == 0) {
    return;  // (no row of this group: nothing to claim)
  }
  const int64_t rows = words[0];
  int32_t err = 0;
  int64_t key;
  if (entry == a.null_entry) {
    key = a.key_null_word;
  } else {
    const int64_t kv = a.key_lo + static_cast<int64_t>(entry);
    key = a.key_form == 1 ? double_to_bits(static_cast<double>(kv)) : kv;
  }
  bh_fold_group_fn(a.plan, table_shape(a.plan), wl, a.kp.groupby_buf[0], a.out_entry_count, s_col_off, key,
                   [&](int w) -> int64_t { return ((a.nword_mask >> w) & 1u) ? rows - words[w] : words[w]; }, err);
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
