// scan_bh_packed.h -- the open-addressing group-by on chip, priced against what an LDS operation costs.
//
// Measured (scripts/microbench/lds_atomics.hip, profiles/microbench/r05_lds_atomics.txt): a CU retires about 3 ds_add_u64
// lane-operations per clock when the lanes of a wave address random groups (10 without conflicts), 5 for 32-bit ones.  The
// reference's BaselineHash benchmark query -- count, sum, max, min, avg of one column by one key, 8 bytes a row -- spends
// five atomics and a key read per row in the plain form (scan_bh_fast.h): 1 B rows x 6 / 1.8e12 per second = 3.3 ms, three
// times the 1.0 ms the columns take to stream.  So the row is made cheaper instead:
//   * COUNT and SUM travel in ONE ds_add_u64: [rows : 24 | sum : 40], the sum signed -- the column statistics and the rows
//     a block can see bound both fields (checked on the host; a value outside the statistics goes through the exact
//     per-row path instead of the table);
//   * MIN and MAX are looked at before they are touched: one ds_read_b64 of [max : min] (32-bit each), an atomic only when
//     the row improves one of them -- after the first rows of a group, never;
//   * keys are 32-bit tags (the key column's value, before any cast), four to a 16-byte bucket: one ds_read_b128 finds
//     the entry of all but a few per cent of the rows (linear probing over buckets for the rest);
// three LDS operations a row, two of them plain reads that broadcast when lanes share a group.
// Second half of the round: DENSE tables -- when the key column's statistics span no more values than the
// table has room for, a row's entry is key - min and there are no tags at all (bh_dense_rows; the tags are written once, when a
// block hands its table on); PLAIN kernels -- an unfiltered plan's kernel holds only the hot form of a tile's steps (the
// general form, the exact-path call and the filter code each cost the hot form registers and waves when they shared its
// kernel); a THREE-LEVEL fold of the blocks' tables (hdk_bh_fold_slabs: slabs -> lists -> output table, every group folded
// into the output once).
// Shape: one group key = an integer column whose values fit 32 bits, as it is or cast to double; every aggregate over ONE
// integer column inside 32 bits with statistics (or COUNT(*) alone); filters `column cmp literal`.
//
// The same row body serves tables that do NOT fit LDS (10 K - 1 M entries): pass A (hdk_bh_scatter) writes 8-byte tuples
// [argument : key] into bins by the key's hash, pass B (hdk_bh_aggregate) runs one block per (bin, XCD) sub-slab with the
// bin's share of the groups in LDS.  Reference being replaced in both cases: get_group_value + agg_*_shared on the final
// table for every row (QE/GroupByRuntime.cpp:31-55, QE/cuda_mapd_rt.cu:167-203,424-478).
#pragma once
#include "watch.h"
#include "plain_quals.h"
#include "part_scatter_batch.h"
#include "scan_agg_fast.h"
#include "scan_bh.h"

namespace hdk {

constexpr uint32_t kBhTagEmpty = 0x7FFFFFFFu;  // a key equal to it takes the exact per-row path
constexpr int kBhSumBits = 40;
constexpr int kBhPackedBlock = 256;
constexpr int kBhAggBlock = 256;

enum BhWordKind : int32_t { BHW_ROWS = 0, BHW_NULLS = 1, BHW_SUM = 2, BHW_MIN = 3, BHW_MAX = 4 };

struct BhPackedArgs {
  const hdk_hip_plan* plan;  // device copy (the folds read the targets)
  KernParams kp;
  uint32_t out_entry_count;
  uint32_t cap_log2;       // tags per replica, >= 2 (buckets of four)
  uint32_t rep;            // replicas, power of two (1 in pass B)
  uint32_t rep_words;      // 32-bit words between replicas (arrays + padding that spreads the replicas over the banks)
  uint32_t off_mm;         // 32-bit word offsets of the replica's arrays behind tags[cap]: mm[cap] u64, packed[cap] u64, nulls[cap] u32
  uint32_t off_packed;
  uint32_t off_nulls;
  uint32_t bins_log2;      // pass A / B: bins by the top hash bits (0: the one-pass kernel)
  int32_t key_buf_idx, val_buf_idx;
  int32_t key_width, val_width;  // bytes of the columns
  int32_t key_form;        // 0: the column's value is the key; 1: cast(integer AS double); 2: column % key_mod (general kernels,
                           // dense tables only: the kernel bounds the key where the planner has no range -- (-m, m))
  int32_t key_nullable;
  int64_t key_null;        // the key column's in-band NULL (widened)
  int64_t key_null_out;    // key_form 1: the cast's NULL (NULL_DOUBLE bits)
  int32_t has_val, val_nullable;
  int64_t val_null;        // the argument column's in-band NULL (widened)
  int32_t val_min, val_max;  // statistics of the argument: what the packed sum was sized for
  int32_t key_min, key_max;  // statistics of the key column when it is 8 bytes wide (values outside take the exact path)
  int32_t want_minmax;
  // dense: the key column's statistics [dense_min, dense_min + dense_n) fit the table: entry = key - dense_min (NULL: dense_n)
  int32_t dense, dense_min;
  uint32_t dense_n, pad_dense_;
  int32_t key_mod;         // key_form 2: the literal divisor m (1 .. 2^15), with its magic for the unsigned quotient
  uint32_t mod_magic, mod_shift, pad_mod_;
  uint32_t flush_rows;     // a block folds its table into the output and starts over before it has seen this many rows: the
                           // bound the packed fields were sized for (rows < 2^24, |sum| < 2^39 per entry)
  int32_t nquals;
  int32_t pad_;
  int32_t wkind[kMaxWordsPerEntry];  // BhWordKind of every word of the layout (agg_common.h)
  ProjFastQual q[kMaxPlainQuals];
  // two-level fold: every scan block leaves its table (replica 0: 6 x cap 32-bit words) in slabs[block]; hdk_bh_fold_slabs
  // merges them slice by slice in LDS and folds each group into the output table a few times instead of once per block
  uint32_t* slabs;         // nullptr: the scan blocks fold into the output table themselves
  uint32_t num_slabs;
  uint32_t fold_slices;    // bucket ranges (power of two, <= buckets)
  uint32_t fold_groups;    // slab groups: fold block (slice, group) takes slabs group, group + fold_groups, ...
  // three-level fold (many slabs): stage 1 merges the slabs of a (slice, group) in LDS and writes its groups as a LIST
  // lists[(slice * fold_groups + group) * list_cap ...] of five 64-bit words [tag, max:min, rows, sum, nulls], their number in
  // list_counts[]; stage 2, ONE block per slice, merges the slice's lists and folds every group into the output table once --
  // no two blocks meet on an entry of the output.  0: the one-kernel form (every (slice, group) block folds into the output)
  int32_t fold_stage;
  uint32_t list_cap;
  uint64_t* lists;
  uint32_t* list_counts;
  // pass A / B
  int64_t* tuples;         // [bins][kPbXcds][cap]
  uint32_t* fill;          // [bins][kPbXcds] x kPbCursorStride
  uint64_t cap;
};

// the decoded partial of one entry
struct BhPartial {
  int64_t rows, nulls, sum, mn, mx;
};
HDK_DEV int64_t bh_partial_word(const BhPartial& b, int32_t kind) {
  return kind == BHW_ROWS ? b.rows : (kind == BHW_NULLS ? b.nulls : (kind == BHW_SUM ? b.sum : (kind == BHW_MIN ? b.mn : b.mx)));
}
HDK_DEV BhPartial bh_decode(uint64_t packed, uint32_t nulls, uint64_t mm) {
  BhPartial b;
  const int64_t sum = static_cast<int64_t>(packed << (64 - kBhSumBits)) >> (64 - kBhSumBits);
  b.sum = sum;
  b.nulls = nulls;
  b.rows = static_cast<int64_t>((packed - static_cast<uint64_t>(sum)) >> kBhSumBits) + nulls;
  b.mn = static_cast<int32_t>(static_cast<uint32_t>(mm));
  b.mx = static_cast<int32_t>(static_cast<uint32_t>(mm >> 32));
  return b;
}

HDK_DEV uint32_t bh_tag_hash(int32_t key) { return static_cast<uint32_t>(key) * 0x9E3779B1u; }

// v % m for a positive invariant m, truncating like C (the sign of v): eval_expr's `a % b` (host_match.h: magic_u32)
HDK_DEV int32_t bh_mod(int32_t v, uint32_t m, uint32_t magic, uint32_t shift) {
  if (m == 1u) {
    return 0;
  }
  const uint32_t n = static_cast<uint32_t>(v < 0 ? -v : v);
  const uint32_t t = __umulhi(magic, n);
  const uint32_t q = (((n - t) >> 1) + t) >> shift;
  const int32_t r = static_cast<int32_t>(n - q * m);
  return v < 0 ? -r : r;
}
// the key word of the output table for a raw column value that goes through the exact path
HDK_DEV int64_t bh_exact_key_word(const BhPackedArgs& a, int64_t kj) {
  const bool knull = a.key_nullable && kj == a.key_null;
  if (a.key_form == 1) {
    return knull ? a.key_null_out : double_to_bits(static_cast<double>(kj));
  }
  if (a.key_form == 2) {
    return knull ? a.key_null_out : kj % static_cast<int64_t>(a.key_mod);
  }
  return kj;
}

// the 64-bit key word of the output table for a key column value
HDK_DEV int64_t bh_key_word(const BhPackedArgs& a, int32_t key) {
  if (a.key_form == 1) {
    return (a.key_nullable && static_cast<int64_t>(key) == a.key_null) ? a.key_null_out : double_to_bits(static_cast<double>(key));
  }
  if (a.key_form == 2) {  // (the table holds the remainder already; the column's NULL stands for the expression's)
    return (a.key_nullable && static_cast<int64_t>(key) == a.key_null) ? a.key_null_out : static_cast<int64_t>(key);
  }
  return static_cast<int64_t>(key);
}

// One row through the reference's own scheme: find_or_claim on the final table, agg_* atomics with the row's value.  Rare
// (a key equal to the tag value, statistics that do not hold, a full LDS table in pass B) and large: ONE out-of-line copy,
// scalar arguments only (the layout and the word kinds are read from LDS).  Returns 0 or the error.
struct BhExactCtx {   // in LDS, filled once per block
  WordLayout wl;
  int32_t wkind[kMaxWordsPerEntry];
  uint64_t col_off[2 * HDK_HIP_MAX_TARGETS];
};
static __device__ __attribute__((noinline)) int32_t bh_exact_row(const hdk_hip_plan* plan, int64_t* buf, uint32_t entry_count, const BhExactCtx* cx,
                                                          int64_t keyword, int64_t val, int32_t is_null) {
  const TableShape shape = table_shape(plan);
  BhPartial b;
  b.rows = 1;
  b.nulls = is_null ? 1 : 0;
  b.sum = b.mn = b.mx = val;
  int32_t err = 0;
  bh_fold_group_fn(plan, shape, cx->wl, buf, entry_count, cx->col_off, keyword,
                   [&](int w) -> int64_t { return bh_partial_word(b, cx->wkind[w]); }, err);
  return err;
}
HDK_DEV void bh_exact_ctx_init(BhExactCtx* cx, const BhPackedArgs& a, int tid) {
  if (tid == 0) {
    make_word_layout(a.plan, &cx->wl);
  }
  if (tid < kMaxWordsPerEntry) {
    cx->wkind[tid] = a.wkind[tid];
  }
  if (tid < 2 * HDK_HIP_MAX_TARGETS) {
    cx->col_off[tid] = a.plan->output_columnar ? columnar_slot_off(a.plan, a.out_entry_count, tid) : 0;
  }
}

// position of `key` in the replica's tags (claiming one when new): buckets of four, linear over buckets; -1: all full
HDK_DEV int32_t bh_tag_probe(uint32_t* tags, int32_t key, uint32_t bucket, uint32_t bmask) {
  int32_t found = -2;
  uint32_t steps = 0;
  while (found == -2) {
    const uint4 t = *reinterpret_cast<const uint4*>(tags + bucket * 4);
    const uint32_t k = static_cast<uint32_t>(key);
    const uint32_t tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (found == -2) {
        uint32_t cur = tv[i];
        if (cur == kBhTagEmpty) {
          cur = atomicCAS(tags + bucket * 4 + i, kBhTagEmpty, k);
          if (cur == kBhTagEmpty) {
            cur = k;
          }
        }
        if (cur == k) {
          found = static_cast<int32_t>(bucket * 4 + i);
        }
      }
    }
    if (found == -2) {
      bucket = (bucket + 1) & bmask;
      if (++steps > bmask) {
        found = -1;
      }
    }
  }
  return found;
}

// arrays of a replica hold cap + 4 entries: entry `cap` is the DUMMY that rows without an update address (bh_packed_rows)
HDK_DEV uint32_t bh_slab_words(const BhPackedArgs& a) { return 6u * ((1u << a.cap_log2) + 4u); }
HDK_DEV void bh_packed_lds_init(uint32_t* lds, const BhPackedArgs& a, int tid, int block) {
  const uint32_t cap = (1u << a.cap_log2) + 4u;
  for (uint32_t r = 0; r < a.rep; ++r) {
    uint32_t* base = lds + r * a.rep_words;
    for (uint32_t i = tid; i < cap; i += block) {
      base[i] = kBhTagEmpty;
      reinterpret_cast<uint64_t*>(base + a.off_mm)[i] = (static_cast<uint64_t>(static_cast<uint32_t>(INT32_MIN)) << 32) | static_cast<uint32_t>(INT32_MAX);
      reinterpret_cast<uint64_t*>(base + a.off_packed)[i] = 0;
      base[a.off_nulls + i] = 0;
    }
  }
}

// What the row body keeps in scalar registers (a copy of the hot fields: the argument struct is large)
struct BhHot {
  uint32_t bmask, bshift_total, bins_log2, cap;   // bucket = ((hash << bins_log2) >> bshift_total) & bmask; cap = the dummy entry
  uint32_t off_mm, off_packed, off_nulls;
  int32_t val_min, val_max;
  int32_t has_val, want_minmax, one_bucket;
  int32_t dense_min, key_null32, key_nullable;    // (dense tables)
  uint32_t dense_n;
};
HDK_DEV BhHot bh_hot(const BhPackedArgs& a) {
  BhHot h;
  h.bmask = (1u << (a.cap_log2 - 2)) - 1;
  h.bshift_total = 32 - (a.cap_log2 - 2);
  h.bins_log2 = a.bins_log2;
  h.cap = 1u << a.cap_log2;
  h.off_mm = a.off_mm;
  h.off_packed = a.off_packed;
  h.off_nulls = a.off_nulls;
  h.val_min = a.val_min;
  h.val_max = a.val_max;
  h.has_val = a.has_val;
  h.want_minmax = a.want_minmax;
  h.one_bucket = a.cap_log2 <= 2;
  h.dense_min = a.dense_min;
  h.dense_n = a.dense_n;
  h.key_nullable = a.key_nullable && a.key_width == 4;  // (an 8-byte column's NULL does not fit 32 bits: the caller's exact path)
  h.key_null32 = static_cast<int32_t>(a.key_null);
  return h;
}
HDK_DEV uint32_t bh_bucket_of(const BhHot& h, int32_t key) {
  // (one bucket: a shift by 32 is not a shift)
  return h.one_bucket ? 0u : ((bh_tag_hash(key) << h.bins_log2) >> h.bshift_total) & h.bmask;
}

// NR rows of one lane, straight-line for the common row: tags of all rows first (NR ds_read_b128 in flight), then
// [max : min] (one ds_read_b64 each), then ONE ds_add_u64 per row.  Rows that do not take part (filtered out, NULL argument,
// handed to the exact path) address the DUMMY entry behind the table with neutral operands instead of branching.  The rare
// cases sit behind wave-uniform tests: a key that is not in its bucket (probe loop, one inlined copy), a row that improves
// MIN / MAX, a NULL argument.  Returns the mask of rows for the caller's exact path: the tag value itself as a key,
// statistics that do not hold for the row, a full table when FULL_IS_ERROR is false.
template <int NR>
HDK_DEV void bh_rows_update(const BhHot& h, uint32_t* rp, const uint32_t (&e)[NR], uint32_t nulls, const int32_t (&val)[NR]);

template <int NR, bool FULL_IS_ERROR, bool ALLOK = false>
HDK_DEV uint32_t bh_packed_rows(const BhHot& h, uint32_t* rp, const int32_t (&key)[NR], const int32_t (&val)[NR], const bool (&ok_in)[NR],
                                const bool (&null_in)[NR], int32_t& err) {
  // (ALLOK: every row takes part -- full, unfiltered tiles: `ok_in` is not read.  Predicates are combined with & and |, never
  // && and ||, and folded into VGPR bit masks row by row: the short-circuit forms became branches around two compares, and four
  // rows of live lane masks pushed scalar registers into spill lanes inside the row loop.)
  uint32_t e[NR];
  uint32_t bucket[NR];
  uint4 t[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    bucket[j] = bh_bucket_of(h, key[j]);
    t[j] = *reinterpret_cast<const uint4*>(rp + bucket[j] * 4);
  }
  const uint32_t span = static_cast<uint32_t>(h.val_max) - static_cast<uint32_t>(h.val_min);
  const bool has_val = h.has_val != 0;
  uint32_t slow = 0, miss = 0, nulls = 0;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const uint32_t bit = 1u << j;
    const uint32_t k = static_cast<uint32_t>(key[j]);
    const bool okj = ALLOK ? true : ok_in[j];
    const bool nullj = null_in[j];
    // strangers: the tag value itself as a key; an argument (not NULL) outside the statistics the packed sum was sized for
    const bool outside = has_val & !nullj & ((static_cast<uint32_t>(val[j]) - static_cast<uint32_t>(h.val_min)) > span);
    const bool stranger = (k == kBhTagEmpty) | outside;
    const bool in = okj & !stranger;
    slow |= (okj & stranger) ? bit : 0u;
    const bool h0 = t[j].x == k, h1 = t[j].y == k, h2 = t[j].z == k, h3 = t[j].w == k;
    const uint32_t s = h0 ? 0u : (h1 ? 1u : (h2 ? 2u : 3u));
    e[j] = in ? bucket[j] * 4 + s : h.cap;
    miss |= (in & !(h0 | h1 | h2 | h3)) ? bit : 0u;
    nulls |= (in & nullj) ? bit : 0u;
  }
  // strangers to their bucket (a group's first rows, keys pushed out of a full bucket): the lane's pending rows one after
  // another through ONE inlined probe loop
  while (__builtin_amdgcn_ballot_w64(miss != 0)) {
    if (miss) {
      const int j = __ffs(miss) - 1;
      miss &= miss - 1;
      int32_t kj = key[0];
      uint32_t bj = bucket[0];
#pragma unroll
      for (int i = 1; i < NR; ++i) {
        kj = i == j ? key[i] : kj;
        bj = i == j ? bucket[i] : bj;
      }
      const int32_t ej = bh_tag_probe(rp, kj, bj, h.bmask);
      if (ej < 0) {
        if (FULL_IS_ERROR) {
          err = HDK_HIP_ERR_OUT_OF_SLOTS;  // more groups than the plan's table holds
        } else {
          slow |= 1u << j;
        }
        nulls &= ~(1u << j);
      }
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        e[i] = i == j ? (ej < 0 ? h.cap : static_cast<uint32_t>(ej)) : e[i];
      }
    }
  }
  bh_rows_update<NR>(h, rp, e, nulls, val);
  return slow;
}

// the updates of NR rows whose entries are known: e[j] = the row's entry, or the dummy (h.cap) when it does not take part;
// `nulls`: rows whose argument is NULL
template <int NR>
HDK_DEV void bh_rows_update(const BhHot& h, uint32_t* rp, const uint32_t (&e)[NR], uint32_t nulls, const int32_t (&val)[NR]) {
  unsigned long long* packed = reinterpret_cast<unsigned long long*>(rp + h.off_packed);
  if (!h.has_val) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      atomicAdd(packed + e[j], 1ull << kBhSumBits);  // (rows that do not take part count into the dummy entry)
    }
    return;
  }
  uint32_t e_add[NR];  // where the row's packed increment goes: its entry, or the dummy
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    e_add[j] = (nulls >> j) & 1u ? h.cap : e[j];  // a NULL argument is counted apart (below), not added
  }
  if (h.want_minmax) {
    uint64_t mm[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      mm[j] = reinterpret_cast<const uint64_t*>(rp + h.off_mm)[e_add[j]];
    }
    uint32_t better = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      // (a row that goes to the dummy entry may "improve" the dummy's pair once: harmless)
      const bool b = (val[j] < static_cast<int32_t>(static_cast<uint32_t>(mm[j]))) | (val[j] > static_cast<int32_t>(static_cast<uint32_t>(mm[j] >> 32)));
      better |= b ? 1u << j : 0u;
    }
    if (__builtin_amdgcn_ballot_w64(better != 0)) {  // (after a group's first rows: never)
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        if (better & (1u << j)) {
          int32_t* mmw = reinterpret_cast<int32_t*>(rp + h.off_mm) + 2 * e_add[j];
          atomicMin(mmw, val[j]);
          atomicMax(mmw + 1, val[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    // (the dummy entry swallows the increments of the rows that do not take part: no select on the operand)
    atomicAdd(packed + e_add[j], (1ull << kBhSumBits) + static_cast<unsigned long long>(static_cast<long long>(val[j])));
  }
  if (__builtin_amdgcn_ballot_w64(nulls != 0)) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (nulls & (1u << j)) {
        atomicAdd(rp + h.off_nulls + e[j], 1u);
      }
    }
  }
}

// DENSE tables: the key column's statistics span no more values than the table has entries, so a row's entry is
// key - dense_min (the NULL key: entry dense_n) -- no tags, no probe; the tags are written once, when the block hands its
// table on (bh_dense_finish), so that the folds see what they see from the tag form.  A key outside the statistics takes
// the exact path.
template <int NR, bool ALLOK>
HDK_DEV uint32_t bh_dense_rows(const BhHot& h, uint32_t* rp, const int32_t (&key)[NR], const int32_t (&val)[NR], const bool (&ok_in)[NR],
                               const bool (&null_in)[NR]) {
  const uint32_t span = static_cast<uint32_t>(h.val_max) - static_cast<uint32_t>(h.val_min);
  const bool has_val = h.has_val != 0, key_nullable = h.key_nullable != 0;
  uint32_t e[NR];
  uint32_t slow = 0, nulls = 0;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const uint32_t bit = 1u << j;
    const bool okj = ALLOK ? true : ok_in[j];
    const bool nullj = null_in[j];
    const uint32_t d = static_cast<uint32_t>(key[j]) - static_cast<uint32_t>(h.dense_min);
    const bool knull = key_nullable & (key[j] == h.key_null32);
    const bool kin = (d < h.dense_n) | knull;
    const bool outside = has_val & !nullj & ((static_cast<uint32_t>(val[j]) - static_cast<uint32_t>(h.val_min)) > span);
    const bool in = okj & kin & !outside;
    slow |= (okj & !(kin & !outside)) ? bit : 0u;
    e[j] = in ? (knull ? h.dense_n : d) : h.cap;
    nulls |= (in & nullj) ? bit : 0u;
  }
  bh_rows_update<NR>(h, rp, e, nulls, val);
  return slow;
}

// replicas 1.. into replica 0 (inside LDS)
template <int BLOCK>
HDK_DEV void bh_packed_merge_replicas(const BhPackedArgs& a, uint32_t* lds, int tid, int32_t& err) {
  const uint32_t cap = 1u << a.cap_log2;
  const uint32_t bmask = (cap >> 2) - 1;
  const uint32_t bshift = 32 - a.bins_log2 - (a.cap_log2 - 2);
  __syncthreads();
  if (a.rep > 1) {
    const uint32_t n = cap * (a.rep - 1);
    for (uint32_t i = tid; i < n; i += BLOCK) {
      const uint32_t r = 1 + i / cap, ei = i % cap;
      uint32_t* rb = lds + r * a.rep_words;
      const uint32_t tag = rb[ei];
      if (tag == kBhTagEmpty) {
        continue;
      }
      const int32_t key = static_cast<int32_t>(tag);
      const uint32_t bucket = a.cap_log2 > 2 ? ((bh_tag_hash(key) << a.bins_log2) >> (bshift + a.bins_log2)) & bmask : 0u;
      const int32_t e0 = bh_tag_probe(lds, key, bucket, bmask);
      if (e0 < 0) {
        err = HDK_HIP_ERR_OUT_OF_SLOTS;
        continue;
      }
      const uint64_t pk = reinterpret_cast<const uint64_t*>(rb + a.off_packed)[ei];
      const uint64_t mm = reinterpret_cast<const uint64_t*>(rb + a.off_mm)[ei];
      const uint32_t nl = rb[a.off_nulls + ei];
      if (pk) atomicAdd(reinterpret_cast<unsigned long long*>(lds + a.off_packed) + e0, static_cast<unsigned long long>(pk));
      if (nl) atomicAdd(lds + a.off_nulls + e0, nl);
      int32_t* mmw = reinterpret_cast<int32_t*>(lds + a.off_mm) + 2 * e0;
      atomicMin(mmw, static_cast<int32_t>(static_cast<uint32_t>(mm)));
      atomicMax(mmw + 1, static_cast<int32_t>(static_cast<uint32_t>(mm >> 32)));
    }
    __syncthreads();
  }
}

// Dense tables at the end of a block (or of its flush interval): replicas 1.. into replica 0 entry by entry, then replica
// 0's tags -- the key of every entry that saw a row.
template <int BLOCK>
HDK_DEV void bh_dense_finish(const BhPackedArgs& a, uint32_t* lds, int tid) {
  const uint32_t cap = 1u << a.cap_log2;
  __syncthreads();
  for (uint32_t ei = tid; ei < cap; ei += BLOCK) {
    uint64_t pk = reinterpret_cast<const uint64_t*>(lds + a.off_packed)[ei];
    uint32_t nl = lds[a.off_nulls + ei];
    const uint64_t mm0 = reinterpret_cast<const uint64_t*>(lds + a.off_mm)[ei];
    int32_t mn = static_cast<int32_t>(static_cast<uint32_t>(mm0)), mx = static_cast<int32_t>(static_cast<uint32_t>(mm0 >> 32));
    for (uint32_t r = 1; r < a.rep; ++r) {
      const uint32_t* rb = lds + r * a.rep_words;
      pk += reinterpret_cast<const uint64_t*>(rb + a.off_packed)[ei];
      nl += rb[a.off_nulls + ei];
      const uint64_t mm = reinterpret_cast<const uint64_t*>(rb + a.off_mm)[ei];
      mn = min(mn, static_cast<int32_t>(static_cast<uint32_t>(mm)));
      mx = max(mx, static_cast<int32_t>(static_cast<uint32_t>(mm >> 32)));
    }
    reinterpret_cast<uint64_t*>(lds + a.off_packed)[ei] = pk;
    lds[a.off_nulls + ei] = nl;
    reinterpret_cast<uint64_t*>(lds + a.off_mm)[ei] = (static_cast<uint64_t>(static_cast<uint32_t>(mx)) << 32) | static_cast<uint32_t>(mn);
    const bool live = (pk | nl) != 0;  // (rows >= 1 makes the packed word positive: |sum| < 2^39)
    const uint32_t key = ei < a.dense_n ? ei + static_cast<uint32_t>(a.dense_min) : static_cast<uint32_t>(static_cast<int32_t>(a.key_null));
    lds[ei] = (live && ei <= a.dense_n) ? key : kBhTagEmpty;
  }
  __syncthreads();
}

// End of a block (or of its flush interval): replicas merged, replica 0's groups into the output table.
template <int BLOCK>
HDK_DEV void bh_packed_flush(const BhPackedArgs& a, const BhExactCtx* cx, uint32_t* lds, int tid, int32_t& err) {
  const uint32_t cap = 1u << a.cap_log2;
  bh_packed_merge_replicas<BLOCK>(a, lds, tid, err);
  const TableShape shape = table_shape(a.plan);
  int64_t* buf = a.kp.groupby_buf[0];
  for (uint32_t ei = tid; ei < cap; ei += BLOCK) {
    const uint32_t tag = lds[ei];
    if (tag == kBhTagEmpty) {
      continue;
    }
    const BhPartial b = bh_decode(reinterpret_cast<const uint64_t*>(lds + a.off_packed)[ei], lds[a.off_nulls + ei],
                                  reinterpret_cast<const uint64_t*>(lds + a.off_mm)[ei]);
    bh_fold_group_fn(a.plan, shape, cx->wl, buf, a.out_entry_count, cx->col_off, bh_key_word(a, static_cast<int32_t>(tag)),
                     [&](int w) -> int64_t { return bh_partial_word(b, cx->wkind[w]); }, err);
  }
}

// ---- the one-pass kernel: the table fits LDS -------------------------------------------------------------------------
// KW / VW: byte width of the key / argument column (VW 0: COUNT(*) only); U steps of 16 bytes per lane and tile
#ifndef HDK_BH_PACKED_WAVES
#define HDK_BH_PACKED_WAVES 0  // > 0: hold the kernel to that many waves per SIMD (A/B builds: make variant DEFS=-DHDK_BH_PACKED_WAVES=4)
#endif
// One step of a tile: the R rows in a lane's 16-byte registers through the row body.  ALLOK: every row takes part (a full tile
// of an unfiltered plan over 4-byte columns) -- the hot form computes no row bounds and no per-row flags at all.  Returns the
// rows for the exact path.
template <int KW, int VW, int R, bool DENSE, bool ALLOK, bool NOQ = false>
HDK_DEV uint32_t bh_tile_step(const BhPackedArgs& a, const BhHot& hot, uint32_t* rp, const uint32_t* kr, const uint32_t* vr, bool full,
                              int64_t rbase, int64_t row0, int64_t nrows, const int8_t* const* cols, int32_t& err) {
  int32_t key[R], val[R];
  bool ok[R], isnull[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    ok[i] = ALLOK || full || rbase + i < nrows;
  }
  if (!ALLOK && !NOQ && a.nquals != 0) {  // (NOQ: a plain kernel -- the plan has no filters: none compiled in)
    int64_t rows[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      rows[i] = ok[i] ? rbase + i : row0;
    }
    if (a.q[0].nprog != 0) {
      plain_quals_pass<R, true>(a.q, a.nquals, cols, rows, ok, true);
    } else {
      // a conjunction: a conjunct on the key or the argument column (WHERE measure > 0, WHERE key < c) reads the registers the
      // tile already holds instead of gathering the column a second time (the filtered BH001: 1.17 -> see DESIGN.md 3.4c)
      for (int qi = 0; qi < a.nquals; ++qi) {
        const ProjFastQual q = a.q[qi];
        const bool plain_int = q.col.kind == HDK_COL_INT && !q.fp;
        const bool on_val = VW != 0 && plain_int && q.col.buf_idx == a.val_buf_idx && q.col.width == VW;
        const bool on_key = plain_int && q.col.buf_idx == a.key_buf_idx && q.col.width == KW;
        if (on_val || on_key) {
          const bool nullable = q.nullable != 0;
#define HDK_BH_REGQ(OP)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < R; ++i) {                                                            \
    const int64_t v = on_val ? (VW ? extract_elem<(VW ? VW : 8)>(vr, i) : 0) : extract_elem<KW>(kr, i);        \
    ok[i] = ok[i] & !(nullable & (v == q.null_val)) & (v OP q.rhs);                                          \
  }
          switch (q.cmp) {
            case HDK_CMP_EQ: HDK_BH_REGQ(==) break;
            case HDK_CMP_NE: HDK_BH_REGQ(!=) break;
            case HDK_CMP_LT: HDK_BH_REGQ(<) break;
            case HDK_CMP_GT: HDK_BH_REGQ(>) break;
            case HDK_CMP_LE: HDK_BH_REGQ(<=) break;
            default: HDK_BH_REGQ(>=) break;
          }
#undef HDK_BH_REGQ
        } else {
          plain_quals_pass<R, false>(&a.q[qi], 1, cols, rows, ok, true);
        }
      }
    }
  }
  uint32_t slow = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int64_t k64 = extract_elem<KW>(kr, i);
    const int64_t v64 = VW ? extract_elem<(VW ? VW : 8)>(vr, i) : 0;
    key[i] = static_cast<int32_t>(k64);
    if (!NOQ && a.key_form == 2) {  // (general kernels only) the key is column % m; a NULL stays the column's NULL
      const bool knull = (a.key_nullable != 0) & (key[i] == static_cast<int32_t>(a.key_null));
      const int32_t r = bh_mod(key[i], static_cast<uint32_t>(a.key_mod), a.mod_magic, a.mod_shift);
      key[i] = knull ? key[i] : r;
    }
    val[i] = static_cast<int32_t>(v64);
    // (a 4-byte column's NULL test in 32 bits: the widened compare was a sign extension and two compares per row)
    isnull[i] = VW == 4 ? (a.val_nullable != 0) & (val[i] == static_cast<int32_t>(a.val_null))
                        : (VW != 0) & (a.val_nullable != 0) & (v64 == a.val_null);
    // 8-byte columns ride as their low 32 bits: what does not fit (a key outside the statistics or NULL, an argument
    // outside them) takes the exact path
    bool wide = false;
    if (KW == 8) {
      wide = (a.key_nullable && k64 == a.key_null) || k64 < a.key_min || k64 > a.key_max;
    }
    if (VW == 8) {
      wide = wide || (!isnull[i] && (v64 < a.val_min || v64 > a.val_max));
    }
    if (!ALLOK && ok[i] && wide) {
      slow |= 1u << i;
      ok[i] = false;
    }
  }
  slow |= DENSE ? bh_dense_rows<R, ALLOK>(hot, rp, key, val, ok, isnull) : bh_packed_rows<R, true, ALLOK>(hot, rp, key, val, ok, isnull, err);
  return slow;
}

// PLAIN: an unfiltered plan -- every row of a full tile takes part, so the kernel holds ONLY the hot form of the steps (and a
// row-at-a-time loop for the ragged tail of a fragment); over 4-byte columns that form has no per-row flags at all.  The general form -- row bounds, filters, 8-byte
// columns whose strangers bypass the table -- needs 178 vector registers with four steps in flight (two waves on a SIMD) and
// held the hot form to that when both sat in one kernel; on its own the hot form takes 122.
template <int KW, int VW, int U, int BLOCK, bool DENSE, bool PLAIN>
HDK_DEV void bh_packed_kernel_body(const BhPackedArgs& a) {
  // (8-byte columns in a plain kernel: their strangers -- values outside 32 bits or the statistics -- still go through `ok`,
  // but there are no row bounds and no filters in the step: HOT32 below is the all-rows-take-part form)
  constexpr bool HOT32 = KW != 8 && VW != 8;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
  __shared__ BhExactCtx s_cx;
  constexpr int WMAX = KW > VW ? KW : VW;
  constexpr int R = 16 / WMAX;
  constexpr int KB = KW * R;
  constexpr int VB = VW * R;
  constexpr int KREGS = KB >= 4 ? KB / 4 : 1;
  constexpr int VREGS = VB >= 4 ? VB / 4 : 1;
  const int tid = threadIdx.x;
  bh_exact_ctx_init(&s_cx, a, tid);
  bh_packed_lds_init(lds32, a, tid, BLOCK);
  __syncthreads();
  uint32_t* rp = lds32 + (tid & (a.rep - 1)) * a.rep_words;
  const BhHot hot = bh_hot(a);
  int64_t* const out_buf = a.kp.groupby_buf[0];
  uint32_t rows_since_flush = 0;
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  constexpr int64_t kTileRows = static_cast<int64_t>(BLOCK) * R * U;
  int32_t err = 0;
  const Watch watch = watch_begin(a.kp);
  const bool filtered = a.nquals != 0;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    const gcol_t kcol = (gcol_t)cols[a.key_buf_idx];
    const gcol_t vcol = VW ? (gcol_t)cols[a.val_buf_idx] : nullptr;
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      if (rows_since_flush + static_cast<uint32_t>(kTileRows) > a.flush_rows) {  // (block-uniform: every thread walks the same tiles)
        if (DENSE) {
          bh_dense_finish<BLOCK>(a, lds32, tid);
        }
        bh_packed_flush<BLOCK>(a, &s_cx, lds32, tid, err);
        __syncthreads();
        bh_packed_lds_init(lds32, a, tid, BLOCK);
        __syncthreads();
        rows_since_flush = 0;
      }
      rows_since_flush += static_cast<uint32_t>(kTileRows);
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      const bool full = row0 + kTileRows <= nrows;
      if (PLAIN && !full) {  // the ragged tail of a fragment, a row per lane and trip
        for (int64_t r = row0 + tid; r < nrows; r += BLOCK) {
          const int64_t kj = load_elem<KW>(kcol, r);
          const int64_t vj = VW ? load_elem<(VW ? VW : 4)>(vcol, r) : 0;
          const uint32_t k1[2] = {static_cast<uint32_t>(kj), static_cast<uint32_t>(static_cast<uint64_t>(kj) >> 32)};
          const uint32_t v1[2] = {static_cast<uint32_t>(vj), static_cast<uint32_t>(static_cast<uint64_t>(vj) >> 32)};
          uint32_t slow1 = bh_tile_step<KW, VW, 1, DENSE, false, true>(a, hot, rp, k1, v1, true, r, r, nrows, cols, err);
          while (__builtin_amdgcn_ballot_w64(slow1 != 0)) {
            if (slow1) {
              slow1 = 0;
              const bool nj = (VW != 0) & (a.val_nullable != 0) & (vj == a.val_null);
              const int64_t kword = bh_exact_key_word(a, kj);
              const int32_t xe = bh_exact_row(a.plan, out_buf, a.out_entry_count, &s_cx, kword, vj, nj);
              err = xe ? xe : err;
            }
          }
        }
        continue;
      }
      uint32_t kr[U][KREGS];
      uint32_t vr[U][VREGS];
      if (full) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R;
          load_bytes<KB, true>(kcol + r * KW, kr[u]);
          if (VW) load_bytes<(VB > 0 ? VB : 4), true>(vcol + r * VW, vr[u]);
        }
      } else {  // ragged tail of a fragment: element loads into the same registers
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t rbase = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            const bool in = rbase + i < nrows;
            const int64_t k = in ? load_elem<KW>(kcol, rbase + i) : 0;
            const int64_t v = (VW && in) ? load_elem<(VW ? VW : 8)>(vcol, rbase + i) : 0;
            if (KW == 8) {
              kr[u][2 * i] = static_cast<uint32_t>(k);
              kr[u][(2 * i + 1) % KREGS] = static_cast<uint32_t>(static_cast<uint64_t>(k) >> 32);
            } else {
              kr[u][i % KREGS] = static_cast<uint32_t>(k);
            }
            if (VW == 8) {
              vr[u][(2 * i) % VREGS] = static_cast<uint32_t>(v);
              vr[u][(2 * i + 1) % VREGS] = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
            } else if (VW == 4) {
              vr[u][i % VREGS] = static_cast<uint32_t>(v);
            }
          }
        }
      }
      uint32_t slow_all = 0;  // bit u * R + i: row i of step u goes through the reference's own scheme (below, after the steps)
      if (PLAIN) {
        // every row takes part -- no row bounds, no filters, no 8-byte columns (whose strangers `ok` carries)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          slow_all |= bh_tile_step<KW, VW, R, DENSE, HOT32, true>(a, hot, rp, kr[u], vr[u], true, 0, 0, 0, cols, err) << (u * R);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t rbase = row0 + (static_cast<int64_t>(u) * BLOCK + tid) * R;
          slow_all |= bh_tile_step<KW, VW, R, DENSE, false>(a, hot, rp, kr[u], vr[u], full, rbase, row0, nrows, cols, err) << (u * R);
        }
      }
      // rows for the reference's own scheme (rare): ONE call site behind the tile's steps -- the callee's registers come on
      // top of what is live at the call, and here that is the tile's column registers and nothing else.  The lane's pending
      // rows one after another.
      while (__builtin_amdgcn_ballot_w64(slow_all != 0)) {
        if (slow_all) {
          const int b = __ffs(slow_all) - 1;
          slow_all &= slow_all - 1;
          int64_t kj = 0, vj = 0;
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
              const bool me = b == u * R + i;
              const int64_t k64 = extract_elem<KW>(kr[u], i);
              const int64_t v64 = VW ? extract_elem<(VW ? VW : 8)>(vr[u], i) : 0;
              kj = me ? k64 : kj;
              vj = me ? v64 : vj;
            }
          }
          const bool nj = (VW != 0) & (a.val_nullable != 0) & (vj == a.val_null);
          const int64_t kword = bh_exact_key_word(a, kj);
          const int32_t xe = bh_exact_row(a.plan, out_buf, a.out_entry_count, &s_cx, kword, vj, nj);
          err = xe ? xe : err;
        }
      }
    }
    frag_tile_begin += ntiles;
  }
  if (DENSE) {
    bh_dense_finish<BLOCK>(a, lds32, tid);  // (the tag-keyed merge below then finds replicas 1.. without tags: nothing to do)
  }
  if (a.slabs) {
    // the block's table to its slab; hdk_bh_fold_slabs folds the slabs into the output table
    bh_packed_merge_replicas<BLOCK>(a, lds32, tid, err);
    const uint32_t words = bh_slab_words(a);
    uint32_t* slab = a.slabs + static_cast<size_t>(blockIdx.x) * words;
    for (uint32_t i = tid; i < words; i += BLOCK) {
      slab[i] = lds32[i];
    }
  } else {
    bh_packed_flush<BLOCK>(a, &s_cx, lds32, tid, err);
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

template <int KW, int VW, int U, int BLOCK>
__global__ __launch_bounds__(BLOCK, HDK_BH_PACKED_WAVES ? HDK_BH_PACKED_WAVES * 256 / BLOCK : 1) void hdk_scan_agg_bh_packed(BhPackedArgs a) {
  bh_packed_kernel_body<KW, VW, U, BLOCK, false, false>(a);
}
// the dense form (entry = key - min by the key column's statistics); 512-thread blocks, one per CU, for tables of 2 K - 4 K entries
template <int KW, int VW, int U, int BLOCK = kBhPackedBlock>
__global__ __launch_bounds__(BLOCK, HDK_BH_PACKED_WAVES ? HDK_BH_PACKED_WAVES * 256 / BLOCK : 1) void hdk_scan_agg_bh_dense(BhPackedArgs a) {
  bh_packed_kernel_body<KW, VW, U, BLOCK, true, false>(a);
}
// the plain forms (unfiltered plans: the reference's benchmark shapes)
template <int KW, int VW, int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hdk_scan_agg_bh_packed_plain(BhPackedArgs a) {
  bh_packed_kernel_body<KW, VW, U, BLOCK, false, true>(a);
}
template <int KW, int VW, int U, int BLOCK = kBhPackedBlock>
__global__ __launch_bounds__(BLOCK) void hdk_scan_agg_bh_dense_plain(BhPackedArgs a) {
  bh_packed_kernel_body<KW, VW, U, BLOCK, true, true>(a);
}

// ---- the fold of the scan blocks' slabs -----------------------------------------------------------------------------------
// Block (slice, group): the entries of bucket range `slice` of slabs group, group + fold_groups, ... merged into an LDS table
// with FULL-WIDTH words (a group's rows over all blocks do not fit the packed fields), then every group of that table folded
// into the output table.  Same tags, same buckets as the scan: a key sits in the same bucket range in every slab (keys pushed
// over a range's end by a full bucket show up in two fold blocks, which fold them one after the other).
// LDS: tags[cap] u32 | mm[cap] u64 | rows[cap] u64 | sum[cap] u64 | nulls[cap] u64
constexpr int kBhFoldBlock = 256;
__global__ __launch_bounds__(kBhFoldBlock) void hdk_bh_fold_slabs(BhPackedArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
  __shared__ BhExactCtx s_cx;
  const int tid = threadIdx.x;
  const uint32_t scan_cap = 1u << a.cap_log2;  // entries of a scan block's table (the slabs)
  // this block's own table: in the staged forms a block meets the keys of ONE slice (<= list_cap, a few more when probing
  // carried keys over a slice border) -- a table of 4 x list_cap entries instead of the scan's capacity (4 096 entries took
  // 144 initialising stores and 16 emit trips per thread for 64 keys); what it cannot hold is folded into the output at once
  uint32_t cap_log2 = a.cap_log2;
  if (a.fold_stage != 0) {
    const uint32_t want = 33 - __clz(a.list_cap * 2 - 1);  // log2(4 x list_cap) for a power of two
    cap_log2 = min(cap_log2, max(want, 6u));
  }
  const uint32_t cap = 1u << cap_log2;
  const uint32_t bmask = (cap >> 2) - 1;
  uint32_t* tags = lds32;
  uint64_t* mm = reinterpret_cast<uint64_t*>(lds32 + cap);
  unsigned long long* rows = reinterpret_cast<unsigned long long*>(lds32 + 3 * cap);
  unsigned long long* sum = reinterpret_cast<unsigned long long*>(lds32 + 5 * cap);
  unsigned long long* nulls = reinterpret_cast<unsigned long long*>(lds32 + 7 * cap);
  bh_exact_ctx_init(&s_cx, a, tid);
  const TableShape shape = table_shape(a.plan);
  int64_t* buf = a.kp.groupby_buf[0];
  for (uint32_t i = tid; i < cap; i += kBhFoldBlock) {
    tags[i] = kBhTagEmpty;
    mm[i] = (static_cast<uint64_t>(static_cast<uint32_t>(INT32_MIN)) << 32) | static_cast<uint32_t>(INT32_MAX);
    rows[i] = 0;
    sum[i] = 0;
    nulls[i] = 0;
  }
  __syncthreads();
  int32_t err = 0;
  const uint32_t slice = blockIdx.x % a.fold_slices, group = blockIdx.x / a.fold_slices;
  const uint32_t per_slice = scan_cap / a.fold_slices;  // entries of a slice (a multiple of 4, or the whole table)
  const uint32_t e_begin = slice * per_slice;
  const uint32_t words = bh_slab_words(a);
  // (slab, entry) pairs of this block's slabs group, group + fold_groups, ... side by side: every trip of a thread is an
  // independent load -- one slab after the other (a few entries each, 128 dependent trips) the kernel took 100 - 240 us, as long
  // as a third of the scan it follows
  auto merge = [&](int32_t key, const BhPartial& b) {
    // (the one-pass kernel's slabs: no bins; the bucket by the top hash bits of THIS table's size)
    const uint32_t bucket = cap_log2 > 2 ? (bh_tag_hash(key) >> (32 - (cap_log2 - 2))) & bmask : 0u;
    const int32_t e0 = bh_tag_probe(tags, key, bucket, bmask);
    if (e0 < 0) {  // this block's table is full: the partial goes into the output table as it is
      bh_fold_group_fn(a.plan, shape, s_cx.wl, buf, a.out_entry_count, s_cx.col_off, bh_key_word(a, key),
                       [&](int w) -> int64_t { return bh_partial_word(b, s_cx.wkind[w]); }, err);
      return;
    }
    atomicAdd(rows + e0, static_cast<unsigned long long>(b.rows));
    atomicAdd(sum + e0, static_cast<unsigned long long>(b.sum));
    if (b.nulls) atomicAdd(nulls + e0, static_cast<unsigned long long>(b.nulls));
    int32_t* mmw = reinterpret_cast<int32_t*>(mm) + 2 * e0;
    atomicMin(mmw, static_cast<int32_t>(b.mn));
    atomicMax(mmw + 1, static_cast<int32_t>(b.mx));
  };
  if (a.fold_stage == 2) {  // (grid = fold_slices: `group` is 0) the lists stage 1 left for this slice, side by side
    const uint32_t lc_log2 = 31 - __clz(a.list_cap);  // (a power of two, as cap and the slices are)
    const uint32_t total2 = a.fold_groups << lc_log2;
    constexpr int kLists = 4;
    for (uint32_t i0 = tid; i0 < total2; i0 += kBhFoldBlock * kLists) {
      uint64_t w[kLists][5];
      bool live[kLists];
#pragma unroll
      for (int j = 0; j < kLists; ++j) {
        const uint32_t idx = i0 + j * kBhFoldBlock;
        const uint32_t li = slice * a.fold_groups + (idx >> lc_log2), i = idx & (a.list_cap - 1);
        live[j] = idx < total2 && i < a.list_counts[li];
        const uint64_t* e = a.lists + (static_cast<size_t>(li) * a.list_cap + i) * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          w[j][k] = live[j] ? e[k] : 0;
        }
      }
#pragma unroll
      for (int j = 0; j < kLists; ++j) {
        if (live[j]) {
          BhPartial b;
          b.mn = static_cast<int32_t>(static_cast<uint32_t>(w[j][1]));
          b.mx = static_cast<int32_t>(static_cast<uint32_t>(w[j][1] >> 32));
          b.rows = static_cast<int64_t>(w[j][2]);
          b.sum = static_cast<int64_t>(w[j][3]);
          b.nulls = static_cast<int64_t>(w[j][4]);
          merge(static_cast<int32_t>(static_cast<uint32_t>(w[j][0])), b);
        }
      }
    }
  }
  const uint32_t nmine = (a.fold_stage != 2 && group < a.num_slabs) ? (a.num_slabs - group + a.fold_groups - 1) / a.fold_groups : 0u;
  const uint32_t ps_log2 = 31 - __clz(per_slice);  // (a power of two: cap and fold_slices are)
  const uint32_t total = nmine << ps_log2;
  constexpr int kInFlight = 4;
  for (uint32_t i0 = tid; i0 < total; i0 += kBhFoldBlock * kInFlight) {
    uint32_t tagv[kInFlight], eiv[kInFlight];
    const uint32_t* slabv[kInFlight];
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) {
      const uint32_t idx = i0 + j * kBhFoldBlock;
      const bool in = idx < total;
      const uint32_t sl = group + (idx >> ps_log2) * a.fold_groups;
      eiv[j] = e_begin + (idx & (per_slice - 1));
      slabv[j] = a.slabs + static_cast<size_t>(in ? sl : group) * words;
      tagv[j] = in ? slabv[j][eiv[j]] : kBhTagEmpty;
    }
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) {
      const uint32_t tag = tagv[j], ei = eiv[j];
      const uint32_t* slab = slabv[j];
      if (tag == kBhTagEmpty) {
        continue;
      }
      const BhPartial b = bh_decode(reinterpret_cast<const uint64_t*>(slab + a.off_packed)[ei], slab[a.off_nulls + ei],
                                    reinterpret_cast<const uint64_t*>(slab + a.off_mm)[ei]);
      merge(static_cast<int32_t>(tag), b);
    }
  }
  __syncthreads();
  __shared__ uint32_t s_list_n;
  if (tid == 0) {
    s_list_n = 0;
  }
  __syncthreads();
  const uint32_t li = slice * a.fold_groups + group;
  uint64_t* list = a.fold_stage == 1 ? a.lists + static_cast<size_t>(li) * a.list_cap * 5 : nullptr;
  for (uint32_t ei = tid; ei < cap; ei += kBhFoldBlock) {
    const uint32_t tag = tags[ei];
    if (tag == kBhTagEmpty) {
      continue;
    }
    BhPartial b;
    b.rows = static_cast<int64_t>(rows[ei]);
    b.sum = static_cast<int64_t>(sum[ei]);
    b.nulls = static_cast<int64_t>(nulls[ei]);
    b.mn = static_cast<int32_t>(static_cast<uint32_t>(mm[ei]));
    b.mx = static_cast<int32_t>(static_cast<uint32_t>(mm[ei] >> 32));
    if (a.fold_stage == 1) {
      const uint32_t pos = atomicAdd(&s_list_n, 1u);
      if (pos < a.list_cap) {  // (else: more keys than a slice's share met here -- probing crossed slice borders; folded right away)
        list[pos * 5] = tag;
        list[pos * 5 + 1] = mm[ei];
        list[pos * 5 + 2] = static_cast<uint64_t>(b.rows);
        list[pos * 5 + 3] = static_cast<uint64_t>(b.sum);
        list[pos * 5 + 4] = static_cast<uint64_t>(b.nulls);
        continue;
      }
    }
    bh_fold_group_fn(a.plan, shape, s_cx.wl, buf, a.out_entry_count, s_cx.col_off, bh_key_word(a, static_cast<int32_t>(tag)),
                     [&](int w) -> int64_t { return bh_partial_word(b, s_cx.wkind[w]); }, err);
  }
  if (a.fold_stage == 1) {
    __syncthreads();
    if (tid == 0) {
      a.list_counts[li] = s_list_n;
    }
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

// ---- tables beyond LDS: pass A, rows -> filters -> tuples [argument : key] -> 256 bins by the key's hash ------------------
// (the argument's NULL travels as INT32_MIN: no value inside the statistics the packed sum accepts is that small)
// (PROG: the plan's filters are an AND / OR / NOT program -- its own instantiation: the program's code costs the pass a resident
// block, 1.23 -> 1.8 ms per 256 M rows, whether a plan has one or not)
template <int VR, bool PROG = false>
__global__ __launch_bounds__(kPbBlock) void hdk_bh_scatter(BhPackedArgs a) {
  constexpr int kTile = kPbBlock * VR;
  __shared__ uint32_t s_cnt[kPbMaxBins];
  __shared__ uint4 s_run[kPbMaxBins];
  __shared__ uint32_t s_total;
  __shared__ int32_t s_watch;
  __shared__ BhExactCtx s_cx;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + kTile);
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
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  bool stop = false;
  for (uint64_t f = 0; f < nfrag && !stop; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTile - 1) / kTile;
    const int8_t* const* cols = a.kp.col_buffers[f];
    const int8_t* kb = cols[a.key_buf_idx];
    const int8_t* vb = a.has_val ? cols[a.val_buf_idx] : nullptr;
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      if (watch.flags) {
        if (const int32_t w = watch_poll_block(watch, &s_watch)) {
          err = w;
          stop = true;
          break;
        }
      }
      const int64_t row0 = (tile - frag_tile_begin) * kTile;
      int64_t row[VR];
      bool live[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        row[r] = row0 + static_cast<int64_t>(r) * kPbBlock + tid;
        live[r] = row[r] < nrows;
        row[r] = live[r] ? row[r] : 0;
      }
      if (a.nquals) {
        plain_quals_pass<VR, PROG>(a.q, a.nquals, cols, row, live, true);
      }
      int64_t k64[VR], v64[VR];
      if (a.key_width == 8) {
#pragma unroll
        for (int r = 0; r < VR; ++r) k64[r] = gload<int64_t>(kb, row[r], true);
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) k64[r] = gload<int32_t>(kb, row[r], true);
      }
      if (!a.has_val) {
#pragma unroll
        for (int r = 0; r < VR; ++r) v64[r] = 0;
      } else if (a.val_width == 8) {
#pragma unroll
        for (int r = 0; r < VR; ++r) v64[r] = gload<int64_t>(vb, row[r], true);
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) v64[r] = gload<int32_t>(vb, row[r], true);
      }
      int64_t tup[VR][1];
      uint32_t bin[VR];
      uint32_t slow = 0;
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const bool isnull = a.has_val && a.val_nullable && v64[r] == a.val_null;
        const bool is_knull = a.key_width == 8 && a.key_nullable && k64[r] == a.key_null;
        // what the 32-bit tuple fields (or the statistics) cannot carry goes through the exact per-row path right here
        const bool key_fits = a.key_width == 4 || (!is_knull && k64[r] >= a.key_min && k64[r] <= a.key_max);
        const bool val_fits = !a.has_val || isnull || (v64[r] >= a.val_min && v64[r] <= a.val_max);
        const int32_t k32 = static_cast<int32_t>(k64[r]);
        if (live[r] && (!key_fits || !val_fits || static_cast<uint32_t>(k32) == kBhTagEmpty)) {
          slow |= 1u << r;
          live[r] = false;
        }
        const int32_t v32 = isnull ? INT32_MIN : static_cast<int32_t>(v64[r]);
        tup[r][0] = static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(v32)) << 32) | static_cast<uint32_t>(k32));
        bin[r] = bh_tag_hash(k32) >> (32 - a.bins_log2);
      }
      while (__builtin_amdgcn_ballot_w64(slow != 0)) {
        if (slow) {
          const int j = __ffs(slow) - 1;
          slow &= slow - 1;
          int64_t kj = k64[0], vj = v64[0];
#pragma unroll
          for (int i = 1; i < VR; ++i) {
            kj = i == j ? k64[i] : kj;
            vj = i == j ? v64[i] : vj;
          }
          const bool vnull = a.has_val && a.val_nullable && vj == a.val_null;
          const int64_t kword = bh_exact_key_word(a, kj);
          const int32_t xe = bh_exact_row(a.plan, out_buf, a.out_entry_count, &s_cx, kword, vj, vnull);
          err = xe ? xe : err;
        }
      }
      pb_scatter_batch<1, VR>(
          tup, bin, live, s_cnt, s_run, &s_total, s_stage, s_binof, a.tuples,
          [&](uint32_t b, uint32_t n, uint32_t* base, uint32_t* nfit) {
            *base = atomicAdd(a.fill + (static_cast<size_t>(b) * kPbXcds + xcd) * kPbCursorStride, n);
            const uint64_t room = *base < a.cap ? a.cap - *base : 0;
            *nfit = n < room ? n : static_cast<uint32_t>(room);
          },
          [&](uint32_t b, uint64_t pos) { return (static_cast<uint64_t>(b) * kPbXcds + xcd) * a.cap + pos; },
          [&](const int64_t* t) {  // no room in the sub-slab (a hot key): the reference's scheme for this row
            const int32_t k32 = static_cast<int32_t>(static_cast<uint32_t>(t[0]));
            const int32_t v32 = static_cast<int32_t>(static_cast<uint64_t>(t[0]) >> 32);
            const bool vnull = a.has_val && a.val_nullable && v32 == INT32_MIN;
            const int32_t xe = bh_exact_row(a.plan, out_buf, a.out_entry_count, &s_cx, bh_key_word(a, k32), vnull ? a.val_null : v32, vnull);
            err = xe ? xe : err;
          });
    }
    frag_tile_begin += ntiles;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

// ---- pass B: one block per bin, the bin's share of the groups in LDS ---------------------------------------------------------
constexpr int kBhAggThreads = 1024;
__global__ __launch_bounds__(kBhAggThreads) void hdk_bh_aggregate(BhPackedArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
  __shared__ BhExactCtx s_cx;
  const int tid = threadIdx.x;
  bh_exact_ctx_init(&s_cx, a, tid);
  bh_packed_lds_init(lds32, a, tid, kBhAggThreads);
  __syncthreads();
  const BhHot hot = bh_hot(a);
  int64_t* const out_buf = a.kp.groupby_buf[0];
  const uint32_t bin = blockIdx.x;
  int32_t err = 0;
  uint32_t rows_since_flush = 0;
  constexpr int R = 2;
  constexpr int U = 4;
  constexpr uint32_t kStep = kBhAggThreads * R * U;
  for (uint32_t x = 0; x < kPbXcds; ++x) {
    const size_t sub = static_cast<size_t>(bin) * kPbXcds + x;
    const uint64_t n64 = min(static_cast<uint64_t>(a.fill[sub * kPbCursorStride]), a.cap);
    const uint32_t n = static_cast<uint32_t>(n64);
    const int64_t* t = a.tuples + sub * a.cap;
    for (uint32_t base = 0; base < n; base += kStep) {
      if (rows_since_flush + kStep > a.flush_rows) {  // (block-uniform)
        bh_packed_flush<kBhAggThreads>(a, &s_cx, lds32, tid, err);
        __syncthreads();
        bh_packed_lds_init(lds32, a, tid, kBhAggThreads);
        __syncthreads();
        rows_since_flush = 0;
      }
      rows_since_flush += kStep;
      bf_i64x2 tv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = base + (static_cast<uint32_t>(u) * kBhAggThreads + tid) * R;
        tv[u].x = 0;
        tv[u].y = 0;
        if (i + 1 < n) {  // (sub-slabs start 16-byte aligned: cap is even)
          tv[u] = __builtin_nontemporal_load(reinterpret_cast<const bf_i64x2*>(t + i));
        } else if (i < n) {
          tv[u].x = t[i];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = base + (static_cast<uint32_t>(u) * kBhAggThreads + tid) * R;
        const int64_t tw[R] = {tv[u].x, tv[u].y};
        int32_t key[R], val[R];
        bool ok[R], isnull[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          key[j] = static_cast<int32_t>(static_cast<uint32_t>(tw[j]));
          val[j] = static_cast<int32_t>(static_cast<uint64_t>(tw[j]) >> 32);
          ok[j] = i + j < n;
          isnull[j] = a.has_val && a.val_nullable && val[j] == INT32_MIN;
        }
        uint32_t slow = bh_packed_rows<R, false>(hot, lds32, key, val, ok, isnull, err);
        while (__builtin_amdgcn_ballot_w64(slow != 0)) {  // (a bin with more groups than its table holds: the reference's scheme)
          if (slow) {
            const int j = __ffs(slow) - 1;
            slow &= slow - 1;
            const int32_t kj = j ? key[1] : key[0], vj = j ? val[1] : val[0];
            const bool nj = j ? isnull[1] : isnull[0];
            const int32_t xe = bh_exact_row(a.plan, out_buf, a.out_entry_count, &s_cx, bh_key_word(a, kj), nj ? a.val_null : vj, nj);
            err = xe ? xe : err;
          }
        }
      }
    }
  }
  bh_packed_flush<kBhAggThreads>(a, &s_cx, lds32, tid, err);
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
