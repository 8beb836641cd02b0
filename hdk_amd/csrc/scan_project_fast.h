// scan_project_fast.h -- filter/project, specialised for the plain shape: conjunction of
// `outer column <cmp> literal` filters, targets that are plain outer columns, no join.
// Same output contract as hdk_scan_project (RowFuncBuilder.cpp:162-215, GroupByRuntime.cpp:248-272:
// row position + targets at densely claimed output rows, TOTAL_MATCHED counts every passing row, rows
// past MAX_MATCHED report a negative code).  What the interpreter cannot do:
//   * no per-batch interpretation at all: the filter columns are decoded and compared with compile-time
//     row loops (the interpreter is instruction-issue bound at ~2 ms per 256 M rows);
//   * target columns are loaded ONLY for passing rows (predicated gathers), so a selective filter reads
//     little more than the filter columns;
//   * no output claims at all: pass 1 (MODE 0) only counts the passing rows of each block's tiles, a
//     one-block scan turns the per-block counts into output offsets (and adds the total to
//     TOTAL_MATCHED), pass 2 (MODE 1) writes at offset + running count.  Pass 1 leaves its verdicts behind
//     as a selection bitmask (one bit per row, rows/8 bytes of stream-ordered scratch), so pass 2 reads
//     1 bit per row instead of decoding the filter columns again; without scratch (or past its end) it
//     re-evaluates the filter.
//     A claim per tile is a same-address atomic on the critical path of every tile (its round trip
//     under contention, not the bandwidth, set the pace: 1.7 ms per 256 M rows at any selectivity);
//     reading the filter columns twice costs 8 B/row here and removes it.  Output rows come out in
//     block-major row order, deterministically.
#pragma once
#include "watch.h"
#include "plain_quals.h"
#include "scan_project.h"

namespace hdk {

constexpr int kProjFastBlock = 512;
constexpr int kProjFastVR = 8;
constexpr int kProjFastMaxQuals = kMaxPlainQuals;

enum ProjFastSrc : int32_t {
  PF_SRC_OUTER = 0,    // a column of the outer table at the row
  PF_SRC_PAYLOAD = 1,  // payload word `col.buf_idx` of the joined row's fused entry (HDK_JOIN_ONE_TO_ONE_FUSED)
  PF_SRC_INNER = 2     // a column of the inner table at the joined row id (the reference's table layout)
};
struct ProjFastTarget {
  ProjFastCol col;
  int32_t slot_width;
  int32_t slot_off;  // row-wise: byte offset inside the row
  int32_t src;       // ProjFastSrc
  int32_t pad_;
};
struct ProjFastArgs {
  KernParams kp;
  uint32_t entry_count;  // == *MAX_MATCHED
  int32_t columnar;
  uint32_t row_size_quad;
  int32_t nquals;
  int32_t ntargets;
  ProjFastQual q[kProjFastMaxQuals];
  ProjFastTarget t[HDK_HIP_MAX_TARGETS];
  uint64_t col_off[HDK_HIP_MAX_TARGETS];  // columnar: byte offset of each target column
  uint32_t* block_counts;                 // [gridDim.x]: pass-1 counts, then exclusive offsets
  uint8_t* sel_mask;                      // pass 1 -> pass 2: one byte (VR pass bits) per thread and tile, or nullptr
  uint64_t sel_tiles;                     // tiles the mask has room for; later tiles re-evaluate the filter in pass 2
  int32_t pairs;                          // every filter column is 8 bytes wide: the *_pairs kernels (rows dealt two at a time)
  // one-pass form (hdk_scan_project_stream): one status word per batch of kProjFastGroup tiles, the batch ticket and the
  // word that arms the two-pass kernels behind it
  uint64_t* status;
  uint64_t status_cap;       // batches the status array has room for
  uint32_t* ticket;          // [0]: next batch; [1]: 1 = the input has more tiles than status words, take the two passes
  const uint32_t* run_if;    // two-pass kernels: nullptr = always run; else only when *run_if == 1
  uint32_t* mode;            // written by hdk_scan_project_offsets: 1 = the sparse writing pass runs, 2 = the dense one
  int32_t keep_cached;       // a filter column is also a target: its lines are gathered again right after the filter
  // ONE inner-like join on a one-to-one perfect-hash table (plain or fused): the counting pass probes the rows that
  // passed the filters -- only those -- and a row without a partner loses its verdict bit; the (sparse) writing pass
  // probes again for the joined columns it projects.  [bucketized_]hash_join_idx[_nullable|_bitwise]
  // (QE/GroupByRuntime.cpp:274-366) through JoinLoop's Singleton form (QE/IRCodegen.cpp:497-667).
  int32_t join;
  ProjFastCol jkey;          // the outer key column
  hdk_hip_join jn;
};

// the joined row of outer row `row`: its row id (-1: no partner) and the slot of its entry
HDK_DEV int64_t pf_join_probe(const ProjFastArgs& a, const int8_t* const* cols, const int64_t* join_hash_tables, int64_t row,
                              int64_t* slot_out) {
  const int64_t key = decode_col_g(cols[a.jkey.buf_idx], a.jkey.width, a.jkey.kind, row, false);
  int64_t k = key;
  int64_t maxk = a.jn.max_key;
  *slot_out = 0;
  if (a.jn.null_mode != HDK_JOIN_NULL_NONE && key == a.jn.null_val) {
    if (a.jn.null_mode == HDK_JOIN_NULL_NULLABLE) {
      return -1;
    }
    k = a.jn.translated_null;
    maxk = a.jn.translated_null;
  }
  if (k < a.jn.min_key || k > maxk) {
    return -1;
  }
  int64_t off = k - a.jn.min_key;
  if (a.jn.bucket > 1) {
    off /= a.jn.bucket;
  }
  *slot_out = off;
  if (a.jn.kind == HDK_JOIN_ONE_TO_ONE_FUSED) {
    return gload<int64_t>(reinterpret_cast<const int8_t*>(join_hash_tables), off * a.jn.fused_stride, false);
  }
  return gload<int32_t>(reinterpret_cast<const int8_t*>(join_hash_tables), off, false);
}

// Row of batch slot r.  R = 1: lane-striped (slot r of lane t = tile row r*BLOCK + t).  R = 2: slots 2k
// and 2k+1 are ADJACENT rows of an 8-byte column, one 16-byte load per lane.  Measured on this kernel the
// paired form did not pay as long as it went through the predicated loaders below (counting pass 0.54 vs ~0.6 ms
// per 256 M rows); the *_pairs kernels take it with pf_filter_full_tile_pairs for the counting pass, and the
// writing pass only reads the verdict bits.
template <int BLOCK, int R>
HDK_DEV int64_t pf_row(int64_t tile_row0, int tid, int r) {
  return R == 1 ? tile_row0 + static_cast<int64_t>(r) * BLOCK + tid
                : tile_row0 + (static_cast<int64_t>(r >> 1) * BLOCK + tid) * 2 + (r & 1);
}

// verdict bits after the join: a passing row without a partner fails (INNER / SEMI)
template <int R>
HDK_DEV uint32_t pf_join_filter(const ProjFastArgs& a, const int8_t* const* cols, int64_t row0, int tid, uint32_t bits) {
  uint32_t left = bits;
  while (left) {
    const int b = __ffs(left) - 1;
    left &= left - 1;
    int64_t slot;
    if (pf_join_probe(a, cols, a.kp.join_hash_tables, pf_row<kProjFastBlock, R>(row0, tid, b), &slot) < 0) {
      bits &= ~(1u << b);
    }
  }
  return bits;
}

typedef long long __attribute__((ext_vector_type(2))) pf_i64x2;

// VR rows of one column: the decoder switch is wave-uniform and sits outside the row loop
// PAIRS8: the rows are dealt in adjacent pairs AND the column is 8 bytes wide (always so for the filter columns of the
// *_pairs kernels; the dense writing pass asks per target column)
template <int VR, int BLOCK, int R, bool PAIRS8 = (R == 2)>
HDK_DEV void load_rows(const int8_t* buf, int width, int kind, int64_t row0, int tid, const bool (&live)[VR], bool nt,
                       int64_t (&out)[VR]) {
  if (PAIRS8) {  // 8-byte columns (int64 / double bits): pairs of adjacent rows
#pragma unroll
    for (int r = 0; r < VR; r += 2) {
      out[r] = 0;
      out[r + 1] = 0;
      const int64_t row = pf_row<BLOCK, R>(row0, tid, r);
      if (live[r + 1]) {  // both rows inside the fragment
        const pf_i64x2 x = gload<pf_i64x2>(buf, row >> 1, nt);
        out[r] = x.x;
        out[r + 1] = x.y;
      } else if (live[r]) {
        out[r] = gload<int64_t>(buf, row, nt);
      }
    }
    return;
  }
#define HDK_PF_ROWS(T, CONV)                                                   \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                             \
    out[r] = 0;                                                                \
    if (live[r]) {                                                             \
      const T x = gload<T>(buf, pf_row<BLOCK, R>(row0, tid, r), nt);          \
      out[r] = CONV;                                                           \
    }                                                                          \
  }
  if (kind == HDK_COL_DOUBLE) {
    HDK_PF_ROWS(int64_t, x)
  } else if (kind == HDK_COL_FLOAT) {
    HDK_PF_ROWS(float, double_to_bits(static_cast<double>(x)))
  } else if (kind == HDK_COL_UNSIGNED) {
    switch (width) {
      case 1: HDK_PF_ROWS(uint8_t, static_cast<int64_t>(x)) break;
      case 2: HDK_PF_ROWS(uint16_t, static_cast<int64_t>(x)) break;
      case 4: HDK_PF_ROWS(uint32_t, static_cast<int64_t>(x)) break;
      default: HDK_PF_ROWS(int64_t, x) break;
    }
  } else {
    switch (width) {
      case 1: HDK_PF_ROWS(int8_t, static_cast<int64_t>(x)) break;
      case 2: HDK_PF_ROWS(int16_t, static_cast<int64_t>(x)) break;
      case 4: HDK_PF_ROWS(int32_t, static_cast<int64_t>(x)) break;
      default: HDK_PF_ROWS(int64_t, x) break;
    }
  }
#undef HDK_PF_ROWS
}

HDK_DEV bool proj_fast_cmp(int cmp, bool fp, int64_t l, int64_t r) {
  if (fp) {
    const double a = bits_to_double(l), b = bits_to_double(r);
    switch (cmp) {
      case HDK_CMP_EQ: return a == b;
      case HDK_CMP_NE: return a != b;
      case HDK_CMP_LT: return a < b;
      case HDK_CMP_GT: return a > b;
      case HDK_CMP_LE: return a <= b;
      default: return a >= b;
    }
  }
  switch (cmp) {
    case HDK_CMP_EQ: return l == r;
    case HDK_CMP_NE: return l != r;
    case HDK_CMP_LT: return l < r;
    case HDK_CMP_GT: return l > r;
    case HDK_CMP_LE: return l <= r;
    default: return l >= r;
  }
}

// The filter over one FULL tile whose filter columns are all 8 bytes wide, rows dealt in adjacent pairs (R = 2): every
// lane reads 16 bytes per load, lane after lane contiguous, no bounds tests, and the verdicts live in ONE register per
// lane (bit r = slot r passes) -- a bool per row carried round the conjunct loop is kept by the compiler as a 0/1 byte
// in a VGPR and converted back and forth every trip.
template <bool NT = true>
HDK_DEV uint32_t pf_filter_full_tile_pairs(const ProjFastArgs& a, const int8_t* const* cols, int64_t row0, int tid) {
  constexpr int VR = kProjFastVR;
  uint32_t m = (1u << VR) - 1u;
#pragma unroll 1
  for (int qi = 0; qi < a.nquals; ++qi) {
    const ProjFastQual q = a.q[qi];
    const uint64_t b = reinterpret_cast<uintptr_t>(cols[q.col.buf_idx]) + static_cast<uint64_t>(row0) * 8;
    const uint32_t b_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b));
    const uint32_t b_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b >> 32));
    const __attribute__((address_space(1))) int8_t* base =
        reinterpret_cast<const __attribute__((address_space(1))) int8_t*>((static_cast<uint64_t>(b_hi) << 32) | b_lo);
    int64_t v[VR];
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) {
      const uint32_t off = static_cast<uint32_t>(u * kProjFastBlock + tid) * 16u;
      const __attribute__((address_space(1))) pf_i64x2* at = reinterpret_cast<const __attribute__((address_space(1))) pf_i64x2*>(base + off);
      const pf_i64x2 x = NT ? __builtin_nontemporal_load(at) : *at;
      v[2 * u] = x.x;
      v[2 * u + 1] = x.y;
    }
    const bool fp = q.fp != 0;  // an 8-byte column compared as doubles is a double column (col_fp == fp here)
    uint32_t fail = 0;
    if (q.nullable) {
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const bool isnull = fp ? bits_to_double(v[r]) == bits_to_double(q.null_val) : v[r] == q.null_val;
        fail |= isnull ? (1u << r) : 0u;
      }
    }
#define HDK_PF_FAIL(OP)                                                                               \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                                                     \
    const bool ok = fp ? (bits_to_double(v[r]) OP bits_to_double(q.rhs)) : (v[r] OP q.rhs);            \
    fail |= ok ? 0u : (1u << r);                                                                       \
  }
    switch (q.cmp) {
      case HDK_CMP_EQ: HDK_PF_FAIL(==) break;
      case HDK_CMP_NE: HDK_PF_FAIL(!=) break;
      case HDK_CMP_LT: HDK_PF_FAIL(<) break;
      case HDK_CMP_GT: HDK_PF_FAIL(>) break;
      case HDK_CMP_LE: HDK_PF_FAIL(<=) break;
      default: HDK_PF_FAIL(>=) break;
    }
#undef HDK_PF_FAIL
    m &= ~fail;
  }
  return m;
}

// The filter over one tile in general: any column width, ragged tiles (rows past `nrows` fail), lane-striped or paired
// rows.  Bit r of the result = slot r passes.
template <int R>
HDK_DEV uint32_t pf_filter_tile(const ProjFastArgs& a, const int8_t* const* cols, int64_t row0, int64_t nrows, int tid, bool nt) {
  constexpr int VR = kProjFastVR;
  bool pass[VR];
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    pass[r] = pf_row<kProjFastBlock, R>(row0, tid, r) < nrows;
  }
  // decode + compare, all VR loads of a column in flight together
  for (int qi = 0; qi < a.nquals; ++qi) {
    const ProjFastQual q = a.q[qi];
    const int8_t* qb = cols[q.col.buf_idx];
    int64_t v[VR];
    load_rows<VR, kProjFastBlock, R>(qb, q.col.width, q.col.kind, row0, tid, pass, nt, v);
    const bool fpc = q.fp != 0;
    const bool col_fp = q.col_fp != 0;
    const bool nullable = q.nullable != 0;
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const bool isnull = nullable && (col_fp ? bits_to_double(v[r]) == bits_to_double(q.null_val) : v[r] == q.null_val);
      pass[r] = pass[r] && !isnull;
      if (fpc && !col_fp) {
        v[r] = double_to_bits(static_cast<double>(v[r]));
      }
    }
    // the comparison operator is wave-uniform: one switch per batch, the row loop inside each case
#define HDK_PF_CMP(OP)                                                                                   \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                                                       \
      pass[r] = pass[r] && (fpc ? (bits_to_double(v[r]) OP bits_to_double(q.rhs)) : (v[r] OP q.rhs));      \
    }
    switch (q.cmp) {
      case HDK_CMP_EQ: HDK_PF_CMP(==) break;
      case HDK_CMP_NE: HDK_PF_CMP(!=) break;
      case HDK_CMP_LT: HDK_PF_CMP(<) break;
      case HDK_CMP_GT: HDK_PF_CMP(>) break;
      case HDK_CMP_LE: HDK_PF_CMP(<=) break;
      default: HDK_PF_CMP(>=) break;
    }
#undef HDK_PF_CMP
  }
  uint32_t bits = 0;
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    bits |= (pass[r] ? 1u : 0u) << r;
  }
  return bits;
}

constexpr int kProjFastGroup = 4;  // writing pass: tiles handled per block-wide scan (one barrier per group, not per tile)

// ---- writing a group of kProjFastGroup tiles -----------------------------------------------------------------------
// Output order inside a group: tile after tile, inside a tile wave after wave, lane after lane, slot after slot.  So the
// rows a WAVE passes from one tile land in one contiguous range of output rows, which is what lets a wave choose, tile by
// tile, between two ways of getting them there:
//   sparse -- few rows pass: the wave lists its passing rows in an LDS strip, lane j then gathers the j-th row from each
//             target column and stores it (every gather and store has all its lanes busy; the target columns are read
//             only where rows pass);
//   dense  -- an eighth of the rows or more pass, so every 128-byte line of a target column is needed anyway: the wave
//             reads its 8 slots per lane of the column with the coalesced loads of the filter pass, compacts the VALUES
//             through LDS and stores them -- no row list, no gathers (at 50 % selectivity the gathers ran at 3.8 TB/s of
//             in + out bytes; measured in DESIGN.md 3.6).
constexpr int kProjFastG = kProjFastGroup;
constexpr uint32_t kProjDenseMin = kWave * kProjFastVR / 8;  // passing rows of a wave and tile from which the dense form pays

struct PfWavePos {
  uint32_t lane_excl[kProjFastG];  // passing rows of the lower lanes of this wave, per tile
  uint32_t wave_tot[kProjFastG];   // the wave's passing rows, per tile
};

// per-tile wave scans of the verdict bits (bit g * VR + r = slot r of tile g passes)
HDK_DEV void pf_scan_tiles(uint32_t bits, int lane, PfWavePos& p) {
#pragma unroll
  for (int g = 0; g < kProjFastG; ++g) {
    const uint32_t mine = __builtin_popcount((bits >> (g * kProjFastVR)) & ((1u << kProjFastVR) - 1u));
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const uint32_t n = __shfl_up(incl, d, kWave);
      if (lane >= d) {
        incl += n;
      }
    }
    p.lane_excl[g] = incl - mine;
    p.wave_tot[g] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), kWave - 1));  // (an SGPR)
  }
}

// lane j takes the j-th entry of the wave's strip (entry: tile of the group << 12 | row inside the tile), U entries per
// lane and trip with the type switches outside the entry loops: U gathers back to back, then U stores
constexpr int kProjFastU = 2;
template <typename RowOf, typename PosOf, typename ColOf>
HDK_DEV void pf_project_rows(const ProjFastArgs& a, int64_t* buf, const uint16_t* strip, uint32_t count, uint32_t max_matched,
                             int lane, RowOf row_of, PosOf pos_of, ColOf col_of, int32_t& slots_err) {
  constexpr int U = kProjFastU;
  const bool columnar = a.columnar != 0;
  const size_t rq = a.row_size_quad;
  for (uint32_t j0 = 0; j0 < count; j0 += U * kWave) {
    uint32_t e[U];
    int64_t row[U];
    uint64_t pos[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t j = j0 + u * kWave + lane;
      ok[u] = j < count;
      e[u] = ok[u] ? strip[j] : 0;
      row[u] = row_of(e[u]);
      pos[u] = pos_of(j, e[u]);
      if (ok[u] && pos[u] >= max_matched) {
        slots_err = -1 - static_cast<int32_t>(row[u] & 0x3fffffff);
        ok[u] = false;
      }
      if (ok[u]) {
        buf[columnar ? static_cast<size_t>(pos[u]) : static_cast<size_t>(pos[u]) * rq] = row[u];
      }
    }
    for (int ti = 0; ti < a.ntargets; ++ti) {
      const ProjFastTarget t = a.t[ti];
      int8_t* base = columnar ? reinterpret_cast<int8_t*>(buf) + a.col_off[ti] : reinterpret_cast<int8_t*>(buf) + t.slot_off;
      const size_t stride = columnar ? static_cast<size_t>(t.slot_width) : rq * 8;
      int64_t v[U];
#define HDK_PF_GATHER(T, CONV)                                          \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                       \
    v[u] = 0;                                                           \
    if (ok[u]) {                                                        \
      const T x = gload<T>(col_of(e[u] >> 12, ti), row[u], true);       \
      v[u] = CONV;                                                      \
    }                                                                   \
  }
      if (t.col.kind == HDK_COL_DOUBLE) {
        HDK_PF_GATHER(int64_t, x)
      } else if (t.col.kind == HDK_COL_UNSIGNED) {
        switch (t.col.width) {
          case 1: HDK_PF_GATHER(uint8_t, static_cast<int64_t>(x)) break;
          case 2: HDK_PF_GATHER(uint16_t, static_cast<int64_t>(x)) break;
          case 4: HDK_PF_GATHER(uint32_t, static_cast<int64_t>(x)) break;
          default: HDK_PF_GATHER(int64_t, x) break;
        }
      } else {  // (float columns are not in this kernel's shape: match_project_fast)
        switch (t.col.width) {
          case 1: HDK_PF_GATHER(int8_t, static_cast<int64_t>(x)) break;
          case 2: HDK_PF_GATHER(int16_t, static_cast<int64_t>(x)) break;
          case 4: HDK_PF_GATHER(int32_t, static_cast<int64_t>(x)) break;
          default: HDK_PF_GATHER(int64_t, x) break;
        }
      }
#undef HDK_PF_GATHER
#define HDK_PF_PUT(T)                                                                                       \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                           \
    if (ok[u]) {                                                                                            \
      *reinterpret_cast<T*>(base + static_cast<size_t>(pos[u]) * stride) = static_cast<T>(v[u]);            \
    }                                                                                                       \
  }
      switch (t.slot_width) {
        case 1: HDK_PF_PUT(int8_t) break;
        case 2: HDK_PF_PUT(int16_t) break;
        case 4: HDK_PF_PUT(int32_t) break;
        default: HDK_PF_PUT(int64_t) break;
      }
#undef HDK_PF_PUT
    }
  }
}

// the wave's passing rows of the tiles in the mask `tiles` into its strip, tile after tile; goff[g] = where tile g's entries start
template <int R>
HDK_DEV void pf_fill_strip(uint16_t* strip, uint32_t bits, int tid, const PfWavePos& p, uint32_t tiles,
                           uint32_t (&goff)[kProjFastG], uint32_t* count) {
  constexpr int VR = kProjFastVR;
  uint32_t at = 0;
#pragma unroll
  for (int g = 0; g < kProjFastG; ++g) {
    goff[g] = at;
    if ((tiles >> g) & 1u) {
      uint32_t j = at + p.lane_excl[g];
      uint32_t left = (bits >> (g * VR)) & ((1u << VR) - 1u);
      while (left) {
        const int b = __ffs(left) - 1;
        left &= left - 1;
        strip[j++] = static_cast<uint16_t>((g << 12) | static_cast<int>(pf_row<kProjFastBlock, R>(0, tid, b)));
      }
      at += p.wave_tot[g];
    }
  }
  *count = at;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One group: `bits` of this lane, the wave's positions, wave_base[g] = output row of the wave's first passing row of tile
// g, trow0[g] = row (inside its fragment) of tile g's slot 0, full_mask bit g = tile g is a whole tile, col_of(g, ti) =
// target ti's column buffer for tile g's fragment.  wave_lds: 4 KB of LDS of the wave's own.
template <int R, typename ColOf>
HDK_DEV void pf_write_group(const ProjFastArgs& a, int64_t* buf, int64_t* wave_lds, uint32_t bits, const PfWavePos& p,
                            const uint64_t (&wave_base)[kProjFastG], const int64_t (&trow0)[kProjFastG], uint32_t full_mask,
                            ColOf col_of, uint32_t max_matched, int tid, int lane, int32_t& slots_err) {
  constexpr int VR = kProjFastVR;
  constexpr int G = kProjFastG;
  uint16_t* strip = reinterpret_cast<uint16_t*>(wave_lds);
  uint32_t dense_mask = 0;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    dense_mask |= ((full_mask >> g) & 1u) && p.wave_tot[g] >= kProjDenseMin ? 1u << g : 0u;
  }
  auto row_of = [&](uint32_t e) {
    const uint32_t g = e >> 12;
    return (g == 0 ? trow0[0] : g == 1 ? trow0[1] : g == 2 ? trow0[2] : trow0[3]) + (e & 4095u);
  };
  static_assert(G == 4, "the selects below spell out four tiles");
  auto sel32 = [](const uint32_t (&x)[G], int g) { return g == 0 ? x[0] : g == 1 ? x[1] : g == 2 ? x[2] : x[3]; };
  auto sel64 = [](const uint64_t (&x)[G], int g) { return g == 0 ? x[0] : g == 1 ? x[1] : g == 2 ? x[2] : x[3]; };
  auto sel64s = [](const int64_t (&x)[G], int g) { return g == 0 ? x[0] : g == 1 ? x[1] : g == 2 ? x[2] : x[3]; };
  const bool columnar = a.columnar != 0;
  const size_t rq = a.row_size_quad;
  // ---- dense tiles, one after the other (g is wave-uniform: the per-tile values come out of their arrays by select, so
  // that the loop need not be unrolled -- four copies of the body cost 60 registers) -------------------------------------
  if (dense_mask) {
    bool live[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      live[r] = true;
    }
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      if (!((dense_mask >> g) & 1u)) {
        continue;
      }
      const uint32_t n = sel32(p.wave_tot, g);
      const uint32_t m = (bits >> (g * VR)) & ((1u << VR) - 1u);
      const uint32_t at0 = sel32(p.lane_excl, g);
      const uint64_t out0 = sel64(wave_base, g);
      const int64_t r0 = sel64s(trow0, g);
      // the values of the lane's passing slots -> LDS at the lane's positions; then lane j stores the j-th value
      auto compact = [&](const int64_t (&v)[VR]) {
        uint32_t at = at0;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if ((m >> r) & 1u) {
            wave_lds[at++] = v[r];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      };
      int64_t v[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        v[r] = r0 + pf_row<kProjFastBlock, R>(0, tid, r);
      }
      compact(v);
      for (uint32_t j = lane; j < n; j += kWave) {
        const int64_t row = wave_lds[j];
        const uint64_t pos = out0 + j;
        if (pos >= max_matched) {
          slots_err = -1 - static_cast<int32_t>(row & 0x3fffffff);
        } else {
          buf[columnar ? static_cast<size_t>(pos) : static_cast<size_t>(pos) * rq] = row;
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int ti = 0; ti < a.ntargets; ++ti) {
        const ProjFastTarget t = a.t[ti];
        int8_t* base = columnar ? reinterpret_cast<int8_t*>(buf) + a.col_off[ti] : reinterpret_cast<int8_t*>(buf) + t.slot_off;
        const size_t stride = columnar ? static_cast<size_t>(t.slot_width) : rq * 8;
        if (R == 2 && t.col.width == 8 && t.col.kind != HDK_COL_FLOAT) {
          load_rows<VR, kProjFastBlock, R, true>(col_of(g, ti), t.col.width, t.col.kind, r0, tid, live, true, v);
        } else {
          load_rows<VR, kProjFastBlock, R, false>(col_of(g, ti), t.col.width, t.col.kind, r0, tid, live, true, v);
        }
        compact(v);
#define HDK_PF_PUT(T)                                                                                       \
  for (uint32_t j = lane; j < n; j += kWave) {                                                              \
    if (out0 + j < max_matched) {                                                                           \
      *reinterpret_cast<T*>(base + static_cast<size_t>(out0 + j) * stride) = static_cast<T>(wave_lds[j]);   \
    }                                                                                                       \
  }
        switch (t.slot_width) {
          case 1: HDK_PF_PUT(int8_t) break;
          case 2: HDK_PF_PUT(int16_t) break;
          case 4: HDK_PF_PUT(int32_t) break;
          default: HDK_PF_PUT(int64_t) break;
        }
#undef HDK_PF_PUT
        __builtin_amdgcn_wave_barrier();  // the values are overwritten by the next column
      }
    }
  }
  // ---- the other tiles together: strip, gather, store ----------------------------------------------------------------
  uint32_t sparse_mask = 0;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    sparse_mask |= (!((dense_mask >> g) & 1u) && p.wave_tot[g] != 0) ? 1u << g : 0u;
  }
  if (sparse_mask) {
    uint32_t goff[G], count;
    pf_fill_strip<R>(strip, bits, tid, p, sparse_mask, goff, &count);
    const uint64_t adj0 = wave_base[0] - goff[0], adj1 = wave_base[1] - goff[1], adj2 = wave_base[2] - goff[2],
                   adj3 = wave_base[3] - goff[3];
    pf_project_rows(a, buf, strip, count, max_matched, lane, row_of,
                    [&](uint32_t j, uint32_t e) {
                      const uint32_t g = e >> 12;
                      return (g == 0 ? adj0 : g == 1 ? adj1 : g == 2 ? adj2 : adj3) + j;
                    },
                    col_of, slots_err);
    __builtin_amdgcn_wave_barrier();  // the strip is rewritten by the next group
  }
}

// MODE 0: count passing rows per block; 1: write them.  R: see pf_row.  DENSE (writing pass): positions tile after tile
// and pf_write_group, for launches in which an eighth of the rows or more pass (decided on the device by
// hdk_scan_project_offsets); else positions lane after lane over the whole group, one strip, gathers -- the order does
// not matter to anyone, but the two forms cost each other registers (81 against 129) and a selective filter (1 %: 0.56 ms
// per 256 M rows against 0.86) wants the occupancy.
template <int MODE, int R, bool DENSE = false>
HDK_DEV void scan_project_direct_body(const ProjFastArgs& a) {
  constexpr int VR = kProjFastVR;
  constexpr int G = kProjFastGroup;
  constexpr int kWaves = kProjFastBlock / kWave;
  __shared__ uint32_t s_wave_tot[2][G][kWaves];
  // MODE 1: 4 KB per wave: the strip of its passing rows, or the values of one tile and column (pf_write_group)
  __shared__ int64_t s_wave_lds[MODE == 1 ? kWaves : 1][MODE == 1 ? kWave * kProjFastVR : 1];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const uint32_t max_matched = static_cast<uint32_t>(*a.kp.max_matched);
  constexpr int64_t kTileRows = static_cast<int64_t>(kProjFastBlock) * VR;
  int64_t* buf = a.kp.groupby_buf[0];
  const bool columnar = a.columnar != 0;
  int32_t slots_err = 0;
  uint32_t iter = 0;
  uint32_t counted = 0;                                        // MODE 0: this thread's passing rows
  uint32_t running = MODE == 1 ? a.block_counts[blockIdx.x] : 0;  // MODE 1: next output row of the block

  if (MODE == 0 && a.run_if && *a.run_if != 1) {
    return;  // (armed behind the one-pass kernel, which took the launch)
  }
  if (MODE == 1 && *a.mode != (DENSE ? 2u : 1u)) {
    return;  // (the other writing pass has this launch, or the one-pass kernel had it)
  }
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  __shared__ int32_t s_watch;
  const Watch watch = watch_begin(a.kp);
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    const int8_t* const* cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      if (watch.flags) {  // (this loop has barriers: the block decides together)
        if (const int32_t w_ = watch_poll_block(watch, &s_watch)) {
          if (tid == 0) {
            record_error(a.kp.error_code, w_);
          }
          tile = INT64_MAX - gridDim.x;
          break;
        }
      }
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      const bool masked = a.sel_mask != nullptr && static_cast<uint64_t>(tile) < a.sel_tiles;  // block-uniform
      uint8_t* mask_at = a.sel_mask + static_cast<size_t>(tile) * kProjFastBlock + tid;
      if (MODE == 0 && R == 2 && row0 + kTileRows <= nrows) {  // the counting pass over a full tile: straight line
        uint32_t m = pf_filter_full_tile_pairs(a, cols, row0, tid);
        if (a.join) {
          m = pf_join_filter<R>(a, cols, row0, tid, m);
        }
        if (masked) {
          *mask_at = static_cast<uint8_t>(m);
        }
        counted += __builtin_popcount(m);
        continue;
      }
      // verdict bits of this lane: bit g * VR + r = slot r of the group's g-th tile passes
      uint32_t bits = 0;
      int group = 1;
      const int64_t group_last = tile + static_cast<int64_t>(G - 1) * gridDim.x;
      if (MODE == 1 && masked && group_last < frag_tile_begin + ntiles && static_cast<uint64_t>(group_last) < a.sel_tiles) {
        // the block's next G tiles of this fragment all have their verdicts in the mask: one scan, one barrier
        group = G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          bits |= static_cast<uint32_t>(__builtin_nontemporal_load(mask_at + static_cast<size_t>(g) * gridDim.x * kProjFastBlock))
                  << (g * VR);
        }
      } else if (MODE == 1 && masked) {
        bits = __builtin_nontemporal_load(mask_at);
      } else {
        bits = pf_filter_tile<R>(a, cols, row0, nrows, tid, MODE == 1);  // pass 1 leaves the lines cached
        if (!DENSE && a.join) {  // (a join plan never takes the dense writing pass: its registers stay what they were)
          bits = pf_join_filter<R>(a, cols, row0, tid, bits);
        }
        if (MODE == 0 && masked) {
          *mask_at = static_cast<uint8_t>(bits);
        }
      }
      if (MODE == 0) {
        counted += __builtin_popcount(bits);
        continue;
      }
      if (!DENSE) {
        // ---- selection vector -> dense output positions inside the block's range, lane after lane ------------------
        const uint32_t mine = __builtin_popcount(bits);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint32_t n = __shfl_up(incl, d, kWave);
          if (lane >= d) {
            incl += n;
          }
        }
        uint32_t out_pos = running + incl - mine;
        uint32_t tile_total = 0;
        {
          const int par = static_cast<int>(iter & 1);  // double buffer: a fast wave may already be one group ahead
          if (lane == kWave - 1) {
            s_wave_tot[par][0][wave] = incl;
          }
          __syncthreads();
#pragma unroll
          for (int w = 0; w < kWaves; ++w) {
            const uint32_t t = s_wave_tot[par][0][w];
            tile_total += t;
            if (w < wave) {
              out_pos += t;
            }
          }
          ++iter;
        }
        running += tile_total;
        // ---- compact: the wave's passing rows, in output order, into its LDS strip ----------------------------------
        // Lane j of the wave then handles the j-th passing row: every gather and store instruction has all its
        // lanes busy and consecutive lanes write consecutive output rows (coalesced), instead of VR predicated
        // load/store groups per lane in which a selective filter leaves 1 lane in 100 active.
        uint16_t* strip = reinterpret_cast<uint16_t*>(s_wave_lds[wave]);
        const uint32_t wave_total = __shfl(incl, kWave - 1, kWave);
        const uint32_t wave_base = out_pos - (incl - mine);  // first output row of this wave's rows
        {
          uint32_t j = incl - mine;
          uint32_t left = bits;
          while (left) {
            const int b = __ffs(left) - 1;
            left &= left - 1;
            strip[j++] = static_cast<uint16_t>(((b / VR) << 12) | static_cast<int>(pf_row<kProjFastBlock, R>(0, tid, b % VR)));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // row of a strip entry: the group's g-th tile is `g * gridDim.x` tiles further on
        const int64_t group_stride_rows = static_cast<int64_t>(gridDim.x) * kTileRows;
        auto row_of = [&](uint32_t e) { return row0 + static_cast<int64_t>(e >> 12) * group_stride_rows + (e & 4095u); };
        // ---- project: row position, then each target column -----------------------------------------------------
        const size_t rq = a.row_size_quad;
        for (uint32_t j0 = 0; j0 < wave_total; j0 += kWave) {
          const uint32_t j = j0 + lane;
          if (j < wave_total) {
            const int64_t row = row_of(strip[j]);
            const uint32_t pos = wave_base + j;
            if (pos >= max_matched) {
              slots_err = -1 - static_cast<int32_t>(row & 0x3fffffff);
            } else {
              buf[columnar ? static_cast<size_t>(pos) : static_cast<size_t>(pos) * rq] = row;
            }
          }
        }
        for (int ti = 0; ti < a.ntargets; ++ti) {
          const ProjFastTarget t = a.t[ti];
          const int8_t* tb = t.src == PF_SRC_PAYLOAD ? nullptr : cols[t.col.buf_idx];
          int8_t* base = columnar ? reinterpret_cast<int8_t*>(buf) + a.col_off[ti] : reinterpret_cast<int8_t*>(buf) + t.slot_off;
          const size_t stride = columnar ? static_cast<size_t>(t.slot_width) : rq * 8;
          for (uint32_t j0 = 0; j0 < wave_total; j0 += kWave) {
            const uint32_t j = j0 + lane;
            const uint32_t pos = wave_base + j;
            if (j < wave_total && pos < max_matched) {
              const int64_t row = row_of(strip[j]);
              int64_t v;
              if (t.src == PF_SRC_OUTER) {
                v = decode_col_g(tb, t.col.width, t.col.kind, row, true);
              } else {  // a column of the joined row (every row of the strip has a partner: the counting pass saw to it)
                int64_t slot;
                const int64_t rid = pf_join_probe(a, cols, a.kp.join_hash_tables, row, &slot);
                v = t.src == PF_SRC_PAYLOAD
                        ? gload<int64_t>(reinterpret_cast<const int8_t*>(a.kp.join_hash_tables), slot * a.jn.fused_stride + t.col.buf_idx, false)
                        : decode_col_g(tb, t.col.width, t.col.kind, rid < 0 ? 0 : rid, false);
              }
              int8_t* dst = base + static_cast<size_t>(pos) * stride;
              switch (t.slot_width) {
                case 1: *reinterpret_cast<int8_t*>(dst) = static_cast<int8_t>(v); break;
                case 2: *reinterpret_cast<int16_t*>(dst) = static_cast<int16_t>(v); break;
                case 4: *reinterpret_cast<int32_t*>(dst) = static_cast<int32_t>(v); break;
                default: *reinterpret_cast<int64_t*>(dst) = v; break;
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();  // the strip is rewritten by the next group
      } else {
        // ---- selection vector -> dense output positions inside the block's range (order: see pf_write_group) ------------
        PfWavePos p;
        pf_scan_tiles(bits, lane, p);
        const int par = static_cast<int>(iter & 1);  // double buffer: a fast wave may already be one group ahead
        if (lane < G) {
          s_wave_tot[par][lane][wave] = lane == 0 ? p.wave_tot[0] : lane == 1 ? p.wave_tot[1] : lane == 2 ? p.wave_tot[2] : p.wave_tot[3];
        }
        __syncthreads();
        ++iter;
        uint64_t wave_base[G];
        {
          uint32_t at = running;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
              const uint32_t t = s_wave_tot[par][g][w];
              total += t;
              before += w < wave ? t : 0;
            }
            wave_base[g] = at + before;
            at += total;
          }
          running = at;
        }
        // row 0 of the group's g-th tile: `g * gridDim.x` tiles further on in the same fragment
        const int64_t group_stride_rows = static_cast<int64_t>(gridDim.x) * kTileRows;
        int64_t trow0[G];
        uint32_t full_mask = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          trow0[g] = row0 + g * group_stride_rows;
          full_mask |= (g < group && trow0[g] + kTileRows <= nrows) ? 1u << g : 0u;
        }
        pf_write_group<R>(a, buf, s_wave_lds[wave], bits, p, wave_base, trow0, full_mask,
                          [&](uint32_t, int ti) { return cols[a.t[ti].col.buf_idx]; }, max_matched, tid, lane, slots_err);
      }
      tile += static_cast<int64_t>(group - 1) * gridDim.x;
    }
    frag_tile_begin += ntiles;
  }
  if (MODE == 0) {
    // block total -> block_counts[block]
    for (int d = kWave / 2; d > 0; d >>= 1) {
      counted += __shfl_down(counted, d, kWave);
    }
    if (lane == 0) {
      s_wave_tot[0][0][wave] = counted;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t t = 0;
      for (int w = 0; w < kWaves; ++w) {
        t += s_wave_tot[0][0][w];
      }
      a.block_counts[blockIdx.x] = t;
    }
  } else if (slots_err) {
    atomicCAS(a.kp.error_code, 0, slots_err);  // negative = ran out of slots (benign under a LIMIT)
  }
}

extern "C" __global__ __launch_bounds__(kProjFastBlock) void hdk_scan_project_count(ProjFastArgs a) {
  scan_project_direct_body<0, 1>(a);
}
extern "C" __global__ __launch_bounds__(kProjFastBlock) void hdk_scan_project_direct(ProjFastArgs a) {
  scan_project_direct_body<1, 1>(a);
}
// rows dealt in adjacent pairs: every filter column is 8 bytes wide (match_project_fast: pairs)
extern "C" __global__ __launch_bounds__(kProjFastBlock) void hdk_scan_project_count_pairs(ProjFastArgs a) {
  scan_project_direct_body<0, 2>(a);
}
extern "C" __global__ __launch_bounds__(kProjFastBlock) void hdk_scan_project_direct_pairs(ProjFastArgs a) {
  scan_project_direct_body<1, 2>(a);
}
// the writing pass for launches in which many rows pass (see scan_project_direct_body)
#ifndef HDK_PROJ_DENSE_WAVES
#define HDK_PROJ_DENSE_WAVES 0  // waves per SIMD the dense writing pass is held to (0: the compiler's choice)
#endif
extern "C" __global__ __launch_bounds__(kProjFastBlock, HDK_PROJ_DENSE_WAVES) void hdk_scan_project_dense(ProjFastArgs a) {
  scan_project_direct_body<1, 1, true>(a);
}
extern "C" __global__ __launch_bounds__(kProjFastBlock, HDK_PROJ_DENSE_WAVES) void hdk_scan_project_dense_pairs(ProjFastArgs a) {
  scan_project_direct_body<1, 2, true>(a);
}
// ---- one pass: filter -> dense output positions by decoupled look-back over batches -> write ------------------
// The two passes above read the filter columns twice (or write and read a bitmask) and need a grid-wide step between
// them.  Here a block takes a BATCH of kProjFastGroup consecutive tiles (a ticket, so that every earlier batch belongs
// to a block that is already running), evaluates the filter, publishes the batch's count in its status word, looks
// back over the status words of the batches before it until it meets one that already knows its prefix, publishes its
// own prefix and writes.  Nothing waits for anything but counts, which depend on nothing: no chain.  Output rows come
// out batch after batch, inside a batch wave by wave and lane by lane: deterministic, as with the two passes.
// status word: [63:62] 0 = nothing yet, 1 = the batch's own count, 2 = the count of everything up to and including the
// batch (= the output row after its last one); [61:0] the count.
constexpr uint64_t kProjStatusCount = 1ull << 62, kProjStatusPrefix = 2ull << 62, kProjStatusValue = (1ull << 62) - 1;

HDK_DEV uint64_t pf_look_back(const uint64_t* status, uint32_t batch, int lane) {
  uint64_t excl = 0;
  for (int64_t idx = static_cast<int64_t>(batch) - 1;; idx -= kWave) {
    const int64_t i = idx - lane;
    uint64_t st = kProjStatusPrefix;  // (before batch 0: nothing; batch 0's own prefix carries what TOTAL_MATCHED held)
    if (i >= 0) {
      for (;;) {
        st = __hip_atomic_load(status + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st >> 62) {
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const uint64_t known = __ballot((st >> 62) == 2);
    const int first = known ? __ffsll(static_cast<unsigned long long>(known)) - 1 : kWave;
    uint64_t v = lane <= first ? (st & kProjStatusValue) : 0;
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
      v += __shfl_xor(v, d, kWave);
    }
    excl += v;
    if (known) {
      return excl;
    }
  }
}

template <int R>
HDK_DEV void scan_project_stream_body(const ProjFastArgs& a) {
  constexpr int VR = kProjFastVR;
  constexpr int G = kProjFastGroup;
  constexpr int kWaves = kProjFastBlock / kWave;
  constexpr int64_t kTileRows = static_cast<int64_t>(kProjFastBlock) * VR;
  static_assert(kTileRows == 4096, "a strip entry is (tile of the batch) << 12 | row inside the tile");
  __shared__ uint32_t s_wave_tot[2][G][kWaves];
  __shared__ int64_t s_wave_lds[kWaves][kWave * kProjFastVR];      // 4 KB per wave (pf_write_group)
  __shared__ uint32_t s_next;                                      // the block's next ticket
  __shared__ uint64_t s_base;                                      // first output row of the batch
  __shared__ int64_t s_g_row0[2][G];                               // batches that straddle fragments: row of each tile's
  __shared__ const int8_t* s_g_tb[2][G][HDK_HIP_MAX_TARGETS];      // first slot, and its fragment's target columns
  __shared__ int32_t s_watch;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const uint32_t max_matched = static_cast<uint32_t>(*a.kp.max_matched);
  int64_t* buf = a.kp.groupby_buf[0];
  const bool columnar = a.columnar != 0;
  const bool keep = a.keep_cached != 0;
  uint64_t total_tiles = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    total_tiles += static_cast<uint64_t>((a.kp.num_rows[f * ntab] + kTileRows - 1) / kTileRows);
  }
  const uint64_t nbatch = (total_tiles + G - 1) / G;
  if (nbatch > a.status_cap) {  // more fragments (ragged tiles) than the status words allow for: the two passes take it
    if (blockIdx.x == 0 && tid == 0) {
      a.ticket[1] = 1;
    }
    return;
  }
  int32_t slots_err = 0;
  uint32_t iter = 0;
  // fragment cursor: tickets only grow, so it only moves forward
  uint64_t cf = 0;
  int64_t c_begin = 0, c_rows = 0, c_tiles = 0;
  const int8_t* const* c_cols = nullptr;
  if (nfrag) {
    c_rows = a.kp.num_rows[0];
    c_tiles = (c_rows + kTileRows - 1) / kTileRows;
    c_cols = a.kp.col_buffers[0];
  }
  const Watch watch = watch_begin(a.kp);
  if (tid == 0) {
    s_next = atomicAdd(a.ticket, 1u);
  }
  __syncthreads();
  uint32_t b = s_next;
  __syncthreads();
  while (b < nbatch) {
    if (watch.flags) {
      if (const int32_t w_ = watch_poll_block(watch, &s_watch)) {
        if (tid == 0) {
          record_error(a.kp.error_code, w_);
          // nobody may wait for this batch: the result is void anyway
          __hip_atomic_store(a.status + b, kProjStatusPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
      }
    }
    uint32_t next_ticket = 0;
    if (tid == 0) {
      next_ticket = atomicAdd(a.ticket, 1u);  // (in flight under the filter's loads)
    }
    const int par = static_cast<int>(iter & 1);
    const int64_t t0 = static_cast<int64_t>(b) * G;
    while (t0 >= c_begin + c_tiles) {
      c_begin += c_tiles;
      ++cf;
      c_rows = a.kp.num_rows[cf * ntab];
      c_tiles = (c_rows + kTileRows - 1) / kTileRows;
      c_cols = a.kp.col_buffers[cf];
    }
    // the batch lies in one fragment (block-uniform): a strip entry is then the row's offset from the first tile's row 0
    const bool same = t0 + G <= c_begin + c_tiles || static_cast<uint64_t>(c_begin + c_tiles) >= total_tiles;
    const int64_t row00 = (t0 - c_begin) * kTileRows;
    // ---- filter: verdict bits of this lane, bit g * VR + r = slot r of the batch's g-th tile passes --------------
    uint32_t bits = 0, full_mask = 0;
    {
      uint64_t lf = cf;
      int64_t l_begin = c_begin, l_rows = c_rows, l_tiles = c_tiles;
      const int8_t* const* l_cols = c_cols;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int64_t tile = t0 + g;
        if (static_cast<uint64_t>(tile) < total_tiles) {
          while (tile >= l_begin + l_tiles) {
            l_begin += l_tiles;
            ++lf;
            l_rows = a.kp.num_rows[lf * ntab];
            l_tiles = (l_rows + kTileRows - 1) / kTileRows;
            l_cols = a.kp.col_buffers[lf];
          }
          const int64_t row0 = (tile - l_begin) * kTileRows;
          full_mask |= row0 + kTileRows <= l_rows ? 1u << g : 0u;
          uint32_t m;
          if (R == 2 && row0 + kTileRows <= l_rows) {
            m = keep ? pf_filter_full_tile_pairs<false>(a, l_cols, row0, tid) : pf_filter_full_tile_pairs<true>(a, l_cols, row0, tid);
          } else {
            m = pf_filter_tile<R>(a, l_cols, row0, l_rows, tid, !keep);
          }
          bits |= m << (g * VR);
          if (!same) {
            if (tid == 0) {
              s_g_row0[par][g] = row0;
            }
            if (tid < a.ntargets) {
              s_g_tb[par][g][tid] = l_cols[a.t[tid].col.buf_idx];
            }
          }
        }
      }
    }
    // ---- dense positions inside the batch (order: see pf_write_group) ------------------------------------------------
    PfWavePos p;
    pf_scan_tiles(bits, lane, p);
    if (lane < G) {
      s_wave_tot[par][lane][wave] = lane == 0 ? p.wave_tot[0] : lane == 1 ? p.wave_tot[1] : lane == 2 ? p.wave_tot[2] : p.wave_tot[3];
    }
    if (tid == 0) {
      s_next = next_ticket;
    }
    __syncthreads();
    uint32_t before[G], tile_before[G], batch_total = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      uint32_t bf = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const uint32_t t = s_wave_tot[par][g][w];
        total += t;
        bf += w < wave ? t : 0;
      }
      before[g] = bf;
      tile_before[g] = batch_total;
      batch_total += total;
    }
    const uint32_t nb = s_next;
    // ---- the batch's first output row: decoupled look-back, by wave 0 --------------------------------------------
    if (wave == 0) {
      uint64_t excl;
      if (b == 0) {
        excl = static_cast<uint32_t>(*a.kp.total_matched);  // this launch appends, like the claiming kernels
      } else {
        if (lane == 0) {
          __hip_atomic_store(a.status + b, kProjStatusCount | batch_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        excl = pf_look_back(a.status, b, lane);
      }
      if (lane == 0) {
        __hip_atomic_store(a.status + b, kProjStatusPrefix | (excl + batch_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base = excl;
        if (b + 1 == nbatch) {
          *a.kp.total_matched = static_cast<int32_t>(excl + batch_total);
        }
      }
    }
    __syncthreads();
    const uint64_t base64 = s_base;
    uint64_t wave_base[G];
    int64_t trow0[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      wave_base[g] = base64 + tile_before[g] + before[g];
      trow0[g] = same ? row00 + g * kTileRows : s_g_row0[par][g];
    }
    pf_write_group<R>(a, buf, s_wave_lds[wave], bits, p, wave_base, trow0, full_mask,
                      [&](uint32_t g, int ti) { return same ? c_cols[a.t[ti].col.buf_idx] : s_g_tb[par][g][ti]; }, max_matched, tid,
                      lane, slots_err);
    ++iter;
    b = nb;
  }
  if (slots_err) {
    atomicCAS(a.kp.error_code, 0, slots_err);  // negative = ran out of slots (benign under a LIMIT)
  }
}

#ifndef HDK_PROJ_STREAM_WAVES
#define HDK_PROJ_STREAM_WAVES 0  // waves per SIMD the one-pass kernels are held to (0: the compiler's choice)
#endif
extern "C" __global__ __launch_bounds__(kProjFastBlock, HDK_PROJ_STREAM_WAVES) void hdk_scan_project_stream(ProjFastArgs a) {
  scan_project_stream_body<1>(a);
}
extern "C" __global__ __launch_bounds__(kProjFastBlock, HDK_PROJ_STREAM_WAVES) void hdk_scan_project_stream_pairs(ProjFastArgs a) {
  scan_project_stream_body<2>(a);
}

// per-block counts -> exclusive offsets (in place), starting at what TOTAL_MATCHED already holds (like the claiming
// kernels of scan_project.h, which append with atomicAdd); the grand total is added to TOTAL_MATCHED
// ... and picks the writing pass: `mode` = 2 (dense) when an eighth of the launch's rows or more pass, else 1 (`force`, if
// not 0, decides instead: A/B measurements)
extern "C" __global__ __launch_bounds__(1024) void hdk_scan_project_offsets(uint32_t* counts, uint32_t n, int32_t* total_matched,
                                                                             const uint32_t* run_if, KernParams kp, uint32_t* mode,
                                                                             uint32_t force) {
  __shared__ uint32_t s_part[1024];
  __shared__ unsigned long long s_rows_total;
  if (run_if && *run_if != 1) {
    return;
  }
  if (threadIdx.x == 0) {
    s_rows_total = 0;
  }
  __syncthreads();
  {
    const uint64_t nfrag = *kp.num_fragments;
    const uint32_t ntab = *kp.num_tables;
    unsigned long long mine = 0;
    for (uint64_t f = threadIdx.x; f < nfrag; f += 1024) {
      mine += static_cast<unsigned long long>(kp.num_rows[f * ntab]);
    }
    if (mine) {
      atomicAdd(&s_rows_total, mine);
    }
  }
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t lo = tid * per;
  const uint32_t hi = lo + per < n ? lo + per : n;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) {
    sum += counts[i];
  }
  s_part[tid] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
    const uint32_t v = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  const uint32_t already = static_cast<uint32_t>(*total_matched);  // (single block: read by all before thread 1023 adds)
  __syncthreads();
  uint32_t run = already + (tid ? s_part[tid - 1] : 0);
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t c = counts[i];
    counts[i] = run;
    run += c;
  }
  if (tid == 1023) {
    atomicAdd(total_matched, static_cast<int32_t>(s_part[1023]));
    *mode = force ? force : (static_cast<unsigned long long>(s_part[1023]) * 8ull >= s_rows_total ? 2u : 1u);
  }
}

}  // namespace hdk
