// scan_bhm.hip -- matcher, launcher and instantiations of the multi-argument / multi-key on-chip group-by (scan_bhm.h).
#include <string.h>

#include <algorithm>

#include "host_match.h"
#include "scan_bhm.h"
#include "scan_bhm_part.h"
#include "scan_bhm_host.h"
#include "scan_bhm_shapes.h"

namespace hdk {

constexpr uint32_t kBhmMaxLdsBytes = 144u << 10;        // one 1024-thread block per CU
constexpr uint32_t kBhmSmallLdsBytes = 40u << 10;       // up to here: 256-thread blocks, four on a CU
constexpr uint32_t kBhmReplicatedBytes = 32u << 10;     // replicas while the table is tiny
constexpr int64_t kBhmMaxAbsVal = (1ll << 19) - 1;      // |argument| below this keeps a block's row budget at 2^20 or more
// (A/B switches; every table here is dense: HDK_HIP_NO_BH_DENSE turns it off with the one-argument dense forms)
static bool bhm_off() { return hdk_sw(SW_NO_BHM) != nullptr || hdk_sw(SW_NO_BH_LDS) != nullptr || hdk_sw(SW_NO_BH_DENSE) != nullptr; }

static uint32_t bits_for(uint64_t codes) {  // bits that hold the values 0 .. codes - 1
  uint32_t b = 1;
  while ((1ull << b) < codes) ++b;
  return b;
}

struct BhmGeom {
  int nk, ns, block, width;
  uint32_t grid_per_cu;
  bool perfect;
  int64_t key_lo;
  uint32_t null_entry;
  uint32_t nword_mask;
  int32_t wop[kMaxWordsPerEntry];
};

// the part of the match that does not depend on where the table lives: keys, arguments, LDS words per entry
static bool match_bhm_shape(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhmArgs* a, BhmGeom* g, uint32_t* per_entry_out, int64_t* amax_out) {
  const bool perfect = p->query_kind == HDK_Q_PERFECT_HASH;
  if (bhm_off() || (!perfect && p->query_kind != HDK_Q_BASELINE_HASH)) return false;
  if (!ko || ko->total_rows == 0 || ko->total_rows >= (1ull << 40)) return false;
  if (ko->flags & (HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS | HDK_HIP_LAUNCH_FORCE_PARTITIONED)) return false;
  if (launch_forces_generic(ko)) return false;
  if (p->num_joins) return false;
  if (p->key_count < 1 || p->key_count > kBhmMaxKeys) return false;
  memset(a, 0, sizeof(*a));
  // plain filters `outer column cmp literal`, alone or under an AND / OR / NOT program (plain_quals.h): the run-time form
  if (p->num_quals || p->num_filter_ops) {
    if (!match_plain_quals(p, a->q, true)) return false;
    a->nquals = p->num_quals;
  }
  int width = 0;  // of every streamed column: 4, or 8 (BIGINT inside 32 bits)
  memset(g, 0, sizeof(*g));
  g->perfect = perfect;
  g->null_entry = 0xFFFFFFFFu;
  // ---- keys: plain 4-byte integer columns with statistics --------------------------------------------------------------
  uint64_t entries = 1;
  for (int k = 0; k < p->key_count; ++k) {
    const hdk_hip_expr& ke = p->keys[k];
    if (ke.leaf0.kind != HDK_LEAF_COL) return false;
    const hdk_hip_col& kc = p->cols[ke.leaf0.col];
    if (kc.table != 0 || kc.kind != HDK_COL_INT || (kc.width != 4 && kc.width != 8)) return false;
    if (width && width != kc.width) return false;  // (one width for all streamed columns)
    width = kc.width;
    if (kc.width == 8 && (!kc.has_stats || kc.min_val <= INT32_MIN || kc.max_val > INT32_MAX)) return false;
    BhmKey& key = a->key[k];
    key.buf_idx = kc.buf_idx;
    key.null32 = kc.width == 8 ? INT32_MIN : static_cast<int32_t>(ke.leaf0.null_val);  // (bhm_narrow maps a BIGINT NULL there)
    if (perfect) {
      // the plan's own index: (key - min) * stride, a NULL under its translated value (perfect_key_hash)
      if (ke.nsteps != 0 || p->key_bucket[k] > 1) return false;
      const bool translate = p->key_has_nulls[k] && ke.nullable;
      const int64_t card = p->key_count == 1 ? static_cast<int64_t>(p->entry_count) : p->key_card[k];
      if (card < 1 || card > (1 << 22) || p->key_min[k] < INT32_MIN / 2 || p->key_min[k] > INT32_MAX / 2) return false;
      key.min = static_cast<int32_t>(p->key_min[k]);
      key.n = static_cast<uint32_t>(card - (translate ? 1 : 0));
      key.nullable = translate;
      const int64_t nd = p->key_null_translated[k] - p->key_min[k];
      if (translate && (nd < 0 || nd >= card)) return false;
      key.null_d = translate ? static_cast<uint32_t>(nd) : 0u;
      key.stride = static_cast<uint32_t>(entries);
      entries *= static_cast<uint64_t>(card);
    } else {
      // open addressing: one key, as it is or cast to double; dense over the key column's statistics
      if (p->key_count != 1 || (p->key_width != 4 && p->key_width != 8)) return false;
      if (ke.nsteps == 1) {
        const hdk_hip_step& sp = ke.steps[0];
        if (sp.op != HDK_OP_CAST_INT_TO_FP || p->key_width != 8) return false;
        a->key_form = 1;
        a->key_null_word = sp.null_out;
      } else if (ke.nsteps != 0) {
        return false;
      } else {
        a->key_null_word = ke.leaf0.null_val;
      }
      if (!kc.has_stats || kc.max_val < kc.min_val || kc.min_val < INT32_MIN / 2 || kc.max_val > INT32_MAX / 2) return false;
      const uint64_t n = static_cast<uint64_t>(kc.max_val - kc.min_val) + 1;
      // (a dense table only pays while it is not much sparser than the groups the output table was sized for)
      if (n > (1u << 20) || n > 4ull * p->entry_count + 64) return false;
      key.min = static_cast<int32_t>(kc.min_val);
      key.n = static_cast<uint32_t>(n);
      // (a NULL key only when the statistics announce one: else it lies outside them like any other stranger -- the flag)
      key.nullable = ke.leaf0.nullable != 0 && kc.has_nulls != 0;
      key.null_d = static_cast<uint32_t>(n);
      key.stride = 1;
      entries = n + (key.nullable ? 1 : 0);
      g->key_lo = kc.min_val;
      g->null_entry = key.nullable ? static_cast<uint32_t>(n) : 0xFFFFFFFFu;
    }
    if (entries > (1u << 22)) return false;
  }
  if (perfect && entries != p->entry_count) return false;
  a->nkeys = p->key_count;
  a->entries = static_cast<uint32_t>(entries);
  // ---- targets ------------------------------------------------------------------------------------------------------------
  WordLayout wl;
  make_word_layout(p, &wl);
  a->wpe = wl.wpe;
  for (int w = 0; w < wl.wpe; ++w) g->wop[w] = wl.wop[w];
  struct DerInfo {
    int64_t rmin, rmax;
    bool want_packed, want_max, want_min;
  } info[kBhmMaxDer];
  memset(info, 0, sizeof(info));
  int word_der[kMaxWordsPerEntry], word_kind[kMaxWordsPerEntry];
  for (int w = 0; w < kMaxWordsPerEntry; ++w) {
    word_der[w] = -1;
    word_kind[w] = BMW_ROWS;
  }
  int64_t amax = 1;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    if (tg.agg == HDK_AGG_SINGLE_VALUE) return false;
    if (tg.agg == HDK_AGG_ID) {
      if (tg.key_idx < 0 || tg.key_idx >= p->key_count || (tg.slot_width != 0 && tg.slot_width != 4 && tg.slot_width != 8)) return false;
      continue;
    }
    if (tg.slot_width != 4 && tg.slot_width != 8) return false;
    if (tg.agg == HDK_AGG_AVG && tg.slot2_width != 4 && tg.slot2_width != 8) return false;
    if (!tg.has_arg) {
      if (tg.agg != HDK_AGG_COUNT) return false;
      continue;
    }
    const hdk_hip_expr& e = tg.arg;
    if (e.nsteps > 1 || e.leaf0.kind != HDK_LEAF_COL || tg.arg_is_fp || e.vclass != HDK_VC_INT) return false;
    const hdk_hip_col& col = p->cols[e.leaf0.col];
    if (col.table != 0 || col.kind != HDK_COL_INT || col.width != width || !col.has_stats || col.max_val < col.min_val) return false;
    if (col.min_val < -kBhmMaxAbsVal || col.max_val > kBhmMaxAbsVal) return false;
    int64_t mul = 1, add = 0, rmin = col.min_val, rmax = col.max_val;
    if (e.nsteps == 1) {
      const hdk_hip_step& sp = e.steps[0];
      if (sp.out_class != HDK_VC_INT || sp.rhs.kind != HDK_LEAF_INT) return false;
      if (sp.op != HDK_OP_ADD && sp.op != HDK_OP_SUB && sp.op != HDK_OP_MUL) return false;
      if (sp.rhs.ival < -kBhmMaxAbsVal || sp.rhs.ival > kBhmMaxAbsVal) return false;
      // a NULL operand gives the step's NULL, which must be what the target skips (as match_fast)
      if (tg.skip_null && (e.null_val != sp.null_out || tg.null_val != sp.null_out)) return false;
      if (!tg.skip_null && e.leaf0.nullable) return false;
      if (sp.op == HDK_OP_MUL) {
        mul = sp.rhs.ival;
      } else {
        add = sp.op == HDK_OP_ADD ? sp.rhs.ival : -sp.rhs.ival;
      }
      const int64_t c0 = rmin * mul + add, c1 = rmax * mul + add;
      rmin = std::min(c0, c1);
      rmax = std::max(c0, c1);
      // (the result's range fits the operation's type: the checked arithmetic of QE/ArithmeticIR.cpp:277-520 cannot fire)
      const int cw = sp.check_width ? sp.check_width : 8;
      const int64_t tmax = cw >= 8 ? INT64_MAX : (1ll << (8 * cw - 1)) - 1;
      if (rmin < -tmax || rmax > tmax) return false;
      if (mul == 1 && add == 0) return false;  // (x + 0: leave the oddities to the interpreter)
    }
    if (rmin < -kBhmMaxAbsVal || rmax > kBhmMaxAbsVal) return false;
    // the column's NULL: skipped when the statistics announce NULLs; else it would be outside them (the flag).  A NULL
    // aggregated as a value (no skip) is far outside the packed fields: not this kernel's
    if (!tg.skip_null && col.has_nulls && e.leaf0.nullable) return false;
    const int nullable = (tg.skip_null && e.leaf0.nullable && col.has_nulls) ? 1 : 0;
    int s = -1;
    for (int i = 0; i < a->nsrc; ++i) {
      if (a->src[i].buf_idx == col.buf_idx) s = i;
    }
    if (s < 0) {
      if (a->nsrc == kBhmMaxSrc) return false;
      s = a->nsrc++;
      BhmSrc& src = a->src[s];
      src.buf_idx = col.buf_idx;
      src.nullable = nullable;
      src.null32 = col.width == 8 ? INT32_MIN : static_cast<int32_t>(e.leaf0.null_val);
      src.raw_min = static_cast<int32_t>(col.min_val);
      src.raw_span = static_cast<uint32_t>(col.max_val - col.min_val);
    } else if (a->src[s].nullable != nullable) {
      return false;
    }
    int d = -1;
    for (int i = 0; i < a->nder; ++i) {
      if (a->der[i].src == s && a->der[i].mul == static_cast<int32_t>(mul) && a->der[i].add == static_cast<int32_t>(add)) d = i;
    }
    if (d < 0) {
      if (a->nder == kBhmMaxDer) return false;
      d = a->nder++;
      a->der[d].src = s;
      a->der[d].mul = static_cast<int32_t>(mul);
      a->der[d].add = static_cast<int32_t>(add);
      a->der[d].packed = -1;
      info[d].rmin = rmin;
      info[d].rmax = rmax;
    }
    amax = std::max<int64_t>(amax, std::max<int64_t>(-rmin, rmax));
    if (tg.agg == HDK_AGG_SUM || tg.agg == HDK_AGG_AVG || tg.agg == HDK_AGG_COUNT) info[d].want_packed = true;
    if (tg.agg == HDK_AGG_MAX) info[d].want_max = true;
    if (tg.agg == HDK_AGG_MIN) info[d].want_min = true;
    if (wl.vword[t] >= 0) {
      word_der[wl.vword[t]] = d;
      word_kind[wl.vword[t]] = tg.agg == HDK_AGG_MIN ? BMW_MIN : (tg.agg == HDK_AGG_MAX ? BMW_MAX : BMW_SUM);
    }
    if (wl.nword[t] >= 0) {
      word_der[wl.nword[t]] = d;
      word_kind[wl.nword[t]] = BMW_NN;
      g->nword_mask |= 1u << wl.nword[t];
    }
  }
  if (a->nsrc == 0) return false;  // (COUNT(*) alone: the keys kernel / the one-argument packed kernels)
  a->col_width = width;
  g->width = width;
  a->qvec = a->nquals > 0;
  bool lean = a->nquals > 0;  // (programs too: bhm_quals_lean carries the three-valued combiner)
  for (int qi = 0; qi < a->nquals; ++qi) {
    const ProjFastCol& c = a->q[qi].col;
    a->qvec = a->qvec && c.kind == HDK_COL_INT && c.width == width;
    lean = lean && a->q[qi].fp == 0 && a->q[qi].col_fp == 0;
  }
  if (a->qvec && lean) a->qvec = 2;
  for (int qi = 0; qi < kMaxPlainQuals; ++qi) {
    a->qsrc[qi] = kBhmQualOwnLoad;
    if (qi >= a->nquals) continue;
    for (int k = 0; k < a->nkeys; ++k) {
      if (a->key[k].buf_idx == a->q[qi].col.buf_idx) a->qsrc[qi] = k;
    }
    for (int s = 0; s < a->nsrc; ++s) {
      if (a->src[s].buf_idx == a->q[qi].col.buf_idx) a->qsrc[qi] = kBhmMaxKeys + s;
    }
  }
  // (one key, one plain argument is scan_bh_packed.h's own shape: its one-pass kernels are asked first, launch_bh_packed; what
  // they cannot hold -- 24 bytes an entry, 4 096 entries -- may still fit here at 12: BH004 / PHS004's 10 000 groups in ONE pass)
  for (int k = 0; k < a->nkeys; ++k) a->any_nullable |= a->key[k].nullable;
  for (int s = 0; s < a->nsrc; ++s) a->any_nullable |= a->src[s].nullable;
  // ---- LDS words: packed [rows : sum] words, MIN / MAX fields in the halves of one 64-bit word ---------------------------------
  a->rows_packed = -1;
  uint32_t bitpos[2] = {0, 0};
  int fhalf_of[kBhmMaxDer][2];
  auto place_field = [&](uint32_t codebits, BhmField* fd, int* half_out) -> bool {
    const uint32_t need = codebits + 1;  // + the guard bit
    for (int h = 0; h < 2; ++h) {
      if (bitpos[h] + need <= 32 && a->nfields < kBhmMaxFields) {
        const uint32_t mask = static_cast<uint32_t>((1ull << codebits) - 1);
        fd->shift = bitpos[h];
        fd->mask_lo = h == 0 ? mask : 0u;
        fd->mask_hi = h == 1 ? mask : 0u;
        a->fshift[a->nfields] = bitpos[h];
        a->fmask[a->nfields] = mask;
        a->fhalf[a->nfields] = static_cast<uint32_t>(h);
        a->nfields++;
        (h == 0 ? a->guard_lo : a->guard_hi) |= 1u << (bitpos[h] + codebits);
        bitpos[h] += need;
        *half_out = h;
        return true;
      }
    }
    return false;
  };
  for (int d = 0; d < a->nder; ++d) {
    BhmDer& der = a->der[d];
    const DerInfo& di = info[d];
    if (di.want_packed) {
      der.packed = a->npacked++;
      if (!a->src[der.src].nullable && a->rows_packed < 0) a->rows_packed = der.packed;
    }
    const uint32_t codebits = bits_for(static_cast<uint64_t>(di.rmax - di.rmin) + 2);
    if (codebits > 30) return false;
    fhalf_of[d][0] = fhalf_of[d][1] = 0;
    if (di.want_max) {
      der.has_mx = 1;
      der.mx.bias = static_cast<int32_t>(di.rmin - 1);
      if (!place_field(codebits, &der.mx, &fhalf_of[d][0])) return false;
    }
    if (di.want_min) {
      der.has_mn = 1;
      der.mn.bias = static_cast<int32_t>(di.rmax + 1);
      if (!place_field(codebits, &der.mn, &fhalf_of[d][1])) return false;
    }
  }
  a->mm_bytes = a->nfields == 0 ? 0 : (bitpos[1] == 0 ? 4 : 8);
  // the slab words
  for (int w = 0; w < wl.wpe; ++w) {
    BhmSlabWord& sw = a->sw[w];
    sw.kind = word_kind[w];
    sw.packed = -1;
    if (w == 0 || word_der[w] < 0) {
      sw.kind = BMW_ROWS;
      sw.packed = a->rows_packed;
      continue;
    }
    const BhmDer& der = a->der[word_der[w]];
    const BhmSrc& src = a->src[der.src];
    auto from_field = [&](bool mx) {
      const BhmField& fd = mx ? der.mx : der.mn;
      sw.in_mm = 1;
      sw.half = fhalf_of[word_der[w]][mx ? 0 : 1];
      sw.shift = fd.shift;
      sw.mask = fd.mask_lo | fd.mask_hi;
      sw.bias = fd.bias;
    };
    if (sw.kind == BMW_SUM) {
      sw.packed = der.packed;
    } else if (sw.kind == BMW_NN) {
      if (!src.nullable) {  // the argument is never NULL: its non-NULL count is the row count
        sw.kind = BMW_ROWS;
        sw.packed = a->rows_packed;
      } else if (der.packed >= 0) {
        sw.packed = der.packed;
      } else {  // MIN / MAX only: "some non-NULL row" is all its count says
        from_field(der.has_mx != 0);
      }
    } else {
      from_field(sw.kind == BMW_MAX);
    }
  }
  *per_entry_out = 8u * static_cast<uint32_t>(a->npacked) + static_cast<uint32_t>(a->mm_bytes) + (a->rows_packed < 0 ? 4u : 0u);
  *amax_out = amax;
  g->nk = a->nkeys;
  g->ns = a->nsrc;
  return true;
}

// byte offsets of a replica's arrays for a table of a->entries entries
static uint64_t bhm_lds_layout(BhmArgs* a, uint32_t per_entry) {
  a->e1 = (a->entries + 1 + 1) & ~1u;
  a->off_mm = 8u * static_cast<uint32_t>(a->npacked) * a->e1;
  a->off_rows = a->off_mm + static_cast<uint32_t>(a->mm_bytes) * a->e1;
  return static_cast<uint64_t>(a->e1) * per_entry;
}

// ONE pass: the whole dense table in a block's LDS
static bool match_bhm(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhmArgs* a, BhmGeom* g) {
  uint32_t per_entry;
  int64_t amax;
  if (!match_bhm_shape(p, ko, a, g, &per_entry, &amax)) return false;
  const uint64_t one = bhm_lds_layout(a, per_entry);
  if (one > kBhmMaxLdsBytes) return false;
  uint32_t rep_bytes = (static_cast<uint32_t>(one) + 15u) & ~15u;
  uint32_t rep = 16;
  while (rep > 1 && static_cast<uint64_t>(rep_bytes + 16) * rep > kBhmReplicatedBytes) rep >>= 1;
  if (rep > 1) {
    while (rep_bytes % 256 != 16) rep_bytes += 16;  // replica r starts 4 r banks further on
  }
  a->rep = rep;
  a->rep_bytes = rep_bytes;
  a->lds_bytes = rep_bytes * rep;
  g->block = a->lds_bytes > kBhmSmallLdsBytes ? 1024 : 256;
  // (256-thread blocks: three on a CU measured best -- msbs1 / msphs1 / phm2 at 256 M rows: 1 -> 1.06 / 1.04 / 0.79 ms, 2 -> 0.66 /
  // 0.63 / 0.60, 3 -> 0.63 / 0.60 / 0.60, 4 -> 0.74 / 0.72 / 0.71, 8 -> 0.94 / 0.93 / 0.89: more waves add LDS contention and slabs)
  g->grid_per_cu = g->block == 1024 ? 1 : 3;
  // rows a block may put into one entry: rows < 2^23, |sum| < 2^39
  const int64_t by_sum = ((1ll << 39) - 1) / amax;
  a->max_rows_per_block = static_cast<uint32_t>(std::min<int64_t>(by_sum, 1ll << 23));
  return true;
}

// TWO passes (scan_bhm_part.h): bins of 2^w entries, 4-byte tuples
constexpr uint32_t kBhmPartLdsBytes = 50u << 10;   // a bin's table in pass B: 4 096 entries of 12 bytes (three 256-thread blocks on a CU)
struct BhmPartLayout {
  size_t cursor_bytes, tuple_bytes, slab_bytes, total;
};
static bool match_bhm_part(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, BhmPartArgs* pg, BhmGeom* g, BhmPartLayout* l) {
  if (hdk_sw(SW_NO_BHM_PARTITIONS) || hdk_sw(SW_NO_BH_PARTITIONS)) return false;
  memset(pg, 0, sizeof(*pg));
  BhmArgs* a = &pg->b;
  uint32_t per_entry;
  int64_t amax;
  if (!match_bhm_shape(p, ko, a, g, &per_entry, &amax)) return false;
  const bool always = hdk_sw(SW_BH_PARTITIONS_ALWAYS) != nullptr;  // (tests: small inputs and tables)
  if (!always && ko->total_rows < (4ull << 20)) return false;
  const uint32_t total = a->entries;
  // the largest bin that fits pass B's LDS, but at least 64 bins while the table has that many groups of 16 (pass B runs
  // 8 blocks per bin: parallelism) -- and never more than 256 bins
  uint32_t w = 4;
  while (w < 14 && (static_cast<uint64_t>((2u << w) + 2) * per_entry) <= kBhmPartLdsBytes) ++w;
  uint32_t min_bins = 128;  // (256 M rows, msbs2 / msphs3 / phm4 / bh5: 8 bins 3.12 / 1.71 / 1.47 / 1.35 ms, 32: 1.56 / 1.71 / 1.47 / 1.35, 64: 1.41 / 1.54 /
                            // 1.37 / 1.21, 128: 1.32 / 1.48 / 1.33 / 1.20: pass B wants blocks)
  if (const char* e = hdk_sw(SW_BHM_PART_MIN_BINS)) min_bins = static_cast<uint32_t>(std::max(1, atoi(e)));  // (measurements)
  while (w > 4 && ((total + (1u << w) - 1) >> w) < min_bins) --w;
  while (((total + (1u << w) - 1) >> w) > kPbMaxBins) ++w;
  if (static_cast<uint64_t>((1u << w) + 2) * per_entry > kBhmPartLdsBytes) return false;  // too many groups for 256 bins
  pg->w = w;
  pg->nbins = (total + (1u << w) - 1) >> w;
  pg->total_entries = total;
  // the tuple: [entry inside the bin | code of every argument column]
  uint32_t pos = w;
  for (int s = 0; s < a->nsrc; ++s) {
    const uint32_t cb = bits_for(static_cast<uint64_t>(a->src[s].raw_span) + 2);
    if (pos + cb > 32) return false;
    pg->cshift[s] = pos;
    pg->cmask[s] = static_cast<uint32_t>((1ull << cb) - 1);
    pos += cb;
  }
  pg->tw = (pos <= 16 && !hdk_sw(SW_BHM_WIDE_TUPLES)) ? 2 : 4;
  // pass B's table: one bin, one replica
  a->entries = 1u << w;
  const uint64_t one = bhm_lds_layout(a, per_entry);
  // ONE replica: replicas of a small bin's table (HDK_HIP_BHM_PART_REPLICAS = 2 / 4 / 8) measured no better -- bh5 1.14 / 1.15 / 1.21 /
  // 1.20 ms, msbs2 1.32 / 1.31 / 1.29 / 1.31, phm4 1.27 / 1.29 / 1.35 / 1.34 per 256 M rows: same-address conflicts are not what
  // bounds pass B (its LDS operations per row are)
  uint32_t prep_bytes = (static_cast<uint32_t>(one) + 15u) & ~15u;
  uint32_t prep = 1;
  if (const char* e = hdk_sw(SW_BHM_PART_REPLICAS)) prep = static_cast<uint32_t>(std::max(1, atoi(e)));  // (measurements; a power of two)
  while (prep > 1 && static_cast<uint64_t>(prep_bytes + 16) * prep > kBhmReplicatedBytes) prep >>= 1;
  if (prep > 1) {
    while (prep_bytes % 256 != 16) prep_bytes += 16;
  }
  a->rep = prep;
  a->rep_bytes = prep_bytes;
  a->lds_bytes = prep_bytes * prep;
  a->max_rows_per_block = 0xFFFFFFFFu;  // (pass B checks its own bound: a sub-slab's capacity)
  // the sub-slabs are sized on the device from a sample of the keys (hdk_bhm_part_sample / _layout): twice what the sample
  // promises a bin.  The allocation has room for a quarter more rows than the launch's bound says there are.
  const int64_t by_sum = ((1ll << 39) - 1) / amax;
  // one LDS table of pass B takes a GENERATION of tuples (rows in 24 bits, sums in 40); a sub-slab with more is flushed in between
  uint64_t generation = static_cast<uint64_t>(std::min<int64_t>(by_sum, 1ll << 23)) & ~4095ull;
  if (const char* e = hdk_sw(SW_BHM_PART_GENERATION)) generation = static_cast<uint64_t>(std::max(1, atoi(e))) * 4096;  // (tests)
  if (generation < 4096) return false;
  pg->generation = static_cast<uint32_t>(generation);
  for (int w = 0; w < a->wpe; ++w) pg->wop[w] = g->wop[w];
  pg->cap_limit = 0xFFFFFFF0ull;
  // pass B cuts a sub-slab of more than four times the average into parts of twice the average (at least 64 K tuples)
  pg->parts = 64;
  const uint64_t avg = ko->total_rows / (static_cast<uint64_t>(pg->nbins) * kPbXcds) + 1;
  pg->part_tuples = static_cast<uint32_t>(std::min<uint64_t>((std::max<uint64_t>(2 * avg, 65536) + 4095) & ~4095ull, 1u << 30));
  if (const char* e = hdk_sw(SW_BHM_PART_TUPLES)) pg->part_tuples = static_cast<uint32_t>(std::max(1, atoi(e))) * 4096u;  // (tests)
  pg->total_rows = ko->total_rows;
  pg->region_max = ((ko->total_rows / kPbXcds) * 5 / 2 + static_cast<uint64_t>(pg->nbins) * 4104 + 7) & ~7ull;
  if (pg->region_max > 0xFFFFFFF0ull) return false;
  // about 1 500 tiles of 1 024 rows are looked at (every tile of a small input; 1 B rows: every 650th -- 41 us at every 128th)
  pg->sample_stride = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(ko->total_rows / (256ull * (16 / g->width)) / 1536, 1), 1u << 16));
  if (const char* e = hdk_sw(SW_BHM_PART_SAMPLE_STRIDE)) pg->sample_stride = static_cast<uint32_t>(std::max(1, atoi(e)));  // (tests)
  l->cursor_bytes = static_cast<size_t>(pg->nbins) * kPbXcds * kPbCursorStride * sizeof(uint32_t) + kBlWords * sizeof(uint32_t);
  l->cursor_bytes = (l->cursor_bytes + 255) & ~static_cast<size_t>(255);
  // (after pass B the tuples are spent and their space takes the ONE slab the eight are reduced to: at least that much)
  // (+ one batch of slack: a clamped claim of the last sub-slab, scan_bhm_part.h)
  l->tuple_bytes = std::max((static_cast<size_t>(kPbXcds) * pg->region_max + kBhmPartBlock * 16) * pg->tw, static_cast<size_t>(total) * a->wpe * 8);
  l->tuple_bytes = (l->tuple_bytes + 255) & ~static_cast<size_t>(255);
  l->slab_bytes = static_cast<size_t>(kPbXcds) * total * a->wpe * 8;
  if (l->slab_bytes > (1ull << 30)) return false;  // (PHM005's 1 M entries: 700 MB of slabs, written and read once)
  l->total = 256 + l->cursor_bytes + l->tuple_bytes + l->slab_bytes;
  return true;
}

// ---- the instantiations (scan_bhm_shapes.h: the list; this file holds <4-byte columns, no filter> and pass B) ------------------
HDK_BHM_DEFINE_KERNELS(4, false, HDK_BHM_SHAPE_FN_NULLS, HDK_BHM_PLAIN_BODY_YES)

struct BhmStaticShape {
  int nk, ns;
  uint32_t code[kBhmMaxDer];
  const void* (*aggregate)(int tw, bool nulls);  // pass B of the two-pass form (the keys are folded into the tuple: NK, W, the filter do not matter)
};
#define HDK_BHM_SHAPE_ROW(NK, NS, D0, D1, D2, D3)                                                                            \
  {NK, NS, {D0, D1, D2, D3}, [](int tw, bool nulls) -> const void* {                                                         \
     if (nulls) {                                                                                                            \
       return tw == 2 ? reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmStatic<D0, D1, D2, D3, true>, NS, 2>)             \
                      : reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmStatic<D0, D1, D2, D3, true>, NS, 4>);            \
     }                                                                                                                       \
     return tw == 2 ? reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmStatic<D0, D1, D2, D3, false>, NS, 2>)              \
                    : reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmStatic<D0, D1, D2, D3, false>, NS, 4>);             \
   }},
static const BhmStaticShape kBhmShapes[] = {HDK_BHM_SHAPES(HDK_BHM_SHAPE_ROW)};
#undef HDK_BHM_SHAPE_ROW
constexpr int kBhmNumShapes = static_cast<int>(sizeof(kBhmShapes) / sizeof(kBhmShapes[0]));

// the (column width, filtered) quarter a plan belongs to
template <class F4, class F4Q, class F8, class F8Q>
static const void* bhm_by_quarter(const BhmArgs& a, const BhmGeom& g, F4 f4, F4Q f4q, F8 f8, F8Q f8q) {
  return g.width == 8 ? (a.nquals ? f8q() : f8()) : (a.nquals ? f4q() : f4());
}
static const void* bhm_scatter_kernel(const BhmArgs& a, const BhmGeom& g) {
  return bhm_by_quarter(
      a, g, [&] { return BhmKernels<4, false>::scatter(g.nk, g.ns); }, [&] { return BhmKernels<4, true>::scatter(g.nk, g.ns); },
      [&] { return BhmKernels<8, false>::scatter(g.nk, g.ns); }, [&] { return BhmKernels<8, true>::scatter(g.nk, g.ns); });
}

template <int W>
static const void* bhm_sample_kernel_w(int nk) {
  return nk == 1 ? reinterpret_cast<const void*>(hdk_bhm_part_sample<1, W>)
                 : (nk == 2 ? reinterpret_cast<const void*>(hdk_bhm_part_sample<2, W>) : reinterpret_cast<const void*>(hdk_bhm_part_sample<3, W>));
}
static const void* bhm_sample_kernel(const BhmGeom& g) { return g.width == 8 ? bhm_sample_kernel_w<8>(g.nk) : bhm_sample_kernel_w<4>(g.nk); }

// the argument descriptors as bhm_update wants them: over codes (column value - (minimum - 1)) instead of values
static uint32_t bhm_plain_flags(const BhmArgs& a);
static void bhm_shift_to_codes(BhmArgs* a) {
  a->plain_flags = bhm_plain_flags(*a);
  for (int d = 0; d < a->nder; ++d) {
    BhmDer& der = a->der[d];
    const BhmSrc& src = a->src[der.src];
    if (der.mul == 1) {  // column, column +- literal: code of the value = code of the column
      der.mx.bias = 0;                                       // (MAX code = the code)
      der.mn.bias = static_cast<int32_t>(src.raw_span) + 2;  // (MIN code = span + 2 - code)
      if (der.packed >= 0) a->psum_k[der.packed] = src.raw_min + der.add - 1;  // (sum of values = sum of codes + rows x (smallest value - 1))
    } else {
      der.add = static_cast<int32_t>(der.add + (static_cast<int64_t>(src.raw_min) - 1) * der.mul);  // (value = (code + min - 1) x mul + add)
    }
  }
}

static uint32_t bhm_code_of(const BhmArgs& a, int i) {
  if (i >= a.nder) return kBhmNone;
  const BhmDer& d = a.der[i];
  if (d.mul != 1 && d.add != 0) return 0xFE;  // (column * literal + literal: never produced by the matcher)
  return bhm_code(d.src, d.packed >= 0, d.has_mx != 0, d.has_mn != 0, d.mul != 1 ? 2 : (d.add != 0 ? 1 : 0));
}

// argument i is column i as it is, for every argument: the flags BhmPlain reads (0: not that family)
static uint32_t bhm_plain_flags(const BhmArgs& a) {
  if (a.nder != a.nsrc) return 0;
  uint32_t f = 0;
  for (int i = 0; i < a.nder; ++i) {
    const BhmDer& d = a.der[i];
    if (d.src != i || d.mul != 1 || d.add != 0) return 0;
    f |= ((d.packed >= 0 ? 1u : 0u) | (d.has_mx ? 2u : 0u) | (d.has_mn ? 4u : 0u)) << (3 * i);
  }
  return f;
}

// compile-time shape when there is one (every field inside 32 bits; with NULLs announced: the unfiltered quarters' twins), else the
// run-time form
static const void* bhm_kernel(const BhmArgs& a, const BhmGeom& g, bool* is_static) {
  *is_static = false;
  if (a.mm_bytes != 8 && !hdk_sw(SW_BHM_DYNAMIC)) {
    const bool nulls = a.any_nullable != 0;
    for (int si = 0; si < kBhmNumShapes; ++si) {
      const BhmStaticShape& sh = kBhmShapes[si];
      bool same = sh.nk == g.nk && sh.ns == g.ns;
      for (int i = 0; same && i < kBhmMaxDer; ++i) same = sh.code[i] == bhm_code_of(a, i);
      if (same) {
        const void* k = bhm_by_quarter(
            a, g, [&] { return BhmKernels<4, false>::fixed(si, g.block, nulls); }, [&] { return BhmKernels<4, true>::fixed(si, g.block, nulls); },
            [&] { return BhmKernels<8, false>::fixed(si, g.block, nulls); }, [&] { return BhmKernels<8, true>::fixed(si, g.block, nulls); });
        if (k) {
          *is_static = true;
          return k;
        }
        break;
      }
    }
  }
  if (bhm_plain_flags(a) && !hdk_sw(SW_BHM_DYNAMIC)) {  // aggregates of plain columns (BhmPlain; the unfiltered quarters)
    const void* k = bhm_by_quarter(
        a, g, [&] { return BhmKernels<4, false>::plain(g.nk, g.ns, g.block); }, [&] { return BhmKernels<4, true>::plain(g.nk, g.ns, g.block); },
        [&] { return BhmKernels<8, false>::plain(g.nk, g.ns, g.block); }, [&] { return BhmKernels<8, true>::plain(g.nk, g.ns, g.block); });
    if (k) return k;
  }
  return bhm_by_quarter(
      a, g, [&] { return BhmKernels<4, false>::dynamic(g.nk, g.ns, g.block); }, [&] { return BhmKernels<4, true>::dynamic(g.nk, g.ns, g.block); },
      [&] { return BhmKernels<8, false>::dynamic(g.nk, g.ns, g.block); }, [&] { return BhmKernels<8, true>::dynamic(g.nk, g.ns, g.block); });
}

template <int TW>
static const void* bhm_aggregate_dynamic(int ns) {
  return ns == 1 ? reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmDynamic, 1, TW>)
                 : (ns == 2 ? reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmDynamic, 2, TW>)
                            : reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmDynamic, 3, TW>));
}
template <int TW>
static const void* bhm_aggregate_plain(int ns) {
  return ns == 1 ? reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmPlain<1>, 1, TW>)
                 : (ns == 2 ? reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmPlain<2>, 2, TW>)
                            : reinterpret_cast<const void*>(hdk_bhm_aggregate<BhmPlain<3>, 3, TW>));
}
static const void* bhm_aggregate_kernel(const BhmArgs& a, const BhmGeom& g, int tw) {
  if (a.mm_bytes != 8 && !hdk_sw(SW_BHM_DYNAMIC)) {
    for (const BhmStaticShape& sh : kBhmShapes) {
      bool same = sh.ns == g.ns;  // (any NK: pass B sees entries, not keys)
      for (int i = 0; same && i < kBhmMaxDer; ++i) same = sh.code[i] == bhm_code_of(a, i);
      if (same) return sh.aggregate(tw, a.any_nullable != 0);
    }
  }
  if (bhm_plain_flags(a) && !hdk_sw(SW_BHM_DYNAMIC)) {
    return tw == 2 ? bhm_aggregate_plain<2>(g.ns) : bhm_aggregate_plain<4>(g.ns);
  }
  return tw == 2 ? bhm_aggregate_dynamic<2>(g.ns) : bhm_aggregate_dynamic<4>(g.ns);
}

const char* bhm_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko) {
  BhmArgs a;
  BhmGeom g;
  if (!match_bhm(p, ko, &a, &g)) {
    BhmPartArgs pg;
    BhmPartLayout l;
    if (match_bhm_part(p, ko, &pg, &g, &l)) {
      return g.perfect ? "hdk_bhm_scatter,hdk_bhm_aggregate,hdk_bhm_reduce_slabs,hdk_finalize"
                       : "hdk_bhm_scatter,hdk_bhm_aggregate,hdk_bhm_reduce_slabs,hdk_bhm_fold";  // (behind hdk_bhm_part_sample, _layout)
    }
    return nullptr;
  }
  return g.perfect ? "hdk_scan_agg_bhm,hdk_bhm_reduce_slabs,hdk_finalize" : "hdk_scan_agg_bhm,hdk_bhm_reduce_slabs,hdk_bhm_fold";
}

// the fold of `fold_count` slabs of the dense table (internal entry i = key_lo + i ...) into the plan's output, and the armed fallback
static int32_t launch_bhm_folds(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                                const hdk_hip_device_properties* props, hipStream_t s, const BhmArgs& a, const BhmGeom& g, uint32_t entries,
                                const int64_t* fold_slabs, uint32_t fold_count) {
  int32_t st;
  if (g.perfect) {
    st = launch_finalize_slabs(d_plan, fold_slabs, kp.groupby_buf, fold_count, plan->entry_count, a.flag, s);
    if (st) return st;
  } else {
    BhmFoldArgs f;
    memset(&f, 0, sizeof(f));
    f.plan = d_plan;
    f.kp = kp;
    f.slabs = fold_slabs;
    f.flag = a.flag;
    f.num_slabs = fold_count;
    f.entries = entries;
    f.out_entry_count = plan->entry_count;
    f.wpe = a.wpe;
    for (int w = 0; w < a.wpe; ++w) f.wop[w] = g.wop[w];
    f.nword_mask = g.nword_mask;
    f.key_form = a.key_form;
    f.key_lo = g.key_lo;
    f.null_entry = g.null_entry;
    f.key_null_word = a.key_null_word;
    if (fold_count <= 16 && entries >= 2048) {  // few slabs of a large table: a thread per entry
      hipLaunchKernelGGL(hdk_bhm_fold<1>, dim3((entries + 255) / 256), dim3(256), 0, s, f);
    } else {
      hipLaunchKernelGGL(hdk_bhm_fold<0>, dim3((entries + 3) / 4), dim3(256), 0, s, f);
    }
    HDK_HIP_CHECK(hipGetLastError());
  }
  if (hdk_sw(SW_BHM_FLAG_IS_ERROR)) {  // (tests: "this input must stay on the fast path" -- the flag becomes an error code)
    hipLaunchKernelGGL(hdk_bhm_flag_is_error<0>, dim3(1), dim3(1), 0, s, a.flag, kp.error_code);
    HDK_HIP_CHECK(hipGetLastError());
    return HDK_HIP_OK;
  }
  // armed: runs only when the flag says the statistics did not hold (the folds skipped then)
  return launch_scan_global_armed(plan, d_plan, kp, ko, props, s, a.flag);
}

static int32_t launch_bhm_part(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                               const hdk_hip_device_properties* props, hipStream_t s, BhmPartArgs& pg, const BhmGeom& g, const BhmPartLayout& l,
                               bool* launched) {
  AsyncScratch scratch(s);
  if (hipMallocAsync(&scratch.p, l.total, s) != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return HDK_HIP_OK;  // no room for the tuples: the other strategies
  }
  int8_t* base = static_cast<int8_t*>(scratch.p);
  HDK_HIP_CHECK(hipMemsetAsync(base, 0, 256 + l.cursor_bytes, s));
  BhmArgs& a = pg.b;
  a.plan = d_plan;
  a.kp = kp;
  a.flag = reinterpret_cast<uint32_t*>(base);
  pg.fill = reinterpret_cast<uint32_t*>(base + 256);
  pg.layout = pg.fill + static_cast<size_t>(pg.nbins) * kPbXcds * kPbCursorStride;  // (zeroed with the cursors)
  pg.tuples = reinterpret_cast<uint32_t*>(base + 256 + l.cursor_bytes);
  a.slabs = reinterpret_cast<int64_t*>(base + 256 + l.cursor_bytes + l.tuple_bytes);
  const void* sk = bhm_scatter_kernel(a, g);
  const size_t scatter_lds = bhm_scatter_lds(g.width);
  HDK_HIP_CHECK(hipFuncSetAttribute(sk, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(scatter_lds)));
  const unsigned g1 = scatter_grid(sk, kBhmPartBlock, scatter_lds, props, kBhmPartWaves * 256 / kBhmPartBlock);
  void* kargs[] = {&pg};
  // the sample of the keys, the sub-slabs made from it
  // (few blocks: every block ends with one global add per bin -- 3 900 blocks met on the 196 counters for 180 us)
  const uint64_t sampled_tiles = ko->total_rows / (256ull * (16 / g.width)) / pg.sample_stride + 1;
  const unsigned g0 = static_cast<unsigned>(std::min<uint64_t>((sampled_tiles + 3) / 4, static_cast<uint64_t>(props->num_cu)));
  HDK_HIP_CHECK(hipLaunchKernel(bhm_sample_kernel(g), dim3(g0), dim3(256), kargs, 0, s));
  HDK_HIP_CHECK(hipLaunchKernel(reinterpret_cast<const void*>(hdk_bhm_part_layout<0>), dim3(1), dim3(kPbMaxBins), kargs, 0, s));
  HDK_HIP_CHECK(hipLaunchKernel(sk, dim3(g1), dim3(kBhmPartBlock), kargs, scatter_lds, s));
  const void* ak = bhm_aggregate_kernel(a, g, static_cast<int>(pg.tw));
  // pass B reads CODES (value - min + 1) out of the tuples: its own copy of the descriptors, shifted to them (pass A codes the
  // columns by their statistics and needs those as they are)
  BhmPartArgs pgb = pg;
  bhm_shift_to_codes(&pgb.b);
  void* kargs_b[] = {&pgb};
  if (a.lds_bytes > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(ak, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(a.lds_bytes)));
  }
  {
    BhmSlabInitArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.slabs = a.slabs;
    ia.words = static_cast<uint64_t>(kPbXcds) * pg.total_entries * static_cast<uint32_t>(a.wpe);
    ia.wpe = a.wpe;
    for (int w = 0; w < a.wpe; ++w) ia.wop[w] = g.wop[w];
    const unsigned gi = static_cast<unsigned>(std::min<uint64_t>((ia.words + 255) / 256, static_cast<uint64_t>(props->num_cu) * 8));
    hipLaunchKernelGGL(hdk_bhm_slab_init<0>, dim3(gi), dim3(256), 0, s, ia);
    HDK_HIP_CHECK(hipGetLastError());
  }
  HDK_HIP_CHECK(hipLaunchKernel(ak, dim3(pg.nbins * kPbXcds, pg.parts), dim3(kBhmAggBlock), kargs_b, a.lds_bytes, s));
  // the eight slabs -> one (a thread per word, coalesced), then the fold with a thread per entry: folding the eight directly
  // took 110 - 135 us for 100 K entries (88 strided words per thread)
  BhmReduceArgs r;
  memset(&r, 0, sizeof(r));
  r.in = a.slabs;
  r.out = reinterpret_cast<int64_t*>(base + 256 + l.cursor_bytes);  // (the tuples are spent: their space takes the one slab)
  r.flag = a.flag;
  r.num_slabs = kPbXcds;
  r.groups = 1;
  r.words = pg.total_entries * static_cast<uint32_t>(a.wpe);
  r.wpe = a.wpe;
  for (int w = 0; w < a.wpe; ++w) r.wop[w] = g.wop[w];
  hipLaunchKernelGGL(hdk_bhm_reduce_slabs<0>, dim3((r.words + 255) / 256, 1), dim3(256), 0, s, r);
  HDK_HIP_CHECK(hipGetLastError());
  const int32_t st = launch_bhm_folds(plan, d_plan, kp, ko, props, s, a, g, pg.total_entries, r.out, 1);
  if (st) return st;
  *launched = true;
  return HDK_HIP_OK;
}

int32_t launch_bhm(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                   const hdk_hip_device_properties* props, hipStream_t s, bool* launched) {
  *launched = false;
  BhmArgs a;
  BhmGeom g;
  if (!match_bhm(plan, ko, &a, &g)) {
    BhmPartArgs pg;
    BhmPartLayout l;
    if (match_bhm_part(plan, ko, &pg, &g, &l)) return launch_bhm_part(plan, d_plan, kp, ko, props, s, pg, g, l, launched);
    return HDK_HIP_OK;
  }
  bool is_static;
  const void* k = bhm_kernel(a, g, &is_static);
  if (a.lds_bytes > (48u << 10)) {
    HDK_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(a.lds_bytes)));
  }
  const uint32_t cu = static_cast<uint32_t>(props->num_cu);
  uint32_t grid = std::min(resident_grid(k, g.block, a.lds_bytes, props), g.grid_per_cu * cu);
  if (const char* e = hdk_sw(SW_BHM_BLOCKS_PER_CU)) grid = std::max(1, atoi(e)) * cu;  // (measurements)
  // a block's rows must stay inside what the packed fields were sized for (twice the even share: tiles are dealt round robin)
  constexpr uint64_t kTileRows256 = 256ull * 4 * kBhmU;
  const uint64_t tile_rows = kTileRows256 * (g.block / 256);
  const uint64_t budget = a.max_rows_per_block > 4 * tile_rows ? a.max_rows_per_block - 4 * tile_rows : tile_rows;
  const uint64_t need = (2 * ko->total_rows + budget - 1) / budget;
  if (need > grid) grid = static_cast<uint32_t>(std::min<uint64_t>(need, 1u << 16));
  if (ko->grid_dim_x) grid = ko->grid_dim_x;
  // the blocks' slabs, and -- when there are many -- the few slabs they are reduced to before the fold
  const uint32_t words = a.entries * static_cast<uint32_t>(a.wpe);
  const uint32_t red_groups = grid > 16 ? 4u : 0u;
  const size_t slab_bytes = (static_cast<size_t>(grid) * words * 8 + 255) & ~static_cast<size_t>(255);
  const size_t red_bytes = static_cast<size_t>(red_groups) * words * 8;
  AsyncScratch scratch(s);
  if (hipMallocAsync(&scratch.p, 256 + slab_bytes + red_bytes, s) != hipSuccess) {
    (void)hipGetLastError();
    scratch.p = nullptr;
    return HDK_HIP_OK;  // no room for the slabs: the other strategies
  }
  HDK_HIP_CHECK(hipMemsetAsync(scratch.p, 0, 256, s));
  a.plan = d_plan;
  a.kp = kp;
  a.flag = static_cast<uint32_t*>(scratch.p);
  a.slabs = reinterpret_cast<int64_t*>(static_cast<int8_t*>(scratch.p) + 256);
  bhm_shift_to_codes(&a);  // (after the kernel was chosen by the arguments' forms)
  void* kargs[] = {&a};
  HDK_HIP_CHECK(hipLaunchKernel(k, dim3(grid), dim3(g.block), kargs, a.lds_bytes, s));
  const int64_t* fold_slabs = a.slabs;
  uint32_t fold_count = grid;
  if (red_groups) {
    BhmReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.in = a.slabs;
    r.out = reinterpret_cast<int64_t*>(static_cast<int8_t*>(scratch.p) + 256 + slab_bytes);
    r.flag = a.flag;
    r.num_slabs = grid;
    r.groups = red_groups;
    r.words = words;
    r.wpe = a.wpe;
    for (int w = 0; w < a.wpe; ++w) r.wop[w] = g.wop[w];
    hipLaunchKernelGGL(hdk_bhm_reduce_slabs<0>, dim3((words + 255) / 256, red_groups), dim3(256), 0, s, r);
    HDK_HIP_CHECK(hipGetLastError());
    fold_slabs = r.out;
    fold_count = red_groups;
  }
  const int32_t st = launch_bhm_folds(plan, d_plan, kp, ko, props, s, a, g, a.entries, fold_slabs, fold_count);
  if (st) return st;
  *launched = true;
  return HDK_HIP_OK;
}

}  // namespace hdk
