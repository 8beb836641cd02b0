// scan_bh_decl.h -- what the batched interpreter's body (scan_agg_vec.h, BH = true) needs to know about the LDS
// open-addressing table of scan_bh.h: types and declarations only, so that the translation units that never instantiate
// BH = true (scan_agg.hip) do not depend on the definitions.
#pragma once
#include "agg_common.h"

namespace hdk {

struct BhGeom {
  uint32_t out_entry_count;  // the plan's table
  uint32_t cap_log2;         // LDS capacity per replica = 1 << cap_log2
  uint32_t rep;              // power of two
  uint32_t lds_bytes;
};

// The words kept behind a key in LDS.  The interpreter keeps the slab words of agg_common.h as they are (nlw = wpe, identity
// map); the streaming kernel keeps every DISTINCT update once (rows, NULLs, sum, min, max of its one argument column) and
// maps the layout's words onto them: COUNT(y), SUM(y), MAX(y), MIN(y), AVG(y) are four LDS atomics per row, not nine.
struct BhLdsLayout {
  int32_t nlw;                         // LDS words per entry, the key word (index nlw) not counted
  int32_t lwop[kMaxWordsPerEntry];     // WordOp of LDS word i
  int32_t lmap[kMaxWordsPerEntry];     // word w of the WordLayout lives in LDS word lmap[w]
};

HDK_DEV void bh_lds_init(int64_t* lds, const BhLdsLayout& ll, uint32_t cap_log2, uint32_t rep, int tid, int block);
HDK_DEV int32_t bh_lds_find_or_claim(int64_t* lds, int64_t key, uint32_t cap_log2, uint32_t estride, uint32_t key_off);
HDK_DEV void bh_layout_identity(const WordLayout& wl, BhLdsLayout* ll);
template <int BLOCK>
HDK_DEV void bh_flush_block(const hdk_hip_plan* p, const WordLayout& wl, const BhLdsLayout& ll, int64_t* lds, const BhGeom& g,
                            int64_t* const* groupby_buf, uint64_t* s_col_off, int tid, int32_t& err, int64_t* slab);

}  // namespace hdk
