// host_match.h -- host-side helpers shared by the scan translation units (scan_agg.hip: API + LDS strategies;
// scan_baseline.hip: open-addressing strategies; scan_project.hip: filter/project), and what they export to each other.
#pragma once
#include <string.h>

#include "host_common.h"
#include "switches.h"
#include "launch_common.h"
#include "plain_quals.h"

namespace hdk {

// Persistent grids are sized from what actually fits: blocks per CU (register / LDS limited) x CUs, so that
// every block is resident and the static tile walk has no second, partly filled round (the batched
// interpreter at 147 VGPRs fits 3 blocks per CU: 1024 blocks ran as 768 + 256 -- taxi Q3 1.43 ms -- while
// 768 blocks take 1.14 ms).
inline uint32_t resident_grid(const void* kernel, int block, size_t lds_bytes, const hdk_hip_device_properties* props) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds_bytes) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    return static_cast<uint32_t>(props->grid_size);
  }
  return static_cast<uint32_t>(per_cu) * static_cast<uint32_t>(props->num_cu);
}

// grid of a radix-scatter pass: resident blocks, held to `per_cu_default` per CU when that is not 0
// (HDK_HIP_SCATTER_BLOCKS_PER_CU overrides; 0 = all resident blocks).  The join's level 1 with three 512-thread blocks on a
// CU is 6-12 % slower than with two (scripts/microbench/scatter_runs.hip: 1.38 / 1.21 ms per 256 M rows, fewer bytes
// written: 2.57 against 2.67 GB -- fewer half-written lines in flight per L2); the group-by's level 1 does not care.
inline uint32_t scatter_grid(const void* kernel, int block, size_t lds_bytes, const hdk_hip_device_properties* props,
                             uint32_t per_cu_default = 0) {
  uint32_t g = resident_grid(kernel, block, lds_bytes, props);
  uint32_t per_cu = per_cu_default;
  if (const char* e = hdk_sw(SW_SCATTER_BLOCKS_PER_CU)) per_cu = static_cast<uint32_t>(atoi(e));
  const uint32_t cap = per_cu * static_cast<uint32_t>(props->num_cu);
  if (cap >= 1 && cap < g) g = cap;
  return g;
}

inline bool plain_outer_col(const hdk_hip_plan* p, const hdk_hip_expr& e, int* col) {
  if (e.nsteps != 0 || e.leaf0.kind != HDK_LEAF_COL) return false;
  const hdk_hip_col& c = p->cols[e.leaf0.col];
  if (c.table != 0) return false;
  if (c.kind == HDK_COL_SMALL_DATE) return false;  // decoded (x 86400, narrow NULL) by the plan interpreters only
  *col = e.leaf0.col;
  return true;
}

// INNER, or SEMI: the same probe over a table whose fill let the first row of a key win
inline bool join_type_inner_like(int32_t type) { return type == HDK_JOIN_INNER || type == HDK_JOIN_SEMI; }

// some column of the plan is a DATE in days: none of the specialised kernels decodes it
inline bool plan_reads_small_dates(const hdk_hip_plan* p) {
  for (int i = 0; i < p->num_cols; ++i) {
    if (p->cols[i].kind == HDK_COL_SMALL_DATE) return true;
  }
  return false;
}

// joins only the row-at-a-time interpreter walks: matching sets with more than one row, keyed tables,
// LEFT joins (the batched interpreter handles inner one-to-one probes)
// an aggregate plan whose ONE join probes a one-to-many perfect-hash table: the batched interpreter replays the batch once
// per match (hdk_scan_agg_vec_many; vec_eval.h: vec_round_v).  Projections claim output rows per tile and stay row at a time.
inline bool plan_is_single_matching_set_join(const hdk_hip_plan* p) {
  return p->num_joins == 1 && p->joins[0].kind == HDK_JOIN_ONE_TO_MANY && p->query_kind != HDK_Q_PROJECTION &&
         !hdk_sw(SW_NO_BATCHED_MATCHING_SETS);
}

inline bool needs_join_loops(const hdk_hip_plan* p) {
  if (plan_is_single_matching_set_join(p)) {
    return false;
  }
  // (OR / NOT filter programs run in the batched interpreters too: vec_eval.h, filter_program_pass_v)
  for (int j = 0; j < p->num_joins; ++j) {
    const hdk_hip_join& jn = p->joins[j];
    // at most one partner per row: INNER / SEMI drop the row without one, LEFT keeps it with NULL inner columns, ANTI keeps
    // only those -- no loop (the batched interpreters, vec_eval.h: rows_pass_v); matching sets need the loop nest
    if (jn.kind != HDK_JOIN_ONE_TO_ONE && jn.kind != HDK_JOIN_ONE_TO_ONE_FUSED && jn.kind != HDK_JOIN_KEYED_ONE_TO_ONE) {
      return true;
    }
  }
  return false;
}

// some join of a plan the batched interpreters take probes a keyed one-to-one table: their *_keyed kernels
inline bool plan_has_keyed_join(const hdk_hip_plan* p) {
  for (int j = 0; j < p->num_joins; ++j) {
    if (p->joins[j].kind == HDK_JOIN_KEYED_ONE_TO_ONE) return true;
  }
  return false;
}

// filters of the form `outer column cmp literal` (plain_quals.h); false when any conjunct has another shape.
// allow_program: the caller's kernel evaluates its filters through plain_quals_pass and nothing else, so an AND / OR / NOT
// program over such leaves (hdk_hip_plan::filter_ops) is fine too: it rides in out[0]
// deepest value stack a postfix filter program needs (hdk_hip_plan::filter_ops): the streaming kernels' evaluators
// (plain_quals.h: plain_quals_program; scan_agg_fast.h: fast_x_program) keep kPlainProgStack values in registers, so a
// program that nests deeper -- `(a AND b) OR ((a AND c) OR (b AND c))` pushes four -- must go to the interpreter
// (vec_eval.h keeps 16).  plan.py deduplicates leaves, so three quals can need any depth.
constexpr int kPlainProgStack = 3;
inline int filter_program_depth(const hdk_hip_plan* p) {
  int depth = 0, deepest = 0;
  for (int i = 0; i < p->num_filter_ops; ++i) {
    const uint32_t op = p->filter_ops[i];
    if (op < HDK_F_AND) {
      ++depth;
    } else if (op != HDK_F_NOT) {
      --depth;
    }
    if (depth > deepest) deepest = depth;
  }
  return deepest;
}

inline bool match_plain_quals(const hdk_hip_plan* p, ProjFastQual* out, bool allow_program = false) {
  if (p->num_quals > kMaxPlainQuals) return false;
  if (p->num_filter_ops && (!allow_program || p->num_filter_ops > kMaxPlainProg || p->num_joins || p->num_quals == 0 ||
                            filter_program_depth(p) > kPlainProgStack)) return false;
  for (int i = 0; i < p->num_quals; ++i) {
    const hdk_hip_qual& q = p->quals[i];
    int c;
    if (!plain_outer_col(p, q.lhs, &c)) return false;
    if (q.rhs.kind != HDK_LEAF_INT && q.rhs.kind != HDK_LEAF_FP) return false;
    const hdk_hip_col& col = p->cols[c];
    ProjFastQual& fq = out[i];
    fq.col.buf_idx = col.buf_idx;
    fq.col.width = col.width;
    fq.col.kind = col.kind;
    fq.cmp = q.cmp;
    fq.nullable = q.lhs.leaf0.nullable;
    fq.null_val = q.lhs.leaf0.null_val;
    fq.col_fp = col.kind == HDK_COL_FLOAT || col.kind == HDK_COL_DOUBLE;
    const bool rhs_fp = q.rhs.kind == HDK_LEAF_FP;
    fq.fp = fq.col_fp || rhs_fp;
    if (fq.fp && !rhs_fp) {
      const double d = static_cast<double>(q.rhs.ival);
      memcpy(&fq.rhs, &d, 8);
    } else {
      fq.rhs = q.rhs.ival;
    }
    fq.nprog = 0;
  }
  if (p->num_filter_ops) {
    out[0].nprog = p->num_filter_ops;
    for (int i = 0; i < p->num_filter_ops; ++i) out[0].prog[i] = p->filter_ops[i];
  }
  return true;
}

inline uint32_t pow2_ceil_log2(uint64_t x) {
  uint32_t l = 0;
  while ((1ull << l) < x) ++l;
  return l;
}

// unsigned 32-bit division by an invariant divisor d >= 2, round-up method in its branch-free form:
//   t = mulhi(magic, n);  q = (((n - t) >> 1) + t) >> shift        (exact for every 32-bit n)
inline void magic_u32(uint32_t d, uint32_t* magic, uint32_t* shift) {
  uint32_t log2d = 31;
  while (!(d >> log2d)) --log2d;
  if ((d & (d - 1)) == 0) {
    *magic = 0;
    *shift = log2d - 1;
    return;
  }
  const uint64_t two_k = 1ull << (32 + log2d);
  const uint64_t m = two_k / d;
  const uint32_t rem = static_cast<uint32_t>(two_k - m * d);
  uint32_t m32 = static_cast<uint32_t>(m) * 2u;
  const uint32_t twice_rem = rem * 2u;
  if (twice_rem >= d || twice_rem < rem) m32 += 1;
  *magic = m32 + 1u;
  *shift = log2d;
}

inline bool launch_forces_generic(const hdk_hip_kernel_options* ko) {
  return ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR));
}

// ---- scan_agg.hip ---------------------------------------------------------------------------------------------------
int32_t validate_plan(const hdk_hip_plan* p);         // everything a launch reads
int32_t validate_plan_layout(const hdk_hip_plan* p);  // the output layout only (reductions, table helpers)
// head of every launch: plan copy + the launch's interrupt / watchdog words into the front of `workspace`
// (kPlanRegionBytes), the 12 launch pointers into a KernParams
int32_t launch_head(const hdk_hip_plan* plan, int8_t* const params[HDK_KP_COUNT], const hdk_hip_kernel_options* ko,
                    int32_t device_id, void* workspace, hipStream_t s, hdk_hip_plan** d_plan_out, KernParams* kp_out);
// HIP events around a launch's dominant kernels (HDK_HIP_LAUNCH_RECORD_EVENTS): begin returns the pair to record
int32_t scan_events_begin(int32_t device_id, hipStream_t s, hipEvent_t* e0, hipEvent_t* e1);

// ---- scan_fast.hip: the streaming kernel hdk_scan_agg_direct (scan_agg_fast.h), key width kw, value width vw ---------
struct FastArgs;
int32_t launch_fast_direct(int kw, int vw, const FastArgs& fa, const LaunchShape& shape, hipStream_t s);

// ---- scan_cols.hip: non-grouped aggregates over several plain columns (scan_agg_cols.h: hdk_scan_agg_cols) -----------
struct ColsArgs;
int32_t launch_cols(const ColsArgs& ca, const LaunchShape& shape, hipStream_t s);

// ---- scan_baseline.hip: GroupByBaselineHash plans and perfect-hash tables too big for LDS (STRAT_GLOBAL) ----------
// persistent grid of the kernel that will run (x 4: random atomics make block run times uneven)
uint32_t baseline_grid(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props);
// kernel names of the launch, comma separated
void baseline_describe(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, char* out, size_t out_len);
// the whole strategy: radix-partitioned passes when the shape allows, else the global-atomics kernels.  `init_output`
// = HDK_HIP_LAUNCH_INIT_OUTPUT (fused into the partitioned pass 3, or the init kernel first)
int32_t launch_baseline(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                        const hdk_hip_kernel_options* ko, const LaunchShape& shape, bool init_output,
                        const hdk_hip_device_properties* props, hipStream_t s);

// multi-GPU tuple exchange (include/hdk_hip.h); `cursors` / `scratch`: the caller's workspace behind the plan region
int32_t exchange_shape(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko, int32_t num_owners,
                       uint32_t owner_entry_count, struct PartArgs* pa, hdk_hip_exchange_shape* out);
int32_t launch_scatter_to_owners(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                                 const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape, int8_t* send,
                                 int8_t* cursors, const hdk_hip_device_properties* props, hipStream_t s);
int32_t launch_aggregate_from_ranks(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                                    const hdk_hip_kernel_options* ko, const hdk_hip_exchange_shape* shape,
                                    const int8_t* recv, int8_t* scratch, const hdk_hip_device_properties* props,
                                    hipStream_t s);

// ---- scan_project.hip: Projection plans (STRAT_PROJECT) ---------------------------------------------------------------
uint32_t project_grid(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props);
void project_describe(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, char* out, size_t out_len);
int32_t launch_project(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                       const hdk_hip_kernel_options* ko, const LaunchShape& shape, const hdk_hip_device_properties* props,
                       hipStream_t s);

}  // namespace hdk
