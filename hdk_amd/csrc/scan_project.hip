// scan_project.hip -- filter/project strategies (Projection plans): the two-pass direct kernels
// (scan_project_fast.h), the batched interpreter with and without join probes and the row-at-a-time one
// (scan_project.h), with their matcher.  A translation unit of its own (see host_match.h).
#include "host_match.h"
#include "scan_project.h"
#include "scan_project_fast.h"

namespace hdk {

// the shape hdk_scan_project_direct takes (scan_project_fast.h)
static bool match_project_fast(const hdk_hip_plan* p, ProjFastArgs* fa) {
  if (p->query_kind != HDK_Q_PROJECTION || p->num_joins > 1 || p->num_quals > kProjFastMaxQuals) return false;
  memset(fa, 0, sizeof(*fa));
  if (!match_plain_quals(p, fa->q)) return false;  // (filters on outer columns only: a filter on a joined column is the interpreter's)
  fa->nquals = p->num_quals;
  if (p->num_joins) {
    // one inner-like join on a one-to-one perfect-hash table, probed with a plain integer column of the outer table
    const hdk_hip_join& jn = p->joins[0];
    int kc;
    if ((jn.kind != HDK_JOIN_ONE_TO_ONE && jn.kind != HDK_JOIN_ONE_TO_ONE_FUSED) || !join_type_inner_like(jn.type) || jn.table_idx != 0 ||
        p->num_filter_ops || hdk_sw(SW_PROJECT_NO_FAST_JOIN)) {
      return false;
    }
    if (!plain_outer_col(p, jn.outer_key, &kc) || (p->cols[kc].kind != HDK_COL_INT && p->cols[kc].kind != HDK_COL_UNSIGNED)) return false;
    fa->join = 1;
    fa->jn = jn;
    fa->jkey.buf_idx = p->cols[kc].buf_idx;
    fa->jkey.width = p->cols[kc].width;
    fa->jkey.kind = p->cols[kc].kind;
  }
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target& tg = p->targets[t];
    int c;
    if (tg.agg != HDK_AGG_ID) return false;
    if (!plain_outer_col(p, tg.arg, &c)) {
      // a column of the joined row: payload word of the fused entry, or the inner column at the row id
      if (!fa->join || tg.arg.nsteps != 0 || tg.arg.leaf0.kind != HDK_LEAF_COL) return false;
      c = tg.arg.leaf0.col;
      const hdk_hip_col& jc = p->cols[c];
      if (jc.table == -1 && fa->jn.kind == HDK_JOIN_ONE_TO_ONE_FUSED && jc.buf_idx >= 1 && jc.buf_idx < fa->jn.fused_stride) {
        fa->t[t].src = PF_SRC_PAYLOAD;
      } else if (jc.table == 1 && fa->jn.kind == HDK_JOIN_ONE_TO_ONE && jc.kind != HDK_COL_SMALL_DATE) {
        fa->t[t].src = PF_SRC_INNER;
      } else {
        return false;
      }
    }
    const hdk_hip_col& col = p->cols[c];
    if (col.kind == HDK_COL_FLOAT) return false;
    fa->t[t].col.buf_idx = col.buf_idx;
    fa->t[t].col.width = col.width;
    fa->t[t].col.kind = col.kind;
    fa->t[t].slot_width = tg.slot_width;
    fa->t[t].slot_off = tg.slot_off;
  }
  fa->ntargets = p->num_targets;
  fa->columnar = p->output_columnar;
  fa->row_size_quad = p->row_size_quad;
  fa->entry_count = p->entry_count;
  // rows dealt in adjacent pairs (16-byte loads, scan_project_fast.h): every filter column is an 8-byte integer or
  // double column compared in its own class
  fa->pairs = fa->nquals > 0;
  for (int i = 0; i < fa->nquals; ++i) {
    const ProjFastQual& fq = fa->q[i];
    if (fq.col.width != 8 || (fq.col.kind != HDK_COL_INT && fq.col.kind != HDK_COL_DOUBLE) || (fq.fp != 0) != (fq.col_fp != 0)) {
      fa->pairs = 0;
    }
  }
  for (int i = 0; i < fa->nquals; ++i) {
    for (int t = 0; t < fa->ntargets; ++t) {
      if (fa->t[t].src == PF_SRC_OUTER && fa->q[i].col.buf_idx == fa->t[t].col.buf_idx) fa->keep_cached = 1;
    }
  }
  // columnar target columns: [int64 row positions][target columns, each aligned to 8]
  size_t off = (static_cast<size_t>(p->entry_count) * 8 + 7) & ~size_t(7);
  for (int t = 0; t < p->num_targets; ++t) {
    off = (off + 7) & ~size_t(7);
    fa->col_off[t] = off;
    off += static_cast<size_t>(p->entry_count) * p->targets[t].slot_width;
  }
  return true;
}

// The one-pass form (HDK_HIP_PROJECT_ONE_PASS=1: measured slower than the two passes at 1 % and at 50 % selectivity,
// DESIGN.md 3.6, and kept behind the switch) needs a status word per batch of tiles, so the launch has to state its row
// count.
static bool project_one_pass(const hdk_hip_kernel_options* ko, const ProjFastArgs* pf = nullptr) {
  return ko && ko->total_rows && hdk_sw(SW_PROJECT_ONE_PASS) && !(pf && pf->join);
}

uint32_t project_grid(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props) {
  const bool scalar = (ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_SCALAR)) || needs_join_loops(p);
  const void* k;
  int block;
  ProjFastArgs pf;
  if (!launch_forces_generic(ko) && match_project_fast(p, &pf)) {
    k = reinterpret_cast<const void*>(hdk_scan_project_direct);
    block = kProjFastBlock;
  } else {
    k = scalar ? reinterpret_cast<const void*>(hdk_scan_project_scalar)
               : (p->num_joins ? (plan_has_keyed_join(p) ? reinterpret_cast<const void*>(hdk_scan_project_keyed)
                                                         : reinterpret_cast<const void*>(hdk_scan_project_join))
                               : reinterpret_cast<const void*>(hdk_scan_project));
    block = scalar ? kProjBlock : (p->num_joins ? kProjBlockJoin : kProjBlockPlain);
  }
  return resident_grid(k, block, 0, props);
}

void project_describe(const hdk_hip_plan* plan, const hdk_hip_kernel_options* ko, char* out, size_t out_len) {
  ProjFastArgs pf;
  if (!launch_forces_generic(ko) && match_project_fast(plan, &pf)) {
    snprintf(out, out_len, "%shdk_scan_project_count,hdk_scan_project_offsets,hdk_scan_project_dense,hdk_scan_project_direct",
             project_one_pass(ko, &pf) ? "hdk_scan_project_stream," : "");
  } else {
    snprintf(out, out_len, "%s", needs_join_loops(plan) ? "hdk_scan_project_scalar"
                                 : plan->num_joins     ? (plan_has_keyed_join(plan) ? "hdk_scan_project_keyed" : "hdk_scan_project_join")
                                                       : "hdk_scan_project");
  }
}

int32_t launch_project(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                       const hdk_hip_kernel_options* ko, const LaunchShape& shape, const hdk_hip_device_properties* props,
                       hipStream_t s) {
  ProjArgs pa;
  pa.plan = d_plan;
  pa.kp = kp;
  pa.entry_count = plan->entry_count;
  ProjFastArgs pf;
  const bool generic = ko && (ko->flags & (HDK_HIP_LAUNCH_FORCE_GENERIC | HDK_HIP_LAUNCH_FORCE_SCALAR));
  if (!generic && match_project_fast(plan, &pf)) {
    pf.kp = kp;
    // ---- one pass (decoupled look-back over batches of tiles), the two passes armed behind it ----------------------
    AsyncScratch status_mem(s);
    bool streamed = false;
    if (project_one_pass(ko, &pf)) {
      // a status word per batch: the stated rows in full tiles, plus one ragged tile per fragment -- 64 K of them
      // (the kernel counts the real tiles and hands the launch to the two passes if they do not fit)
      uint64_t slack = 65536;
      if (const char* e = hdk_sw(SW_PROJECT_STATUS_SLACK)) slack = strtoull(e, nullptr, 10);  // (tests)
      const uint64_t tiles = ko->total_rows / (static_cast<uint64_t>(kProjFastBlock) * kProjFastVR) + slack;
      const uint64_t cap = tiles / kProjFastGroup + 1;
      const size_t bytes = 16 + cap * sizeof(uint64_t);
      if (hipMallocAsync(&status_mem.p, bytes, s) == hipSuccess) {
        HDK_HIP_CHECK(hipMemsetAsync(status_mem.p, 0, bytes, s));
        pf.ticket = static_cast<uint32_t*>(status_mem.p);
        pf.status = reinterpret_cast<uint64_t*>(static_cast<int8_t*>(status_mem.p) + 16);
        pf.status_cap = cap;
        const void* k = pf.pairs ? reinterpret_cast<const void*>(hdk_scan_project_stream_pairs)
                                 : reinterpret_cast<const void*>(hdk_scan_project_stream);
        const uint32_t grid = resident_grid(k, kProjFastBlock, 0, props);
        if (pf.pairs) {
          hipLaunchKernelGGL(hdk_scan_project_stream_pairs, dim3(grid), dim3(kProjFastBlock), 0, s, pf);
        } else {
          hipLaunchKernelGGL(hdk_scan_project_stream, dim3(grid), dim3(kProjFastBlock), 0, s, pf);
        }
        pf.run_if = pf.ticket + 1;
        streamed = true;
      } else {
        (void)hipGetLastError();
        status_mem.p = nullptr;
      }
    }
    AsyncScratch counts_mem(s), mask_mem(s);  // stream-ordered scratch: pass-1 counts / pass-2 offsets per block
    HDK_HIP_CHECK(hipMallocAsync(&counts_mem.p, (static_cast<size_t>(shape.grid) + 1) * sizeof(uint32_t), s));
    uint32_t* counts = static_cast<uint32_t*>(counts_mem.p);
    pf.block_counts = counts;
    pf.mode = counts + shape.grid;  // (0 until the offsets kernel decides: if it does not run, neither does a writing pass)
    HDK_HIP_CHECK(hipMemsetAsync(pf.mode, 0, sizeof(uint32_t), s));
    // selection bitmask handed from the counting pass to the writing pass: rows/8 bytes when the caller
    // states the row count (plus room for one partial tile per fragment, up to 1024 fragments; tiles past
    // the end re-evaluate the filter).  No scratch, no mask: pass 2 then decodes the filter columns again.
    pf.sel_mask = nullptr;
    pf.sel_tiles = 0;
    if (ko && ko->total_rows && !streamed) {  // (armed behind the one-pass kernel the passes do without it)
      const uint64_t tiles = ko->total_rows / (static_cast<uint64_t>(kProjFastBlock) * kProjFastVR) + 1024;
      if (hipMallocAsync(&mask_mem.p, tiles * kProjFastBlock, s) == hipSuccess) {
        pf.sel_mask = static_cast<uint8_t*>(mask_mem.p);
        pf.sel_tiles = tiles;
      } else {
        (void)hipGetLastError();
        mask_mem.p = nullptr;
      }
    }
    if (pf.pairs) {
      hipLaunchKernelGGL(hdk_scan_project_count_pairs, dim3(shape.grid), dim3(kProjFastBlock), 0, s, pf);
    } else {
      hipLaunchKernelGGL(hdk_scan_project_count, dim3(shape.grid), dim3(kProjFastBlock), 0, s, pf);
    }
    uint32_t force = 0;  // (HDK_HIP_PROJECT_WRITER=sparse|dense: A/B measurements)
    if (const char* e = hdk_sw(SW_PROJECT_WRITER)) force = e[0] == 'd' ? 2u : e[0] == 's' ? 1u : 0u;
    if (pf.join) force = 1u;  // joined columns are gathered row by row: the sparse writing pass
    hipLaunchKernelGGL(hdk_scan_project_offsets, dim3(1), dim3(1024), 0, s, counts, shape.grid, kp.total_matched, pf.run_if, kp,
                       pf.mode, force);
    // the two writing passes: the offsets kernel has picked one, the other returns at once
    if (pf.pairs) {
      hipLaunchKernelGGL(hdk_scan_project_dense_pairs, dim3(shape.grid), dim3(kProjFastBlock), 0, s, pf);
      hipLaunchKernelGGL(hdk_scan_project_direct_pairs, dim3(shape.grid), dim3(kProjFastBlock), 0, s, pf);
    } else {
      hipLaunchKernelGGL(hdk_scan_project_dense, dim3(shape.grid), dim3(kProjFastBlock), 0, s, pf);
      hipLaunchKernelGGL(hdk_scan_project_direct, dim3(shape.grid), dim3(kProjFastBlock), 0, s, pf);
    }
    HDK_HIP_CHECK(hipGetLastError());
  } else if (needs_join_loops(plan) || (ko && (ko->flags & HDK_HIP_LAUNCH_FORCE_SCALAR))) {
    hipLaunchKernelGGL(hdk_scan_project_scalar, dim3(shape.grid), dim3(kProjBlock), 0, s, pa);
  } else if (plan->num_joins && plan_has_keyed_join(plan)) {
    hipLaunchKernelGGL(hdk_scan_project_keyed, dim3(shape.grid), dim3(kProjBlockJoin), 0, s, pa);
  } else if (plan->num_joins) {
    hipLaunchKernelGGL(hdk_scan_project_join, dim3(shape.grid), dim3(kProjBlockJoin), 0, s, pa);
  } else {
    hipLaunchKernelGGL(hdk_scan_project, dim3(shape.grid), dim3(kProjBlockPlain), 0, s, pa);
  }
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

}  // namespace hdk
