// scan_project.h -- filter/project: one output row per input row that passes the filter (and the
// inner-join probes), compacted through a selection vector.
//
// Reference behaviour (QE/RowFuncBuilder.cpp:162-215, QE/GroupByRuntime.cpp:248-272): every passing
// row does `old = atomicAdd(total_matched, 1)` and writes its position + targets at output row `old`
// (get_scan_output_slot / get_columnar_scan_output_offset); rows past max_matched report -pos.
//
// MI355X form: the batch interpreter (vec_eval.h) produces pass[VR] per lane; a wave-level exclusive
// prefix sum over the lanes' pass counts (DPP-free shuffle scan, 64 lanes) turns the wave's selection
// vector into dense output positions with ONE global atomic per wave per batch instead of one per row.
// Output row order is therefore a permutation of the reference's (which is itself scheduling-dependent
// on a GPU); the set of rows is identical.
#pragma once
#include "watch.h"
#include "vec_eval.h"

namespace hdk {

constexpr int kProjBlock = 256;       // row-at-a-time kernel
constexpr int kProjBlockPlain = 1024;  // batched kernel without joins
constexpr int kProjBlockJoin = 512;    // batched kernel with inner one-to-one probes

struct ProjArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  uint32_t entry_count;  // == *MAX_MATCHED
};

HDK_DEV void store_slot(int8_t* p, int width, int64_t v) {
  switch (width) {
    case 1: *reinterpret_cast<int8_t*>(p) = static_cast<int8_t>(v); break;
    case 2: *reinterpret_cast<int16_t*>(p) = static_cast<int16_t>(v); break;
    case 4: *reinterpret_cast<int32_t*>(p) = static_cast<int32_t>(v); break;
    default: *reinterpret_cast<int64_t*>(p) = v; break;
  }
}

template <bool J, int BLOCK, bool KEYED = false>
HDK_DEV void scan_project_body(const ProjArgs& a) {
  __shared__ uint64_t s_col_off[HDK_HIP_MAX_TARGETS];
  __shared__ uint32_t s_wave_tot[2][BLOCK / kWave];
  __shared__ uint32_t s_block_base[2];
  const int wave = threadIdx.x / kWave;
  uint32_t iter = 0;
  const cplan_t p = to_const_as(a.plan);
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const bool columnar = p->output_columnar;
  const int nt = p->num_targets;
  if (columnar && tid < HDK_HIP_MAX_TARGETS) {
    s_col_off[tid] = columnar_slot_off(a.plan, a.entry_count, tid);
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const uint32_t max_matched = static_cast<uint32_t>(*a.kp.max_matched);
  constexpr int64_t kTileRows = static_cast<int64_t>(BLOCK) * VR;
  int64_t* buf = a.kp.groupby_buf[0];

  VecCtxT<J, KEYED> c;
  vec_ctx_init(c, p, tid, BLOCK);
  int32_t err = 0;
  int32_t slots_err = 0;
  __shared__ int32_t s_watch;
  const Watch watch = watch_begin(a.kp);

  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    c.cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      if (watch.flags) {  // (this loop has barriers: the block decides together)
        if (const int32_t w_ = watch_poll_block(watch, &s_watch)) {
          err = w_;
          tile = INT64_MAX - gridDim.x;
          break;
        }
      }
      const int64_t row0 = (tile - frag_tile_begin) * kTileRows;
      bool pass[VR];
      vec_ctx_tile(c, row0, nrows, pass);
      rows_pass_v(c, a.kp.join_hash_tables, pass, err);
      // ---- selection vector -> dense output positions ------------------------------------------
      uint32_t mine = 0;
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        mine += pass[r] ? 1u : 0u;
      }
      uint32_t incl = mine;  // inclusive scan over the wave's 64 lanes
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t n = __shfl_up(incl, d, kWave);
        if (lane >= d) {
          incl += n;
        }
      }
      // ONE claim per block and tile (2048 rows), not per wave: every claim is an atomic on the same
      // address (TOTAL_MATCHED) and those serialise at ~10 ns each -- at one per wave-batch (512 rows) they
      // alone took 5 ms per 256 M rows, whatever the selectivity.  Double-buffered by tile parity so that a
      // fast wave cannot overwrite the totals a slow wave is still reading.
      const uint32_t wave_total = __shfl(incl, kWave - 1, kWave);
      const int par = static_cast<int>(iter & 1);
      if (lane == 0) {
        s_wave_tot[par][wave] = wave_total;
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) {
          t += s_wave_tot[par][w];
        }
        s_block_base[par] = t ? static_cast<uint32_t>(atomicAdd(a.kp.total_matched, static_cast<int32_t>(t))) : 0u;
      }
      __syncthreads();
      uint32_t wave_base = s_block_base[par];
#pragma unroll
      for (int w = 0; w < BLOCK / kWave; ++w) {
        if (w < wave) {
          wave_base += s_wave_tot[par][w];
        }
      }
      ++iter;
      uint32_t out_pos = wave_base + incl - mine;
      // ---- project ------------------------------------------------------------------------------
      uint32_t pos_r[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        pos_r[r] = out_pos;
        if (pass[r]) {
          ++out_pos;
          if (pos_r[r] >= max_matched) {
            slots_err = -1 - static_cast<int32_t>(vrow(c, r) & 0x3fffffff);  // out of slots: negative
            pass[r] = false;
          }
        }
      }
      // All layout decisions below are wave-uniform and sit OUTSIDE the row loops (one scalar branch per
      // batch instead of one per row): the kernel is instruction-issue bound, not bandwidth bound.
      const size_t rq = p->row_size_quad;
      if (columnar) {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (pass[r]) {
            buf[pos_r[r]] = vrow(c, r);  // get_columnar_scan_output_offset
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          if (pass[r]) {
            buf[static_cast<size_t>(pos_r[r]) * rq] = vrow(c, r);  // get_scan_output_slot
          }
        }
      }
      for (int t = 0; t < nt; ++t) {
        ctarget_t tg = p->targets[t];
        int64_t v[VR];
        eval_expr_v(c, tg.arg, v, pass, err);
        const int w = tg.slot_width;
        int8_t* base = columnar ? reinterpret_cast<int8_t*>(buf) + s_col_off[t] : reinterpret_cast<int8_t*>(buf) + tg.slot_off;
        const size_t stride = columnar ? static_cast<size_t>(w) : rq * 8;
#define HDK_STORE_ROWS(T)                                                             \
  _Pragma("unroll") for (int r = 0; r < VR; ++r) {                                    \
    if (pass[r]) {                                                                    \
      *reinterpret_cast<T*>(base + static_cast<size_t>(pos_r[r]) * stride) = static_cast<T>(v[r]); \
    }                                                                                 \
  }
        switch (w) {
          case 1: HDK_STORE_ROWS(int8_t) break;
          case 2: HDK_STORE_ROWS(int16_t) break;
          case 4: HDK_STORE_ROWS(int32_t) break;
          default: HDK_STORE_ROWS(int64_t) break;
        }
#undef HDK_STORE_ROWS
      }
    }
    frag_tile_begin += ntiles;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  } else if (slots_err) {
    atomicCAS(a.kp.error_code, 0, slots_err);  // negative = ran out of slots (benign under a LIMIT)
  }
}

// Large blocks: the output claim is one same-address atomic per block and tile, so rows per tile =
// BLOCK x VR is what bounds the claim rate (1024 threads: 8192 rows per claim).  The join variant
// needs ~190 VGPRs, which caps its block at 512 threads.
extern "C" __global__ __launch_bounds__(kProjBlockPlain) void hdk_scan_project(ProjArgs a) {
  scan_project_body<false, kProjBlockPlain>(a);
}
extern "C" __global__ __launch_bounds__(kProjBlockJoin) void hdk_scan_project_join(ProjArgs a) {
  scan_project_body<true, kProjBlockJoin>(a);
}
extern "C" __global__ __launch_bounds__(kProjBlockJoin) void hdk_scan_project_keyed(ProjArgs a) {  // (vec_eval.h: keyed_probe_one)
  scan_project_body<true, kProjBlockJoin, true>(a);
}

// General form: any join the library accepts (one-to-many and keyed tables, LEFT joins), one row at
// a time through the join loop nest of device_common.h.  Every surviving row combination claims an
// output row; lanes that reach the claim together share one atomic (the active-lane mask is the
// selection vector: ballot + popcount inside the divergent loop).
extern "C" __global__ __launch_bounds__(kProjBlock) void hdk_scan_project_scalar(ProjArgs a) {
  __shared__ uint64_t s_col_off[HDK_HIP_MAX_TARGETS];
  const hdk_hip_plan* __restrict__ p = a.plan;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const bool columnar = p->output_columnar;
  const int nt = p->num_targets;
  if (columnar && tid < HDK_HIP_MAX_TARGETS) {
    s_col_off[tid] = columnar_slot_off(p, a.entry_count, tid);
  }
  __syncthreads();
  const uint64_t nfrag = *a.kp.num_fragments;
  const uint32_t ntab = *a.kp.num_tables;
  const uint32_t max_matched = static_cast<uint32_t>(*a.kp.max_matched);
  int64_t* buf = a.kp.groupby_buf[0];
  constexpr int64_t kTileRows = kProjBlock;

  RowCtx c;
  c.plan = p;
  c.join_row[0] = 0;
  c.join_row[1] = 0;
  int32_t err = 0;
  int32_t slots_err = 0;
  int64_t tile = blockIdx.x;
  int64_t frag_tile_begin = 0;
  const Watch watch = watch_begin(a.kp);
  for (uint64_t f = 0; f < nfrag; ++f) {
    const int64_t nrows = a.kp.num_rows[f * ntab];
    const int64_t ntiles = (nrows + kTileRows - 1) / kTileRows;
    c.cols = a.kp.col_buffers[f];
    for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
      HDK_WATCH_TILE(watch, err, tile)
      const int64_t row = (tile - frag_tile_begin) * kTileRows + tid;
      if (row >= nrows) {
        continue;
      }
      c.pos = row;
      for_each_row_match(c, a.kp.join_hash_tables, err, [&]() {
        const uint64_t active = __ballot(1);
        const int leader = __ffsll(static_cast<long long>(active)) - 1;
        uint32_t base = 0;
        if (lane == leader) {
          base = static_cast<uint32_t>(atomicAdd(a.kp.total_matched, static_cast<int32_t>(__popcll(active))));
        }
        base = __shfl(base, leader, kWave);
        const uint32_t pos = base + static_cast<uint32_t>(__popcll(active & ((1ull << lane) - 1)));
        if (pos >= max_matched) {
          slots_err = -1 - static_cast<int32_t>(row & 0x3fffffff);
          return;
        }
        if (columnar) {
          buf[pos] = row;
        } else {
          buf[static_cast<size_t>(pos) * p->row_size_quad] = row;
        }
        for (int t = 0; t < nt; ++t) {
          const hdk_hip_target& tg = p->targets[t];
          const int64_t v = eval_expr(c, tg.arg, err);
          const int w = tg.slot_width;
          int8_t* dst = columnar ? reinterpret_cast<int8_t*>(buf) + s_col_off[t] + static_cast<size_t>(pos) * w
                                 : reinterpret_cast<int8_t*>(buf + static_cast<size_t>(pos) * p->row_size_quad) + tg.slot_off;
          store_slot(dst, w, v);
        }
      });
    }
    frag_tile_begin += ntiles;
  }
  if (err) {
    record_error(a.kp.error_code, err);
  } else if (slots_err) {
    atomicCAS(a.kp.error_code, 0, slots_err);
  }
}

}  // namespace hdk
