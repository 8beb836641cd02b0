// part_scatter_batch.h -- the run-staging scatter step shared by the partitioned join-table build (join_build_part.h)
// and the partitioned perfect-hash group-by (scan_agg_perfect_part.h): a block's batch of tuples -> LDS histogram by bin ->
// one claim per bin -> runs ordered by bin in LDS -> consecutive positions behind each bin's cursor.  (The sliced join
// and the open-addressing group-by have older copies of the same step welded to their tuple formats.)
#pragma once

namespace hdk {

constexpr int kPbBlock = 512;             // scatter passes
constexpr int kPbMaxBins = 256;
constexpr int kPbXcds = 8;
constexpr uint32_t kPbCursorStride = 32;  // level-1 cursors: one per 128-byte line
constexpr uint32_t kPbCursor2Stride = 8;  // level-2 cursors

// A batch of a block's tuples -> runs ordered by bin in LDS -> consecutive positions behind each bin's cursor.
// dynamic LDS: [kTile][TW] staging | uint8 bin of every staging slot
template <int TW, int VR>
struct PbStage {
  static constexpr int kTile = kPbBlock * VR;
  static constexpr size_t lds_bytes() { return static_cast<size_t>(kTile) * TW * 8 + kTile + 16; }
};

struct PbNoDrop {
  HDK_DEV void operator()(const int64_t*) const {}
};

// dest(bin, n) -> claims n positions of bin's slab and returns {first position, positions that exist}; drop(tuple): called
// for every tuple that found no position (its slab is full)
template <int TW, int VR, typename Claim, typename Addr, typename Drop = PbNoDrop>
HDK_DEV void pb_scatter_batch(const int64_t (&tup)[VR][TW], const uint32_t (&bin)[VR], const bool (&live)[VR], uint32_t* s_cnt,
                              uint4* s_run, uint32_t* s_total, int64_t* s_stage, uint8_t* s_binof, int64_t* out, Claim claim,
                              Addr addr, Drop drop = Drop()) {
  const int tid = threadIdx.x;
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
      claim(static_cast<uint32_t>(tid), n, &base, &nfit);
    }
    s_run[tid].y = nfit;
    s_run[tid].z = base;
  }
  if (tid < kWave) {  // exclusive scan of the counts
    uint32_t carry = 0;
    for (int c0 = 0; c0 < kPbMaxBins; c0 += kWave) {
      const uint32_t n = s_cnt[c0 + tid];
      uint32_t incl = n;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, kWave);
        if (tid >= d) {
          incl += v;
        }
      }
      s_run[c0 + tid].x = carry + incl - n;
      carry += __shfl(incl, kWave - 1, kWave);
    }
    if (tid == 0) {
      *s_total = carry;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < VR; ++r) {
    if (live[r]) {
      const uint32_t si = s_run[bin[r]].x + rank[r];
      s_binof[si] = static_cast<uint8_t>(bin[r]);
#pragma unroll
      for (int w = 0; w < TW; ++w) {
        s_stage[static_cast<size_t>(si) * TW + w] = tup[r][w];
      }
    }
  }
  if (tid < kPbMaxBins) {
    s_cnt[tid] = 0;
  }
  __syncthreads();
  const uint32_t total = *s_total;
  for (uint32_t i = tid; i < total; i += kPbBlock) {
    const uint32_t b = s_binof[i];
    const uint4 run = s_run[b];
    const uint32_t r = i - run.x;
    if (r < run.y) {
      int64_t* o = out + addr(b, static_cast<uint64_t>(run.z) + r) * TW;
#pragma unroll
      for (int w = 0; w < TW; ++w) {
        o[w] = s_stage[static_cast<size_t>(i) * TW + w];
      }
    } else {
      drop(s_stage + static_cast<size_t>(i) * TW);
    }
  }
  __syncthreads();
}

}  // namespace hdk
