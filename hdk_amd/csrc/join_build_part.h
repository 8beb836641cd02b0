// join_build_part.h -- one-to-one perfect-hash join table built by PARTITIONING instead of random atomics.
//
// k_join_build<BUILD_ONE_TO_ONE> does one 4-byte compare-and-swap per inner row at a random slot of the table: for a
// 100 M-key dimension that is 1e8 memory-side atomics = 4.4 ms (the device does ~2.4e10 of them a second, whatever the
// table size -- profiles/microbench/atomic_window.txt), and k_build_fused then gathers every payload through the row id:
// one 128-byte line per slot, 2.9 ms.  Here the rows become tuples [row id : slot | payload words], are scattered by
// slot range (one or two levels, the same run-staging scatter as the sliced join probe, scan_join_sliced.h), and ONE
// block builds each 32 768-slot slice in LDS: compare-and-swap on LDS words (a taken slot is the reference's -1,
// JoinHashImpl.h:55-80), then the slice of the table -- and of the fused table, payloads read from the tuples, not
// gathered -- is written front to back.  Everything is streamed: 8 (16 with a payload) bytes per row and level.
//
// Contract: fill_hash_join_buff_on_device_bucketized (QE/JoinHashTable/Runtime/HashJoinRuntimeGpu.cu:57-75) for a
// table that is NOT built for a semi join (first-row-wins needs no error and may see any number of duplicates: the
// atomic kernel keeps that case).  Sub-slab capacities come from the row count; a key distribution that overflows one
// raises a flag, the slice pass leaves the table alone, and the atomic kernels -- armed behind it -- do the work.
#pragma once
#include "part_scatter_batch.h"

namespace hdk {

constexpr int kPbBuildBlock = 1024;       // slice pass of the table alone: one block per CU
constexpr int kPbBuildBlockFused = 512;   // ... with payloads: up to three blocks per CU
constexpr int kPbSliceLog2 = 15;          // the table alone: 32 768 slots per slice, 128 KB of row ids in LDS
constexpr int kPbSliceLog2Fused = 12;     // with payloads: 4 096 slots per slice, row id + payload words in LDS (48 / 80 / 112 KB)
constexpr int kPbMaxPayload = 3;

struct PbArgs {
  const hdk_hip_join_chunk* chunks;
  size_t num_chunks;
  hdk_hip_join_column_type_info ti;
  int64_t bucket;
  int64_t hash_entry_count;  // slots of the table
  int32_t invalid_slot_val;
  int32_t np;                // payload columns (0: the table only)
  int32_t* dev_err;
  const int8_t* pcols[kPbMaxPayload];
  int32_t pwidths[kPbMaxPayload];
  int32_t pkinds[kPbMaxPayload];
  uint32_t slice_log2;       // slots per slice = 1 << slice_log2
  uint32_t nslices;          // ceil(slots / slots per slice)
  uint32_t fpc_log2;         // fine slices per level-1 bin = 1 << fpc_log2 (0: one level)
  uint32_t nb1;              // level-1 bins
  uint32_t two_level;
  uint32_t members2;         // level 2: blocks per level-1 bin
  uint32_t pad1_;
  uint64_t cap1;             // tuples of a (bin, XCD) sub-slab
  uint64_t cap2;             // tuples of a slice (level 2): its slot count -- more means a duplicate
  int64_t* tuples1;          // [nb1][kPbXcds][cap1][TW]
  int64_t* tuples2;          // [nslices][cap2][TW]
  uint32_t* fill1;           // [nb1][kPbXcds] x kPbCursorStride
  uint32_t* fill2;           // [nslices] x kPbCursor2Stride
  uint32_t* flag;            // 0: slices; 1: a sub-slab overflowed, the armed atomic kernels build the table
  int32_t* buff;
  int64_t* fused;            // nullptr: the table only
};

// ---- level 1: inner rows -> tuples, scattered by slot range -----------------------------------------------------------
template <int TW, int VR>
__global__ __launch_bounds__(kPbBlock) void k_pb_scatter(PbArgs a) {
  constexpr int kTile = kPbBlock * VR;
  __shared__ uint32_t s_cnt[kPbMaxBins];
  __shared__ uint4 s_run[kPbMaxBins];
  __shared__ uint32_t s_total;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + static_cast<size_t>(kTile) * TW);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kPbXcds - 1);
  for (int i = tid; i < kPbMaxBins; i += kPbBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  const int elem_sz = static_cast<int>(a.ti.elem_sz);
  const uint32_t bin_shift = a.slice_log2 + a.fpc_log2;
  bool stale = false;
  size_t index_base = 0;
  for (size_t c = 0; c < a.num_chunks; ++c) {
    const hdk_hip_join_chunk ch = a.chunks[c];
    const size_t ntiles = (ch.num_elems + kTile - 1) / kTile;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int64_t tup[VR][TW];
      uint32_t bin[VR];
      bool live[VR];
      // all key loads of the batch first (an 8-byte signed key -- the usual dimension key -- without the decoder's switches
      // between the loads)
      int64_t ev[VR];
      if (elem_sz == 8 && a.ti.column_type == HDK_JC_SIGNED) {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const size_t i = tile * kTile + static_cast<size_t>(r) * kPbBlock + tid;
          ev[r] = i < ch.num_elems ? __builtin_nontemporal_load(reinterpret_cast<const int64_t*>(ch.col_buff) + i) : 0;
        }
      } else {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const size_t i = tile * kTile + static_cast<size_t>(r) * kPbBlock + tid;
          ev[r] = i < ch.num_elems ? join_elem(ch.col_buff, i, elem_sz, a.ti.column_type) : 0;
        }
      }
      // payload words: the same, all loads of a column first (8-byte integer / double columns without the decoder)
      int64_t pv[TW > 1 ? TW - 1 : 1][VR];
#pragma unroll
      for (int w = 1; w < TW; ++w) {
        const int8_t* pb = a.pcols[w - 1];
        if (a.pwidths[w - 1] == 8 && a.pkinds[w - 1] != HDK_COL_FLOAT) {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            const size_t i = tile * kTile + static_cast<size_t>(r) * kPbBlock + tid;
            pv[w - 1][r] = i < ch.num_elems ? __builtin_nontemporal_load(reinterpret_cast<const int64_t*>(pb) + index_base + i) : 0;
          }
        } else {
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            const size_t i = tile * kTile + static_cast<size_t>(r) * kPbBlock + tid;
            pv[w - 1][r] = i < ch.num_elems ? decode_col(pb, a.pwidths[w - 1], a.pkinds[w - 1], static_cast<int64_t>(index_base + i)) : 0;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const size_t i = tile * kTile + static_cast<size_t>(r) * kPbBlock + tid;
        live[r] = i < ch.num_elems;
        int64_t elem = ev[r];
        if (live[r] && elem == a.ti.null_val) {
          if (a.ti.uses_bw_eq) {
            elem = a.ti.translated_null_val;
          } else {
            live[r] = false;
          }
        }
        int64_t slot = elem - a.ti.min_val;
        if (a.bucket > 1) {
          slot /= a.bucket;
        }
        if (live[r] && static_cast<uint64_t>(slot) >= static_cast<uint64_t>(a.hash_entry_count)) {
          stale = true;  // key outside [min, max]: the metadata the table was sized from is stale
          live[r] = false;
        }
        const uint32_t rid = static_cast<uint32_t>(index_base + i);
        const uint32_t slot32 = live[r] ? static_cast<uint32_t>(slot) : 0u;
        bin[r] = slot32 >> bin_shift;
        tup[r][0] = static_cast<int64_t>((static_cast<uint64_t>(rid) << 32) | slot32);
#pragma unroll
        for (int w = 1; w < TW; ++w) {
          tup[r][w] = live[r] ? pv[w - 1][r] : 0;
        }
      }
      pb_scatter_batch<TW, VR>(
          tup, bin, live, s_cnt, s_run, &s_total, s_stage, s_binof, a.tuples1,
          [&](uint32_t b, uint32_t n, uint32_t* base, uint32_t* nfit) {
            *base = atomicAdd(a.fill1 + (static_cast<size_t>(b) * kPbXcds + xcd) * kPbCursorStride, n);
            *nfit = static_cast<uint64_t>(*base) >= a.cap1 ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(n), a.cap1 - *base));
            if (*nfit < n) {
              atomicMax(a.flag, 1u);  // this key distribution does not fit the slabs: the atomic kernels take over
            }
          },
          [&](uint32_t b, uint64_t pos) { return (static_cast<uint64_t>(b) * kPbXcds + xcd) * a.cap1 + pos; });
    }
    index_base += ch.num_elems;
  }
  if (__any(stale) && (threadIdx.x & (kWave - 1)) == 0) {
    atomicMin(a.dev_err, -2);
  }
}

// ---- level 2: one block per (level-1 bin, XCD) sub-slab, scattered by fine slice -------------------------------------------
template <int TW, int VR>
__global__ __launch_bounds__(kPbBlock) void k_pb_scatter2(PbArgs a) {
  constexpr int kTile = kPbBlock * VR;
  __shared__ uint32_t s_cnt[kPbMaxBins];
  __shared__ uint4 s_run[kPbMaxBins];
  __shared__ uint32_t s_total, s_stop;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + static_cast<size_t>(kTile) * TW);
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_stop = __hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int i = tid; i < kPbMaxBins; i += kPbBlock) {
    s_cnt[i] = 0;
  }
  __syncthreads();
  if (s_stop) {
    return;
  }
  const uint32_t fmask = (1u << a.fpc_log2) - 1u;
  bool dup = false;
  // Blocks b with b % 8 == c % 8 work on level-1 bin c, `members` of them per bin (block ids are dealt to the XCDs round
  // robin: all writers of a slice's slab then sit behind ONE L2 and the partial lines at the ends of their runs merge there;
  // an affinity for speed, any placement is correct).  The bin's eight sub-slabs are one sequence of tiles, dealt to the
  // members in turn.
  const uint32_t members = a.members2;
  const uint32_t lane8 = blockIdx.x % kPbXcds, idx8 = blockIdx.x / kPbXcds;
  const uint32_t member = idx8 % members;
  for (uint32_t b1 = lane8 + kPbXcds * (idx8 / members); b1 < a.nb1; b1 += kPbXcds * (gridDim.x / (kPbXcds * members))) {
   uint32_t turn = 0;
   for (uint32_t x8 = 0; x8 < kPbXcds; ++x8) {
    const uint32_t sub = b1 * kPbXcds + x8;
    const uint64_t n = min(static_cast<uint64_t>(a.fill1[static_cast<size_t>(sub) * kPbCursorStride]), a.cap1);
    const int64_t* src = a.tuples1 + static_cast<uint64_t>(sub) * a.cap1 * TW;
    for (uint64_t t0 = 0; t0 < n; t0 += kTile, ++turn) {
      if (turn % members != member) {
        continue;
      }
      int64_t tup[VR][TW];
      uint32_t bin[VR];
      bool live[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint64_t i = t0 + static_cast<uint64_t>(r) * kPbBlock + tid;
        live[r] = i < n;
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          tup[r][w] = live[r] ? __builtin_nontemporal_load(src + i * TW + w) : 0;
        }
        bin[r] = (static_cast<uint32_t>(tup[r][0]) >> a.slice_log2) & fmask;
      }
      pb_scatter_batch<TW, VR>(
          tup, bin, live, s_cnt, s_run, &s_total, s_stage, s_binof, a.tuples2,
          [&](uint32_t f, uint32_t cnt, uint32_t* base, uint32_t* nfit) {
            const uint32_t slice = (b1 << a.fpc_log2) + f;
            *base = atomicAdd(a.fill2 + static_cast<size_t>(slice) * kPbCursor2Stride, cnt);
            *nfit = static_cast<uint64_t>(*base) >= a.cap2 ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(cnt), a.cap2 - *base));
            if (*nfit < cnt) {
              dup = true;  // more rows than the slice has slots: some slot is taken twice
            }
          },
          [&](uint32_t f, uint64_t pos) { return static_cast<uint64_t>((b1 << a.fpc_log2) + f) * a.cap2 + pos; });
    }
   }
  }
  if (__any(dup) && (threadIdx.x & (kWave - 1)) == 0) {
    atomicMin(a.dev_err, -1);
  }
}

// ---- slices: one block builds a slice in LDS and writes it out front to back ----------------------------------------------
// dynamic LDS: int64 payload[slots][TW - 1] | int32 row id[slots]
template <int TW>
__global__ __launch_bounds__(TW == 1 ? kPbBuildBlock : kPbBuildBlockFused) void k_pb_build(PbArgs a) {
  constexpr int NP = TW - 1;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  const uint32_t slots = 1u << a.slice_log2;
  int64_t* s_pay = s_dyn;
  int32_t* s_rid = reinterpret_cast<int32_t*>(s_dyn + static_cast<size_t>(slots) * NP);
  __shared__ uint32_t s_stop;
  const uint32_t tid = threadIdx.x;
  const uint32_t nthr = blockDim.x;
  if (tid == 0) {
    s_stop = __hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_stop) {
    return;  // (the atomic kernels behind this one build the table)
  }
  const int32_t inv = a.invalid_slot_val;
  bool dup = false;
  for (uint32_t s = blockIdx.x; s < a.nslices; s += gridDim.x) {
    const uint32_t base = s << a.slice_log2;
    const uint32_t nslots = static_cast<uint32_t>(min<int64_t>(slots, a.hash_entry_count - static_cast<int64_t>(base)));
    for (uint32_t i = tid; i < slots; i += nthr) {
      s_rid[i] = inv;
    }
    __syncthreads();
    const int nsrc = a.two_level ? 1 : kPbXcds;
    for (int x = 0; x < nsrc; ++x) {
      const int64_t* src;
      uint64_t n;
      if (a.two_level) {
        src = a.tuples2 + static_cast<uint64_t>(s) * a.cap2 * TW;
        n = min(static_cast<uint64_t>(a.fill2[static_cast<size_t>(s) * kPbCursor2Stride]), a.cap2);
      } else {
        const uint64_t sub = static_cast<uint64_t>(s) * kPbXcds + x;
        src = a.tuples1 + sub * a.cap1 * TW;
        n = min(static_cast<uint64_t>(a.fill1[sub * kPbCursorStride]), a.cap1);
      }
      for (uint64_t i = tid; i < n; i += nthr) {
        int64_t t[TW];
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          t[w] = __builtin_nontemporal_load(src + i * TW + w);
        }
        const uint32_t local = static_cast<uint32_t>(t[0]) - base;
        const int32_t rid = static_cast<int32_t>(static_cast<uint64_t>(t[0]) >> 32);
        if (local < nslots) {
          // fill_one_to_one_hashtable (JoinHashImpl.h:55-66): the slot is free, or the key is not unique
          const int32_t old = atomicCAS(&s_rid[local], inv, rid);
          if (old == inv) {
#pragma unroll
            for (int w = 0; w < NP; ++w) {
              s_pay[static_cast<size_t>(local) * NP + w] = t[1 + w];  // (the row that holds the slot)
            }
          } else {
            dup = true;
          }
        }
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nslots; i += nthr) {
      const int32_t rid = s_rid[i];
      a.buff[base + i] = rid;
      if (a.fused) {  // HDK_JOIN_ONE_TO_ONE_FUSED: [row id | payload words], empty slots carry zeros
        int64_t* o = a.fused + (static_cast<int64_t>(base) + i) * TW;
        o[0] = rid;
#pragma unroll
        for (int w = 0; w < NP; ++w) {
          o[1 + w] = rid == inv ? 0 : s_pay[static_cast<size_t>(i) * NP + w];
        }
      }
    }
    __syncthreads();
  }
  if (__any(dup) && (threadIdx.x & (kWave - 1)) == 0) {
    atomicMin(a.dev_err, -1);
  }
}

}  // namespace hdk
