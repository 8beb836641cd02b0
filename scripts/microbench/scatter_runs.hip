// Micro-benchmark: level 1 of the sliced join's scatter (scan_join_sliced.h: hdk_join_scatter_slices<NARROW>) with the
// write side done four ways -- what do the partial lines at the ends of a block's runs cost in HBM writes, and is there
// a cheaper way to write them?  (10.6 GB written for 8.0 GB of tuples per 1 B rows.)
//
//   mode 0  production: a block's tuples of a bin go to consecutive positions behind the bin's cursor (8-byte granular)
//   mode 1  runs start on an ALIGN-tuple boundary and are padded to one with sentinel tuples (ALIGN = 4 / 8 / 16:
//           32 / 64 / 128 bytes)
//   mode 2  software write-combining: a block carries the tuples of a bin that do not fill a CHUNK (8 or 16 tuples) to
//           its next batch in LDS and writes whole aligned chunks only
// Usage: scatter_runs [rows] [bins]; rocprofv3 --pmc WRITE_SIZE over it gives the bytes (kernel names tell the modes).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef long long __attribute__((ext_vector_type(2))) i64x2;
constexpr int kBlock = 512;
constexpr int VR = 8;
constexpr int kTile = kBlock * VR;
constexpr int kMaxBins = 256;
constexpr int kXcds = 8;
constexpr uint32_t kCursorStride = 32;
constexpr int kWave = 64;
constexpr int64_t kSentinel = -1;

struct Args {
  const int64_t* key;
  const int64_t* x;
  int64_t n;
  int64_t* tuples;   // [nbins][kXcds][sub]
  uint32_t* fill;    // [nbins][kXcds] x kCursorStride
  uint64_t sub;
  uint32_t nbins;
  uint32_t magic, shift;  // bin = key / slice_keys by multiplication
};

__device__ inline uint32_t bin_of(const Args& a, uint32_t d32) {
  const uint32_t t = __umulhi(a.magic, d32);
  return (((d32 - t) >> 1) + t) >> a.shift;
}

// MODE 0 / 1 (ALIGN > 1: padded runs).  LDS: stage [kTile + kMaxBins * (ALIGN - 1)] tuples | bin of every slot
// SOA6 (round 6): the tuple leaves as a 4-byte payload and a 2-byte key-inside-the-slice in two arrays (6 bytes a row, power-of-two
// elements) instead of one 8-byte word -- what VERDICT r5 task 8 asks to price for C3 / C5
template <int ALIGN, int VR = 8, int kBlock = 512, bool SOA6 = false>
__global__ __launch_bounds__(kBlock) void k_scatter_runs(Args a) {
  constexpr int kTile = kBlock * VR;
  constexpr int kSlots = kTile + kMaxBins * (ALIGN - 1);
  __shared__ uint32_t s_cnt[kMaxBins];
  __shared__ uint4 s_run[kMaxBins];  // .x start in the stage, .y padded length, .z slab position
  __shared__ uint32_t s_total;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + kSlots);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kXcds - 1);
  for (int i = tid; i < kMaxBins; i += kBlock) s_cnt[i] = 0;
  __syncthreads();
  const int64_t ntiles = a.n / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t k[VR], x[VR];
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) {
      const int64_t p = (tile * kTile >> 1) + static_cast<int64_t>(u) * kBlock + tid;
      const i64x2 kk = __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.key) + p);
      const i64x2 xx = __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.x) + p);
      k[2 * u] = kk.x; k[2 * u + 1] = kk.y; x[2 * u] = xx.x; x[2 * u + 1] = xx.y;
    }
    uint32_t bin[VR], rank[VR];
    int64_t tup[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const uint32_t d32 = static_cast<uint32_t>(k[r]);
      bin[r] = bin_of(a, d32);
      tup[r] = static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(x[r])) << 32) | d32);
      rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
    }
    __syncthreads();
    if (tid < kMaxBins) {
      const uint32_t n = s_cnt[tid];
      const uint32_t npad = (n + ALIGN - 1) / ALIGN * ALIGN;
      uint32_t base = 0;
      if (n) base = atomicAdd(a.fill + (static_cast<size_t>(tid) * kXcds + xcd) * kCursorStride, npad);
      s_run[tid].y = npad;
      s_run[tid].z = base;
      s_run[tid].w = n;
    }
    __syncthreads();
    if (tid < kWave) {
      uint32_t carry = 0;
      for (int c0 = 0; c0 < kMaxBins; c0 += kWave) {
        const uint32_t n = s_run[c0 + tid].y;
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint32_t v = __shfl_up(incl, d, kWave);
          if (tid >= d) incl += v;
        }
        s_run[c0 + tid].x = carry + incl - n;
        carry += __shfl(incl, kWave - 1, kWave);
      }
      if (tid == 0) s_total = carry;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const uint32_t si = s_run[bin[r]].x + rank[r];
      s_binof[si] = static_cast<uint8_t>(bin[r]);
      s_stage[si] = tup[r];
    }
    if (ALIGN > 1 && tid < kMaxBins) {
      const uint4 run = s_run[tid];
      for (uint32_t i = run.w; i < run.y; ++i) {
        s_binof[run.x + i] = static_cast<uint8_t>(tid);
        s_stage[run.x + i] = kSentinel;
      }
    }
    if (tid < kMaxBins) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t total = s_total;
    for (uint32_t i = tid; i < total; i += kBlock) {
      const uint32_t b = s_binof[i];
      const uint4 run = s_run[b];
      const uint64_t pos = static_cast<uint64_t>(run.z) + (i - run.x);
      if (SOA6) {
        if (pos < a.sub) {
          const uint64_t at = (static_cast<uint64_t>(b) * kXcds + xcd) * a.sub + pos;
          const uint64_t t = static_cast<uint64_t>(s_stage[i]);
          reinterpret_cast<uint32_t*>(a.tuples)[at] = static_cast<uint32_t>(t >> 32);                                   // payload
          reinterpret_cast<uint16_t*>(a.tuples + static_cast<uint64_t>(a.nbins) * kXcds * a.sub / 2)[at] =
              static_cast<uint16_t>(static_cast<uint32_t>(t) - b * 39936u);                                             // key inside the slice
        }
      } else if (pos < a.sub) {
        a.tuples[(static_cast<uint64_t>(b) * kXcds + xcd) * a.sub + pos] = s_stage[i];
      }
    }
    __syncthreads();
  }
}

// reading side of pass 2: the sub-slabs streamed back, 8-byte tuples against 4 + 2 bytes
template <bool SOA6>
__global__ __launch_bounds__(256) void k_read_back(Args a, unsigned long long* sink) {
  unsigned long long acc = 0;
  const uint64_t total = static_cast<uint64_t>(a.nbins) * kXcds * a.sub;
  for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x) * 2; i + 1 < total; i += static_cast<uint64_t>(gridDim.x) * 256 * 2) {
    if (SOA6) {
      const unsigned long long p = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(reinterpret_cast<const uint32_t*>(a.tuples) + i));
      const uint32_t k = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.tuples + total / 2) + i));
      acc += p + k;
    } else {
      const i64x2 t = __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.tuples + i));
      acc += static_cast<unsigned long long>(t.x) + static_cast<unsigned long long>(t.y);
    }
  }
  if (acc == 0x1234567) *sink = acc;
}

// MODE 2: whole CHUNK-tuple chunks only; the rest of a bin is carried in the stage's head to the next batch.
// LDS: stage [kTile + kMaxBins * (CHUNK - 1)] | bin of every slot | carry [kMaxBins][CHUNK - 1... CHUNK]
template <int CHUNK>
__global__ __launch_bounds__(kBlock) void k_scatter_wc(Args a) {
  constexpr int kSlots = kTile + kMaxBins * CHUNK;
  __shared__ uint32_t s_cnt[kMaxBins];   // new tuples of the batch
  __shared__ uint32_t s_res[kMaxBins];   // carried tuples
  __shared__ uint4 s_run[kMaxBins];      // .x start in the stage, .y tuples written out (whole chunks), .z slab position, .w all tuples (carried + new)
  __shared__ uint32_t s_total;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  int64_t* s_carry = s_dyn + kSlots;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_carry + kMaxBins * CHUNK);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kXcds - 1);
  for (int i = tid; i < kMaxBins; i += kBlock) { s_cnt[i] = 0; s_res[i] = 0; }
  __syncthreads();
  const int64_t ntiles = a.n / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t k[VR], x[VR];
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) {
      const int64_t p = (tile * kTile >> 1) + static_cast<int64_t>(u) * kBlock + tid;
      const i64x2 kk = __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.key) + p);
      const i64x2 xx = __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.x) + p);
      k[2 * u] = kk.x; k[2 * u + 1] = kk.y; x[2 * u] = xx.x; x[2 * u + 1] = xx.y;
    }
    uint32_t bin[VR], rank[VR];
    int64_t tup[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const uint32_t d32 = static_cast<uint32_t>(k[r]);
      bin[r] = bin_of(a, d32);
      tup[r] = static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(x[r])) << 32) | d32);
      rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
    }
    __syncthreads();
    if (tid < kMaxBins) {
      const uint32_t all = s_res[tid] + s_cnt[tid];
      const uint32_t out = all / CHUNK * CHUNK;
      uint32_t base = 0;
      if (out) base = atomicAdd(a.fill + (static_cast<size_t>(tid) * kXcds + xcd) * kCursorStride, out);
      s_run[tid].y = out;
      s_run[tid].z = base;
      s_run[tid].w = all;
    }
    __syncthreads();
    if (tid < kWave) {
      uint32_t carry = 0;
      for (int c0 = 0; c0 < kMaxBins; c0 += kWave) {
        const uint32_t n = s_run[c0 + tid].w;
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint32_t v = __shfl_up(incl, d, kWave);
          if (tid >= d) incl += v;
        }
        s_run[c0 + tid].x = carry + incl - n;
        carry += __shfl(incl, kWave - 1, kWave);
      }
      if (tid == 0) s_total = carry;
    }
    __syncthreads();
    // stage: [carried | new] per bin
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const uint32_t si = s_run[bin[r]].x + s_res[bin[r]] + rank[r];
      s_binof[si] = static_cast<uint8_t>(bin[r]);
      s_stage[si] = tup[r];
    }
    for (int i = tid; i < kMaxBins * CHUNK; i += kBlock) {
      const uint32_t b = i / CHUNK, j = i % CHUNK;
      if (j < s_res[b]) {
        const uint32_t si = s_run[b].x + j;
        s_binof[si] = static_cast<uint8_t>(b);
        s_stage[si] = s_carry[i];
      }
    }
    __syncthreads();
    const uint32_t total = s_total;
    for (uint32_t i = tid; i < total; i += kBlock) {
      const uint32_t b = s_binof[i];
      const uint4 run = s_run[b];
      const uint32_t r = i - run.x;
      if (r < run.y) {
        const uint64_t pos = static_cast<uint64_t>(run.z) + r;
        if (pos < a.sub) a.tuples[(static_cast<uint64_t>(b) * kXcds + xcd) * a.sub + pos] = s_stage[i];
      } else {
        s_carry[b * CHUNK + (r - run.y)] = s_stage[i];
      }
    }
    if (tid < kMaxBins) {
      s_res[tid] = s_run[tid].w - s_run[tid].y;
      s_cnt[tid] = 0;
    }
    __syncthreads();
  }
  // the block's last partial chunks, padded with sentinels
  if (tid < kMaxBins && s_res[tid]) {
    const uint32_t base = atomicAdd(a.fill + (static_cast<size_t>(tid) * kXcds + xcd) * kCursorStride, static_cast<uint32_t>(CHUNK));
    for (uint32_t j = 0; j < CHUNK; ++j) {
      const uint64_t pos = static_cast<uint64_t>(base) + j;
      if (pos < a.sub) a.tuples[(static_cast<uint64_t>(tid) * kXcds + xcd) * a.sub + pos] = j < s_res[tid] ? s_carry[tid * CHUNK + j] : kSentinel;
    }
  }
}

template <typename K>
static void run(const char* name, K kernel, size_t lds, Args a, int blocks_per_cu_cap, unsigned grid_override = 0, int block = kBlock) {
  int per_cu = 0;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds));
  if (per_cu > blocks_per_cu_cap) per_cu = blocks_per_cu_cap;
  const unsigned grid = grid_override ? grid_override : 256u * per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    CK(hipMemset(a.fill, 0, static_cast<size_t>(a.nbins) * kXcds * kCursorStride * 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  std::vector<uint32_t> f(static_cast<size_t>(a.nbins) * kXcds * kCursorStride);
  CK(hipMemcpy(f.data(), a.fill, f.size() * 4, hipMemcpyDeviceToHost));
  uint64_t written = 0, over = 0;
  for (size_t i = 0; i < f.size(); i += kCursorStride) { written += f[i]; if (f[i] > a.sub) over++; }
  printf("%-22s %d blocks/CU grid %u  %.3f ms  %.1f GB/s (16 B read + 8 B written per row)  slots claimed %.4f x rows%s\n", name, per_cu, grid, best,
         a.n * 24.0 / best * 1e-6, static_cast<double>(written) / a.n, over ? "  SUB-SLAB OVERFLOW" : "");
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 256000000;
  const uint32_t nbins = argc > 2 ? atoi(argv[2]) : 251;
  const uint32_t slice_keys = 39936;
  const uint32_t range = nbins * slice_keys;
  int64_t *key, *x, *tuples; uint32_t* fill;
  CK(hipMalloc(&key, n * 8)); CK(hipMalloc(&x, n * 8));
  const uint64_t sub = ((static_cast<uint64_t>(n) / nbins / kXcds) * 5 / 4 + 4096) / 16 * 16;
  CK(hipMalloc(&tuples, static_cast<size_t>(nbins) * kXcds * sub * 8));
  CK(hipMalloc(&fill, static_cast<size_t>(nbins) * kXcds * kCursorStride * 4));
  {
    std::vector<int64_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = static_cast<int64_t>(s % range); }
    CK(hipMemcpy(key, h.data(), n * 8, hipMemcpyHostToDevice));
    for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = static_cast<int64_t>(s % 1000000); }
    CK(hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice));
  }
  Args a{key, x, n, tuples, fill, sub, nbins, 0, 0};
  {  // division by slice_keys through multiplication (round-up method)
    uint32_t sh = 0; while ((1ull << sh) < slice_keys) ++sh;
    const uint64_t m = ((1ull << (32 + sh)) + slice_keys - 1) / slice_keys;  // 33 bits
    a.magic = static_cast<uint32_t>(m - (1ull << 32));
    a.shift = sh - 1;
  }
  printf("rows %lld  bins %u  sub-slab %llu tuples\n", static_cast<long long>(n), nbins, static_cast<unsigned long long>(sub));
  auto lds01 = [](int align) { return static_cast<size_t>(kTile + kMaxBins * (align - 1)) * 9; };
  auto lds2 = [](int chunk) { return static_cast<size_t>(kTile + kMaxBins * chunk) * 9 + static_cast<size_t>(kMaxBins) * chunk * 8; };
  run("runs (production)", k_scatter_runs<1>, lds01(1), a, 8);
  run("runs padded to 32 B", k_scatter_runs<4>, lds01(4), a, 8);
  run("runs padded to 64 B", k_scatter_runs<8>, lds01(8), a, 8);
  run("runs padded to 128 B", k_scatter_runs<16>, lds01(16), a, 8);
  run("whole 64 B chunks", k_scatter_wc<8>, lds2(8), a, 8);
  run("whole 128 B chunks", k_scatter_wc<16>, lds2(16), a, 8);
  run("runs, 2 blocks/CU", k_scatter_runs<1>, lds01(1), a, 2);
  run("runs, 1 block/CU", k_scatter_runs<1>, lds01(1), a, 1);
  run("runs x16, 2 blocks/CU", k_scatter_runs<1, 16>, static_cast<size_t>(kBlock) * 16 * 9, a, 2);
  run("runs x16, 1 block/CU", k_scatter_runs<1, 16>, static_cast<size_t>(kBlock) * 16 * 9, a, 1);
  run("runs x12, 2 blocks/CU", k_scatter_runs<1, 12>, static_cast<size_t>(kBlock) * 12 * 9, a, 2);
  // round 6: 6 bytes a row as 4 + 2 (two arrays) against the 8-byte tuple, writing and reading back
  run("runs, 2/CU (8 B)", k_scatter_runs<1>, lds01(1), a, 2);
  run("runs, 2/CU (4 + 2 B)", k_scatter_runs<1, 8, 512, true>, lds01(1), a, 2);
  run("runs x16, 2/CU (4 + 2 B)", k_scatter_runs<1, 16, 512, true>, static_cast<size_t>(kBlock) * 16 * 9, a, 2);
  {
    unsigned long long* sink;
    CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int soa = 0; soa < 2; ++soa) {
      float best = 1e9f;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        if (soa) hipLaunchKernelGGL(k_read_back<true>, dim3(2048), dim3(256), 0, 0, a, sink);
        else hipLaunchKernelGGL(k_read_back<false>, dim3(2048), dim3(256), 0, 0, a, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
      }
      printf("read back %s  %.3f ms (sub-slabs with their slack: %.2f x the rows)\n", soa ? "4 + 2 B" : "8 B    ", best,
             static_cast<double>(a.nbins) * kXcds * a.sub / a.n);
    }
  }
  if (argc > 3) return 0;  // (a third argument: only the comparisons above)
  for (unsigned g : {320u, 384u, 448u, 512u, 576u, 640u, 704u}) {
    char nm[64]; snprintf(nm, sizeof nm, "runs, grid %u", g);
    run(nm, k_scatter_runs<1>, lds01(1), a, 3, g);
  }
  run("1024 thr x8, 1/CU", k_scatter_runs<1, 8, 1024>, static_cast<size_t>(1024) * 8 * 9, a, 1, 0, 1024);
  run("1024 thr x4, 1/CU", k_scatter_runs<1, 4, 1024>, static_cast<size_t>(1024) * 4 * 9, a, 1, 0, 1024);
  run("1024 thr x4, 2/CU", k_scatter_runs<1, 4, 1024>, static_cast<size_t>(1024) * 4 * 9, a, 2, 0, 1024);
  run("256 thr x8, 4/CU", k_scatter_runs<1, 8, 256>, static_cast<size_t>(256) * 8 * 9, a, 4, 0, 256);
  run("256 thr x16, 4/CU", k_scatter_runs<1, 16, 256>, static_cast<size_t>(256) * 16 * 9, a, 4, 0, 256);
  return 0;
}
