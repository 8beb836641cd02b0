// Micro-benchmark (round 5, VERDICT r4 task 3): can the 256 MiB Infinity Cache carry the intermediate tuples of the
// multi-pass paths (sliced join C3: 16 of its 34 GB; radix-partitioned group-by C5: 32 of its 53 GB)?
//
// Kernel A reads two 8-byte columns (16 B/row) and writes one 8-byte tuple per row -- either streamed (coalesced, the
// best case) or scattered into 256 bins the way hdk_join_scatter_slices / hdk_part_scatter do (runs behind per-(bin, XCD)
// cursors).  Kernel B reads the tuples back.  ONE pass over all rows (tuples far larger than the cache) is compared with
// the same rows cut into chunks, A(chunk k) then B(chunk k) back to back on one stream, the tuple buffer reused by every
// chunk: between the write of a tuple line and its read lie at most the chunk's input + 2 x its tuples
// (MI355X_MICROARCH.md, Infinity Cache residency rule: table + every byte moved between two uses <= ~256 MiB).
//
// Usage: mall_pingpong [rows = 512 Mi] [shape 0 stream | 1 scatter] [policy 0 nt/pl | 1 nt/nt | 2 pl/pl] [chunk MiB, 0 = one pass]; under rocprofv3 --kernel-trace --pmc FETCH_SIZE (and WRITE_SIZE) the per-kernel
// counters tell whether the chunked B is served without HBM fetches (FETCH_SIZE counts Infinity-Cache hits too according
// to the guide, so TIME is the verdict; the counters are reported for completeness).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef long long __attribute__((ext_vector_type(2))) i64x2;
constexpr int kBlock = 512;
constexpr int VR = 8;
constexpr int kTile = kBlock * VR;
constexpr int kMaxBins = 256;
constexpr int kXcds = 8;
constexpr uint32_t kCursorStride = 32;
constexpr int kWave = 64;

struct Args {
  const int64_t* key;
  const int64_t* x;
  int64_t row0, row1;  // the chunk (multiples of kTile)
  int64_t* tuples;     // streamed: [rows of the chunk]; scattered: [nbins][kXcds][sub]
  uint32_t* fill;      // [nbins][kXcds] x kCursorStride (this chunk's)
  uint64_t sub;
  uint32_t nbins;
  unsigned long long* sink;
};

__global__ void k_gen(int64_t* key, int64_t* x, int64_t n, uint32_t range) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    uint64_t s = static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ull + 88172645463325252ull;
    s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    key[i] = static_cast<int64_t>(s % range);
    s *= 0x94D049BB133111EBull; s ^= s >> 31;
    x[i] = static_cast<int64_t>(s % 1000000);
  }
}

// A, streamed: tuple i of the chunk = [x : key] of row row0 + i
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_a_stream(Args a) {
  const int tid = threadIdx.x;
  const int64_t ntiles = (a.row1 - a.row0) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    i64x2 kk[VR / 2], xx[VR / 2];
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) {
      const int64_t p = ((a.row0 + tile * kTile) >> 1) + static_cast<int64_t>(u) * kBlock + tid;
      kk[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.key) + p) : reinterpret_cast<const i64x2*>(a.key)[p];
      xx[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.x) + p) : reinterpret_cast<const i64x2*>(a.x)[p];
    }
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) {
      i64x2 t;
      t.x = static_cast<long long>((static_cast<uint64_t>(static_cast<uint32_t>(xx[u].x)) << 32) | static_cast<uint32_t>(kk[u].x));
      t.y = static_cast<long long>((static_cast<uint64_t>(static_cast<uint32_t>(xx[u].y)) << 32) | static_cast<uint32_t>(kk[u].y));
      reinterpret_cast<i64x2*>(a.tuples)[((tile * kTile) >> 1) + static_cast<int64_t>(u) * kBlock + tid] = t;
    }
  }
}

// B, streamed: read the chunk's tuples, 16 B per lane
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_b_stream(Args a) {
  const int tid = threadIdx.x;
  const int64_t ntiles = (a.row1 - a.row0) / kTile;
  unsigned long long acc = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    i64x2 t[VR / 2];
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) {
      const int64_t p = ((tile * kTile) >> 1) + static_cast<int64_t>(u) * kBlock + tid;
      t[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.tuples) + p) : reinterpret_cast<const i64x2*>(a.tuples)[p];
    }
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) acc += static_cast<unsigned long long>(t[u].x) + static_cast<unsigned long long>(t[u].y);
  }
  if (acc == 0x1234567ull) atomicAdd(a.sink, acc);  // (keeps the loads alive)
}

// A, scattered: the production shape (scripts/microbench/scatter_runs.hip mode 0; scan_join_sliced.h)
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_a_scatter(Args a) {
  __shared__ uint32_t s_cnt[kMaxBins];
  __shared__ uint4 s_run[kMaxBins];  // .x start in the stage, .y length, .z slab position
  __shared__ uint32_t s_total;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + kTile);
  const int tid = threadIdx.x;
  const uint32_t xcd = static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kXcds - 1);
  for (int i = tid; i < kMaxBins; i += kBlock) s_cnt[i] = 0;
  __syncthreads();
  const int64_t ntiles = (a.row1 - a.row0) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t k[VR], x[VR];
#pragma unroll
    for (int u = 0; u < VR / 2; ++u) {
      const int64_t p = ((a.row0 + tile * kTile) >> 1) + static_cast<int64_t>(u) * kBlock + tid;
      const i64x2 kk = NT ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.key) + p) : reinterpret_cast<const i64x2*>(a.key)[p];
      const i64x2 xx = NT ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(a.x) + p) : reinterpret_cast<const i64x2*>(a.x)[p];
      k[2 * u] = kk.x; k[2 * u + 1] = kk.y; x[2 * u] = xx.x; x[2 * u + 1] = xx.y;
    }
    uint32_t bin[VR], rank[VR];
    int64_t tup[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const uint32_t d32 = static_cast<uint32_t>(k[r]);
      bin[r] = d32 & (a.nbins - 1);  // (keys are uniform: any 8 bits do)
      tup[r] = static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(x[r])) << 32) | d32);
      rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
    }
    __syncthreads();
    if (tid < kMaxBins) {
      const uint32_t n = s_cnt[tid];
      uint32_t base = 0;
      if (n) base = atomicAdd(a.fill + (static_cast<size_t>(tid) * kXcds + xcd) * kCursorStride, n);
      s_run[tid].y = n;
      s_run[tid].z = base;
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
    if (tid < kMaxBins) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t total = s_total;
    for (uint32_t i = tid; i < total; i += kBlock) {
      const uint32_t b = s_binof[i];
      const uint4 run = s_run[b];
      const uint64_t pos = static_cast<uint64_t>(run.z) + (i - run.x);
      if (pos < a.sub) a.tuples[(static_cast<uint64_t>(b) * kXcds + xcd) * a.sub + pos] = s_stage[i];
    }
    __syncthreads();
  }
}

// B, scattered: every (bin, XCD) sub-slab read by whole blocks, 8 B per lane (what the aggregating pass does)
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_b_scatter(Args a) {
  const int tid = threadIdx.x;
  unsigned long long acc = 0;
  const uint32_t nsub = a.nbins * kXcds;
  // blocks of one bin on the XCD that wrote it (launch-order mapping block b -> XCD b % 8)
  for (uint32_t s = blockIdx.x; s < nsub; s += gridDim.x) {
    const uint32_t xcd = s % kXcds, bin = s / kXcds;
    const uint32_t n = min(a.fill[(static_cast<size_t>(bin) * kXcds + xcd) * kCursorStride], static_cast<uint32_t>(a.sub));
    const int64_t* t = a.tuples + (static_cast<uint64_t>(bin) * kXcds + xcd) * a.sub;
    uint32_t i = tid * 2;
    for (; i + 1 < n; i += kBlock * 2) {
      const i64x2 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(t + i)) : *reinterpret_cast<const i64x2*>(t + i);
      acc += static_cast<unsigned long long>(v.x) + static_cast<unsigned long long>(v.y);
    }
    if (i < n) acc += static_cast<unsigned long long>(t[i]);
  }
  if (acc == 0x1234567ull) atomicAdd(a.sink, acc);
}

struct Result { float ms; };

int main(int argc, char** argv) {
  const int64_t n = (argc > 1 ? atoll(argv[1]) : (512ll << 20)) / kTile * kTile;
  const uint32_t nbins = 256;
  int64_t *key, *x, *tuples; uint32_t* fill; unsigned long long* sink;
  CK(hipMalloc(&key, n * 8)); CK(hipMalloc(&x, n * 8));
  CK(hipMalloc(&sink, 8)); CK(hipMemset(sink, 0, 8));
  hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, key, x, n, 10000000u);
  CK(hipDeviceSynchronize());
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const unsigned cu = prop.multiProcessorCount;
  const size_t cursor_bytes = static_cast<size_t>(nbins) * kXcds * kCursorStride * 4;
  const int kMaxChunks = 1024;
  CK(hipMalloc(&fill, cursor_bytes * kMaxChunks));
  // tuple buffer for the monolithic scattered pass: [nbins][8][sub(n)]
  auto sub_for = [&](int64_t rows) { return ((static_cast<uint64_t>(rows) / nbins / kXcds) * 5 / 4 + 4096) / 16 * 16; };
  const size_t tuples_bytes = static_cast<size_t>(nbins) * kXcds * sub_for(n) * 8;
  CK(hipMalloc(&tuples, tuples_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# rows %lld = %.1f GiB of input (two 8-byte columns), tuples %.1f GiB; %u CUs\n", static_cast<long long>(n), n * 16.0 / (1 << 30),
         n * 8.0 / (1 << 30), cu);
  printf("# %-9s %-6s %10s %7s %9s %9s %12s %s\n", "shape", "loads", "chunk_MiB", "chunks", "total_ms", "ms/GiB_in", "moved_GB/s", "(moved = input + tuples written + tuples read)");

  auto run = [&](bool scatter, bool nt, int64_t chunk_rows, int reps, bool ntb) {
    const int64_t nchunks = (n + chunk_rows - 1) / chunk_rows;
    if (nchunks > kMaxChunks) return;
    float best = 1e30f;
    for (int rep = 0; rep < reps + 1; ++rep) {
      if (scatter) CK(hipMemsetAsync(fill, 0, cursor_bytes * nchunks, 0));
      CK(hipEventRecord(e0, 0));
      for (int64_t c = 0; c < nchunks; ++c) {
        Args a{key, x, c * chunk_rows, std::min(n, (c + 1) * chunk_rows), tuples, fill + c * (cursor_bytes / 4), sub_for(chunk_rows), nbins, sink};
        const unsigned ga = 2 * cu, gb = 4 * cu;
        if (scatter) {
          const size_t lds = static_cast<size_t>(kTile) * 9;
          if (nt) hipLaunchKernelGGL(k_a_scatter<true>, dim3(ga), dim3(kBlock), lds, 0, a); else hipLaunchKernelGGL(k_a_scatter<false>, dim3(ga), dim3(kBlock), lds, 0, a);
          if (ntb) hipLaunchKernelGGL(k_b_scatter<true>, dim3(gb), dim3(kBlock), 0, 0, a); else hipLaunchKernelGGL(k_b_scatter<false>, dim3(gb), dim3(kBlock), 0, 0, a);
        } else {
          if (nt) hipLaunchKernelGGL(k_a_stream<true>, dim3(ga), dim3(kBlock), 0, 0, a); else hipLaunchKernelGGL(k_a_stream<false>, dim3(ga), dim3(kBlock), 0, 0, a);
          if (ntb) hipLaunchKernelGGL(k_b_stream<true>, dim3(gb), dim3(kBlock), 0, 0, a); else hipLaunchKernelGGL(k_b_stream<false>, dim3(gb), dim3(kBlock), 0, 0, a);
        }
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = std::min(best, ms);
    }
    const double in_gib = n * 16.0 / (1 << 30);
    printf("  %-9s %-6s %10.0f %7lld %9.3f %9.4f %12.1f\n", scatter ? "scatter" : "stream", nt ? (ntb ? "nt/nt" : "nt/pl") : (ntb ? "pl/nt" : "pl/pl"), chunk_rows * 16.0 / (1 << 20),
           static_cast<long long>(nchunks), best, best / in_gib, n * 32.0 / best / 1e6);
    fflush(stdout);
  };
  if (argc > 4) {  // one configuration only (for rocprofv3 --kernel-trace --stats: per-kernel times of exactly this variant)
    const int scatter = atoi(argv[2]), pol = atoi(argv[3]);
    const int64_t mib = atoll(argv[4]);
    const int64_t rows = mib ? (mib << 20) / 16 / kTile * kTile : n;
    run(scatter, pol != 2, rows < n ? rows : n, 3, pol == 1);
    return 0;
  }
  // A or B alone over everything (the HBM-rate baseline of each kernel): chunk = n with the other kernel's time subtracted is
  // not needed -- the monolithic row IS A + B at HBM rates.
  for (int scatter = 0; scatter < 2; ++scatter) {
    // loads of A's input / of B's tuples: non-temporal (nt) or default policy (pl)
    for (int pol = 0; pol < 3; ++pol) {
      const bool nt = pol != 2, ntb = pol == 1;
      run(scatter, nt, n, 3, ntb);  // one pass: tuples far beyond the cache
      for (int64_t mib : {1024, 512, 256, 192, 128, 96, 64, 48, 32, 16}) {
        const int64_t rows = (mib << 20) / 16 / kTile * kTile;
        if (rows < n) run(scatter, nt, rows, 3, ntb);
      }
    }
  }
  // mixed policy: input non-temporal (read once), tuples with the default policy (they are to be re-read) -- the
  // combination the production kernels would use
  printf("# done\n");
  return 0;
}
