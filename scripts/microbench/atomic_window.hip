// Micro-benchmark for an L2-resident aggregation phase (C5 design question): tuples (key, val) streamed from HBM, each
// one updating a 16-byte entry [key | sum] of a hash-table WINDOW of W bytes that only the blocks of one XCD touch
// (block b -> window b % 8, the round-robin block-to-XCD placement, checked against XCC_ID).  How many updates per second
// do device-scope atomics sustain when the window sits in that XCD's 4 MB L2 -- or in the 256 MB Infinity Cache behind
// it -- instead of HBM?   ./atomic_window [tuples]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define GPTR(T, p) reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p))
typedef long long __attribute__((ext_vector_type(2))) i64x2;

__device__ inline uint32_t mix32(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
  return (uint32_t)k;
}

// MODE 0: atomicAdd on the sum word only (the group exists);  1: plain key load + compare, CAS when empty, then atomicAdd
// (first-touch claims included);  2: one 128-bit... (not available) -> skipped
template <int MODE>
__global__ __launch_bounds__(256) void k_update(const i64x2* tuples, int64_t n, int64_t* windows, uint32_t entries_per_window,
                                                int nwindows, unsigned* xcd_mismatch) {
  const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, bits 3:0
  if (threadIdx.x == 0 && (xcc & 7) != (int)(blockIdx.x & 7)) atomicAdd(xcd_mismatch, 1u);
  int64_t* win = windows + (size_t)(blockIdx.x % nwindows) * entries_per_window * 2;
  const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per_block;
  const int64_t hi = lo + per_block < n ? lo + per_block : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256 * 4) {
    i64x2 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + u * 256;
      t[u] = j < hi ? __builtin_nontemporal_load(GPTR(i64x2, tuples) + j) : i64x2{-1, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (t[u].x < 0) continue;
      const uint32_t e = (uint32_t)(((uint64_t)mix32((uint64_t)t[u].x) * entries_per_window) >> 32);
      int64_t* row = win + (size_t)e * 2;
      if (MODE == 1) {
        const int64_t seen = __hip_atomic_load(row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen != t[u].x) {
          atomicCAS((unsigned long long*)row, 0x7fffffffffffffffull, (unsigned long long)t[u].x);  // collisions ignored here
        }
      }
      atomicAdd((unsigned long long*)(row + 1), (unsigned long long)t[u].y);
    }
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 128ll << 20;
  i64x2* tuples;
  CK(hipMalloc(&tuples, n * 16));
  {
    i64x2* h = (i64x2*)malloc(n * 16);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      h[i] = i64x2{(long long)(s >> 2), (long long)(i & 1023)};
    }
    CK(hipMemcpy(tuples, h, n * 16, hipMemcpyHostToDevice));
    free(h);
  }
  const size_t max_total = 2048ull << 20;
  int64_t* windows;
  CK(hipMalloc(&windows, max_total));
  unsigned* mism;
  CK(hipMalloc(&mism, 4));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * 8;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("# %lld tuples (16 B, streamed nt), grid %d x 256; windows of W bytes, one per XCD (nwindows = 8) or one per block group\n", (long long)n, grid);
  for (int mode = 0; mode < 2; ++mode) {
    for (int nwin : {8, 64}) {
      for (size_t w_mb : {1, 2, 3, 4, 8, 16, 32, 64, 128, 256}) {
        const size_t w_bytes = w_mb << 20;
        if (w_bytes * nwin > max_total) continue;
        const uint32_t epw = (uint32_t)(w_bytes / 16);
        CK(hipMemset(windows, 0x7f, w_bytes * nwin));
        CK(hipMemset(mism, 0, 4));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(a));
          if (mode == 0) k_update<0><<<grid, 256>>>(tuples, n, windows, epw, nwin, mism);
          else k_update<1><<<grid, 256>>>(tuples, n, windows, epw, nwin, mism);
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          best = ms < best ? ms : best;
        }
        unsigned mm; CK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
        printf("mode %d  windows %2d x %3zu MB  %.3f ms  %.2f G updates/s  (%.0f GB/s of tuples)  xcd mismatches %u/%d\n", mode,
               nwin, w_mb, best, n / best / 1e6, n * 16 / best / 1e6, mm / 3, grid);
      }
    }
  }
  return 0;
}
