// Micro-benchmark: one radix-scatter pass of the partitioned group-by (scan_agg_partitioned.h), with
// the phases switched off one at a time, to see which one the pass is waiting for.
//   V0   the product's pass 1: per 2048-tuple batch LDS histogram -> one global cursor atomic per bin ->
//        LDS staging ordered by bin -> copy-out of the runs
//   ABL bit 1: no global cursor atomics (positions from a block-private counter: same write pattern)
//       bit 2: no copy-out (nothing is written)
//       bit 4: no staging and no copy-out (load + hash + histogram only)
//       bit 8: no LDS rank atomics (rank = lane-derived)
//       bit 16: no global cursor atomics, every block appends to PRIVATE sub-slabs (cursor in LDS): what exact
//               offsets from a counting pass -- or block-private chunk claims -- would give
//       bit 32 / 64: global cursors padded to one per 128-B / 256-B line (instead of 128 cursors in 512 bytes)
//       bit 128: 16-byte LDS staging and copy-out (ds_write_b128 / global_store_dwordx4)
//       bit 256: with bit 16, every sub-slab starts 64 bytes into a 128-byte line (runs aligned to 64 B only)
//       bit 512: with bit 16, every sub-slab starts 32 bytes into a line (runs aligned to 32 B only)
//       bit 1024: one global cursor and one sub-slab per (bin, XCD = block % 8), cursors one per 128-B line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define GPTR(T, p) reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p))

constexpr int kBlock = 512;
constexpr int kMaxBins = 256;

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t hash64(int64_t key) {  // MurmurHash3_x86_32 of 8 bytes, seed 0
  uint32_t h1 = 0;
  const uint32_t c1 = 0xcc9e2d51, c2 = 0x1b873593;
  uint32_t k1 = (uint32_t)key;
  k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64;
  k1 = (uint32_t)((uint64_t)key >> 32);
  k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64;
  h1 ^= 8; h1 ^= h1 >> 16; h1 *= 0x85ebca6b; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35; h1 ^= h1 >> 16;
  return h1;
}

struct Args {
  const int64_t* key;
  const int64_t* val;
  int64_t n;
  int64_t* out;      // [nbins][cap][2]
  uint32_t* fill;    // [nbins]
  uint64_t cap;
  uint32_t nbins, fine_count, p2;
};

template <int VR, int ABL>
__global__ __launch_bounds__(kBlock) void k_scatter_v0(Args a) {
  constexpr int T = kBlock * VR;
  __shared__ uint32_t s_cnt[kMaxBins], s_lpos[kMaxBins], s_base[kMaxBins];
  extern __shared__ __attribute__((aligned(16))) int64_t s_stage[];  // [T][2]
  __shared__ uint32_t s_pos[T];
  __shared__ uint16_t s_bin[T];
  const int tid = threadIdx.x;
  for (int i = tid; i < kMaxBins; i += kBlock) s_cnt[i] = 0;
  __syncthreads();
  const int64_t ntiles = (a.n + T - 1) / T;
  uint32_t fake = blockIdx.x * 7919u;
  __shared__ uint32_t s_cur[kMaxBins];
  for (int i = tid; i < kMaxBins; i += kBlock) s_cur[i] = 0;
  constexpr int CSTRIDE = (ABL & 32) ? 32 : ((ABL & 64) ? 64 : 1);
  const uint64_t sub_cap = (a.cap / gridDim.x) & ~7ull;  // bit 16: tuples per (block, bin) sub-slab, whole 128-B lines
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * T + tid;
    bool live[VR];
    int64_t k[VR], v[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const int64_t row = row0 + (int64_t)r * kBlock;
      live[r] = row < a.n;
      k[r] = live[r] ? __builtin_nontemporal_load(GPTR(int64_t, a.key) + row) : 0;
      v[r] = live[r] ? __builtin_nontemporal_load(GPTR(int64_t, a.val) + row) : 0;
    }
    uint32_t bin[VR], rank[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      bin[r] = 0; rank[r] = 0;
      if (live[r]) {
        const uint32_t f = (uint32_t)(((uint64_t)hash64(k[r]) * a.fine_count) >> 32);
        bin[r] = f / a.p2;
        if (ABL & 8) rank[r] = 0; else rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
      }
    }
    if (ABL & 8) {  // keep the counts plausible: every bin gets T / nbins
      if (tid < (int)a.nbins) s_cnt[tid] = T / a.nbins;
#pragma unroll
      for (int r = 0; r < VR; ++r) { bin[r] = (tid + r * 37) % a.nbins; rank[r] = (tid / a.nbins) * VR + r; }
    }
    __syncthreads();
    if (ABL & 4) {
      if (tid == 0 && k[0] == 0x123456789) a.out[0] = v[0] + bin[0] + rank[0];
      __syncthreads();
      for (int i = tid; i < kMaxBins; i += kBlock) s_cnt[i] = 0;
      __syncthreads();
      continue;
    }
    if (tid < kMaxBins) {
      const uint32_t n = tid < (int)a.nbins ? s_cnt[tid] : 0;
      if (ABL & 16) {
        const uint32_t c = s_cur[tid];
        s_base[tid] = (uint32_t)(blockIdx.x * sub_cap) + (c + n <= sub_cap ? c : 0) + ((ABL & 256) ? 4 : 0) + ((ABL & 512) ? 2 : 0);
        s_cur[tid] = c + n <= sub_cap ? c + n : n;
      } else if (ABL & 1) {
        s_base[tid] = (fake + tid * 131u) % (uint32_t)(a.cap - T);
      } else if (ABL & 1024) {  // one cursor and one sub-slab per (bin, XCD): a line is only ever written by one XCD's L2
        const uint32_t x = blockIdx.x & 7u;
        const uint32_t sub = (uint32_t)((a.cap / 8) & ~7ull);
        s_base[tid] = x * sub + (n ? atomicAdd(a.fill + (tid * 8 + x) * 32, n) : 0u);
      } else {
        s_base[tid] = n ? atomicAdd(a.fill + tid * CSTRIDE, n) : 0u;
      }
    }
    fake += 4099u;
    if (tid < 64) {
      uint32_t carry = 0;
      for (int c0 = 0; c0 < kMaxBins; c0 += 64) {
        const uint32_t n = s_cnt[c0 + tid];
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t x = __shfl_up(incl, d, 64);
          if (tid >= d) incl += x;
        }
        s_lpos[c0 + tid] = carry + incl - n;
        carry += __shfl(incl, 63, 64);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      if (live[r]) {
        const uint32_t si = s_lpos[bin[r]] + rank[r];
        const uint64_t pos = (uint64_t)s_base[bin[r]] + rank[r];
        s_bin[si] = (uint16_t)bin[r];
        s_pos[si] = (uint32_t)(pos < a.cap ? pos : a.cap - 1);
        if (ABL & 128) {
          typedef long long __attribute__((ext_vector_type(2))) i64x2;
          i64x2 t; t.x = k[r]; t.y = v[r];
          reinterpret_cast<i64x2*>(s_stage)[si] = t;
        } else {
          s_stage[(size_t)si * 2] = k[r];
          s_stage[(size_t)si * 2 + 1] = v[r];
        }
      }
    }
    __syncthreads();
    if (!(ABL & 2)) {
      const uint32_t total = s_lpos[kMaxBins - 1] + s_cnt[kMaxBins - 1];
      for (uint32_t i = tid; i < total; i += kBlock) {
        const uint32_t b = s_bin[i];
        int64_t* q = a.out + ((size_t)b * a.cap + s_pos[i]) * 2;
        if (ABL & 128) {
          typedef long long __attribute__((ext_vector_type(2))) i64x2;
          *reinterpret_cast<i64x2*>(q) = reinterpret_cast<const i64x2*>(s_stage)[i];
        } else {
          q[0] = s_stage[(size_t)i * 2];
          q[1] = s_stage[(size_t)i * 2 + 1];
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < kMaxBins; i += kBlock) s_cnt[i] = 0;
    __syncthreads();
  }
}

template <int VR, int ABL>
static void run_v0(Args a, int grid, const char* label) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = (size_t)kBlock * VR * 16;
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemsetAsync(a.fill, 0, kMaxBins * 4 * 64 * 8, 0));
    CK(hipEventRecord(e0));
    k_scatter_v0<VR, ABL><<<grid, kBlock, lds>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  printf("%-34s VR=%d grid=%-5d %.3f ms  %.3e rows/s  %.0f GB/s (32 B/row)\n", label, VR, grid, best, a.n / (best * 1e-3),
         a.n * 32.0 / (best * 1e-3) / 1e9);
  fflush(stdout);
}

int main() {
  const int64_t n = 256000000;
  Args a;
  int64_t *key, *val;
  CK(hipMalloc(&key, n * 8)); CK(hipMalloc(&val, n * 8));
  std::vector<int64_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int64_t)(s % 25600000ull); }
  CK(hipMemcpy(key, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(val, 1, n * 8));
  a.key = key; a.val = val; a.n = n;
  a.nbins = 128; a.p2 = 105; a.fine_count = 128 * 105;
  a.cap = (n / a.nbins + n / (a.nbins * 16) + 8192) & ~7ull;  // whole 128-B lines: slab bases stay line-aligned
  CK(hipMalloc(&a.out, (size_t)a.nbins * a.cap * 16));
  CK(hipMalloc(&a.fill, kMaxBins * 4 * 64 * 8));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cu = prop.multiProcessorCount;
  for (int g : {cu * 3}) {
    run_v0<4, 32 + 128>(a, g, "v0 padded 128 B + 16-B stores");
    run_v0<4, 1024 + 128>(a, g, "v0 per-XCD cursors and sub-slabs");
    run_v0<4, 32 + 128>(a, g, "v0 padded 128 B + 16-B stores");
    run_v0<4, 1024 + 128>(a, g, "v0 per-XCD cursors and sub-slabs");
    run_v0<4, 0>(a, g, "v0 full");
    run_v0<4, 64 + 128>(a, g, "v0 padded 256 B + 16-B stores");
    run_v0<4, 16 + 128>(a, g, "v0 private sub-slabs + 16-B stores");
    run_v0<4, 3>(a, g, "v0 no atomics, no copy-out");
  }
  for (int g : {cu * 3}) {
    run_v0<4, 8 + 16 + 128>(a, g, "aligned 256-B runs, private, 16-B st");
    run_v0<4, 8 + 16 + 128 + 256>(a, g, "256-B runs at 64-B alignment");
    run_v0<4, 8 + 16 + 128 + 512>(a, g, "256-B runs at 32-B alignment");
    run_v0<4, 8 + 16 + 128 + 256 + 512>(a, g, "256-B runs at 96-B offset");
  }
  // exact 16-tuple runs per bin and batch (no LDS rank), private sub-slabs: every run is two aligned 128-B lines
  run_v0<4, 8 + 16>(a, cu * 3, "aligned 256-B runs, private, 8-B st");
  run_v0<4, 8 + 16 + 128>(a, cu * 3, "aligned 256-B runs, private, 16-B st");
  run_v0<4, 8 + 64 + 128>(a, cu * 3, "aligned 256-B runs, padded cursors");
  run_v0<4, 8 + 2>(a, cu * 3, "no LDS rank, no copy-out");
  // longer runs: 4096-tuple batches (32 tuples = 512 B per bin and batch), LDS-limited to one block per CU
  run_v0<8, 8 + 16 + 128>(a, cu, "aligned 512-B runs, private, 1/CU");
  run_v0<8, 1024 + 128>(a, cu, "per-XCD cursors, 4096 batch, 1/CU");
  run_v0<8, 1024 + 128>(a, cu * 2, "per-XCD cursors, 4096 batch, 2/CU");
  run_v0<8, 0>(a, cu * 2, "v0 full");
  run_v0<8, 64 + 128>(a, cu * 2, "v0 padded 256 B + 16-B stores");
  run_v0<8, 16 + 128>(a, cu * 2, "v0 private sub-slabs + 16-B stores");
  return 0;
}
