// Micro-benchmark: ceiling for a dependent random gather (join probe shape) on gfx950.
//   stream idx[i] (int64, non-temporal) -> gather table[idx[i]] (8 or 16 bytes) -> sum.
// Sweeps the table size and the number of independent gathers a lane keeps in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef long long __attribute__((ext_vector_type(2))) i64x2;

template <int U, int W>  // W = words gathered per probe (1 or 2)
__global__ __launch_bounds__(256) void k_gather(const int64_t* __restrict__ idx, const int64_t* __restrict__ table,
                                               int64_t n, int stride_words, unsigned long long* out) {
  const int64_t tid = blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;
  int64_t acc = 0;
  for (int64_t base = tid * U; base + U <= n; base += nthreads * U) {
    int64_t k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = __builtin_nontemporal_load(idx + base + u);
    if (W == 2) {
      i64x2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const i64x2*>(table + k[u] * stride_words);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    } else {
      int64_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = table[k[u] * stride_words];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
}

template <int U, int W>
static void run(const int64_t* idx, const int64_t* table, int64_t n, int stride, unsigned long long* out, int grid,
                const char* label, int64_t nd) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_gather<U, W><<<grid, 256>>>(idx, table, n, stride, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int reps = 3;
  for (int r = 0; r < reps; ++r) k_gather<U, W><<<grid, 256>>>(idx, table, n, stride, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  printf("%s nd=%lld U=%d W=%d grid=%d  %.3f ms  %.3e rows/s\n", label, (long long)nd, U, W, grid, ms, n / (ms * 1e-3));
}

int main(int argc, char** argv) {
  const int64_t n = 128000000;
  int64_t* idx; int64_t* table; unsigned long long* out;
  CK(hipMalloc(&idx, n * 8));
  CK(hipMalloc(&out, 8));
  std::vector<int64_t> h(n);
  const int64_t nds[] = {100000, 1000000, 10000000};
  for (int64_t nd : nds) {
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int64_t)(s % (uint64_t)nd); }
    CK(hipMemcpy(idx, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&table, nd * 16));
    CK(hipMemset(table, 1, nd * 16));
    for (int grid : {1024, 2048, 4096}) {
      run<1, 1>(idx, table, n, 2, out, grid, "rand", nd);
      run<4, 1>(idx, table, n, 2, out, grid, "rand", nd);
      run<8, 1>(idx, table, n, 2, out, grid, "rand", nd);
      run<16, 1>(idx, table, n, 2, out, grid, "rand", nd);
      run<4, 2>(idx, table, n, 2, out, grid, "rand", nd);
      run<8, 2>(idx, table, n, 2, out, grid, "rand", nd);
      run<16, 2>(idx, table, n, 2, out, grid, "rand", nd);
    }
    // sorted keys: the locality a radix partition would buy
    for (int64_t i = 0; i < n; ++i) h[i] = (int64_t)((__int128)i * nd / n);
    CK(hipMemcpy(idx, h.data(), n * 8, hipMemcpyHostToDevice));
    run<8, 2>(idx, table, n, 2, out, 2048, "sorted", nd);
    // partition-local randomness: keys random within 4096-slot (64 KB) windows, windows in order
    for (int64_t i = 0; i < n; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      int64_t w = (int64_t)((__int128)i * nd / n) & ~4095ll;
      int64_t k = w + (int64_t)(s & 4095);
      h[i] = k < nd ? k : nd - 1;
    }
    CK(hipMemcpy(idx, h.data(), n * 8, hipMemcpyHostToDevice));
    run<8, 2>(idx, table, n, 2, out, 2048, "win64K", nd);
    CK(hipFree(table));
  }
  return 0;
}
