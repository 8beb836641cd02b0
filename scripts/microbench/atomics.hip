// Micro-benchmark: random 64-bit atomic adds into a table (the open-addressing group-by's inner op),
// by table size, atomic scope and with/without a preceding key load -- the ceilings the baseline-hash
// kernel is priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int SCOPE, int MODE>  // MODE 0: atomic add only; 1: key load + add; 2: CAS(key) + add; 3: plain load only
__global__ __launch_bounds__(256) void k_atomic(const int64_t* __restrict__ idx, unsigned long long* table, int64_t n) {
  const int64_t tid = blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;
  unsigned long long sink = 0;
  for (int64_t i = tid; i < n; i += nthreads) {
    const int64_t k = __builtin_nontemporal_load(idx + i);
    unsigned long long* e = table + 2 * k;
    if (MODE == 1) {
      sink += __hip_atomic_load(e, __ATOMIC_RELAXED, SCOPE);
    } else if (MODE == 2) {
      unsigned long long expected = ~0ull;
      __hip_atomic_compare_exchange_strong(e, &expected, (unsigned long long)k, __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE);
    } else if (MODE == 3) {
      sink += e[0] + e[1];
      continue;
    }
    __hip_atomic_fetch_add(e + 1, 1ull, __ATOMIC_RELAXED, SCOPE);
  }
  if (sink == 0x1234567) table[0] = sink;
}

template <int SCOPE, int MODE>
static void run(const int64_t* idx, unsigned long long* table, int64_t n, int64_t entries, const char* label) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_atomic<SCOPE, MODE><<<2048, 256>>>(idx, table, n);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k_atomic<SCOPE, MODE><<<2048, 256>>>(idx, table, n);
  k_atomic<SCOPE, MODE><<<2048, 256>>>(idx, table, n);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= 2;
  printf("%-28s entries=%-10lld %.3f ms  %.3e rows/s\n", label, (long long)entries, ms, n / (ms * 1e-3));
}

int main() {
  const int64_t n = 128000000;
  int64_t* idx; unsigned long long* table;
  CK(hipMalloc(&idx, n * 8));
  std::vector<int64_t> h(n);
  for (int64_t entries : {4096ll, 65536ll, 1000000ll, 51200000ll}) {
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int64_t)(s % (uint64_t)entries); }
    CK(hipMemcpy(idx, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&table, entries * 16));
    CK(hipMemset(table, 0, entries * 16));
    run<__HIP_MEMORY_SCOPE_AGENT, 3>(idx, table, n, entries, "load16 only");
    run<__HIP_MEMORY_SCOPE_AGENT, 0>(idx, table, n, entries, "add agent");
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 0>(idx, table, n, entries, "add workgroup");
    run<__HIP_MEMORY_SCOPE_AGENT, 1>(idx, table, n, entries, "load+add agent");
    run<__HIP_MEMORY_SCOPE_AGENT, 2>(idx, table, n, entries, "cas+add agent");
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 2>(idx, table, n, entries, "cas+add workgroup");
    // sorted indices: what a partition pass would buy (all hits of an entry adjacent)
    for (int64_t i = 0; i < n; ++i) h[i] = (int64_t)((__int128)i * entries / n);
    CK(hipMemcpy(idx, h.data(), n * 8, hipMemcpyHostToDevice));
    run<__HIP_MEMORY_SCOPE_AGENT, 0>(idx, table, n, entries, "add agent, sorted idx");
    CK(hipFree(table));
  }
  return 0;
}
