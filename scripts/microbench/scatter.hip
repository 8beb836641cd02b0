// Micro-benchmark: scatter 16-byte tuples (key, value) into P partitions through per-partition cursors
// -- the write side of a radix-partitioned group-by.  P sweeps from "few, well coalesced" to "as many
// bins as LDS-sized sub-tables need"; dst = precomputed exact position (as after a counting pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef long long __attribute__((ext_vector_type(2))) i64x2;

// direct: each lane writes its tuple to dst[pos[i]] (pos = random permutation grouped by bin)
__global__ __launch_bounds__(256) void k_scatter(const i64x2* __restrict__ src, const int64_t* __restrict__ pos,
                                                 i64x2* __restrict__ dst, int64_t n) {
  const int64_t tid = blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;
  for (int64_t i = tid; i < n; i += nthreads) {
    const i64x2 t = __builtin_nontemporal_load(src + i);
    dst[__builtin_nontemporal_load(pos + i)] = t;
  }
}

int main() {
  const int64_t n = 128000000;
  i64x2 *src, *dst; int64_t* pos;
  CK(hipMalloc(&src, n * 16)); CK(hipMalloc(&dst, n * 16)); CK(hipMalloc(&pos, n * 8));
  CK(hipMemset(src, 1, n * 16));
  std::vector<int64_t> h(n);
  std::vector<uint32_t> bin(n);
  for (int64_t P : {256ll, 4096ll, 16384ll, 65536ll, 0ll}) {
    // bins of equal size; tuple i goes to bin b(i) (random), position = bin start + arrival rank
    uint64_t s = 88172645463325252ull;
    if (P == 0) {  // fully random permutation-ish: position = random bijection via multiplicative hash
      for (int64_t i = 0; i < n; ++i) h[i] = (int64_t)(((unsigned __int128)i * 0x9E3779B97F4A7C15ull) % (uint64_t)n);
    } else {
      std::vector<int64_t> cnt(P, 0), start(P + 1, 0);
      for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; bin[i] = (uint32_t)(s % (uint64_t)P); cnt[bin[i]]++; }
      for (int64_t b = 0; b < P; ++b) start[b + 1] = start[b] + cnt[b];
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int64_t i = 0; i < n; ++i) h[i] = start[bin[i]] + cnt[bin[i]]++;
    }
    CK(hipMemcpy(pos, h.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_scatter<<<2048, 256>>>(src, pos, dst, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_scatter<<<2048, 256>>>(src, pos, dst, n);
    k_scatter<<<2048, 256>>>(src, pos, dst, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 2;
    printf("scatter16 P=%-8lld %.3f ms  %.3e rows/s\n", (long long)P, ms, n / (ms * 1e-3));
  }
  return 0;
}
