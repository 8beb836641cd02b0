// Micro-benchmark for the join probe (C3): what one probe costs by ENTRY SIZE and TABLE SIZE, with the
// fact table's two 8-byte columns streamed beside it (non-temporal), as in the real kernel.
//   stream fk[i], val[i] (int64, nt) -> gather table[fk[i]] (E bytes) -> sum(val + payload)
// Run plain for timings, and under `rocprofv3 --kernel-trace --pmc ...` for the TCC counters (each
// variant is its own kernel instantiation, so the counter CSV separates them by name).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef long long __attribute__((ext_vector_type(2))) i64x2;
#define GPTR(T, p) reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p))

// E = entry bytes (4, 8, 16); NT = 1: table loads carry the nt hint too
template <int E, int NT, int U>
__global__ __launch_bounds__(256) void k_probe(const int64_t* fk, const int64_t* val, const int8_t* table, int64_t n,
                                               unsigned long long* out) {
  const int64_t tid = blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;
  int64_t acc = 0;
  for (int64_t base = tid; base + (U - 1) * nthreads < n; base += nthreads * U) {
    int64_t k[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      k[u] = __builtin_nontemporal_load(GPTR(int64_t, fk) + base + u * nthreads);
      v[u] = __builtin_nontemporal_load(GPTR(int64_t, val) + base + u * nthreads);
    }
    int64_t p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (E == 4) {
        const auto* q = GPTR(int32_t, table) + k[u];
        p[u] = NT ? __builtin_nontemporal_load(q) : *q;
      } else if (E == 8) {
        const auto* q = GPTR(int64_t, table) + k[u];
        p[u] = NT ? __builtin_nontemporal_load(q) : *q;
      } else {
        const auto* q = GPTR(i64x2, table) + k[u];
        const i64x2 w = NT ? __builtin_nontemporal_load(q) : *q;
        p[u] = w.x + w.y;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u] + p[u];
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
}

template <int E, int NT, int U>
static void run(const int64_t* fk, const int64_t* val, const int8_t* table, int64_t n, unsigned long long* out,
                int grid, int64_t nd, const char* order) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_probe<E, NT, U><<<grid, 256>>>(fk, val, table, n, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int reps = 3;
  for (int r = 0; r < reps; ++r) k_probe<E, NT, U><<<grid, 256>>>(fk, val, table, n, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  printf("probe E=%-2d nt=%d U=%d %-6s nd=%-9lld table=%4lld MB grid=%d  %.3f ms  %.3e rows/s\n", E, NT, U, order,
         (long long)nd, (long long)(nd * E >> 20), grid, ms, n / (ms * 1e-3));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int64_t n = 256000000;
  int64_t *fk, *val; int8_t* table; unsigned long long* out;
  CK(hipMalloc(&fk, n * 8));
  CK(hipMalloc(&val, n * 8));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(val, 1, n * 8));
  std::vector<int64_t> h(n);
  std::vector<int64_t> nds = {1000000, 10000000, 30000000};
  if (argc > 1) {  // gather2 <entries> [norand]: one table size (so that a counter run attributes per size)
    nds.clear();
    nds.push_back(atoll(argv[1]));
  }
  const int grid = 2048;
  for (int64_t nd : nds) {
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int64_t)(s % (uint64_t)nd); }
    CK(hipMemcpy(fk, h.data(), n * 8, hipMemcpyHostToDevice));
    // table memory: default (cached in L2), "uc" = uncached MTYPE, "fg" = fine-grained
    const char* mode = argc > 2 ? argv[2] : "default";
    if (mode[0] == 'u') {
      CK(hipExtMallocWithFlags((void**)&table, nd * 16, hipDeviceMallocUncached));
    } else if (mode[0] == 'f') {
      CK(hipExtMallocWithFlags((void**)&table, nd * 16, hipDeviceMallocFinegrained));
    } else {
      CK(hipMalloc(&table, nd * 16));
    }
    printf("# table memory: %s\n", mode);
    CK(hipMemset(table, 1, nd * 16));
    run<4, 0, 4>(fk, val, table, n, out, grid, nd, "rand");
    run<4, 1, 4>(fk, val, table, n, out, grid, nd, "rand");
    run<8, 0, 4>(fk, val, table, n, out, grid, nd, "rand");
    run<8, 1, 4>(fk, val, table, n, out, grid, nd, "rand");
    run<16, 0, 4>(fk, val, table, n, out, grid, nd, "rand");
    run<16, 1, 4>(fk, val, table, n, out, grid, nd, "rand");
    run<8, 0, 8>(fk, val, table, n, out, grid, nd, "rand");
    if (nd == 10000000) {
      // keys random inside windows of W entries, windows visited in order: what a radix partition of the
      // fact rows by key range would give the probe (window bytes = W * E)
      for (int64_t wbits : {12, 16, 18}) {
        const int64_t W = 1ll << wbits;
        for (int64_t i = 0; i < n; ++i) {
          s ^= s << 13; s ^= s >> 7; s ^= s << 17;
          int64_t w0 = (int64_t)((__int128)i * nd / n) & ~(W - 1);
          int64_t k = w0 + (int64_t)(s & (uint64_t)(W - 1));
          h[i] = k < nd ? k : nd - 1;
        }
        CK(hipMemcpy(fk, h.data(), n * 8, hipMemcpyHostToDevice));
        char lab[16];
        snprintf(lab, sizeof lab, "win%lld", (long long)wbits);
        run<8, 0, 4>(fk, val, table, n, out, grid, nd, lab);
        run<16, 0, 4>(fk, val, table, n, out, grid, nd, lab);
      }
    }
    CK(hipFree(table));
  }
  return 0;
}
