// Micro-benchmark (round 5): what does an LDS atomic cost on gfx950?  The on-chip group-by kernels (scan_agg_fast.h,
// scan_bh_fast.h) spend one ds atomic per aggregate per row; this prices them: lane-operations per clock and CU for
//   add_u64 / add_u32 / min_i64 / read_b64 / read_b128, with the 64 lanes of a wave on
//   (a) 64 different words (no bank or address conflict), (b) ONE word, (c) G random groups x REP interleaved replicas
//   (the layout of the group-by tables: word (g * REP + lane % REP)).
// Usage: lds_atomics  -> a table; every kernel runs `blocks_per_cu` x CUs blocks of 256 threads, ITER x 4 ops per lane.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int kBlock = 256;
constexpr int kIter = 2048;

enum Op { ADD64, ADD32, MIN64, READ64, READ128, CAS64 };

template <int OP>
__device__ __forceinline__ void one(unsigned long long* lds, uint32_t idx, unsigned long long v, unsigned long long& sink) {
  if (OP == ADD64) atomicAdd(lds + idx, v);
  else if (OP == ADD32) atomicAdd(reinterpret_cast<unsigned int*>(lds) + idx, static_cast<unsigned int>(v));
  else if (OP == MIN64) atomicMin(reinterpret_cast<long long*>(lds) + idx, static_cast<long long>(v));
  else if (OP == READ64) sink += reinterpret_cast<volatile unsigned long long*>(lds)[idx];
  else if (OP == READ128) {
    typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;
    u64x2 t;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds + (idx & ~1u)))));
    sink += t.x + t.y;
  }
  else sink += atomicCAS(lds + idx, 0x7fffffffffffffffull, v);
}

// MODE 0: lane -> its own word; 1: all lanes one word; 2: (group, replica) words, group from a per-lane LCG
template <int OP, int MODE>
__global__ __launch_bounds__(kBlock) void k_lds(unsigned long long* out, uint32_t groups, uint32_t rep) {
  __shared__ unsigned long long lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += kBlock) lds[i] = OP == MIN64 ? 0x7fffffffffffffffull : 0;
  __syncthreads();
  unsigned long long sink = 0;
  uint32_t s = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
  const uint32_t my_rep = tid & (rep - 1);
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t idx;
      if (MODE == 0) idx = tid + j * kBlock;
      else if (MODE == 1) idx = j;
      else {
        s = s * 1664525u + 1013904223u;
        idx = (__umulhi(s, groups) * 4 + j) * rep + my_rep;  // 4 words per group, as an entry's value words
      }
      one<OP>(lds, idx, static_cast<unsigned long long>(it + j + 1), sink);
    }
  }
  __syncthreads();
  if (sink == 0x1234567ull || lds[tid] == 0xdeadbeefull) out[0] = sink;
}

template <int OP, int MODE>
static void run(const char* name, unsigned cu, unsigned per_cu, unsigned long long* out, uint32_t groups, uint32_t rep, double ghz) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_lds<OP, MODE>), dim3(cu * per_cu), dim3(kBlock), 0, 0, out, groups, rep);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best) best = ms;
  }
  const double ops = static_cast<double>(cu) * per_cu * kBlock * kIter * 4;
  printf("  %-34s %2u blocks/CU  %8.3f ms  %7.2f lane-ops/clk/CU  %9.3e lane-ops/s chip\n", name, per_cu, best,
         ops / (best * 1e-3) / cu / (ghz * 1e9), ops / (best * 1e-3));
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const unsigned cu = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  unsigned long long* out; CK(hipMalloc(&out, 64));
  printf("# %u CUs at %.2f GHz (hipDeviceProp clockRate); 256-thread blocks, %d x 4 ops per lane\n", cu, ghz, kIter);
  for (unsigned per_cu : {2u, 4u, 8u}) {
    run<ADD64, 0>("add_u64, 64 different words", cu, per_cu, out, 1, 1, ghz);
    run<ADD32, 0>("add_u32, 64 different words", cu, per_cu, out, 1, 1, ghz);
    run<MIN64, 0>("min_i64, 64 different words", cu, per_cu, out, 1, 1, ghz);
    run<CAS64, 0>("cmpswap_b64 (rtn), different words", cu, per_cu, out, 1, 1, ghz);
    run<READ64, 0>("read_b64, 64 different words", cu, per_cu, out, 1, 1, ghz);
    run<READ128, 0>("read_b128", cu, per_cu, out, 1, 1, ghz);
    run<ADD64, 1>("add_u64, one word", cu, per_cu, out, 1, 1, ghz);
    run<ADD32, 1>("add_u32, one word", cu, per_cu, out, 1, 1, ghz);
    run<READ64, 1>("read_b64, one word (broadcast)", cu, per_cu, out, 1, 1, ghz);
    run<ADD64, 2>("add_u64, 10 groups x 16 replicas", cu, per_cu, out, 10, 16, ghz);
    run<ADD64, 2>("add_u64, 10 groups x 32 replicas", cu, per_cu, out, 10, 32, ghz);
    run<ADD64, 2>("add_u64, 10 groups x 1 replica", cu, per_cu, out, 10, 1, ghz);
    run<ADD64, 2>("add_u64, 100 groups x 4 replicas", cu, per_cu, out, 100, 4, ghz);
    run<ADD64, 2>("add_u64, 1000 groups x 1 replica", cu, per_cu, out, 1000, 1, ghz);
    run<ADD32, 2>("add_u32, 10 groups x 16 replicas", cu, per_cu, out, 10, 16, ghz);
    run<ADD32, 2>("add_u32, 1000 groups x 1 replica", cu, per_cu, out, 1000, 1, ghz);
    run<MIN64, 2>("min_i64, 10 groups x 16 replicas", cu, per_cu, out, 10, 16, ghz);
    run<READ64, 2>("read_b64, 10 groups x 16 replicas", cu, per_cu, out, 10, 16, ghz);
    run<READ64, 2>("read_b64, 1000 groups x 1 replica", cu, per_cu, out, 1000, 1, ghz);
  }
  return 0;
}
